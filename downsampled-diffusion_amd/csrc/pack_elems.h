// pack_elems.h -- one element of each kernel-layout weight copy, shared by the single-tensor pack kernels (layout_pack.hip,
// conv_wgrad.hip, conv_wino.hip) and the job-table kernel that refreshes every copy of a model in one launch (pack_jobs.hip):
// the two paths cannot drift apart, a copy refreshed by either holds the same bits.
#pragma once

namespace ddk {

// dst[o][tap][i_pad] <- w[o][i][ky][kx].  The input channels may come from TWO sources that are each padded on their own (the concat
// of unet.py:97 at widths that are not multiples of 32): channels [0, split) sit at [0, split), channels [split, I) at
// [split_pad, split_pad + I - split); split == I, split_pad == i_pad is the plain single-source layout.
__device__ __forceinline__ void pack_conv_weight_elem(const float* __restrict__ w, float* __restrict__ dst, long long idx, int I, int taps,
                                                      int i_pad, int split, int split_pad) {
    // 32-bit index arithmetic (a packed weight has far fewer than 2^31 elements; the callers check): the 64-bit divisions of the first
    // version were ~240 instructions per element -- 285 us for the 48 M elements of a model's copies
    const unsigned u = (unsigned)idx;
    const int ip = (int)(u % (unsigned)i_pad);
    const unsigned r = u / (unsigned)i_pad;
    const int tap = (int)(r % (unsigned)taps);
    const long long o = r / (unsigned)taps;
    const int i = ip < split_pad ? (ip < split ? ip : -1) : (ip - split_pad < I - split ? split + ip - split_pad : -1);
    dst[idx] = i >= 0 ? w[(o * I + i) * taps + tap] : 0.f;
}

// dst[phase][o][tap][i] <- w[i][o][ky][kx], ky = 1 - py + 2a, kx = 1 - px + 2b (phase = py*2+px, tap = a*2+b)
// (Ip, Op: channel counts padded to 32 at widths that are not multiples of 32 -- the padding rows / columns are zero)
__device__ __forceinline__ void pack_convT_weight_elem(const float* __restrict__ w, float* __restrict__ dst, long long idx, int I, int O,
                                                       int Ip, int Op) {
    const unsigned u = (unsigned)idx;
    const int i = (int)(u % (unsigned)Ip);
    unsigned r = u / (unsigned)Ip;
    const int tap = (int)(r & 3u); r >>= 2;
    const int o = (int)(r % (unsigned)Op);
    const int phase = (int)(r / (unsigned)Op);
    const int py = phase >> 1, px = phase & 1, a = tap >> 1, b = tap & 1;
    const int ky = 1 - py + 2 * a, kx = 1 - px + 2 * b;
    dst[idx] = (i < I && o < O) ? w[(((long long)i * O + o) * 4 + ky) * 4 + kx] : 0.f;
}

// dst[i][t][o] = w[o][i][T-1-t] (i < I), zero rows up to i_pad; o padded to o_pad with zeros
__device__ __forceinline__ void pack_dgrad_elem(const float* __restrict__ w, float* __restrict__ dst, long long idx, int O, int I, int taps,
                                                int o_pad) {
    const unsigned u = (unsigned)idx;
    const int o = (int)(u % (unsigned)o_pad);
    const unsigned r = u / (unsigned)o_pad;
    const int t = (int)(r % (unsigned)taps);
    const int i = (int)(r / (unsigned)taps);
    dst[idx] = (i < I && o < O) ? w[((long long)o * I + i) * taps + (taps - 1 - t)] : 0.f;
}

// G g G^T for one (n, c): G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; dst is [i_pad/32][16][O][32].
// DGRAD: the filter of the INPUT-gradient conv for the forward conv's input channels [lo, lo + O): again a 3x3 stride-1 conv, of
// dY, with g'[n][c][a][b] = w[c][lo + n][2-a][2-b] (taps flipped, channel roles swapped); w is the forward OIHW tensor with
// `wi` input channels, I = its output channels.  One kernel instead of flip + transpose + contiguous + pack.
template <bool DGRAD>
__device__ __forceinline__ void pack_wino_elem(const float* __restrict__ w, float* __restrict__ dst, long long idx, int O, int I, int i_pad,
                                               int lo, int wi) {
    const int c = (int)((unsigned)idx % (unsigned)i_pad);
    const int n = (int)((unsigned)idx / (unsigned)i_pad);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (DGRAD) g[a][b] = c < I ? w[(((long long)c * wi + lo + n) * 3 + (2 - a)) * 3 + (2 - b)] : 0.f;
            else g[a][b] = c < I ? w[(((long long)n * I + c) * 3 + a) * 3 + b] : 0.f;
        }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
        t[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0], u1 = 0.5f * ((t[i][0] + t[i][1]) + t[i][2]), u2 = 0.5f * ((t[i][0] - t[i][1]) + t[i][2]), u3 = t[i][2];
        const float u[4] = {u0, u1, u2, u3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[((((long long)(c >> 5)) * 16 + (4 * i + j)) * O + n) * 32 + (c & 31)] = u[j];
    }
}

}  // namespace ddk
