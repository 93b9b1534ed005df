// conv_first.hip -- the UNet's first Block conv: Conv2d(C_in, N, 3, padding=1) with C_in = image / latent channels (1..8).
//
// Reference: models/unet/unet.py:43-50 (dims[0] = unet_in: 1 MNIST, 3 CIFAR / CelebA, 8 dDDPM latents), models/unet/blocks.py:78
// (the conv of Block), :79 (the GroupNorm whose statistics the epilogue prepares).
//
// The generic kernels need C_in % 32 == 0, so this conv used to run zero-padded to 32 input channels (K = 288 per tap set, 512
// in the Winograd form) behind a pad kernel: 23.5 us for an op SURVEY.md section 8d prices at 2.2 us (HBM) / 3.8 us (FLOP).
// Here K = 9 * C_in exactly (72 at C_in = 8) and the input is read unpadded.
//
//   GEMM view: D[n][pixel] = sum_k W[n][k] X[pixel][k],  k = tap * C_in + c,  v_mfma_f32_32x32x2_f32 with the WEIGHTS as the
//   A operand (rows = output channels) and the gathered input patch as the B operand (columns = pixels): a lane then ends up
//   with 4 consecutive output channels of its pixel per accumulator quad -> float4 NHWC stores straight from registers.
//   Workgroup = 128 consecutive pixels x all N channels, 8 waves = 4 pixel groups of 32 x {even, odd} 32-channel blocks (the two
//   waves of a SIMD alternate between their MFMA run and their epilogue); a lane gathers its pixel's K values once (they stay in
//   registers for all its channel blocks); the packed filter ([N/32][K/2][64 lanes], 36 KB at N = 128, C_in = 8) and the tile's
//   input neighbourhood (in row-major order ONE contiguous range of 130 + 2 W pixels) are staged in LDS once per workgroup.
//   Epilogue: + bias, stores, and per (128-pixel tile, GroupNorm group) {mean, M2 about that mean} by Chan's pairwise merge
//   (quad -> 32 lanes -> 4 waves -> quads of the group: one pass, no cancellation), the format gn_apply_parts_kernel consumes.
//   In the sampler the workgroup 0 also does the step's bookkeeping (t_cur[b] <- counter; counter -= 1): it is the first kernel
//   of a reverse step, so the separate prepare / pad kernel disappears.
#include <type_traits>

#include "conv_common.h"

namespace ddk {

struct FirstParams {
    const float* x;        // [B][H][W][CIN], unpadded
    const float* wp;       // [N/32][K2][64]
    const float* bias;     // [N] or nullptr
    float* out;            // [B][H][W][N]
    float2* gn_part;       // [B*HW/128][groups] or nullptr
    int H, W, HW, N, NB;
    int cpg, groups;
    int64_t* counter;      // sampler bookkeeping (nullptr outside the sampler)
    int64_t* t_cur;
    int B;
    // FUSE: GroupNorm + Mish + time shift finished inside the launch (the image's tiles exchange their statistics); out = the activation
    const float* gamma;
    const float* beta;
    const float* temb;             // [rows][temb_stride] or null
    const long long* temb_rows;    // row of image b, or null: row b
    int temb_stride;
    float eps;
    unsigned long long* cl_rec;    // [tile][16] records {mean, M2} per group, 8 bytes each
    unsigned* cl_cnt;              // [image][8 * 16]: word 0 arrivals, word 1 departures (re-armed by the last one out)
    unsigned* cl_fail;
    int tpi;                       // tiles (workgroups) per image
};

// dst[(nb*K2 + s)*64 + l] = w[n = nb*32 + (l & 31)][k = 2s + (l >> 5)],  k = tap*I + c  <-  OIHW w[n][c][tap]
__global__ __launch_bounds__(256) void pack_conv_weight_first_kernel(const float* __restrict__ w, float* __restrict__ dst, int O, int I,
                                                                      int K2, long long total) {
    for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int l = (int)(idx & 63);
        const long long r = idx >> 6;
        const int s = (int)(r % K2), nb = (int)(r / K2);
        const int n = nb * 32 + (l & 31), k = 2 * s + (l >> 5);
        float v = 0.f;
        if (k < 9 * I && n < O) {
            const int tap = k / I, c = k - tap * I;
            v = w[((long long)n * I + c) * 9 + tap];
        }
        dst[idx] = v;
    }
}

// One level of the pairwise {mean, M2} merge inside a row of 16 lanes, on the DPP path (no LDS crossbar): CTRL pairs every lane
// with a partner (an involution), both sides compute the same merged pair.
template <int CTRL>
__device__ __forceinline__ float dpp_partner(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

constexpr unsigned long long FIRST_TIMEOUT_TICKS = 2000000ull;   // 20 ms of the 100 MHz s_memrealtime clock

// FUSE (round 6): the Block's GroupNorm + Mish + time shift in the same launch.  A 128-pixel tile's statistics cannot normalise it --
// the group spans the image's HW / 128 tiles -- so the workgroups of an image (consecutive block ids, dispatched together) exchange
// their per-group {mean, M2} records exactly as conv3x3_wino2_kernel<.., true> does (row 1 of MI355X_MICROARCH.md's sc1 table: 8-byte
// sc1 record stores by wave 0, drained, one arrival; one lane polls; workgroup barrier; sc1 loads), keep their 128 x N outputs in
// registers meanwhile, and store the ACTIVATION: the raw tensor (16.8 MB at cfg4) is never written or re-read and the
// gn_apply_parts_kernel launch behind this kernel disappears (14.6 + 10.1 us -> one launch).  N = 128 only (two channel blocks per wave).
template <int CIN, bool FUSE = false>
__global__ __launch_bounds__(512) void conv_first_kernel(const FirstParams p) {
    constexpr int K = 9 * CIN, K2 = (K + 1) / 2;
    constexpr int XS = CIN | 1;                      // LDS pixel stride of the staged input: odd -> conflict-free 4-byte gathers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                                 // NB * K2 * 64 floats
    float* bl = smem + p.NB * K2 * 64;                                // N floats: the bias (zeros when there is none)
    float2* qs = reinterpret_cast<float2*>(bl + p.N);                 // [4 pixel groups][N / 4 quads] {mean, M2} over 32 pixels x 4 channels
    float* xl = reinterpret_cast<float*>(qs + 4 * (p.N >> 2));       // (130 + 2 W) pixels x XS: the tile's input neighbourhood
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pg = wave & 3, par = wave >> 2;        // 8 waves: pixel group (32 pixels) x parity of the channel blocks it multiplies --
    const int pl = lane & 31, h = lane >> 5;         // the two waves of a SIMD alternate between their MFMA run and their epilogue

    // FUSE in the sampler: every workgroup needs the step's t (the time shift's row) and t_cur is only being written by workgroup 0 of
    // this very launch -- so each reads the counter itself, and nobody decrements it during this kernel (the step's last kernel does)
    const int64_t t_step = (FUSE && p.counter) ? *p.counter : 0;
    if (p.counter && blockIdx.x == 0) {          // first kernel of a reverse step: nobody else touches the counter now
        const int64_t v = *p.counter;
        for (int b = tid; b < p.B; b += 512) p.t_cur[b] = v;
        if constexpr (!FUSE) {
            __syncthreads();
            if (tid == 0) *p.counter = v - 1;
        }
    }
    // In row-major order the 3x3 neighbourhoods of 128 consecutive pixels of an image are ONE contiguous pixel range,
    // [first - W - 1, first + 128 + W + 1): staged once, coalesced; what falls outside the image is zero (and masked anyway).
    const int tile_pix = blockIdx.x * 128;
    const int bimg = tile_pix / p.HW, r0 = tile_pix - bimg * p.HW;        // the tile's image and its first pixel in it
    const int lo = r0 - p.W - 1, span = 130 + 2 * p.W;
    {
        const float* xi = p.x + (long long)bimg * p.HW * CIN;
        for (int i = tid; i < span * CIN; i += 512) {
            const int pix = i / CIN, c = i - pix * CIN, gp = lo + pix;
            xl[pix * XS + c] = (unsigned)gp < (unsigned)p.HW ? xi[gp * CIN + c] : 0.f;
        }
        // the packed filter and the bias, once per workgroup
        const int n4 = p.NB * K2 * 16;
        const float4* src = reinterpret_cast<const float4*>(p.wp);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < n4; i += 512) dst[i] = src[i];
        for (int i = tid; i < p.N; i += 512) bl[i] = p.bias ? p.bias[i] : 0.f;
    }
    // this lane's pixel and which of its 9 taps lie inside the image
    const int g = tile_pix + pg * 32 + pl;
    const int r = r0 + pg * 32 + pl;
    const int y = r / p.W, x = r - y * p.W;
    int toff[9];
    unsigned tmask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const bool ok = (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
        toff[t] = (r + dy * p.W + dx - lo) * XS;      // always inside the staged range
        if (ok) tmask |= 1u << t;
    }
    __syncthreads();                              // input range, filter and bias staged
    float bv[K2];
#pragma unroll
    for (int s = 0; s < K2; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int t0 = k0 / CIN, c0 = k0 % CIN;
        const int t1 = k1 < K ? k1 / CIN : 0, c1 = k1 < K ? k1 % CIN : 0;
        const bool ok = h ? (k1 < K && ((tmask >> t1) & 1u)) : ((tmask >> t0) & 1u);
        const int idx = h ? toff[t1] + c1 : toff[t0] + c0;
        const float v = xl[idx];
        bv[s] = ok ? v : 0.f;
    }

    const long long orow = (long long)g * p.N;
    float4 vk[2][4];             // FUSE: this wave's two channel blocks stay in registers until the image's statistics are known
    (void)vk;
    // (FUSE: exactly two channel blocks per wave, a compile-time trip count -- vk must be indexed by constants to stay in registers)
    auto block = [&](int nb, auto itc) {
        constexpr int it = decltype(itc)::value;
        (void)it;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* wrow = wl + nb * K2 * 64 + lane;
#pragma unroll
        for (int s = 0; s < K2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[s * 64], bv[s], acc, 0, 0, 0);
        // accumulator register 4q + i of this lane = channel nb*32 + 8q + 4h + i of pixel pl
        const int cb = nb * 32 + 4 * h;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bl + cb + 8 * q);
            v[q] = make_float4(acc[4 * q] + bb.x, acc[4 * q + 1] + bb.y, acc[4 * q + 2] + bb.z, acc[4 * q + 3] + bb.w);
            if constexpr (!FUSE) *reinterpret_cast<float4*>(p.out + orow + cb + 8 * q) = v[q];
            else vk[it][q] = v[q];
        }
        if (FUSE || p.gn_part) {
            // {mean, M2} of each quad, then merged over the 32 lanes that hold the same channels (equal counts at every level);
            // the four quads' trees run side by side
            float m[4], m2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m[q] = ((v[q].x + v[q].y) + (v[q].z + v[q].w)) * 0.25f;
                const float d0 = v[q].x - m[q], d1 = v[q].y - m[q], d2 = v[q].z - m[q], d3 = v[q].w - m[q];
                m2[q] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            // lanes 1, 2 apart (quad permutes), then the mirrored quad of the 8-lane group and the mirrored half of the 16-lane
            // row (DPP row_half_mirror / row_mirror: any involution that crosses the halves is a valid pairing), then the other row
            auto merge = [&](float (&mo)[4], float (&qo)[4], float half_cnt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = mo[q] - m[q];
                    m2[q] = (m2[q] + qo[q]) + d * d * half_cnt;
                    m[q] = 0.5f * (m[q] + mo[q]);
                }
            };
            float mo[4], qo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { mo[q] = dpp_partner<0xB1>(m[q]); qo[q] = dpp_partner<0xB1>(m2[q]); }     // quad_perm [1,0,3,2]
            merge(mo, qo, 2.0f);                  // n_a * n_b / (n_a + n_b) with n_a = n_b = 4, then 8, ...
#pragma unroll
            for (int q = 0; q < 4; ++q) { mo[q] = dpp_partner<0x4E>(m[q]); qo[q] = dpp_partner<0x4E>(m2[q]); }     // quad_perm [2,3,0,1]
            merge(mo, qo, 4.0f);
#pragma unroll
            for (int q = 0; q < 4; ++q) { mo[q] = dpp_partner<0x141>(m[q]); qo[q] = dpp_partner<0x141>(m2[q]); }   // row_half_mirror
            merge(mo, qo, 8.0f);
#pragma unroll
            for (int q = 0; q < 4; ++q) { mo[q] = dpp_partner<0x140>(m[q]); qo[q] = dpp_partner<0x140>(m2[q]); }   // row_mirror
            merge(mo, qo, 16.0f);
#pragma unroll
            for (int q = 0; q < 4; ++q) { mo[q] = __shfl_xor(m[q], 16, 64); qo[q] = __shfl_xor(m2[q], 16, 64); }
            merge(mo, qo, 32.0f);
            if (pl == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) qs[pg * (p.N >> 2) + ((cb + 8 * q) >> 2)] = make_float2(m[q], m2[q]);
            }
        }
    };
    if constexpr (FUSE) {
        block(par, std::integral_constant<int, 0>{});
        block(par + 2, std::integral_constant<int, 1>{});
    } else {
        for (int nb = par; nb < p.NB; nb += 2) block(nb, std::integral_constant<int, 0>{});
    }
    if (FUSE || p.gn_part) {
        __syncthreads();
        float2* ts = qs;                  // FUSE: [groups] merged {mean, rstd} of the image, over the quad records once they are consumed
        float gmean = 0.f, gm2 = 0.f;
        if (tid < p.groups) {
            // fixed order: pixel groups, then the group's quads (Chan et al., unequal counts)
            const int qpg = p.cpg >> 2, q0 = tid * qpg;
            float n = 0.f, mean = 0.f, m2 = 0.f;
            const float ni = 128.0f;              // 32 pixels x 4 channels per item
            for (int w = 0; w < 4; ++w)
                for (int q = 0; q < qpg; ++q) {
                    const float2 t = qs[w * (p.N >> 2) + q0 + q];
                    const float tot = n + ni, d = t.x - mean;
                    mean += d * (ni / tot);
                    m2 += t.y + d * d * (n * ni / tot);
                    n = tot;
                }
            if constexpr (!FUSE) p.gn_part[(long long)blockIdx.x * p.groups + tid] = make_float2(mean, m2);
            gmean = mean;
            gm2 = m2;
        }
        if constexpr (FUSE) {
            int* gave_up = reinterpret_cast<int*>(bl);       // the bias is consumed: its first word carries this workgroup's give-up flag
            unsigned* cnt = p.cl_cnt + (size_t)bimg * 8 * 16;
            if (tid < p.groups) {
                const unsigned long long bits = (unsigned long long)__float_as_uint(gmean) | ((unsigned long long)__float_as_uint(gm2) << 32);
                __hip_atomic_store(p.cl_rec + (size_t)blockIdx.x * 16 + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave == 0) {              // every record was stored by this wave (groups <= 64): drain, ONE arrival, one lane polls
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    *gave_up = 0;
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.tpi) {
                        const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
                        for (;;) {
                            __builtin_amdgcn_s_sleep(2);
                            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.tpi) break;
                            if (__builtin_amdgcn_s_memrealtime() - t_begin > FIRST_TIMEOUT_TICKS) {
                                if (p.cl_fail) __hip_atomic_fetch_add(p.cl_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                *gave_up = 1;
                                break;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < p.groups) {
                // the image's tiles of this group, equal counts: mean of means, M2 = sum M2_i + n_i sum (mean_i - mean)^2
                const unsigned long long* r0 = p.cl_rec + (size_t)bimg * p.tpi * 16 + tid;
                float rm[8], rq[8], ms = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rm[i] = rq[i] = 0.f;
                    if (i < p.tpi) {
                        const unsigned long long bits = __hip_atomic_load(r0 + (size_t)i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        rm[i] = __uint_as_float((unsigned)bits);
                        rq[i] = __uint_as_float((unsigned)(bits >> 32));
                        ms += rm[i];
                    }
                }
                const float mean = *gave_up ? __builtin_nanf("") : ms / (float)p.tpi;
                float m2 = 0.f, d2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (i < p.tpi) { m2 += rq[i]; d2 += (rm[i] - mean) * (rm[i] - mean); }
                const float n_i = 128.0f * (float)p.cpg;
                ts[tid] = make_float2(mean, 1.0f / sqrtf((m2 + n_i * d2) / ((float)p.tpi * n_i) + p.eps));
            }
            if (tid == 0) {               // departure: the last one out re-arms the image's counters for the next launch
                const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)p.tpi - 1u) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            const float* trow = nullptr;
            if (p.temb) trow = p.temb + (p.counter ? (long long)t_step : p.temb_rows ? p.temb_rows[bimg] : (long long)bimg) * p.temb_stride;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int cb = (par + 2 * it) * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = cb + 8 * q;
                    const float2 st = ts[ch / p.cpg];
                    const float4 ga = *reinterpret_cast<const float4*>(p.gamma + ch), be = *reinterpret_cast<const float4*>(p.beta + ch);
                    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (trow) sh = *reinterpret_cast<const float4*>(trow + ch);
                    const float4 v = vk[it][q];
                    *reinterpret_cast<float4*>(p.out + orow + ch) =
                        make_float4(mish_f((v.x - st.x) * st.y * ga.x + be.x) + sh.x, mish_f((v.y - st.x) * st.y * ga.y + be.y) + sh.y,
                                    mish_f((v.z - st.x) * st.y * ga.z + be.z) + sh.z, mish_f((v.w - st.x) * st.y * ga.w + be.w) + sh.w);
                }
            }
        }
    }
}

static size_t first_lds_bytes(int cin, int N, int W) {
    const int K2 = (9 * cin + 1) / 2;
    return (size_t)(N / 32) * K2 * 64 * sizeof(float) + (size_t)N * sizeof(float) + (size_t)4 * (N / 4) * sizeof(float2) +
           (size_t)(130 + 2 * W) * (cin | 1) * sizeof(float);
}

bool conv_first_ok(int cin, int N, int H, int W, int groups) {
    if (cin < 1 || cin > 8 || N < 32 || N % 32 || N > 256 || H < 1 || W < 1) return false;
    if ((long long)H * W % 128) return false;                      // a workgroup's 128 pixels lie inside one image
    if (first_lds_bytes(cin, N, W) > 96 * 1024) return false;
    if (groups <= 0 || groups > 64 || N % groups || (N / groups) % 4) return false;
    return true;
}

#define FIRST_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

int conv_first_init_device() {
#define X(C) DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_first_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
             DDK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_first_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    FIRST_CASES(X)
#undef X
    return DDK_OK;
}

// the first Block whole (conv + GroupNorm + Mish + time shift) in one launch: N = 128, at most 8 tiles of 128 pixels per image
bool conv_first_gn_ok(int cin, int N, int H, int W, int groups) {
    return conv_first_ok(cin, N, H, W, groups) && N == 128 && (long long)H * W / 128 <= 8 && groups <= 16;
}
size_t conv_first_gn_ws_floats(int B, int H, int W) { return (size_t)B * ((size_t)H * W / 128) * 32; }    // 16 eight-byte records per tile

int conv_first_gn(const float* x, const float* wp, const float* bias, const float* gamma, const float* beta, const float* temb,
                  int temb_stride, const long long* temb_rows, float eps, float* out, int B, int H, int W, int cin, int N, int groups,
                  float* records, unsigned* counters, unsigned* fail, int64_t* counter, int64_t* t_cur, hipStream_t st) {
    DDK_REQUIRE(x && wp && out && gamma && beta && records && counters && B > 0, "conv_first_gn: null pointer / B");
    DDK_REQUIRE(conv_first_gn_ok(cin, N, H, W, groups), "conv_first_gn: needs conv_first's shape with N == 128 and H*W <= 1024");
    DDK_REQUIRE(aligned16(wp) && aligned16(out) && aligned16(bias) && aligned16(gamma) && aligned16(beta) && aligned16(temb) &&
                    (temb_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(records) & 7u) == 0 && (reinterpret_cast<uintptr_t>(x) & 3u) == 0,
                "conv_first_gn: alignment");
    DDK_REQUIRE((long long)B * H * W * 8 < (1ll << 31), "conv_first_gn: B*H*W too large for 32-bit pixel offsets");
    DDK_REQUIRE((counter == nullptr) == (t_cur == nullptr), "conv_first_gn: counter and t_cur go together");
    FirstParams p{};
    p.x = x; p.wp = wp; p.bias = bias; p.out = out; p.gn_part = nullptr;
    p.H = H; p.W = W; p.HW = H * W; p.N = N; p.NB = N / 32;
    p.groups = groups; p.cpg = N / groups;
    p.counter = counter; p.t_cur = t_cur; p.B = B;
    p.gamma = gamma; p.beta = beta; p.temb = temb; p.temb_rows = temb_rows; p.temb_stride = temb_stride; p.eps = eps;
    p.cl_rec = reinterpret_cast<unsigned long long*>(records); p.cl_cnt = counters; p.cl_fail = fail; p.tpi = H * W / 128;
    const dim3 grid((unsigned)((long long)B * H * W / 128));
    const size_t lds = first_lds_bytes(cin, N, W);
    switch (cin) {
#define X(C) case C: hipLaunchKernelGGL((conv_first_kernel<C, true>), grid, dim3(512), lds, st, p); break;
        FIRST_CASES(X)
#undef X
        default: return fail_arg("conv_first_gn: C_in");
    }
    return check_launch("conv_first_kernel<FUSE>");
}

int conv_first(const float* x, const float* wp, const float* bias, float* out, float* gn_partials, int B, int H, int W, int cin, int N,
               int groups, int64_t* counter, int64_t* t_cur, hipStream_t st) {
    DDK_REQUIRE(x && wp && out && B > 0, "conv_first: null pointer / B");
    DDK_REQUIRE(conv_first_ok(cin, N, H, W, groups), "conv_first: needs 1 <= C_in <= 8, N % 32 == 0, N <= 256, H*W % 128 == 0, "
                "N / groups a multiple of 4");
    DDK_REQUIRE(aligned16(wp) && aligned16(out) && aligned16(bias) && (reinterpret_cast<uintptr_t>(gn_partials) & 7u) == 0 &&
                    (reinterpret_cast<uintptr_t>(x) & 3u) == 0, "conv_first: alignment");
    DDK_REQUIRE((long long)B * H * W * 8 < (1ll << 31), "conv_first: B*H*W too large for 32-bit pixel offsets");
    DDK_REQUIRE((counter == nullptr) == (t_cur == nullptr), "conv_first: counter and t_cur go together");
    FirstParams p{};
    p.x = x; p.wp = wp; p.bias = bias; p.out = out; p.gn_part = reinterpret_cast<float2*>(gn_partials);
    p.H = H; p.W = W; p.HW = H * W; p.N = N; p.NB = N / 32;
    p.groups = groups; p.cpg = N / groups;
    p.counter = counter; p.t_cur = t_cur; p.B = B;
    const dim3 grid((unsigned)((long long)B * H * W / 128));
    const size_t lds = first_lds_bytes(cin, N, W);
    switch (cin) {
#define X(C) case C: hipLaunchKernelGGL(conv_first_kernel<C>, grid, dim3(512), lds, st, p); break;
        FIRST_CASES(X)
#undef X
        default: return fail_arg("conv_first: C_in");
    }
    return check_launch("conv_first_kernel");
}

}  // namespace ddk

using namespace ddk;

extern "C" int ddk_pack_conv_weight_first(const float* w_oihw, float* dst, int O, int I, ddk_stream_t s) {
    DDK_REQUIRE(w_oihw && dst && O > 0 && O % 32 == 0 && I >= 1 && I <= 8, "pack_conv_weight_first: arguments (O % 32 == 0, 1 <= I <= 8)");
    const int K2 = (9 * I + 1) / 2;
    const long long total = (long long)(O / 32) * K2 * 64;
    const long long blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(pack_conv_weight_first_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, as_stream(s), w_oihw, dst,
                       O, I, K2, total);
    return check_launch("pack_conv_weight_first_kernel");
}

extern "C" int ddk_conv_first_ok(int cin, int N, int H, int W, int groups) { return conv_first_ok(cin, N, H, W, groups) ? 1 : 0; }

extern "C" int ddk_conv_first(const float* x, const float* w_first, const float* bias, float* out, float* gn_partials, int B, int H, int W,
                              int cin, int N, int groups, ddk_stream_t s) {
    DDK_TRY(ensure_device_init());
    return conv_first(x, w_first, bias, out, gn_partials, B, H, W, cin, N, groups, nullptr, nullptr, as_stream(s));
}
