// backward.hip -- backward kernels of the non-conv ops on the training path, plus train-mode forward variants.
//
// Reference: the autograd graph torch builds for models/unet/blocks.py (GroupNorm+Mish+shift+Dropout, channel
// LayerNorm, LinearAttention, time MLP), models/unet/unet.py:71 (final 1x1), models/diffusion/ddpm.py:275-288 (loss),
// models/downsampled/convblocks.py:110-130 (Mish, avg_pool2d, nearest x2), dddpm.py:99,110 (tanh), driven by
// trainers/trainer_ddpm.py:124-128 (objective.backward()).
//
// Per-channel parameter gradients are produced as per-workgroup partial rows ([parts][C]) in a workspace and folded
// by ddk_rows_sum_accum in a fixed order: deterministic, and they ACCUMULATE into the gradient tensors so the
// reference's 2-micro-batch accumulation needs no extra pass.
#include "ddk_internal.h"

namespace ddk {

// Philox-based keep mask for Dropout(p): one 32-bit draw per element (4 per call)
struct U4b { uint32_t x, y, z, w; };
__device__ __forceinline__ U4b philox_u4(unsigned long long idx4, uint32_t a, uint32_t b, uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    U4b c{(uint32_t)idx4, (uint32_t)(idx4 >> 32), a, b};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4b{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    }
    return c;
}
// Device-resident epoch mixed into every dropout key.  Eager training passes a fresh host seed per forward and leaves
// the epoch at 0; a captured training step (trainers/graph_step.py) has its seed baked into the graph, so it bumps the
// epoch with a one-thread kernel at the start of each forward instead -- replays then draw fresh masks, and the backward
// of the same forward (same epoch) regenerates the identical mask.
__device__ unsigned long long g_drop_epoch = 0;
__global__ void drop_epoch_kernel(unsigned long long set_to, int bump) {
    if (bump) g_drop_epoch += 1;
    else g_drop_epoch = set_to;
}

__device__ __forceinline__ float4 dropout_scale4(long long elem4, float p, uint64_t seed, uint32_t layer) {
    // keep with probability 1-p, scale kept values by 1/(1-p)  (nn.Dropout semantics)
    seed += g_drop_epoch * 0x9E3779B97F4A7C15ull;
    const U4b r = philox_u4((unsigned long long)elem4, layer, 0x44524F50u /* 'DROP' */, seed);
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    const float s = 1.0f / (1.0f - p);
    return make_float4(r.x >= thr ? s : 0.f, r.y >= thr ? s : 0.f, r.z >= thr ? s : 0.f, r.w >= thr ? s : 0.f);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm+Mish(+temb)(+dropout)(+addend): train-mode forward and backward, register-resident per (b, group).
//   u = gn(x) * gamma + beta;  y = drop(mish(u) + temb) + addend
// backward (x recomputed from the saved conv output, nothing else is stored):
//   g1 = dy * dropmask;  dtemb[b][c] = sum_hw g1;  du = g1 * mish'(u);  dbeta += sum du;  dgamma += sum du * xhat
//   dx = rstd * (du*gamma - mean_g(du*gamma) - xhat * mean_g(du*gamma*xhat))
// v += slab 1 + slab 2 + ... (in that order) of the float4 at p; four loads are in flight at a time (a (sample, group) workgroup has
// little else to hide their latency behind)
__device__ __forceinline__ void add_slabs(float4& v, const float* __restrict__ p, int nslab, long long slab_stride) {
    for (int sl = 1; sl < nslab; sl += 4) {
        float4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = sl + j < nslab ? sl + j : nslab - 1;
            t[j] = *reinterpret_cast<const float4*>(p + q * slab_stride);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (sl + j < nslab) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
    }
}

template <int VPT, int NT, bool BWD>
__global__ __launch_bounds__(NT) void gn_train_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ temb,
                                                      int temb_stride, const float* __restrict__ addend, float drop_p,
                                                      uint64_t seed, uint32_t layer, const float* __restrict__ dy,
                                                      float* __restrict__ out /* fwd: y; bwd: dx */,
                                                      float* __restrict__ part /* bwd: [4][B][C] dtemb, dgamma, dbeta, sum dx */, int B, int HW,
                                                      int C, int groups, float eps, int nslab, long long slab_stride,
                                                      const float* __restrict__ conv_bias, float* __restrict__ raw_out) {
    // nslab > 1: the tensor this kernel reads first (fwd: x, bwd: dy) still lies as `nslab` split-K partial slabs of the conv that
    // produced it; they are summed in slab order while loading (the order of splitk_reduce_kernel), fwd adds conv_bias and also
    // writes the sum to raw_out -- the pre-normalisation tensor the backward reads.
    __shared__ float red[32];
    __shared__ float csum[16][8][16];  // [wave][cu][4 ch x {dtemb, dgamma, dbeta, dx}]
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const int upr = cpg >> 2;
    const int units = HW * upr;
    const long long base = (long long)b * HW * C + g * cpg;

    float4 v[VPT];
    float4 gin[BWD ? VPT : 1];       // backward: dy, requested together with x (one memory round trip, not two)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u < units) {
            const int row = div_upr(u, upr), cu = u - row * upr;
            const long long o = base + (long long)row * C + cu * 4;
            v[i] = *reinterpret_cast<const float4*>(x + o);
            if (BWD) {
                gin[i] = *reinterpret_cast<const float4*>(dy + o);
                if (nslab > 1) add_slabs(gin[i], dy + o, nslab, slab_stride);
            }
            if (!BWD && nslab > 1) {
                add_slabs(v[i], x + o, nslab, slab_stride);
                if (conv_bias) {
                    const float4 cb = *reinterpret_cast<const float4*>(conv_bias + g * cpg + cu * 4);
                    v[i].x += cb.x; v[i].y += cb.y; v[i].z += cb.z; v[i].w += cb.w;
                }
                *reinterpret_cast<float4*>(raw_out + o) = v[i];
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float inv_n = 1.0f / (float)(HW * cpg);
    const float mean = block_sum(s, red) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u < units) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float var = block_sum(q, red) * inv_n;
    const float rstd = 1.0f / sqrtf(var + eps);
    const int cu_t = threadIdx.x - div_upr((int)threadIdx.x, upr) * upr;  // NT % upr == 0: every unit of this thread has the same channel quad
    const int c0 = g * cpg + cu_t * 4;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
    const float4 be = *reinterpret_cast<const float4*>(beta + c0);

    if (!BWD) {
        float4 tb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (temb) tb = *reinterpret_cast<const float4*>(temb + (long long)b * temb_stride + c0);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int u = threadIdx.x + i * NT;
            if (u < units) {
                const int row = div_upr(u, upr);
                const long long o = base + (long long)row * C + cu_t * 4;
                float4 y;
                y.x = mish_f((v[i].x - mean) * rstd * ga.x + be.x) + tb.x;
                y.y = mish_f((v[i].y - mean) * rstd * ga.y + be.y) + tb.y;
                y.z = mish_f((v[i].z - mean) * rstd * ga.z + be.z) + tb.z;
                y.w = mish_f((v[i].w - mean) * rstd * ga.w + be.w) + tb.w;
                if (drop_p > 0.f) {
                    const float4 m = dropout_scale4(o >> 2, drop_p, seed, layer);
                    y.x *= m.x; y.y *= m.y; y.z *= m.z; y.w *= m.w;
                }
                if (addend) {
                    const float4 r = *reinterpret_cast<const float4*>(addend + o);
                    y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
                }
                *reinterpret_cast<float4*>(out + o) = y;
            }
        }
        return;
    }

    // ---- backward
    float4 du[VPT];
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f), sg = st, sb = st;  // per-channel sums: dtemb, dgamma, dbeta
    float s1 = 0.f, s2 = 0.f;                                        // group sums of du*gamma and du*gamma*xhat
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        du[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < units) {
            const int row = div_upr(u, upr);
            const long long o = base + (long long)row * C + cu_t * 4;
            float4 g1 = gin[BWD ? i : 0];
            if (drop_p > 0.f) {
                const float4 m = dropout_scale4(o >> 2, drop_p, seed, layer);
                g1.x *= m.x; g1.y *= m.y; g1.z *= m.z; g1.w *= m.w;
            }
            st.x += g1.x; st.y += g1.y; st.z += g1.z; st.w += g1.w;
            const float xh0 = (v[i].x - mean) * rstd, xh1 = (v[i].y - mean) * rstd, xh2 = (v[i].z - mean) * rstd,
                        xh3 = (v[i].w - mean) * rstd;
            float4 d;
            d.x = g1.x * mish_grad_f(xh0 * ga.x + be.x);
            d.y = g1.y * mish_grad_f(xh1 * ga.y + be.y);
            d.z = g1.z * mish_grad_f(xh2 * ga.z + be.z);
            d.w = g1.w * mish_grad_f(xh3 * ga.w + be.w);
            sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
            sg.x += d.x * xh0; sg.y += d.y * xh1; sg.z += d.z * xh2; sg.w += d.w * xh3;
            d.x *= ga.x; d.y *= ga.y; d.z *= ga.z; d.w *= ga.w;  // d xhat
            s1 += (d.x + d.y) + (d.z + d.w);
            s2 += (d.x * xh0 + d.y * xh1) + (d.z * xh2 + d.w * xh3);
            du[i] = d;
            v[i] = make_float4(xh0, xh1, xh2, xh3);
        }
    }
    const float m1 = block_sum(s1, red) * inv_n;
    const float m2 = block_sum(s2, red) * inv_n;
    float4 sx = make_float4(0.f, 0.f, 0.f, 0.f);                     // per-channel sum of dx: bias gradient of the producing conv
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u < units) {
            const int row = div_upr(u, upr);
            const long long o = base + (long long)row * C + cu_t * 4;
            float4 r;
            r.x = rstd * (du[i].x - m1 - v[i].x * m2);
            r.y = rstd * (du[i].y - m1 - v[i].y * m2);
            r.z = rstd * (du[i].z - m1 - v[i].z * m2);
            r.w = rstd * (du[i].w - m1 - v[i].w * m2);
            sx.x += r.x; sx.y += r.y; sx.z += r.z; sx.w += r.w;
            *reinterpret_cast<float4*>(out + o) = r;
        }
    }
    // per-channel sums: lanes with equal (lane % upr) hold the same channel quad -> xor-reduce over the others
    float vals[16] = {st.x, st.y, st.z, st.w, sg.x, sg.y, sg.z, sg.w, sb.x, sb.y, sb.z, sb.w, sx.x, sx.y, sx.z, sx.w};
#pragma unroll
    for (int k = 0; k < 16; ++k)
        for (int o = upr; o < 64; o <<= 1) vals[k] += __shfl_xor(vals[k], o, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane < upr)
#pragma unroll
        for (int k = 0; k < 16; ++k) csum[wid][lane][k] = vals[k];
    __syncthreads();
    if (threadIdx.x < upr * 16) {
        const int cu = threadIdx.x / 16, k = threadIdx.x % 16;
        float t = 0.f;
        for (int w = 0; w < NT / 64; ++w) t += csum[w][cu][k];
        const int which = k >> 2, ch = g * cpg + cu * 4 + (k & 3);
        part[((long long)which * B + b) * C + ch] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Large group slabs (full-resolution DDPM: 256x256 x 16 channels = 1M elements per (b, group)): the same arithmetic
// streamed from memory by `ns` workgroups per (b, group).
//   stats:   gn_stats_partials (Welford partials, norm_act.hip) -> gn_big_finalize_kernel -> (mean, rstd) per (b, group)
//   forward: gn_big_fwd_kernel, one elementwise pass
//   backward: pass A accumulates, per split, the group sums of du*gamma and du*gamma*xhat and the per-channel
//            (dtemb, dgamma, dbeta) rows; gn_big_finalize2 adds the splits in order; pass B recomputes du and writes dx.
__global__ void gn_big_finalize_kernel(const float* __restrict__ part, int ns, float* __restrict__ stat, int n_bg, float eps) {
    const int bg = blockIdx.x * blockDim.x + threadIdx.x;
    if (bg >= n_bg) return;
    const float* pp = part + (long long)bg * ns * 3;
    float n = pp[0], mean = pp[1], m2 = pp[2];   // Chan et al. combination, fixed order
    for (int s = 1; s < ns; ++s) {
        const float nb = pp[3 * s], mb = pp[3 * s + 1], qb = pp[3 * s + 2];
        if (nb > 0) {
            const float tot = n + nb, delta = mb - mean;
            mean += delta * (nb / tot);
            m2 += qb + delta * delta * (n * nb / tot);
            n = tot;
        }
    }
    stat[2 * bg] = mean;
    stat[2 * bg + 1] = 1.0f / sqrtf(m2 / n + eps);
}

__global__ __launch_bounds__(256) void gn_big_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ temb, int temb_stride, const float* __restrict__ addend,
                                                         float drop_p, uint64_t seed, uint32_t layer, float* __restrict__ out, int HW, int C,
                                                         int groups, long long total4) {
    const int c4 = C >> 2, cpg = C / groups;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        const long long pix = i / c4;
        const int b = (int)(pix / HW);
        const int c0 = cq * 4, g = c0 / cpg;
        const float mean = stat[2 * (b * groups + g)], rstd = stat[2 * (b * groups + g) + 1];
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
        const float4 be = *reinterpret_cast<const float4*>(beta + c0);
        float4 y;
        y.x = mish_f((v.x - mean) * rstd * ga.x + be.x);
        y.y = mish_f((v.y - mean) * rstd * ga.y + be.y);
        y.z = mish_f((v.z - mean) * rstd * ga.z + be.z);
        y.w = mish_f((v.w - mean) * rstd * ga.w + be.w);
        if (temb) {
            const float4 t = *reinterpret_cast<const float4*>(temb + (long long)b * temb_stride + c0);
            y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
        }
        if (drop_p > 0.f) {
            const float4 m = dropout_scale4(i, drop_p, seed, layer);
            y.x *= m.x; y.y *= m.y; y.z *= m.z; y.w *= m.w;
        }
        if (addend) {
            const float4 r = reinterpret_cast<const float4*>(addend)[i];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        reinterpret_cast<float4*>(out)[i] = y;
    }
}

// PASS 0: group sums + per-channel rows of this split.  PASS 1: dx (m1, m2 read from gsum, already summed over splits).
template <int PASS>
__global__ __launch_bounds__(256) void gn_big_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float drop_p,
                                                         uint64_t seed, uint32_t layer, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ gpart /* [bg][ns][2] */,
                                                         const float* __restrict__ gsum /* [bg][2] */,
                                                         float* __restrict__ cpart /* [ns][4][B][C] */, int B, int HW, int C, int groups,
                                                         int ns) {
    __shared__ float red[32];
    __shared__ float csum[4][8][12];
    float4 sx = make_float4(0.f, 0.f, 0.f, 0.f);
    const int bg = blockIdx.x, sp = blockIdx.y;
    const int b = bg / groups, g = bg % groups;
    const int cpg = C / groups, upr = cpg >> 2;
    const long long units = (long long)HW * upr;
    long long per = (units + ns - 1) / ns;
    per = (per + 255) / 256 * 256;               // multiple of 256 (and of upr): a thread keeps one channel quad
    const long long u0 = sp * per, u1 = (u0 + per < units) ? u0 + per : units;
    const long long base = (long long)b * HW * C + g * cpg;
    const float mean = stat[2 * bg], rstd = stat[2 * bg + 1];
    const int cu_t = threadIdx.x - div_upr((int)threadIdx.x, upr) * upr;
    const int c0 = g * cpg + cu_t * 4;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
    const float4 be = *reinterpret_cast<const float4*>(beta + c0);
    const float inv_n = 1.0f / (float)((long long)HW * cpg);
    float m1 = 0.f, m2 = 0.f;
    if (PASS == 1) { m1 = gsum[2 * bg] * inv_n; m2 = gsum[2 * bg + 1] * inv_n; }

    float4 st = make_float4(0.f, 0.f, 0.f, 0.f), sg = st, sb = st;
    float s1 = 0.f, s2 = 0.f;
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        const long long row = div_upr(u, upr);
        const long long o = base + row * C + cu_t * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        float4 g1 = *reinterpret_cast<const float4*>(dy + o);
        if (drop_p > 0.f) {
            const float4 m = dropout_scale4(o >> 2, drop_p, seed, layer);
            g1.x *= m.x; g1.y *= m.y; g1.z *= m.z; g1.w *= m.w;
        }
        const float xh0 = (v.x - mean) * rstd, xh1 = (v.y - mean) * rstd, xh2 = (v.z - mean) * rstd, xh3 = (v.w - mean) * rstd;
        float4 d;
        d.x = g1.x * mish_grad_f(xh0 * ga.x + be.x);
        d.y = g1.y * mish_grad_f(xh1 * ga.y + be.y);
        d.z = g1.z * mish_grad_f(xh2 * ga.z + be.z);
        d.w = g1.w * mish_grad_f(xh3 * ga.w + be.w);
        if (PASS == 0) {
            st.x += g1.x; st.y += g1.y; st.z += g1.z; st.w += g1.w;
            sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
            sg.x += d.x * xh0; sg.y += d.y * xh1; sg.z += d.z * xh2; sg.w += d.w * xh3;
        }
        d.x *= ga.x; d.y *= ga.y; d.z *= ga.z; d.w *= ga.w;
        if (PASS == 0) {
            s1 += (d.x + d.y) + (d.z + d.w);
            s2 += (d.x * xh0 + d.y * xh1) + (d.z * xh2 + d.w * xh3);
        } else {
            float4 r;
            r.x = rstd * (d.x - m1 - xh0 * m2);
            r.y = rstd * (d.y - m1 - xh1 * m2);
            r.z = rstd * (d.z - m1 - xh2 * m2);
            r.w = rstd * (d.w - m1 - xh3 * m2);
            sx.x += r.x; sx.y += r.y; sx.z += r.z; sx.w += r.w;
            *reinterpret_cast<float4*>(dx + o) = r;
        }
    }
    if (PASS == 1) {   // row kind 3 of this split: per-channel sum of dx
        float vx[4] = {sx.x, sx.y, sx.z, sx.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (int o = upr; o < 64; o <<= 1) vx[k] += __shfl_xor(vx[k], o, 64);
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane < upr)
#pragma unroll
            for (int k = 0; k < 4; ++k) csum[wid][lane][k] = vx[k];
        __syncthreads();
        if (threadIdx.x < upr * 4) {
            const int cu = threadIdx.x / 4, k = threadIdx.x % 4;
            const float t = (csum[0][cu][k] + csum[1][cu][k]) + (csum[2][cu][k] + csum[3][cu][k]);
            cpart[(((long long)sp * 4 + 3) * B + b) * C + g * cpg + cu * 4 + k] = t;
        }
        return;
    }
    const float t1 = block_sum(s1, red), t2 = block_sum(s2, red);
    if (threadIdx.x == 0) { gpart[((long long)bg * ns + sp) * 2] = t1; gpart[((long long)bg * ns + sp) * 2 + 1] = t2; }
    float vals[12] = {st.x, st.y, st.z, st.w, sg.x, sg.y, sg.z, sg.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
    for (int k = 0; k < 12; ++k)
        for (int o = upr; o < 64; o <<= 1) vals[k] += __shfl_xor(vals[k], o, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane < upr)
#pragma unroll
        for (int k = 0; k < 12; ++k) csum[wid][lane][k] = vals[k];
    __syncthreads();
    if (threadIdx.x < upr * 12) {
        const int cu = threadIdx.x / 12, k = threadIdx.x % 12;
        const float t = (csum[0][cu][k] + csum[1][cu][k]) + (csum[2][cu][k] + csum[3][cu][k]);
        const int which = k >> 2, ch = g * cpg + cu * 4 + (k & 3);
        cpart[(((long long)sp * 4 + which) * B + b) * C + ch] = t;
    }
}

// gsum[bg][k] = sum_s gpart[bg][s][k];  part[i] = sum_s cpart[s][i]   (fixed order)
__global__ void gn_big_finalize2_kernel(const float* __restrict__ gpart, float* __restrict__ gsum, int n_bg, const float* __restrict__ cpart,
                                        float* __restrict__ part, long long n_c, int ns) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < 2LL * n_bg) {
        const long long bg = i >> 1;
        const int k = (int)(i & 1);
        float t = 0.f;
        for (int s = 0; s < ns; ++s) t += gpart[(bg * ns + s) * 2 + k];
        gsum[i] = t;
    }
    if (i < n_c) {
        float t = 0.f;
        for (int s = 0; s < ns; ++s) t += cpart[(long long)s * n_c + i];
        part[i] = t;
    }
}

// out[k][n] (+)= sum_r rows[k*batch_stride + r*row_stride + n]  (fixed order: 4 interleaved row groups, then the groups)
// One workgroup per 64 columns of one batch entry k; 4 row groups of 64 lanes keep 4 loads in flight per column.
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ rows, int nrows, long long row_stride,
                                                       long long batch_stride, float* __restrict__ out, int n, int accumulate) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const float* base = rows + (long long)blockIdx.y * batch_stride;
    float s = 0.f;
    if (col < n) {
        // 16 loads in flight per thread (the adds stay in row order): 1024 partial rows of a 4-workgroup grid took 22 us with 4
        int r = rg;
        for (; r + 60 < nrows; r += 64) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = base[(long long)(r + 4 * u) * row_stride + col];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += t[u];
        }
        for (; r < nrows; r += 4) s += base[(long long)r * row_stride + col];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && col < n) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        float* o = out + (long long)blockIdx.y * n + col;
        *o = accumulate ? *o + t : t;
    }
}

__global__ __launch_bounds__(256) void multi_add_kernel(const float* __restrict__ src, const long long* __restrict__ table) {
    const long long off = table[blockIdx.y * 3], n = table[blockIdx.y * 3 + 2];
    float* dst = reinterpret_cast<float*>(table[blockIdx.y * 3 + 1]);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] += src[off + i];
}

// the same with one target pointer per batch entry (null = skip): the GroupNorm backward's dgamma / dbeta / conv-bias rows go
// into three different gradient buffers in one launch
struct RowsTargets { float* out[4]; };
__global__ __launch_bounds__(256) void rows_sum_targets_kernel(const float* __restrict__ rows, int nrows, long long row_stride,
                                                               long long batch_stride, RowsTargets tg, int n, int accumulate) {
    __shared__ float red[4][64];
    float* out = tg.out[blockIdx.y];
    if (!out) return;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const float* base = rows + (long long)blockIdx.y * batch_stride;
    float s = 0.f;
    if (col < n) {
#pragma unroll 4
        for (int r = rg; r < nrows; r += 4) s += base[r * row_stride + col];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && col < n) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[col] = accumulate ? out[col] + t : t;
    }
}

// Many such row sums in one launch (ddk_rows_sum_jobs): the GroupNorm / LayerNorm backward kernels of a whole backward pass leave
// their partial rows in buffers of their own and a record each; the records travel as kernel arguments (48 per launch), a block
// finds its job by binary search over the first-block prefix sums and is then one block of rows_sum_targets_kernel with
// accumulate = 1 -- 82 launches of ~4.5 us per cfg3 optimiser step before.
constexpr int RS_JOBS_PER_LAUNCH = 48;
struct RowsJobPack {
    int n, pad;
    ddk_rows_sum_job j[RS_JOBS_PER_LAUNCH];
};
__global__ __launch_bounds__(256) void rows_sum_jobs_kernel(const RowsJobPack pk) {
    __shared__ float red[4][64];
    const long long blk = blockIdx.x;
    int lo = 0, hi = pk.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pk.j[mid].block0 <= blk) lo = mid;
        else hi = mid - 1;
    }
    const ddk_rows_sum_job& j = pk.j[lo];
    const int local = (int)(blk - j.block0), gx = (j.n + 63) >> 6;
    const int bx = local % gx, by = local / gx;
    float* out = j.out[by];
    if (!out) return;
    const int col = bx * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const float* base = j.rows + (long long)by * j.batch_stride;
    float s = 0.f;
    if (col < j.n) {
#pragma unroll 4
        for (int r = rg; r < j.nrows; r += 4) s += base[r * j.row_stride + col];
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && col < j.n) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[col] += t;
    }
}

// ------------------------------------------------------------------------------------------------
// Channel LayerNorm backward.  y = d/s*g + b, d = x - mean, s = sqrt(var) + eps:
//   dx = dyg/s - mean(dyg)/s - d * sum(dyg*d) / (C * sigma * s^2),  dyg = dy*g;  dg = sum_pix dy*d/s;  db = sum_pix dy
// Grid-stride over pixel groups; each lane keeps per-channel partial sums, one partial row per workgroup.
template <int LPP, int VPL>
__global__ __launch_bounds__(256) void chan_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ dy, float* __restrict__ dx,
                                                                 float* __restrict__ part /* [2][gridDim][C] */, long long M, int C,
                                                                 float eps, const float* __restrict__ addend /* optional: dx += addend */) {
    constexpr int PPW = 64 / LPP;
    __shared__ float acc_s[4][64][VPL * 8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane % LPP;
    float4 ga[VPL], sgm[VPL], sbt[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        ga[i] = *reinterpret_cast<const float4*>(g + (sub + i * LPP) * 4);
        sgm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        sbt[i] = sgm[i];
    }
    const long long waves_total = (long long)gridDim.x * 4;
    for (long long wv = blockIdx.x * 4LL + wid;; wv += waves_total) {
        const long long pix0 = wv * PPW;
        if (pix0 >= M) break;
        const long long pix = pix0 + lane / LPP;
        const bool ok = pix < M;
        float4 v[VPL], d4[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i] = ok ? *reinterpret_cast<const float4*>(x + pix * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            d4[i] = ok ? *reinterpret_cast<const float4*>(dy + pix * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)C;
        float q = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            const float a0 = d4[i].x * ga[i].x, a1 = d4[i].y * ga[i].y, a2 = d4[i].z * ga[i].z, a3 = d4[i].w * ga[i].w;
            t1 += (a0 + a1) + (a2 + a3);
            t2 += (a0 * v[i].x + a1 * v[i].y) + (a2 * v[i].z + a3 * v[i].w);
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) {
            q += __shfl_xor(q, o, 64);
            t1 += __shfl_xor(t1, o, 64);
            t2 += __shfl_xor(t2, o, 64);
        }
        const float sigma = sqrtf(q / (float)C), sden = sigma + eps;
        const float inv_s = 1.0f / sden;
        const float k1 = t1 / (float)C * inv_s;
        const float k2 = t2 / ((float)C * sigma * sden * sden);
        if (ok) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                float4 r;
                r.x = d4[i].x * ga[i].x * inv_s - k1 - v[i].x * k2;
                r.y = d4[i].y * ga[i].y * inv_s - k1 - v[i].y * k2;
                r.z = d4[i].z * ga[i].z * inv_s - k1 - v[i].z * k2;
                r.w = d4[i].w * ga[i].w * inv_s - k1 - v[i].w * k2;
                if (addend) {
                    const float4 ad = *reinterpret_cast<const float4*>(addend + pix * C + (sub + i * LPP) * 4);
                    r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
                }
                *reinterpret_cast<float4*>(dx + pix * C + (sub + i * LPP) * 4) = r;
                sgm[i].x += d4[i].x * v[i].x * inv_s; sgm[i].y += d4[i].y * v[i].y * inv_s;
                sgm[i].z += d4[i].z * v[i].z * inv_s; sgm[i].w += d4[i].w * v[i].w * inv_s;
                sbt[i].x += d4[i].x; sbt[i].y += d4[i].y; sbt[i].z += d4[i].z; sbt[i].w += d4[i].w;
            }
        }
    }
    // fold the PPW pixel slots of a wave, then the 4 waves
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float* a = acc_s[wid][lane] + i * 8;
        a[0] = sgm[i].x; a[1] = sgm[i].y; a[2] = sgm[i].z; a[3] = sgm[i].w;
        a[4] = sbt[i].x; a[5] = sbt[i].y; a[6] = sbt[i].z; a[7] = sbt[i].w;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * 2; idx += 256) {
        const int which = idx / C, c = idx % C;
        const int cq = c >> 2, ssub = cq % LPP, i = cq / LPP;
        float t = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int pslot = 0; pslot < PPW; ++pslot) t += acc_s[w][pslot * LPP + ssub][i * 8 + which * 4 + (c & 3)];
        part[((long long)which * gridDim.x + blockIdx.x) * C + c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Linear attention backward.  Forward: ks = softmax_n(k); ctx = ks^T v; out = q ctx.
//   dctx[d][e] = sum_n q[n][d] dout[n][e]                      (linattn_dctx_kernel)
//   dq = dout ctx^T;  dv = ks dctx;  dks = v dctx^T;  dk = ks * (dks - S),  S[d] = sum_e dctx[d][e] ctx[d][e]
// ks is recomputed from k with the forward's column max / sum (stats[b][h][0][d] = max, [1][d] = sum of exp).
constexpr int DHB = 32;

// Both reductions over the pixels are SPLIT over gridDim.y workgroups per (image, head) (one workgroup walking the 65 536 pixels of a
// 256x256 map took 4.6 ms for the statistics and 2.4 ms for dctx, 23 % of the full-resolution training step): workgroup (bh, sp)
// handles pixels [sp * per, (sp + 1) * per) and writes a partial; a fixed-order finish combines them.  gridDim.y == 1 writes the
// final result directly.
//   stats partial [bh][sp][2][32] = (max over the range, sum of exp(k - that max))
// (the body is shared with linattn_dctx_kernel, which computes the statistics itself where one workgroup covers all pixels)
__device__ __forceinline__ void linattn_stats_body(const float* __restrict__ qkv, float* __restrict__ o, int HW, int heads, int b, int h,
                                                   int n_lo, int n_hi, float* sm /* [8 * 32] */);

__global__ __launch_bounds__(256) void linattn_stats_kernel(const float* __restrict__ qkv, float* __restrict__ out, int HW, int heads,
                                                            int per) {
    __shared__ float sm[8 * DHB];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int n_lo = blockIdx.y * per, n_hi = min(HW, n_lo + per);
    linattn_stats_body(qkv, out + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 2 * DHB, HW, heads, b, h, n_lo, n_hi, sm);
}

__device__ __forceinline__ void linattn_stats_body(const float* __restrict__ qkv, float* __restrict__ o, int HW, int heads, int b, int h,
                                                   int n_lo, int n_hi, float* sm) {
    const int HC = heads * DHB, RS = 3 * HC;
    const float* kp = qkv + (long long)b * HW * RS + HC + h * DHB;
    const int d = threadIdx.x & 31, ng = threadIdx.x >> 5;
    float m = -INFINITY;
    for (int n = n_lo + ng; n < n_hi; n += 8) m = fmaxf(m, kp[(long long)n * RS + d]);
    sm[ng * DHB + d] = m;
    __syncthreads();
    float mm = sm[d];
#pragma unroll
    for (int j = 1; j < 8; ++j) mm = fmaxf(mm, sm[j * DHB + d]);
    __syncthreads();
    float s = 0.f;
    for (int n = n_lo + ng; n < n_hi; n += 8) s += expf(kp[(long long)n * RS + d] - mm);
    sm[ng * DHB + d] = s;
    __syncthreads();
    if (threadIdx.x < DHB) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += sm[j * DHB + threadIdx.x];
        o[threadIdx.x] = mm;
        o[DHB + threadIdx.x] = t;
    }
}
// stats[bh] = (M = max_s m_s, sum_s s_s * exp(m_s - M)), splits in index order
__global__ __launch_bounds__(64) void linattn_stats_merge_kernel(const float* __restrict__ part, float* __restrict__ stats, int splits) {
    const int d = threadIdx.x;
    if (d >= DHB) return;
    const float* p = part + (long long)blockIdx.x * splits * 2 * DHB;
    float M = -INFINITY;
    for (int sp = 0; sp < splits; ++sp) M = fmaxf(M, p[sp * 2 * DHB + d]);
    float t = 0.f;
    for (int sp = 0; sp < splits; ++sp) t += p[sp * 2 * DHB + DHB + d] * expf(p[sp * 2 * DHB + d] - M);
    stats[(long long)blockIdx.x * 2 * DHB + d] = M;
    stats[(long long)blockIdx.x * 2 * DHB + DHB + d] = t;
}

// dctx[b][h][d][e] = sum_n q[n][h*32+d] * dout[n][h*32+e]; partial [sp][bh][32][32] when gridDim.y > 1
// stats_out != null (gridDim.y == 1 only): this workgroup also leaves the softmax statistics of k for its (image, head) -- the backward
// then needs nothing from the forward but qkv and ctx, and the forward no statistics launch
__global__ __launch_bounds__(256) void linattn_dctx_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           float* __restrict__ out, int HW, int heads, int per,
                                                           float* __restrict__ stats_out) {
    __shared__ __attribute__((aligned(16))) float qs[64 * DHB];
    __shared__ __attribute__((aligned(16))) float ds[64 * DHB];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    if (stats_out) {
        linattn_stats_body(qkv, stats_out + (long long)blockIdx.x * 2 * DHB, HW, heads, b, h, 0, HW, qs);
        __syncthreads();
    }
    const int HC = heads * DHB, RS = 3 * HC, tid = threadIdx.x;
    const float* qp = qkv + (long long)b * HW * RS + h * DHB;
    const float* dp = dout + (long long)b * HW * HC + h * DHB;
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    const int n_lo = blockIdx.y * per, n_hi = min(HW, n_lo + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx4 = tid + j * 256, row = idx4 >> 3, c = (idx4 & 7) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
            if (n0 + row < n_hi) {
                a = *reinterpret_cast<const float4*>(qp + (long long)(n0 + row) * RS + c);
                bb = *reinterpret_cast<const float4*>(dp + (long long)(n0 + row) * HC + c);
            }
            *reinterpret_cast<float4*>(qs + row * DHB + c) = a;
            *reinterpret_cast<float4*>(ds + row * DHB + c) = bb;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < 64; ++n) {
            const float qd = qs[n * DHB + d];
            const float4 v4 = *reinterpret_cast<const float4*>(ds + n * DHB + e0);
            acc.x += qd * v4.x; acc.y += qd * v4.y; acc.z += qd * v4.z; acc.w += qd * v4.w;
        }
        __syncthreads();
    }
    *reinterpret_cast<float4*>(out + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * DHB + d) * DHB + e0) = acc;
}

// per (pixel, head): dq, dk, dv -> dqkv [B][HW][3*heads*32].  A workgroup takes 16 pixels of one sample; 8 lanes share a (pixel, head):
// lane j of them owns rows d = j + 8r (r < 4) of ctx / dctx in pass 1 and columns e = 4j..4j+3 of dctx in pass 2 (one thread per
// (pixel, head) walked 3 x 32 x 32 products alone: 24-40 us on the 64-workgroup grids of the small maps at batch 64).  Rows are pitched
// 36 floats in LDS so that the 8 lanes' float4 reads of 8 different rows fall into different banks.  Same sums in the same order.
__global__ __launch_bounds__(256) void linattn_bwd_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                                const float* __restrict__ stats, float* __restrict__ dqkv, int HW,
                                                                int heads, int tiles_per_sample) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int RP = DHB + 4, HP = DHB * RP;
    float* cs = sm;                      // ctx   [heads][32][RP]
    float* dcs = sm + heads * HP;        // dctx  [heads][32][RP]
    float* S = sm + 2 * heads * HP;      // [heads][32]
    const int b = blockIdx.x / tiles_per_sample, tile = blockIdx.x % tiles_per_sample;
    const int HC = heads * DHB, RS = 3 * HC, nthreads = 64 * heads;
    for (int i = threadIdx.x; i < heads * DHB * DHB / 4; i += nthreads) {
        const int hh = (i * 4) / (DHB * DHB), r = (i * 4) % (DHB * DHB);
        const int o = hh * HP + (r / DHB) * RP + (r % DHB);
        *reinterpret_cast<float4*>(cs + o) = *reinterpret_cast<const float4*>(ctx + ((long long)b * heads + hh) * DHB * DHB + r);
        *reinterpret_cast<float4*>(dcs + o) = *reinterpret_cast<const float4*>(dctx + ((long long)b * heads + hh) * DHB * DHB + r);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < heads * DHB; i += nthreads) {
        const int hh = i / DHB, d = i % DHB;
        float t = 0.f;
        for (int e = 0; e < DHB; ++e) t += dcs[hh * HP + d * RP + e] * cs[hh * HP + d * RP + e];
        S[i] = t;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = threadIdx.x & 7, h = (threadIdx.x >> 3) % heads;
    const float* st = stats + ((long long)b * heads + h) * 2 * DHB;
    const float* ch = cs + h * HP;
    const float* dch = dcs + h * HP;
    for (int pl = threadIdx.x / (8 * heads); pl < 16; pl += 8) {
        const int n = tile * 16 + pl;
        if (n >= HW) break;
        const float* base = qkv + ((long long)b * HW + n) * RS;
        const float* dop = dout + ((long long)b * HW + n) * HC + h * DHB;
        float* outp = dqkv + ((long long)b * HW + n) * RS;
        float dO[DHB], vv[DHB], ksl[4];
#pragma unroll
        for (int i = 0; i < DHB / 4; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(dop + i * 4);
            const float4 v4 = *reinterpret_cast<const float4*>(base + 2 * HC + h * DHB + i * 4);
            dO[4 * i] = a.x; dO[4 * i + 1] = a.y; dO[4 * i + 2] = a.z; dO[4 * i + 3] = a.w;
            vv[4 * i] = v4.x; vv[4 * i + 1] = v4.y; vv[4 * i + 2] = v4.z; vv[4 * i + 3] = v4.w;
        }
        // pass 1 (rows d): dq[d] = dout . ctx[d][:],  dks[d] = v . dctx[d][:],  dk[d] = ks[d] (dks[d] - S[d])
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = j + 8 * r;
            float dq = 0.f, dks = 0.f;
            ksl[r] = expf(base[HC + h * DHB + d] - st[d]) / st[DHB + d];
#pragma unroll
            for (int i = 0; i < DHB / 4; ++i) {
                const float4 c4 = *reinterpret_cast<const float4*>(ch + d * RP + i * 4);
                const float4 g4 = *reinterpret_cast<const float4*>(dch + d * RP + i * 4);
                dq += (dO[4 * i] * c4.x + dO[4 * i + 1] * c4.y) + (dO[4 * i + 2] * c4.z + dO[4 * i + 3] * c4.w);
                dks += (vv[4 * i] * g4.x + vv[4 * i + 1] * g4.y) + (vv[4 * i + 2] * g4.z + vv[4 * i + 3] * g4.w);
            }
            outp[h * DHB + d] = dq;
            outp[HC + h * DHB + d] = ksl[r] * (dks - S[h * DHB + d]);
        }
        // pass 2 (columns e = 4j..4j+3): dv[e] = sum_d ks[d] dctx[d][e]; ks[d] lives in lane (d & 7) of this lane's group of 8
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int d = 0; d < DHB; ++d) {
            const float kd = __shfl(ksl[d >> 3], (lane & ~7) | (d & 7), 64);
            const float4 g4 = *reinterpret_cast<const float4*>(dch + d * RP + j * 4);
            acc.x += kd * g4.x; acc.y += kd * g4.y; acc.z += kd * g4.z; acc.w += kd * g4.w;
        }
        *reinterpret_cast<float4*>(outp + 2 * HC + h * DHB + j * 4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ elementwise backward
template <int OP>  // 0: dx = dy * mish'(x)   1: dx = dy * (1 - y^2) (y = tanh output)   2: out = a * s (scale)
__global__ __launch_bounds__(256) void unary_bwd_kernel(const float* __restrict__ xy, const float* __restrict__ dy, float* __restrict__ dx,
                                                        long long n4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(xy)[i], g = reinterpret_cast<const float4*>(dy)[i];
        float4 r;
        if (OP == 0) { r.x = g.x * mish_grad_f(a.x); r.y = g.y * mish_grad_f(a.y); r.z = g.z * mish_grad_f(a.z); r.w = g.w * mish_grad_f(a.w); }
        else { r.x = g.x * (1.f - a.x * a.x); r.y = g.y * (1.f - a.y * a.y); r.z = g.z * (1.f - a.z * a.z); r.w = g.w * (1.f - a.w * a.w); }
        reinterpret_cast<float4*>(dx)[i] = r;
    }
}

// avg_pool2d(2) backward: dx[b][y][x][c] = 0.25 * dy[b][y/2][x/2][c];   nearest x2 backward: dx[b][y][x] = sum of the 2x2 block of dy
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                                           long long total4) {
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        long long p = i / c4;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const long long b = p / H;
        const float4 g = *reinterpret_cast<const float4*>(dy + ((b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + cq * 4);
        reinterpret_cast<float4*>(dx)[i] = make_float4(0.25f * g.x, 0.25f * g.y, 0.25f * g.z, 0.25f * g.w);
    }
}
__global__ __launch_bounds__(256) void upnearest2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                                             long long total4) {
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        long long p = i / c4;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const long long b = p / H;
        const float* s = dy + ((b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + cq * 4;
        const float4 a = *reinterpret_cast<const float4*>(s), bb = *reinterpret_cast<const float4*>(s + C);
        const float4 c = *reinterpret_cast<const float4*>(s + 2LL * W * C), d = *reinterpret_cast<const float4*>(s + 2LL * W * C + C);
        reinterpret_cast<float4*>(dx)[i] = make_float4((a.x + bb.x) + (c.x + d.x), (a.y + bb.y) + (c.y + d.y), (a.z + bb.z) + (c.z + d.z),
                                                       (a.w + bb.w) + (c.w + d.w));
    }
}

// loss backward: d eps_hat = -2 (eps - eps_hat) * scale[b]
__global__ __launch_bounds__(256) void sq_err_grad_kernel(const float* __restrict__ a, const float* __restrict__ bh, const float* __restrict__ scale,
                                                          float* __restrict__ out, long long per4, long long total4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float sc = -2.0f * scale[i / per4];
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(bh)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(sc * (u.x - v.x), sc * (u.y - v.y), sc * (u.z - v.z), sc * (u.w - v.w));
    }
}

// objective of the dDDPM autoencoder from the per-sample losses (see ddk_ae_objective)
__global__ __launch_bounds__(256) void ae_objective_kernel(const float* __restrict__ l_ddpm, const float* __restrict__ l_rec,
                                                           const int64_t* __restrict__ t, int t_rec_max, int B, float* __restrict__ out) {
    __shared__ float red[32];
    float a = 0.f, r = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        a += l_ddpm[b];
        r += t[b] < t_rec_max ? l_rec[b] : 0.f;
    }
    const float sa = block_sum(a, red), sr = block_sum(r, red);
    if (threadIdx.x == 0) {
        const float latent = sa / (float)B, recon = sr / (float)B;
        out[0] = latent + recon;
        out[1] = latent;
        out[2] = recon;
    }
}
__global__ __launch_bounds__(256) void ae_objective_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ t, int t_rec_max, int B,
                                                               float* __restrict__ d_ddpm, float* __restrict__ d_rec) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float v = g[0] / (float)B;
    d_ddpm[b] = v;
    d_rec[b] = t[b] < t_rec_max ? v : 0.f;
}

// out[i] = x[i] * scale[i / per]   (q_sample backward wrt x: sqrt_acp[t_b] * dy; also generic per-sample scaling)
__global__ __launch_bounds__(256) void scale_per_sample_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               float* __restrict__ out, long long per4, long long total4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float sc = scale[i / per4];
        const float4 u = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(sc * u.x, sc * u.y, sc * u.z, sc * u.w);
    }
}

// ------------------------------------------------------------------------------------------------
// small-N 1x1 conv backward (final_conv.1): da[m][c] = sum_co dy[m][co] w[co][c];  dw[co][c] += sum_m dy[m][co] a[m][c];
// db[co] += sum_m dy[m][co].  Partial rows per workgroup: part[blk][n_out*C + n_out].
template <int LPP, int VPL, int NOUT_MAX>
__global__ __launch_bounds__(256) void conv1x1_small_n_bwd_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                                  const float* __restrict__ dy, float* __restrict__ da,
                                                                  float* __restrict__ part, long long M, int C, int n_out) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane % LPP;
    // cb: the LPP * VPL * 4 channels of this trip.  Every listed width is one trip; other multiples of 32 (padded generic widths:
    // 224, 288, 352, ...) walk the pixels once per 32 channels -- the products are per channel, so the result does not depend on it
    for (int cb = 0; cb < C; cb += LPP * VPL * 4) {
    float4 dw[NOUT_MAX][VPL];
    float dbv[NOUT_MAX];
#pragma unroll
    for (int co = 0; co < NOUT_MAX; ++co) {
        dbv[co] = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) dw[co][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long waves_total = (long long)gridDim.x * 4;
    // four pixel groups per trip, their loads issued together: a wave walks 64 groups of the 64x64x64-pixel decoder output with nothing
    // else on its SIMD to hide a load behind (70 us with one group per trip).  Same accumulation order.
    constexpr int UNR = 4;
    for (long long wv0 = blockIdx.x * 4LL + wid;; wv0 += waves_total * UNR) {
        if (wv0 * PPW >= M) break;
        float4 av[UNR][VPL];
        float g[UNR][NOUT_MAX];
        long long pix[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            pix[u] = (wv0 + u * waves_total) * PPW + lane / LPP;
            const bool ok = pix[u] < M;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                av[u][i] = ok ? *reinterpret_cast<const float4*>(a + pix[u] * C + cb + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int co = 0; co < NOUT_MAX; ++co) g[u][co] = (ok && co < n_out) ? dy[pix[u] * n_out + co] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float4 r[VPL];
#pragma unroll
            for (int i = 0; i < VPL; ++i) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int co = 0; co < NOUT_MAX; ++co) {
                if (co < n_out) {
                    const float gg = g[u][co];
                    if (sub == 0) dbv[co] += gg;
#pragma unroll
                    for (int i = 0; i < VPL; ++i) {
                        const float4 ww = *reinterpret_cast<const float4*>(w + (long long)co * C + cb + (sub + i * LPP) * 4);
                        r[i].x += gg * ww.x; r[i].y += gg * ww.y; r[i].z += gg * ww.z; r[i].w += gg * ww.w;
                        dw[co][i].x += gg * av[u][i].x; dw[co][i].y += gg * av[u][i].y; dw[co][i].z += gg * av[u][i].z;
                        dw[co][i].w += gg * av[u][i].w;
                    }
                }
            }
            if (pix[u] < M)
#pragma unroll
                for (int i = 0; i < VPL; ++i) *reinterpret_cast<float4*>(da + pix[u] * C + cb + (sub + i * LPP) * 4) = r[i];
        }
    }
    // fold the PPW pixel slots of the wave by xor shuffles over lane bits >= log2(LPP); one partial row per wave
    const int row = blockIdx.x * 4 + wid;
    float* prow = part + (long long)row * (n_out * C + n_out);
#pragma unroll
    for (int co = 0; co < NOUT_MAX; ++co) {
        if (co < n_out) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                float4 t = dw[co][i];
                for (int o = LPP; o < 64; o <<= 1) {
                    t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
                }
                if (lane < LPP) *reinterpret_cast<float4*>(prow + (long long)co * C + cb + (sub + i * LPP) * 4) = t;
            }
            float tb = dbv[co];
            for (int o = 1; o < 64; o <<= 1) tb += __shfl_xor(tb, o, 64);
            if (lane == 0 && cb == 0) prow[(long long)n_out * C + co] = tb;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Small dense products of the time-embedding path:  C = op(A) op(B), 32x32 output tile per workgroup, k in chunks of 32
// through LDS (coalesced loads for every mode), each of the 256 threads owns a 1x4 strip.
//   mode 0: C[M][N]  = A[M][K]   B[K][N]            mode 1: C[M][N] = A[M][K] B[N][K]^T
//   mode 2: C[M][N] (+)= A[K][M]^T B[K][N]   (accumulating, for weight gradients)
// gridDim.z > 1: the contraction is split, workgroup z takes k in [z * kper, (z + 1) * kper) and writes its partial tile to slab z
// of Cm ([z][M][ldc], not accumulating); the host sums the slabs in order (a long-K product on a 2 x 16 grid of tiles took 380 us).
typedef float sg_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void small_gemm_kernel(int mode, const float* __restrict__ A, const float* __restrict__ Bm,
                                                         float* __restrict__ Cm, int M, int N, int K, int lda, int ldb, int ldc,
                                                         int accumulate, int kper) {
    // Round 5: the products run on the matrix pipe.  The VALU form read 160 LDS words per thread and 32-chunk for 128 FMAs (LDS-bound:
    // 8 launches of ~40 us in the merged cfg3 step); here wave w multiplies the k rows [8 w, 8 w + 8) of every staged chunk with four
    // v_mfma_f32_32x32x2_f32 (8 LDS words per lane and chunk) and the four waves' partial tiles meet in LDS once, at the end, in wave
    // order -- deterministic.
    __shared__ float As[32][33];   // [k][m]
    __shared__ float Bs[32][33];   // [k][n]
    __shared__ float Pt[4 * 32 * 33];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int k_lo = blockIdx.z * kper;
    if (gridDim.z > 1) { Cm += (long long)blockIdx.z * M * ldc; K = min(K, k_lo + kper); }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // ty: 0..7
    const int mrow = threadIdx.x >> 3, nq = (threadIdx.x & 7) * 4;   // output strip: row mrow, columns nq..nq+3
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, kh = lane >> 5;
    sg_f32x16 macc;
#pragma unroll
    for (int r = 0; r < 16; ++r) macc[r] = 0.f;
    for (int k0 = k_lo; k0 < K; k0 += 32) {
        // stage A as [k][m] and B as [k][n]; the fast thread index follows the operand's contiguous dimension
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ty + r * 8;
            if (mode == 2) {       // A[K][M]: contiguous in m
                const int k = k0 + i, m = m0 + tx;
                As[i][tx] = (k < K && m < M) ? A[(long long)k * lda + m] : 0.f;
            } else {               // A[M][K]: contiguous in k
                const int m = m0 + i, k = k0 + tx;
                As[tx][i] = (k < K && m < M) ? A[(long long)m * lda + k] : 0.f;
            }
            if (mode == 1) {       // B[N][K]: contiguous in k
                const int n = n0 + i, k = k0 + tx;
                Bs[tx][i] = (k < K && n < N) ? Bm[(long long)n * ldb + k] : 0.f;
            } else {               // B[K][N]: contiguous in n
                const int k = k0 + i, n = n0 + tx;
                Bs[i][tx] = (k < K && n < N) ? Bm[(long long)k * ldb + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = wv * 8 + 2 * j + kh;
            macc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][l31], Bs[k][l31], macc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Pt[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = macc[r];
    __syncthreads();
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc[j] = ((Pt[mrow * 33 + nq + j] + Pt[(32 + mrow) * 33 + nq + j]) + Pt[(64 + mrow) * 33 + nq + j]) + Pt[(96 + mrow) * 33 + nq + j];
    const int m = m0 + mrow;
    if (m < M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + nq + j;
            if (n < N) {
                float* c = Cm + (long long)m * ldc + n;
                *c = accumulate ? *c + acc[j] : acc[j];
            }
        }
    }
}

// sinusoidal embedding + pre-activations of the time MLP, saved for its backward: e [B][dim], u1 [B][4dim] (pre-Mish), tv [B][dim]
__global__ __launch_bounds__(256) void sincos_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ e,
                                                     int B, int dim) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i % dim, half = dim >> 1;
    const float a = (float)t[b] * freqs[j < half ? j : j - half];
    e[i] = j < half ? sinf(a) : cosf(a);
}
// y = x + bias[col];  optionally act = mish(y)
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, float* __restrict__ act,
                                                       long long total, int N) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const float v = y[i] + bias[i % N];
    y[i] = v;
    if (act) act[i] = mish_f(v);
}

// ------------------------------------------------------------------------------------------------ optimiser (flat fp32 buffers)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[32];
    float s = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// norm = sqrt(sum part); coef = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ part, int nparts, float max_norm, float* __restrict__ out /* [norm, coef] */) {
    __shared__ float red[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        out[0] = norm;
        out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
}
// torch.optim.Adam (no amsgrad / weight decay) on flat buffers; the clip coefficient is read from device memory
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float b2, float omb1,
                                                   float omb2, float eps, float bc2_sqrt, const float* __restrict__ clip) {
    const float coef = clip ? clip[1] : 1.0f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] + (gi - m[i]) * omb1;            // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + omb2 * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);                // addcdiv_(exp_avg, denom, value=-lr/bc1)
    }
}
// p_ema = p_ema * decay + (1 - decay) * p   (trainers/ema.py:41-44)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pe, const float* __restrict__ p, long long n, float decay) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        pe[i] = pe[i] * decay + (1.0f - decay) * p[i];
}

// ------------------------------------------------------------------------------------------------
// Training at widths that are not multiples of 32 (reference blocks.py:75: GroupNorm(8, C) takes any C % 8 == 0; blocks.py:57-60).
// Activations keep a pitch CP = pad32(C) with zero padding, so every conv kernel -- forward, input gradient, weight gradient -- runs
// unchanged on zero-padded weights; only the two normalisations see the real channel count.  Plain passes over global memory, one
// workgroup per (image, group); the same arithmetic as gn_train_kernel, correctness first.
//   thread = (channel cc = tid % CG, row r = tid / CG) with CG the power of two >= C / groups: a thread's elements all lie in ONE
//   channel, so the per-channel sums of the backward are per-thread registers, folded over the rows in fixed order through LDS.
template <bool BWD>
__global__ __launch_bounds__(256) void gn_generic_train_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ temb, int temb_stride,
                                                               const float* __restrict__ addend, float drop_p, uint64_t seed, uint32_t layer,
                                                               const float* __restrict__ dy, float* __restrict__ out /* fwd: y; bwd: dx */,
                                                               float* __restrict__ part /* bwd: [4][B][C] dtemb, dgamma, dbeta, sum dx */,
                                                               int B, int HW, int CP, int C, int groups, float eps, int CG) {
    __shared__ float red[32];
    __shared__ float chs[256][4];
    const int b = blockIdx.x / (groups + 1), g = blockIdx.x % (groups + 1), tid = threadIdx.x;
    const long long img = (long long)b * HW * CP;
    if (g == groups) {                                   // the padding channels of the output stay exactly zero
        const int np = CP - C;
        for (long long e = tid; e < (long long)HW * np; e += 256) out[img + (e / np) * CP + C + (int)(e % np)] = 0.f;
        return;
    }
    const int cpg = C / groups, c0 = g * cpg;
    const int cc = tid % CG, r0 = tid / CG, R = 256 / CG;
    const bool act = cc < cpg;
    const int c = c0 + (act ? cc : 0);
    const float n = (float)((long long)HW * cpg);
    float s = 0.f;
    if (act) for (int px = r0; px < HW; px += R) s += x[img + (long long)px * CP + c];
    const float mean = block_sum(s, red) / n;
    float q = 0.f;
    if (act) for (int px = r0; px < HW; px += R) { const float d = x[img + (long long)px * CP + c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / n + eps);
    const float ga = gamma[c], be = beta[c];
    auto keep = [&](long long o) -> float {             // the Dropout scale of element o (the mask gn_train_kernel draws for it)
        const float4 m = dropout_scale4(o >> 2, drop_p, seed, layer);
        const int k = (int)(o & 3);
        return k == 0 ? m.x : k == 1 ? m.y : k == 2 ? m.z : m.w;
    };
    if (!BWD) {
        const float tb = temb ? temb[(long long)b * temb_stride + c] : 0.f;
        if (act) for (int px = r0; px < HW; px += R) {
            const long long o = img + (long long)px * CP + c;
            float y = mish_f((x[o] - mean) * rstd * ga + be) + tb;
            if (drop_p > 0.f) y *= keep(o);
            if (addend) y += addend[o];
            out[o] = y;
        }
        return;
    }
    float st = 0.f, sg = 0.f, sb = 0.f, sdx = 0.f, s1 = 0.f, s2 = 0.f;
    if (act) for (int px = r0; px < HW; px += R) {
        const long long o = img + (long long)px * CP + c;
        float g1 = dy[o];
        if (drop_p > 0.f) g1 *= keep(o);
        st += g1;
        const float xh = (x[o] - mean) * rstd;
        const float d = g1 * mish_grad_f(xh * ga + be);
        sb += d;
        sg += d * xh;
        s1 += d * ga;
        s2 += d * ga * xh;
    }
    const float m1 = block_sum(s1, red) / n, m2 = block_sum(s2, red) / n;
    if (act) for (int px = r0; px < HW; px += R) {
        const long long o = img + (long long)px * CP + c;
        float g1 = dy[o];
        if (drop_p > 0.f) g1 *= keep(o);
        const float xh = (x[o] - mean) * rstd;
        const float dxh = g1 * mish_grad_f(xh * ga + be) * ga;
        const float dxv = rstd * (dxh - m1 - xh * m2);
        out[o] = dxv;
        sdx += dxv;
    }
    chs[tid][0] = st; chs[tid][1] = sg; chs[tid][2] = sb; chs[tid][3] = sdx;
    __syncthreads();
    if (tid < cpg) {                                     // fixed-order fold over the R rows of this channel
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < R; ++r) {
            a0 += chs[r * CG + tid][0]; a1 += chs[r * CG + tid][1]; a2 += chs[r * CG + tid][2]; a3 += chs[r * CG + tid][3];
        }
        const long long bc = (long long)b * C + c0 + tid, BC = (long long)B * C;
        part[bc] = a0; part[BC + bc] = a1; part[2 * BC + bc] = a2; part[3 * BC + bc] = a3;
    }
}

// Channel LayerNorm backward over the C real channels of CP-pitched rows (the arithmetic of chan_layernorm_bwd_kernel): one wave
// per pixel, a lane owns channels lane, lane + 64, ... (CP <= 512: unet_create admits no level wider than 512 channels); per-workgroup
// partial rows of dg, db.
template <int NS = 8>
__global__ __launch_bounds__(256) void chan_layernorm_generic_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                         const float* __restrict__ dy, const float* __restrict__ addend,
                                                                         float* __restrict__ dx, float* __restrict__ part /* [2][grid][C] */,
                                                                         long long M, int CP, int C, float eps) {
    __shared__ float acc[4][2][64 * NS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float sg[NS], sb[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) sg[i] = sb[i] = 0.f;
    for (long long pix = blockIdx.x * 4LL + wid; pix < M; pix += (long long)gridDim.x * 4) {
        const float* xr = x + pix * CP;
        const float* dr = dy + pix * CP;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f, t1 = 0.f, t2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = xr[c] - mean, a = dr[c] * g[c];
            q += d * d; t1 += a; t2 += a * d;
        }
        q = wave_sum(q); t1 = wave_sum(t1); t2 = wave_sum(t2);
        const float sigma = sqrtf(q / (float)C), sden = sigma + eps, inv_s = 1.0f / sden;
        const float k1 = t1 / (float)C * inv_s, k2 = t2 / ((float)C * sigma * sden * sden);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = lane + 64 * i;
            if (c < CP) {
                float rv = 0.f;
                if (c < C) {
                    const float d = xr[c] - mean;
                    rv = dr[c] * g[c] * inv_s - k1 - d * k2;
                    sg[i] += dr[c] * d * inv_s;
                    sb[i] += dr[c];
                }
                if (addend) rv += addend[pix * CP + c];      // (the Residual's gradient: zero in the padding like everything else)
                dx[pix * CP + c] = rv;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { acc[wid][0][lane + 64 * i] = sg[i]; acc[wid][1][lane + 64 * i] = sb[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part[(long long)blockIdx.x * C + c] = (acc[0][0][c] + acc[1][0][c]) + (acc[2][0][c] + acc[3][0][c]);
        part[((long long)gridDim.x + blockIdx.x) * C + c] = (acc[0][1][c] + acc[1][1][c]) + (acc[2][1][c] + acc[3][1][c]);
    }
}

static int gn_generic_cg(int cpg) { int cg = 1; while (cg < cpg) cg <<= 1; return cg; }

static int grid_for(long long n) {
    const long long b = ceil_div(n > 0 ? n : 1, 256);
    return (int)(b < 4096 ? b : 4096);
}

}  // namespace ddk

using namespace ddk;

extern "C" {

/* Train-mode forward of GroupNorm+Mish: y = dropout_p(mish(gn(x)) + temb) + addend (blocks.py:106-111). */
int ddk_groupnorm_mish_train_fwd(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                                 const float* addend, float drop_p, uint64_t seed, uint32_t layer, float* out, int B, int HW, int C,
                                 int groups, float eps, void* workspace, size_t workspace_bytes, ddk_stream_t s);
/* Backward of the same: dx, and partial rows part[4][B][C] = (dtemb, dgamma, dbeta, sum_hw dx) per sample. */
int ddk_groupnorm_mish_bwd(const float* x, const float* gamma, const float* beta, float drop_p, uint64_t seed, uint32_t layer,
                           const float* dy, float* dx, float* part, int B, int HW, int C, int groups, float eps, void* workspace,
                           size_t workspace_bytes, ddk_stream_t s);

// Workspace of the large-slab path, in floats: [stats partials 3*ns][stat 2][gpart 2*ns][gsum 2] per (b, group), then
// the per-channel rows [ns][3][B][C].  0 for slabs the register-resident kernel handles.
static size_t gn_train_ws_floats(int B, int HW, int C, int groups, int& ns) {
    ns = gn_train_nsplit(HW, C / groups);
    if (ns == 0) return 0;
    const size_t n_bg = (size_t)B * groups;
    return n_bg * (3 * (size_t)ns + 2 + 2 * (size_t)ns + 2) + (size_t)ns * 4 * B * C;
}

static int gn_train_launch(bool bwd, const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                           const float* addend, float drop_p, uint64_t seed, uint32_t layer, const float* dy, float* out, float* part,
                           int B, int HW, int C, int groups, float eps, void* ws, size_t ws_bytes, hipStream_t st, int nslab = 1,
                           long long slab_stride = 0, const float* conv_bias = nullptr, float* raw_out = nullptr) {
    DDK_REQUIRE(x && gamma && beta && out, "groupnorm_train: null pointer");
    DDK_REQUIRE(nslab >= 1 && nslab <= 256, "groupnorm_train: slab count");
    if (nslab > 1) {
        DDK_REQUIRE(slab_stride >= (long long)B * HW * C && slab_stride % 4 == 0, "groupnorm_train: slab stride");
        DDK_REQUIRE(bwd || (raw_out && aligned16(raw_out)), "groupnorm_train: the slab form of the forward needs raw_out");
        DDK_REQUIRE(!conv_bias || aligned16(conv_bias), "groupnorm_train: conv_bias alignment");
    }
    DDK_REQUIRE(B > 0 && HW > 0 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "groupnorm_train: C/groups % 4");
    DDK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "groupnorm_train: dropout p");
    const int cpg = C / groups, upr = cpg / 4;
    DDK_REQUIRE(upr <= 8 && (upr & (upr - 1)) == 0, "groupnorm_train: channels per group must be 4, 8, 16 or 32");
    const long long units = (long long)HW * upr;
    int ns = 0;
    const size_t need = gn_train_ws_floats(B, HW, C, groups, ns) * sizeof(float);
    if (ns > 0) {
        DDK_REQUIRE(nslab == 1, "groupnorm_train: the slab forms exist only for group slabs the register-resident kernel takes "
                                "(ddk_groupnorm_train_workspace_bytes() == 0)");
        if (!ws || ws_bytes < need) {
            set_error("groupnorm_train: workspace too small (%zu < %zu)", ws_bytes, need);
            return DDK_ERR_WORKSPACE;
        }
        DDK_REQUIRE(aligned16(ws), "groupnorm_train: workspace alignment");
        const int n_bg = B * groups;
        float* w_part = static_cast<float*>(ws);
        float* w_stat = w_part + (size_t)n_bg * 3 * ns;
        float* w_gpart = w_stat + (size_t)n_bg * 2;
        float* w_gsum = w_gpart + (size_t)n_bg * 2 * ns;
        float* w_cpart = w_gsum + (size_t)n_bg * 2;
        DDK_TRY(gn_stats_partials(x, w_part, B, HW, C, groups, ns, st));
        hipLaunchKernelGGL(gn_big_finalize_kernel, dim3((unsigned)ceil_div(n_bg, 64)), dim3(64), 0, st, w_part, ns, w_stat, n_bg, eps);
        DDK_TRY(check_launch("gn_big_finalize_kernel"));
        const long long total4 = (long long)B * HW * C / 4;
        if (!bwd) {
            const int blocks = (int)(ceil_div(total4, 256) < 8192 ? ceil_div(total4, 256) : 8192);
            hipLaunchKernelGGL(gn_big_fwd_kernel, dim3(blocks), dim3(256), 0, st, x, w_stat, gamma, beta, temb, temb_stride, addend, drop_p,
                               seed, layer, out, HW, C, groups, total4);
            return check_launch("gn_big_fwd_kernel");
        }
        hipLaunchKernelGGL(gn_big_bwd_kernel<0>, dim3(n_bg, ns), dim3(256), 0, st, x, w_stat, gamma, beta, drop_p, seed, layer, dy, out,
                           w_gpart, w_gsum, w_cpart, B, HW, C, groups, ns);
        DDK_TRY(check_launch("gn_big_bwd_kernel<0>"));
        // group sums first (pass 1 needs them), the per-channel rows of all 4 kinds after pass 1 has written kind 3
        hipLaunchKernelGGL(gn_big_finalize2_kernel, dim3((unsigned)ceil_div(2LL * n_bg, 256)), dim3(256), 0, st, w_gpart, w_gsum, n_bg,
                           w_cpart, part, 0LL, ns);
        DDK_TRY(check_launch("gn_big_finalize2_kernel"));
        hipLaunchKernelGGL(gn_big_bwd_kernel<1>, dim3(n_bg, ns), dim3(256), 0, st, x, w_stat, gamma, beta, drop_p, seed, layer, dy, out,
                           w_gpart, w_gsum, w_cpart, B, HW, C, groups, ns);
        DDK_TRY(check_launch("gn_big_bwd_kernel<1>"));
        const long long n_c = 4LL * B * C;
        hipLaunchKernelGGL(gn_big_finalize2_kernel, dim3((unsigned)ceil_div(n_c, 256)), dim3(256), 0, st, w_gpart, w_gsum, 0, w_cpart, part,
                           n_c, ns);
        return check_launch("gn_big_finalize2_kernel");
    }
    DDK_REQUIRE(units <= 4096, "groupnorm_train: internal: resident path selected for a large slab");
    dim3 grid(B * groups);
#define GT(V, NT)                                                                                                                 \
    do {                                                                                                                          \
        if (bwd) hipLaunchKernelGGL((gn_train_kernel<V, NT, true>), grid, dim3(NT), 0, st, x, gamma, beta, temb, temb_stride, addend, \
                                    drop_p, seed, layer, dy, out, part, B, HW, C, groups, eps, nslab, slab_stride, conv_bias, raw_out); \
        else hipLaunchKernelGGL((gn_train_kernel<V, NT, false>), grid, dim3(NT), 0, st, x, gamma, beta, temb, temb_stride, addend,   \
                                drop_p, seed, layer, dy, out, part, B, HW, C, groups, eps, nslab, slab_stride, conv_bias, raw_out);  \
    } while (0)
    const bool many = (long long)B * groups >= 512;      // a training batch: several 256-thread workgroups per CU instead of one of 1024
    if (units <= 256) GT(1, 256);
    else if (units <= 512) GT(2, 256);
    else if (units <= 1024) { if (many) GT(4, 256); else GT(1, 1024); }
    else if (units <= 2048) GT(2, 1024);
    else GT(4, 1024);
#undef GT
    return check_launch("gn_train_kernel");
}

/* GroupNorm(groups, C) + Mish (+ temb[b][c]) (+ Dropout) (+ addend) on CP-pitched NHWC rows whose channels [C, CP) are zero padding
 * (blocks.py:75-84,106-111 at widths that are not multiples of 32): training forward (the input is what the backward recomputes from). */
int ddk_groupnorm_mish_generic_train_fwd(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                                         const float* addend, float drop_p, unsigned long long seed, unsigned layer, float* out, int B, int HW,
                                         int CP, int C, int groups, float eps, ddk_stream_t s) {
    DDK_REQUIRE(x && gamma && beta && out && B > 0 && HW > 0 && groups > 0 && C > 0 && C % groups == 0 && CP >= C && C / groups <= 256,
                "groupnorm_generic_train_fwd: arguments");
    DDK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "groupnorm_generic_train_fwd: dropout probability");
    hipLaunchKernelGGL((gn_generic_train_kernel<false>), dim3((unsigned)(B * (groups + 1))), dim3(256), 0, as_stream(s), x, gamma, beta, temb,
                       temb_stride, addend, drop_p, (uint64_t)seed, (uint32_t)layer, static_cast<const float*>(nullptr), out,
                       static_cast<float*>(nullptr), B, HW, CP, C, groups, eps, gn_generic_cg(C / groups));
    return check_launch("gn_generic_train_kernel");
}

/* its backward: dx (padding zero) and part [4][B][C] = per (image, channel) sums (dtemb, dgamma, dbeta, sum of dx over the pixels) */
int ddk_groupnorm_mish_generic_bwd(const float* x, const float* gamma, const float* beta, float drop_p, unsigned long long seed, unsigned layer,
                                   const float* dy, float* dx, float* part, int B, int HW, int CP, int C, int groups, float eps,
                                   ddk_stream_t s) {
    DDK_REQUIRE(x && gamma && beta && dy && dx && part && B > 0 && HW > 0 && groups > 0 && C > 0 && C % groups == 0 && CP >= C &&
                    C / groups <= 256, "groupnorm_generic_bwd: arguments");
    hipLaunchKernelGGL((gn_generic_train_kernel<true>), dim3((unsigned)(B * (groups + 1))), dim3(256), 0, as_stream(s), x, gamma, beta,
                       static_cast<const float*>(nullptr), 0, static_cast<const float*>(nullptr), drop_p, (uint64_t)seed, (uint32_t)layer, dy, dx,
                       part, B, HW, CP, C, groups, eps, gn_generic_cg(C / groups));
    return check_launch("gn_generic_train_kernel");
}

/* channel LayerNorm over the C real channels of CP-pitched rows (padding written as zero), and its backward: dx (+ addend) and the
 * partial rows part [2][nparts][C] of (dg, db); *nparts_out workgroups were used (<= max_parts) */
int ddk_chan_layernorm_generic(const float* x, const float* g, const float* b, float* out, long long M, int CP, int C, float eps,
                               ddk_stream_t s) {
    return chan_layernorm_generic(x, g, b, out, M, CP, C, eps, as_stream(s));
}
int ddk_chan_layernorm_generic_bwd(const float* x, const float* g, const float* dy, const float* addend, float* dx, float* part,
                                   int max_parts, int* nparts_out, long long M, int CP, int C, float eps, ddk_stream_t s) {
    DDK_REQUIRE(x && g && dy && dx && part && nparts_out && M > 0 && max_parts > 0 && C > 0 && C <= 512 && CP >= C && CP <= 512,
                "layernorm_generic_bwd: arguments (pitch <= 512, the widest level unet_create admits)");
    long long blocks = ceil_div(M, 4);
    if (blocks > max_parts) blocks = max_parts;
    if (blocks > 512) blocks = 512;
    *nparts_out = (int)blocks;
    hipLaunchKernelGGL(chan_layernorm_generic_bwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), x, g, dy, addend, dx, part, M,
                       CP, C, eps);
    return check_launch("chan_layernorm_generic_bwd_kernel");
}

int ddk_dropout_epoch(unsigned long long set_to, int bump, ddk_stream_t s) {
    hipLaunchKernelGGL(drop_epoch_kernel, dim3(1), dim3(1), 0, as_stream(s), set_to, bump);
    return check_launch("drop_epoch_kernel");
}

size_t ddk_groupnorm_train_workspace_bytes(int B, int HW, int C, int groups) {
    if (B <= 0 || HW <= 0 || groups <= 0 || C % groups) return 0;
    int ns;
    return gn_train_ws_floats(B, HW, C, groups, ns) * sizeof(float);
}

int ddk_groupnorm_mish_train_fwd(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                                 const float* addend, float drop_p, uint64_t seed, uint32_t layer, float* out, int B, int HW, int C,
                                 int groups, float eps, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    return gn_train_launch(false, x, gamma, beta, temb, temb_stride, addend, drop_p, seed, layer, nullptr, out, nullptr, B, HW, C, groups,
                           eps, workspace, workspace_bytes, as_stream(s));
}

int ddk_groupnorm_mish_bwd(const float* x, const float* gamma, const float* beta, float drop_p, uint64_t seed, uint32_t layer,
                           const float* dy, float* dx, float* part, int B, int HW, int C, int groups, float eps, void* workspace,
                           size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(dy && part, "groupnorm_bwd: null pointer");
    return gn_train_launch(true, x, gamma, beta, nullptr, 0, nullptr, drop_p, seed, layer, dy, dx, part, B, HW, C, groups, eps,
                           workspace, workspace_bytes, as_stream(s));
}

/* The two above with the tensor they read first still in `nslab` split-K slabs (ddk_conv_args.defer_reduce): the forward sums
 * x = sum slabs + conv_bias while loading and also writes it to raw_out; the backward sums dy.  Register-resident slabs only. */
int ddk_groupnorm_mish_train_fwd_slabs(const float* slabs, int nslab, long long slab_stride, const float* conv_bias, float* raw_out,
                                       const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                                       float drop_p, uint64_t seed, uint32_t layer, float* out, int B, int HW, int C, int groups, float eps,
                                       ddk_stream_t s) {
    DDK_REQUIRE(nslab >= 2, "groupnorm_train_fwd_slabs: fewer than two slabs (use ddk_groupnorm_mish_train_fwd)");
    return gn_train_launch(false, slabs, gamma, beta, temb, temb_stride, addend, drop_p, seed, layer, nullptr, out, nullptr, B, HW, C, groups,
                           eps, nullptr, 0, as_stream(s), nslab, slab_stride, conv_bias, raw_out);
}

int ddk_groupnorm_mish_bwd_slabs(const float* x, const float* gamma, const float* beta, float drop_p, uint64_t seed, uint32_t layer,
                                 const float* dy_slabs, int nslab, long long slab_stride, float* dx, float* part, int B, int HW, int C,
                                 int groups, float eps, ddk_stream_t s) {
    DDK_REQUIRE(dy_slabs && part, "groupnorm_bwd_slabs: null pointer");
    DDK_REQUIRE(nslab >= 2, "groupnorm_bwd_slabs: fewer than two slabs (use ddk_groupnorm_mish_bwd)");
    return gn_train_launch(true, x, gamma, beta, nullptr, 0, nullptr, drop_p, seed, layer, dy_slabs, dx, part, B, HW, C, groups, eps,
                           nullptr, 0, as_stream(s), nslab, slab_stride, nullptr, nullptr);
}

/* out[n] (+)= sum_r rows[r*row_stride + n] */
int ddk_rows_sum(const float* rows, int nrows, long long row_stride, float* out, int n, int accumulate, ddk_stream_t s) {
    DDK_REQUIRE(rows && out && nrows > 0 && n > 0, "rows_sum: arguments");
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)ceil_div(n, 64), 1), dim3(256), 0, as_stream(s), rows, nrows, row_stride, 0LL, out, n,
                       accumulate);
    return check_launch("rows_sum_kernel");
}

/* out[k][n] (+)= sum_r rows[k*batch_stride + r*row_stride + n], k < nbatch: several row reductions in one launch */
int ddk_rows_sum_batched(const float* rows, int nbatch, long long batch_stride, int nrows, long long row_stride, float* out, int n,
                         int accumulate, ddk_stream_t s) {
    DDK_REQUIRE(rows && out && nbatch > 0 && nrows > 0 && n > 0, "rows_sum_batched: arguments");
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)ceil_div(n, 64), (unsigned)nbatch), dim3(256), 0, as_stream(s), rows, nrows,
                       row_stride, batch_stride, out, n, accumulate);
    return check_launch("rows_sum_kernel");
}

/* out_k[n] (+)= sum_r rows[k*batch_stride + r*row_stride + n] for k < nbatch <= 4 with one target per k (null = skipped) */
int ddk_rows_sum_targets(const float* rows, int nbatch, long long batch_stride, int nrows, long long row_stride, float* out0, float* out1,
                         float* out2, float* out3, int n, int accumulate, ddk_stream_t s) {
    DDK_REQUIRE(rows && nbatch > 0 && nbatch <= 4 && nrows > 0 && n > 0, "rows_sum_targets: arguments");
    RowsTargets tg{{out0, out1, out2, out3}};
    hipLaunchKernelGGL(rows_sum_targets_kernel, dim3((unsigned)ceil_div(n, 64), (unsigned)nbatch), dim3(256), 0, as_stream(s), rows, nrows,
                       row_stride, batch_stride, tg, n, accumulate);
    return check_launch("rows_sum_targets_kernel");
}

/* ddk_rows_sum_targets (accumulate = 1) of many calls in a few launches: the records come from HOST memory and travel as kernel
 * arguments, 48 to a launch; same summation order, same bits.  The `rows` buffers must stay untouched until the launch has run. */
int ddk_rows_sum_jobs(const ddk_rows_sum_job* jobs, int n, ddk_stream_t s) {
    DDK_REQUIRE(jobs && n > 0, "rows_sum_jobs: arguments");
    for (int k0 = 0; k0 < n; k0 += RS_JOBS_PER_LAUNCH) {
        RowsJobPack pk{};
        pk.n = n - k0 < RS_JOBS_PER_LAUNCH ? n - k0 : RS_JOBS_PER_LAUNCH;
        long long blocks = 0;
        for (int k = 0; k < pk.n; ++k) {
            ddk_rows_sum_job j = jobs[k0 + k];
            DDK_REQUIRE(j.rows && j.nbatch > 0 && j.nbatch <= 4 && j.nrows > 0 && j.n > 0, "rows_sum_jobs: job");
            j.block0 = blocks;
            blocks += (long long)ceil_div(j.n, 64) * j.nbatch;
            pk.j[k] = j;
        }
        DDK_REQUIRE(blocks < (1LL << 31), "rows_sum_jobs: too many blocks");
        hipLaunchKernelGGL(rows_sum_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), pk);
        DDK_TRY(check_launch("rows_sum_jobs_kernel"));
    }
    return DDK_OK;
}

/* dst_k[i] += src[off_k + i] for every segment k of `table` ([nseg][3] int64 on the device: {source offset in floats, destination
 * address, count}): the parameter gradients a backward produced side by side in ONE buffer go into their places in the flat
 * gradient bucket with one launch (the time MLP alone returns 38 of them: 38 torch adds per micro-batch before). */
int ddk_multi_add(const float* src, const long long* table, int nseg, long long max_count, ddk_stream_t s) {
    DDK_REQUIRE(src && table && nseg > 0 && max_count > 0, "multi_add: arguments");
    long long bx = ceil_div(max_count, 1024);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(multi_add_kernel, dim3((unsigned)bx, (unsigned)nseg), dim3(256), 0, as_stream(s), src, table);
    return check_launch("multi_add_kernel");
}

/* Channel LayerNorm backward: dx and partial rows part[2][nparts][C] (dg, db); returns nparts via *nparts_out. */
int ddk_chan_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* part, int max_parts, int* nparts_out,
                           long long M, int C, float eps, ddk_stream_t s) {
    return ddk_chan_layernorm_bwd_add(x, g, dy, nullptr, dx, part, max_parts, nparts_out, M, C, eps, s);
}

/* the same with dx += addend (the gradient that reaches x over the Residual around PreNorm, blocks.py:13-14: no separate add launch) */
int ddk_chan_layernorm_bwd_add(const float* x, const float* g, const float* dy, const float* addend, float* dx, float* part, int max_parts,
                               int* nparts_out, long long M, int C, float eps, ddk_stream_t s) {
    DDK_REQUIRE(x && g && dy && dx && part && nparts_out && M > 0 && max_parts > 0, "layernorm_bwd: arguments");
    DDK_REQUIRE(aligned16(x) && aligned16(g) && aligned16(dy) && aligned16(dx) && (!addend || aligned16(addend)), "layernorm_bwd: alignment");
    hipStream_t st = as_stream(s);
#define LB(LPP, VPL)                                                                                                     \
    do {                                                                                                                 \
        long long blocks = ceil_div(ceil_div(M, 64 / LPP), 4);                                                           \
        if (blocks > max_parts) blocks = max_parts;                                                                      \
        if (blocks > 512) blocks = 512;                                                                                  \
        *nparts_out = (int)blocks;                                                                                       \
        hipLaunchKernelGGL((chan_layernorm_bwd_kernel<LPP, VPL>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, dy, dx, part, M, C, eps, addend); \
        return check_launch("chan_layernorm_bwd_kernel");                                                                \
    } while (0)
    switch (C) {
        case 32: LB(8, 1);
        case 64: LB(16, 1);
        case 128: LB(32, 1);
        case 256: LB(64, 1);
        case 512: LB(64, 2);
        default: break;
    }
#undef LB
    return fail_arg("layernorm_bwd: unsupported channel count (32, 64, 128, 256, 512)");
}

// pixel splits of the two reductions above: ~1024 workgroups, at least 256 pixels each
static int linattn_train_splits(int B, int HW, int heads) {
    long long s = ceil_div(1024, (long long)B * heads);
    const long long max_s = HW / 256 > 0 ? HW / 256 : 1;
    if (s > max_s) s = max_s;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

/* workspace of ddk_linattn_stats / ddk_linattn_bwd (partials of the pixel-split reductions); 0 when one workgroup per (image, head)
 * walks the map */
size_t ddk_linattn_train_workspace_bytes(int B, int HW, int heads) {
    const int sp = linattn_train_splits(B, HW, heads);
    return sp > 1 ? (size_t)sp * B * heads * DHB * DHB * sizeof(float) : 0;
}

/* softmax statistics of k saved by the training forward: stats[b][h][2][32] = (column max, sum of exp) */
int ddk_linattn_stats(const float* qkv, float* stats, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(qkv && stats && B > 0 && HW > 0 && heads > 0, "linattn_stats: arguments");
    const int sp = linattn_train_splits(B, HW, heads);
    const int per = (int)(ceil_div(ceil_div(HW, sp), 64) * 64);
    if (sp == 1) {
        hipLaunchKernelGGL(linattn_stats_kernel, dim3(B * heads, 1), dim3(256), 0, as_stream(s), qkv, stats, HW, heads, per);
        return check_launch("linattn_stats_kernel");
    }
    DDK_REQUIRE(workspace && workspace_bytes >= (size_t)sp * B * heads * 2 * DHB * sizeof(float), "linattn_stats: workspace too small");
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(linattn_stats_kernel, dim3(B * heads, sp), dim3(256), 0, as_stream(s), qkv, part, HW, heads, per);
    DDK_TRY(check_launch("linattn_stats_kernel"));
    hipLaunchKernelGGL(linattn_stats_merge_kernel, dim3(B * heads), dim3(64), 0, as_stream(s), static_cast<const float*>(part), stats, sp);
    return check_launch("linattn_stats_merge_kernel");
}

/* dqkv from dout (grad of the attention output before to_out); dctx is scratch [B][heads][32][32] */
static int linattn_bwd_impl(const float* qkv, const float* dout, const float* ctx, const float* stats, float* stats_scratch, float* dctx,
                            float* dqkv, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s);
int ddk_linattn_bwd(const float* qkv, const float* dout, const float* ctx, const float* stats, float* dctx, float* dqkv, int B, int HW,
                    int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(stats, "linattn_bwd: null stats");
    return linattn_bwd_impl(qkv, dout, ctx, stats, nullptr, dctx, dqkv, B, HW, heads, workspace, workspace_bytes, s);
}
/* the same without statistics saved by the forward: they are recomputed from qkv into stats_scratch [B][heads][2][32] -- inside the dctx
 * launch where one workgroup covers the pixels of an (image, head), by ddk_linattn_stats' launches on large maps */
int ddk_linattn_bwd_recompute(const float* qkv, const float* dout, const float* ctx, float* stats_scratch, float* dctx, float* dqkv, int B,
                              int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(stats_scratch, "linattn_bwd_recompute: null stats_scratch");
    return linattn_bwd_impl(qkv, dout, ctx, nullptr, stats_scratch, dctx, dqkv, B, HW, heads, workspace, workspace_bytes, s);
}
static int linattn_bwd_impl(const float* qkv, const float* dout, const float* ctx, const float* stats, float* stats_scratch, float* dctx,
                            float* dqkv, int B, int HW, int heads, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(qkv && dout && ctx && dctx && dqkv && B > 0 && HW > 0 && heads >= 1 && heads <= 4, "linattn_bwd: arguments (heads <= 4)");
    DDK_REQUIRE(aligned16(qkv) && aligned16(dout) && aligned16(ctx) && aligned16(dctx) && aligned16(dqkv) && aligned16(workspace),
                "linattn_bwd: alignment");
    hipStream_t st = as_stream(s);
    const int sp = linattn_train_splits(B, HW, heads);
    const int per = (int)(ceil_div(ceil_div(HW, sp), 64) * 64);
    if (sp == 1) {
        hipLaunchKernelGGL(linattn_dctx_kernel, dim3(B * heads, 1), dim3(256), 0, st, qkv, dout, dctx, HW, heads, per,
                           stats ? static_cast<float*>(nullptr) : stats_scratch);
        DDK_TRY(check_launch("linattn_dctx_kernel"));
        if (!stats) stats = stats_scratch;
    } else {
        if (!stats) {
            DDK_TRY(ddk_linattn_stats(qkv, stats_scratch, B, HW, heads, workspace, workspace_bytes, s));
            stats = stats_scratch;
        }
        const int n = B * heads * DHB * DHB;
        DDK_REQUIRE(workspace && workspace_bytes >= (size_t)sp * n * sizeof(float), "linattn_bwd: workspace too small");
        float* part = static_cast<float*>(workspace);
        hipLaunchKernelGGL(linattn_dctx_kernel, dim3(B * heads, sp), dim3(256), 0, st, qkv, dout, part, HW, heads, per,
                           static_cast<float*>(nullptr));
        DDK_TRY(check_launch("linattn_dctx_kernel"));
        hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)ceil_div(n, 64), 1), dim3(256), 0, st, static_cast<const float*>(part), sp,
                           (long long)n, 0LL, dctx, n, 0);
        DDK_TRY(check_launch("rows_sum_kernel"));
    }
    const int tiles = (int)ceil_div(HW, 16);
    const size_t lds = ((size_t)2 * heads * DHB * (DHB + 4) + heads * DHB) * sizeof(float);
    hipLaunchKernelGGL(linattn_bwd_apply_kernel, dim3(B * tiles), dim3(64 * heads), lds, st, qkv, dout, ctx, dctx, stats, dqkv, HW, heads,
                       tiles);
    return check_launch("linattn_bwd_apply_kernel");
}

int ddk_mish_bwd(const float* x, const float* dy, float* dx, long long n, ddk_stream_t s) {
    DDK_REQUIRE(x && dy && dx && n > 0 && n % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx), "mish_bwd: arguments (n % 4)");
    hipLaunchKernelGGL(unary_bwd_kernel<0>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(s), x, dy, dx, n / 4);
    return check_launch("mish_bwd");
}
int ddk_tanh_bwd(const float* y, const float* dy, float* dx, long long n, ddk_stream_t s) {
    DDK_REQUIRE(y && dy && dx && n > 0 && n % 4 == 0 && aligned16(y) && aligned16(dy) && aligned16(dx), "tanh_bwd: arguments (n % 4)");
    hipLaunchKernelGGL(unary_bwd_kernel<1>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(s), y, dy, dx, n / 4);
    return check_launch("tanh_bwd");
}
/* dy [B][H/2][W/2][C] -> dx [B][H][W][C] */
int ddk_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, ddk_stream_t s) {
    DDK_REQUIRE(dy && dx && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "avgpool2_bwd: shape");
    const long long total4 = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(s), dy, dx, H, W, C, total4);
    return check_launch("avgpool2_bwd");
}
/* dy [B][2H][2W][C] -> dx [B][H][W][C] */
int ddk_upsample_nearest2_bwd(const float* dy, float* dx, int B, int H, int W, int C, ddk_stream_t s) {
    DDK_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample_nearest2_bwd: shape");
    const long long total4 = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(upnearest2_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(s), dy, dx, H, W, C, total4);
    return check_launch("upnearest2_bwd");
}
/* out = -2 (a - b) * scale[sample]: gradient of sum((a-b)^2) wrt b times an upstream per-sample factor */
int ddk_sq_err_grad(const float* a, const float* b, const float* scale, float* out, int B, long long per, ddk_stream_t s) {
    DDK_REQUIRE(a && b && scale && out && B > 0 && per > 0 && per % 4 == 0, "sq_err_grad: arguments");
    const long long total4 = B * per / 4;
    hipLaunchKernelGGL(sq_err_grad_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(s), a, b, scale, out, per / 4, total4);
    return check_launch("sq_err_grad");
}
/* The dDDPM autoencoder objective of the 'simple' loss (dddpm.py:155-177) from the two per-sample losses: out[0] = obj = latent + recon,
 * out[1] = latent = mean_b l_ddpm[b], out[2] = recon = mean_b (t[b] < t_rec_max ? l_rec[b] : 0); one workgroup, sums in index order per
 * lane then a fixed tree (ten tiny torch launches before).  Backward: d l_ddpm[b] = g / B, d l_rec[b] = (t[b] < t_rec_max) g / B. */
int ddk_ae_objective(const float* l_ddpm, const float* l_rec, const int64_t* t, int t_rec_max, int B, float* out, ddk_stream_t s) {
    DDK_REQUIRE(l_ddpm && l_rec && t && out && B > 0, "ae_objective: arguments");
    hipLaunchKernelGGL(ae_objective_kernel, dim3(1), dim3(256), 0, as_stream(s), l_ddpm, l_rec, t, t_rec_max, B, out);
    return check_launch("ae_objective_kernel");
}
int ddk_ae_objective_bwd(const float* g, const int64_t* t, int t_rec_max, int B, float* d_ddpm, float* d_rec, ddk_stream_t s) {
    DDK_REQUIRE(g && t && d_ddpm && d_rec && B > 0, "ae_objective_bwd: arguments");
    hipLaunchKernelGGL(ae_objective_bwd_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, as_stream(s), g, t, t_rec_max, B, d_ddpm, d_rec);
    return check_launch("ae_objective_bwd_kernel");
}
int ddk_scale_per_sample(const float* x, const float* scale, float* out, int B, long long per, ddk_stream_t s) {
    DDK_REQUIRE(x && scale && out && B > 0 && per > 0 && per % 4 == 0, "scale_per_sample: arguments");
    const long long total4 = B * per / 4;
    hipLaunchKernelGGL(scale_per_sample_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(s), x, scale, out, per / 4, total4);
    return check_launch("scale_per_sample");
}

/* final 1x1 backward; part: [max_rows][n_out*C + n_out] partial rows, *nrows_out of them are written */
int ddk_conv1x1_small_n_bwd(const float* a, const float* w, const float* dy, float* da, float* part, int max_rows, int* nrows_out,
                            long long M, int C, int n_out, ddk_stream_t s) {
    DDK_REQUIRE(a && w && dy && da && part && nrows_out && M > 0 && n_out > 0 && n_out <= 8, "conv1x1_small_n_bwd: arguments (n_out <= 8)");
    DDK_REQUIRE(aligned16(a) && aligned16(w) && aligned16(da), "conv1x1_small_n_bwd: alignment");
    hipStream_t st = as_stream(s);
#define CB(LPP, VPL)                                                                                                          \
    do {                                                                                                                      \
        long long blocks = ceil_div(ceil_div(M, 64 / LPP), 4);                                                                \
        if (blocks > max_rows / 4) blocks = max_rows / 4;                                                                     \
        if (blocks > 256) blocks = 256;                                                                                       \
        if (blocks < 1) return fail_arg("conv1x1_small_n_bwd: max_rows < 4");                                                 \
        *nrows_out = (int)blocks * 4;                                                                                         \
        hipLaunchKernelGGL((conv1x1_small_n_bwd_kernel<LPP, VPL, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a, w, dy, da, part, M, C, \
                           n_out);                                                                                            \
        return check_launch("conv1x1_small_n_bwd_kernel");                                                                    \
    } while (0)
    switch (C) {
        case 32: CB(8, 1);
        case 64: CB(16, 1);
        case 128: CB(32, 1);
        case 256: CB(64, 1);
        case 96: CB(8, 3);       // d_chans = 96, 160, 192, ...: the dDDPM decoder's last 1x1 at widths that are multiples of 32 only
        case 160: CB(8, 5);
        case 192: CB(16, 3);
        case 320: CB(16, 5);
        case 384: CB(32, 3);
        case 512: CB(64, 2);
        default: break;
    }
    if (C > 0 && C % 32 == 0) CB(8, 1);     // any other padded width: 32 channels per trip over the pixels
#undef CB
    return fail_arg("conv1x1_small_n_bwd: the channel count must be a multiple of 32");
}

/* C = op(A) op(B) for the tiny time-embedding matrices; see small_gemm_kernel for the modes */
static int small_gemm_splits(int M, int N, int K, int ldc) {
    const long long tiles = ceil_div(N, 32) * ceil_div(M, 32);
    // a workgroup walks its k range 32 at a time with the load latency exposed (nothing else runs on its CU): ~3.8 us per step on the
    // 4 x 2 tiles of the [64][512] x [512][128] product of the time MLP (62 us); from K = 256 on the range is cut into pieces of >= 64
    if (ldc != N || tiles >= 128 || K < 256) return 1;
    long long s = ceil_div(256, tiles);
    if (s > K / 64) s = K / 64;
    return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}
size_t ddk_small_gemm_workspace_bytes(int M, int N, int K, int ldc) {
    const int sp = small_gemm_splits(M, N, K, ldc);
    return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}
int ddk_small_gemm(int mode, const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc, int accumulate,
                   void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(A && Bm && Cm && M > 0 && N > 0 && K > 0 && mode >= 0 && mode <= 2, "small_gemm: arguments");
    const int sp = small_gemm_splits(M, N, K, ldc);
    if (sp == 1 || !workspace || workspace_bytes < (size_t)sp * M * N * sizeof(float)) {
        hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32), 1), dim3(256), 0, as_stream(s), mode,
                           A, Bm, Cm, M, N, K, lda, ldb, ldc, accumulate, K);
        return check_launch("small_gemm_kernel");
    }
    const int kper = (int)(ceil_div(ceil_div(K, sp), 32) * 32);
    const int zs = (int)ceil_div(K, kper);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32), (unsigned)zs), dim3(256), 0, as_stream(s),
                       mode, A, Bm, part, M, N, K, lda, ldb, N, 0, kper);
    DDK_TRY(check_launch("small_gemm_kernel"));
    const int n = M * N;
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)ceil_div(n, 64), 1), dim3(256), 0, as_stream(s), static_cast<const float*>(part), zs,
                       (long long)n, 0LL, Cm, n, accumulate);
    return check_launch("rows_sum_kernel");
}
int ddk_sincos_embed(const int64_t* t, const float* freqs, float* e, int B, int dim, ddk_stream_t s) {
    DDK_REQUIRE(t && freqs && e && B > 0 && dim > 0 && dim % 2 == 0, "sincos_embed: arguments");
    hipLaunchKernelGGL(sincos_kernel, dim3((unsigned)ceil_div((long long)B * dim, 256)), dim3(256), 0, as_stream(s), t, freqs, e, B, dim);
    return check_launch("sincos_kernel");
}
/* y[m][n] += bias[n] in place; act = mish(y) when act != NULL */
int ddk_bias_act(float* y, const float* bias, float* act, long long M, int N, ddk_stream_t s) {
    DDK_REQUIRE(y && bias && M > 0 && N > 0, "bias_act: arguments");
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)ceil_div(M * N, 256)), dim3(256), 0, as_stream(s), y, bias, act, M * N, N);
    return check_launch("bias_act_kernel");
}

/* global gradient norm + clip coefficient: out2 = [norm, min(1, max_norm/(norm+1e-6))]; workspace >= 1024 floats */
int ddk_grad_norm_clip(const float* g, long long n, float max_norm, float* out2, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(g && out2 && workspace && n > 0 && workspace_bytes >= 1024 * sizeof(float), "grad_norm_clip: arguments");
    const int blocks = (int)(ceil_div(n, 256 * 16) < 1024 ? ceil_div(n, 256 * 16) : 1024);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(s), g, n, part);
    DDK_TRY(check_launch("sumsq_partial_kernel"));
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, as_stream(s), part, blocks, max_norm, out2);
    return check_launch("clip_coef_kernel");
}
/* one Adam step on flat buffers; step counts from 1; clip2 (device [norm, coef]) may be NULL */
int ddk_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps, int step,
                  const float* clip2, ddk_stream_t s) {
    DDK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), p, g, m, v, n, (float)(lr / bc1), (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)bc2s, clip2);
    return check_launch("adam_kernel");
}
int ddk_ema_update(float* p_ema, const float* p, long long n, float decay, ddk_stream_t s) {
    DDK_REQUIRE(p_ema && p && n > 0, "ema_update: arguments");
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(s), p_ema, p, n, decay);
    return check_launch("ema_kernel");
}
}
