// diffusion.hip -- noise-schedule arithmetic of the DDPM (HBM-bound elementwise kernels).
//
// Reference: models/diffusion/ddpm.py:256-273 (q_sample), :149-158 (predict_x_from_eps, clamp),
// :177-185 (q_posterior mean), :203-227 (p_sample), :241 (x_T ~ N(0,1)), :279 + utils/utils.py:34-40
// (per-sample squared-error sum); models/utils/helpers.py:31-40 (extract, noise_like).
//
// The coefficient tables are the module's [T] fp32 buffers; each sample gathers its own t.  Products and
// sums use __fmul_rn/__fadd_rn in the reference's evaluation order (no FMA contraction), so given the
// same eps_hat and noise the update is bit-identical to the torch expression.
#include "ddk_internal.h"

// hipcc contracts a*b+c into an FMA by default (-ffp-contract=fast), even through __fmul_rn/__fadd_rn; the
// schedule arithmetic below must round every product like the reference's separate torch ops do, so this
// file is compiled with -ffp-contract=off (see Makefile).

namespace ddk {

// ---- Philox4x32-10 (Salmon et al. SC'11; Random123 philox4x32_R(10)) -----------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    }
    return c;
}

__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f; }  // 2^-24

__device__ __forceinline__ float4 philox_normal4(unsigned long long idx4, uint32_t step, uint32_t stream, uint64_t seed) {
    const U4 r = philox4x32_10(U4{(uint32_t)idx4, (uint32_t)(idx4 >> 32), step, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float two_pi = 6.283185307179586f;
    float4 z;
    float sn, cs;
    float rad = sqrtf(-2.0f * logf(u01(r.x)));
    sincosf(two_pi * u01(r.y), &sn, &cs);
    z.x = rad * cs; z.y = rad * sn;
    rad = sqrtf(-2.0f * logf(u01(r.z)));
    sincosf(two_pi * u01(r.w), &sn, &cs);
    z.z = rad * cs; z.w = rad * sn;
    return z;
}

static int grid1d(long long n) {
    const long long b = ceil_div(n > 0 ? n : 1, 256);
    return (int)(b < 2048 ? b : 2048);
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long long n4, long long n, uint64_t seed, uint32_t step,
                                                    uint32_t stream) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 z = philox_normal4((unsigned long long)i, step, stream, seed);
        if (i * 4 + 3 < n) {
            reinterpret_cast<float4*>(out)[i] = z;
        } else {
            const float zz[4] = {z.x, z.y, z.z, z.w};
            for (int j = 0; j < 4 && i * 4 + j < n; ++j) out[i * 4 + j] = zz[j];
        }
    }
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                       const int64_t* __restrict__ t, const float* __restrict__ ca,
                                                       const float* __restrict__ cb, float* __restrict__ out, long long per4,
                                                       long long total4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int64_t tb = t[i / per4];
        const float a = ca[tb], b = cb[tb];
        const float4 xv = reinterpret_cast<const float4*>(x)[i], ev = reinterpret_cast<const float4*>(eps)[i];
        float4 o;
        o.x = __fadd_rn(__fmul_rn(a, xv.x), __fmul_rn(b, ev.x));
        o.y = __fadd_rn(__fmul_rn(a, xv.y), __fmul_rn(b, ev.y));
        o.z = __fadd_rn(__fmul_rn(a, xv.z), __fmul_rn(b, ev.z));
        o.w = __fadd_rn(__fmul_rn(a, xv.w), __fmul_rn(b, ev.w));
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

__device__ __forceinline__ float p_step(float x, float e, float z, float cr, float crm1, float c1, float c2, float sg) {
    float x0 = __fsub_rn(__fmul_rn(cr, x), __fmul_rn(crm1, e));     // ddpm.py:152-155
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                             // ddpm.py:157 clamp_(-1, 1)
    const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, x));  // ddpm.py:177-180
    return __fadd_rn(mean, __fmul_rn(sg, z));                       // ddpm.py:227 (sg already carries the t>0 mask)
}

__global__ __launch_bounds__(256) void p_sample_kernel(float* __restrict__ x, const float* __restrict__ eps_hat,
                                                       const float* __restrict__ noise, long long noise_step_stride, int t_first,
                                                       const int64_t* __restrict__ t,
                                                       const float* __restrict__ c_recip, const float* __restrict__ c_recipm1,
                                                       const float* __restrict__ c1, const float* __restrict__ c2,
                                                       const float* __restrict__ sigma, long long per4, long long total4,
                                                       uint64_t seed, uint32_t stream, const int64_t* __restrict__ chain_state,
                                                       int64_t* dec_counter) {
    if (dec_counter && blockIdx.x == 0 && threadIdx.x == 0) *dec_counter -= 1;     // the step counter, by the step's last kernel (see final_tail)
    if (chain_state) {   // sampler: the Philox key lives in device memory, so one captured graph serves every seed
        seed = (uint64_t)chain_state[1];
        stream = (uint32_t)chain_state[2];
    }
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int64_t tb = t[i / per4];
        const float cr = c_recip[tb], crm1 = c_recipm1[tb], a1 = c1[tb], a2 = c2[tb];
        const float sg = tb > 0 ? sigma[tb] : 0.0f;  // nonzero_mask * exp(0.5 logvar), ddpm.py:220-227
        const float4 xv = reinterpret_cast<const float4*>(x)[i], ev = reinterpret_cast<const float4*>(eps_hat)[i];
        // injected noise: draw k = t_first - t of a [n_steps][...] array (stride 0: a single tensor)
        const float4 zv = noise ? reinterpret_cast<const float4*>(noise + (long long)(t_first - tb) * noise_step_stride)[i]
                                : philox_normal4((unsigned long long)i, (uint32_t)tb, stream, seed);
        float4 o;
        o.x = p_step(xv.x, ev.x, zv.x, cr, crm1, a1, a2, sg);
        o.y = p_step(xv.y, ev.y, zv.y, cr, crm1, a1, a2, sg);
        o.z = p_step(xv.z, ev.z, zv.z, cr, crm1, a1, a2, sg);
        o.w = p_step(xv.w, ev.w, zv.w, cr, crm1, a1, a2, sg);
        reinterpret_cast<float4*>(x)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// The end of a forward in ONE launch (reference models/unet/unet.py:69-72 after the final Block's conv, blocks.py:79-80;
// in the sampler also models/diffusion/ddpm.py:149-158,177-185,216-227): GroupNorm (statistics from the final conv's per-tile
// partials) -> Mish -> 1x1 projection to n_out <= 8 channels -> eps_hat, and -- when x is given -- the reverse-step update of x
// in place.  Replaces gn_apply_parts_kernel + conv1x1_n8_kernel + p_sample_kernel: the normalised activation (16.8 MB at
// cfg4) and eps_hat never go to memory.
//   workgroup (16 waves) = one 128-pixel tile of one image; phase 1: LPP lanes share a pixel (conv1x1_n8_kernel's butterfly), eps_hat of
//   the tile goes to LDS; phase 2: the tile's 128 * n_out latent elements are updated with p_sample_kernel's exact arithmetic
//   and Philox indexing (bit-identical given the same eps_hat).
struct TailParams {
    const float* raw;          // [B][HW][C] output of the final Block's conv
    const float2* part;        // [B*np][G] {mean, M2} per (128-pixel tile, group)
    const float* gamma;
    const float* beta;
    const float* w;            // [n_out][C]
    const float* bias;         // [n_out] or nullptr
    float* eps_out;            // [B][HW][n_out] or nullptr
    float* x;                  // [B][HW][n_out] or nullptr: updated in place
    const float* noise;
    long long noise_step_stride;
    int t_first;
    const int64_t* t;
    const float *c_recip, *c_recipm1, *c1, *c2, *sigma;
    const int64_t* chain_state;   // sampler: {counter, Philox seed, stream id} in device memory; else seed / stream below
    uint64_t seed;
    uint32_t stream;
    int np, HW, C, cpg, n_out;
    float eps;
    int64_t* dec_counter;         // the step counter, decremented HERE (the step's last kernel) when the first kernel left it alone, or null
};

template <int LPP, int VPL>
__global__ __launch_bounds__(1024) void final_tail_kernel(const TailParams p) {
    constexpr int PPW = 64 / LPP;                    // pixels per wave and iteration
    constexpr int PPI = 16 * PPW;                    // ... per iteration of the 16-wave workgroup
    constexpr int NIT = 128 / PPI;                   // 4 at C = 128 / 256, 2 at C = 64, 1 at C = 32
    __shared__ float2 mr[64];
    __shared__ float2 sp[1024];
    __shared__ __attribute__((aligned(16))) float es[128 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane % LPP, pl = lane / LPP;
    const int b = blockIdx.x / p.np, tile = blockIdx.x - b * p.np;
    const int G = p.C / p.cpg;
    const long long pix0 = (long long)b * p.HW + tile * 128;
    if (p.dec_counter && blockIdx.x == 0 && tid == 0) *p.dec_counter -= 1;     // nobody reads the counter in this kernel (t comes from t_cur)
    // every pixel this wave will touch is requested up front, before the statistics are merged: one latency, not NIT
    float4 v[NIT][VPL];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            v[it][i] = *reinterpret_cast<const float4*>(p.raw + (pix0 + it * PPI + wave * PPW + pl) * p.C + (sub + i * LPP) * 4);
    const float2* pb = p.part + (long long)b * p.np * G;
    for (int i = tid; i < p.np * G; i += 1024) sp[i] = pb[i];
    // ... and so is everything else that does not depend on the statistics
    float4 ga[VPL], be[VPL], ww[8][VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (sub + i * LPP) * 4;
        ga[i] = *reinterpret_cast<const float4*>(p.gamma + c0);
        be[i] = *reinterpret_cast<const float4*>(p.beta + c0);
#pragma unroll
        for (int co = 0; co < 8; ++co)
            ww[co][i] = co < p.n_out ? *reinterpret_cast<const float4*>(p.w + (long long)co * p.C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool h2 = sub & (LPP / 2), h4 = sub & (LPP / 4), h8 = sub & (LPP / 8);
    const int my_co = (h2 ? 4 : 0) + (h4 ? 2 : 0) + (h8 ? 1 : 0);
    const float my_bias = (p.bias && my_co < p.n_out) ? p.bias[my_co] : 0.f;
    // ... including phase 2's operands of the threads that have one (128 * n_out / 4 <= 256 float4 per tile): the latent, the schedule
    // scalars and the noise draw depend on nothing computed here -- one memory latency and the Philox rounds less behind the last barrier
    const int cnt4 = 128 * p.n_out / 4;
    const long long e4 = pix0 * p.n_out / 4;          // host: (128 * n_out) % 4 == 0
    float4 xv0 = make_float4(0.f, 0.f, 0.f, 0.f), zv0 = xv0;
    float cr = 0.f, crm1 = 0.f, a1 = 0.f, a2 = 0.f, sg = 0.f;
    int64_t tb = 0;
    if (p.x && tid < cnt4) {
        const uint64_t seed = p.chain_state ? (uint64_t)p.chain_state[1] : p.seed;
        const uint32_t stream = p.chain_state ? (uint32_t)p.chain_state[2] : p.stream;
        tb = p.t[b];
        cr = p.c_recip[tb]; crm1 = p.c_recipm1[tb]; a1 = p.c1[tb]; a2 = p.c2[tb];
        sg = tb > 0 ? p.sigma[tb] : 0.0f;
        const long long i = e4 + tid;
        xv0 = reinterpret_cast<const float4*>(p.x)[i];
        zv0 = p.noise ? reinterpret_cast<const float4*>(p.noise + (long long)(p.t_first - tb) * p.noise_step_stride)[i]
                      : philox_normal4((unsigned long long)i, (uint32_t)tb, stream, seed);
    }
    __syncthreads();
    if (tid < G) {          // the same fixed-order merge as gn_apply_parts_kernel
        float ms = 0.f;
        for (int i = 0; i < p.np; ++i) ms += sp[i * G + tid].x;
        const float mean = ms / (float)p.np;
        float m2 = 0.f, d2 = 0.f;
        for (int i = 0; i < p.np; ++i) {
            const float2 t = sp[i * G + tid];
            m2 += t.y;
            d2 += (t.x - mean) * (t.x - mean);
        }
        const float n_i = 128.0f * (float)p.cpg;
        const float var = (m2 + n_i * d2) / ((float)p.np * n_i);
        mr[tid] = make_float2(mean, 1.0f / sqrtf(var + p.eps));
    }
    __syncthreads();
    float2 st[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) st[i] = mr[((sub + i * LPP) * 4) / p.cpg];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int lp = it * PPI + wave * PPW + pl;   // pixel within the tile
        float s[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) s[co] = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float4 y;
            y.x = mish_f((v[it][i].x - st[i].x) * st[i].y * ga[i].x + be[i].x);
            y.y = mish_f((v[it][i].y - st[i].x) * st[i].y * ga[i].y + be[i].y);
            y.z = mish_f((v[it][i].z - st[i].x) * st[i].y * ga[i].z + be[i].z);
            y.w = mish_f((v[it][i].w - st[i].x) * st[i].y * ga[i].w + be[i].w);
#pragma unroll
            for (int co = 0; co < 8; ++co) s[co] += (y.x * ww[co][i].x + y.y * ww[co][i].y) + (y.z * ww[co][i].z + y.w * ww[co][i].w);
        }
        // conv1x1_n8_kernel's packed butterfly: lane l ends with output my_co summed over the pixel's LPP lanes
        float t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = h2 ? s[4 + i] : s[i], send = h2 ? s[i] : s[4 + i];
            t4[i] = keep + __shfl_xor(send, LPP / 2, 64);
        }
        float t2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = h4 ? t4[2 + i] : t4[i], send = h4 ? t4[i] : t4[2 + i];
            t2[i] = keep + __shfl_xor(send, LPP / 4, 64);
        }
        float r = (h8 ? t2[1] : t2[0]) + __shfl_xor(h8 ? t2[0] : t2[1], LPP / 8, 64);
#pragma unroll
        for (int o = LPP / 16; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if ((sub & (LPP / 8 - 1)) == 0 && my_co < p.n_out) es[lp * p.n_out + my_co] = r + my_bias;
    }
    __syncthreads();
    // phase 2: the tile's 128 * n_out elements, contiguous in the NHWC latent (cnt4 <= 256: at most one float4 per thread)
    if (tid < cnt4) {
        const float4 ev = reinterpret_cast<const float4*>(es)[tid];
        if (p.eps_out) reinterpret_cast<float4*>(p.eps_out)[e4 + tid] = ev;
        if (p.x) {
            float4 o;
            o.x = p_step(xv0.x, ev.x, zv0.x, cr, crm1, a1, a2, sg);
            o.y = p_step(xv0.y, ev.y, zv0.y, cr, crm1, a1, a2, sg);
            o.z = p_step(xv0.z, ev.z, zv0.z, cr, crm1, a1, a2, sg);
            o.w = p_step(xv0.w, ev.w, zv0.w, cr, crm1, a1, a2, sg);
            reinterpret_cast<float4*>(p.x)[e4 + tid] = o;
        }
    }
}

bool final_tail_ok(int HW, int C, int groups, int n_out, int np) {
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false;
    if (groups <= 0 || groups > 64 || C % groups || (C / groups) % 4) return false;
    if (n_out < 1 || n_out > 8 || (128 * n_out) % 4) return false;
    return np > 0 && HW == np * 128 && np * groups <= 1024;
}

int final_tail(const float* raw, const float* part, int np, const float* gamma, const float* beta, float eps, const float* w,
               const float* bias, int n_out, float* eps_out, float* x, const float* noise, long long noise_step_stride, int t_first,
               const int64_t* t, const float* c_recip, const float* c_recipm1, const float* c1, const float* c2, const float* sigma,
               const int64_t* chain_state, uint64_t seed, uint32_t stream_id, int B, int HW, int C, int groups, hipStream_t st,
               int64_t* dec_counter) {
    DDK_REQUIRE(raw && part && gamma && beta && w && (eps_out || x) && B > 0, "final_tail: null pointer");
    DDK_REQUIRE(final_tail_ok(HW, C, groups, n_out, np), "final_tail: needs C in {32,64,128,256}, n_out <= 8, H*W == tiles * 128");
    DDK_REQUIRE(aligned16(raw) && aligned16(gamma) && aligned16(beta) && aligned16(w) && aligned16(eps_out) && aligned16(x) &&
                    aligned16(noise) && noise_step_stride % 4 == 0, "final_tail: alignment");
    DDK_REQUIRE(!x || (t && c_recip && c_recipm1 && c1 && c2 && sigma), "final_tail: the update needs t and the schedule tables");
    TailParams p{};
    p.raw = raw; p.part = reinterpret_cast<const float2*>(part); p.gamma = gamma; p.beta = beta; p.w = w; p.bias = bias;
    p.eps_out = eps_out; p.x = x; p.noise = noise; p.noise_step_stride = noise_step_stride; p.t_first = t_first; p.t = t;
    p.c_recip = c_recip; p.c_recipm1 = c_recipm1; p.c1 = c1; p.c2 = c2; p.sigma = sigma; p.chain_state = chain_state;
    p.seed = seed; p.stream = stream_id;
    p.np = np; p.HW = HW; p.C = C; p.cpg = C / groups; p.n_out = n_out; p.eps = eps;
    p.dec_counter = dec_counter;
    const dim3 grid((unsigned)(B * np));
    if (C == 32) hipLaunchKernelGGL((final_tail_kernel<8, 1>), grid, dim3(1024), 0, st, p);
    else if (C == 64) hipLaunchKernelGGL((final_tail_kernel<16, 1>), grid, dim3(1024), 0, st, p);
    else if (C == 128) hipLaunchKernelGGL((final_tail_kernel<32, 1>), grid, dim3(1024), 0, st, p);
    else hipLaunchKernelGGL((final_tail_kernel<32, 2>), grid, dim3(1024), 0, st, p);
    return check_launch("final_tail_kernel");
}

// one workgroup per sample: fixed summation tree -> run-to-run deterministic
__global__ __launch_bounds__(1024) void sq_err_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ per_sample, long long per4) {
    __shared__ float red[32];
    const float4* ap = reinterpret_cast<const float4*>(a) + blockIdx.x * per4;
    const float4* bp = reinterpret_cast<const float4*>(b) + blockIdx.x * per4;
    float s = 0.f;
    for (long long i = threadIdx.x; i < per4; i += blockDim.x) {
        const float4 u = ap[i], v = bp[i];
        const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
        s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) per_sample[blockIdx.x] = s;
}

// Evaluation-time variational bound, one term per sample (reference models/diffusion/ddpm.py:317-366 after the UNet call,
// models/utils/losses.py:17-109, utils/utils.py:43-48), fused into ONE pass over x, x_t, eps_hat (and eps for L_simple):
//   x0 = clamp(c_recip x_t - c_recipm1 eps_hat); pred_mean = c1 x0 + c2 x_t; true_mean = c1 x + c2 x_t      (q_posterior twice)
//   t > 0:  kl  = 0.5 (lv - lv - 1 + exp(lv - lv) + (true_mean - pred_mean)^2 exp(-lv))                      (normal_kl)
//   t == 0: nll = -log p(x | pred_mean, exp(0.5 lv)) of the discretised Gaussian (tanh CDF approximation)
//   vlb[b] = mean_chw(.) / ln 2 (flat_bits);  sqerr[b] = sum_chw (eps - eps_hat)^2.
// The reference evaluates ~25 elementwise torch ops + 2 reductions per timestep (and runs the UNet twice on identical
// inputs); this is one launch, 16 B read per element, one workgroup per sample with a fixed summation tree.
__device__ __forceinline__ float std_normal_cdf_approx(float v) {
    return 0.5f * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));      // sqrt(2/pi)
}

// Each sample is split over `ns` workgroups (a batch of 8 full-resolution images would otherwise occupy 8 of 256 CUs); a slice
// leaves {sum of terms, sum of squared errors} in the workspace (sc1 stores), adds to the sample's arrival counter, and the
// workgroup whose add came last sums the ns partials IN SLICE ORDER -- deterministic whatever the arrival order -- writes the
// outputs and re-arms the counter (MI355X_MICROARCH.md, sc1 table, first row: "the workgroup whose add came last").
__global__ __launch_bounds__(1024) void vlb_terms_kernel(const float* __restrict__ x, const float* __restrict__ x_t,
                                                         const float* __restrict__ eps_hat, const float* __restrict__ eps,
                                                         const int64_t* __restrict__ t, const float* __restrict__ c_recip,
                                                         const float* __restrict__ c_recipm1, const float* __restrict__ c1,
                                                         const float* __restrict__ c2, const float* __restrict__ logvar,
                                                         float* __restrict__ vlb, float* __restrict__ sqerr, long long per, int ns,
                                                         unsigned* __restrict__ counters, float* __restrict__ partials) {
    __shared__ float red[32];
    __shared__ unsigned last;
    const int b = blockIdx.y, sl = blockIdx.x;
    const int64_t tb = t[b];
    const float cr = c_recip[tb], crm1 = c_recipm1[tb], a1 = c1[tb], a2 = c2[tb], lv = logvar[tb];
    const float inv_var = expf(-lv), inv_std = expf(-(0.5f * lv));
    const float kl0 = (lv - lv - 1.0f) + expf(lv - lv);
    const long long base = (long long)b * per;
    const long long chunk = (per + ns - 1) / ns, i0 = sl * chunk, i1 = i0 + chunk < per ? i0 + chunk : per;
    float acc = 0.f, sq = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float xv = x[base + i], xt = x_t[base + i], eh = eps_hat[base + i];
        float x0 = __fsub_rn(__fmul_rn(cr, xt), __fmul_rn(crm1, eh));
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float pred = __fadd_rn(__fmul_rn(a1, x0), __fmul_rn(a2, xt));
        float term;
        if (tb == 0) {
            const float c = xv - pred;
            const float cdf_plus = std_normal_cdf_approx(inv_std * (c + 1.0f / 255.0f));
            const float cdf_min = std_normal_cdf_approx(inv_std * (c - 1.0f / 255.0f));
            const float lp = xv < -0.999f ? logf(fmaxf(cdf_plus, 1e-12f))
                                          : (xv > 0.999f ? logf(fmaxf(1.0f - cdf_min, 1e-12f)) : logf(fmaxf(cdf_plus - cdf_min, 1e-12f)));
            term = -lp;
        } else {
            const float tm = __fadd_rn(__fmul_rn(a1, xv), __fmul_rn(a2, xt));
            const float d = tm - pred;
            term = 0.5f * (kl0 + (d * d) * inv_var);
        }
        acc += term;
        if (eps) { const float e = eps[base + i] - eh; sq += e * e; }
    }
    acc = block_sum(acc, red);
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) {
        float* part = partials + ((long long)b * ns + sl) * 2;
        __hip_atomic_store(part, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == (unsigned)ns - 1u;
        if (last) {
            float a = 0.f, s2 = 0.f;
            for (int k = 0; k < ns; ++k) {
                a += __hip_atomic_load(partials + ((long long)b * ns + k) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(partials + ((long long)b * ns + k) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            vlb[b] = (a / (float)per) / 0.6931471805599453f;
            if (sqerr) sqerr[b] = s2;
            __hip_atomic_store(counters + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int vlb_slices(int B, long long per) {
    int ns = (int)((256 + B - 1) / B);                    // fill the chip ...
    const long long cap = (per + 4095) / 4096;            // ... but keep at least ~4 elements per thread and slice
    if (ns > cap) ns = (int)cap;
    if (ns > 64) ns = 64;
    return ns < 1 ? 1 : ns;
}

int p_sample_update(float* x, const float* eps_hat, const float* noise, long long noise_step_stride, int t_first,
                    const int64_t* t, const float* c_recip, const float* c_recipm1, const float* c1, const float* c2,
                    const float* sigma, int B, long long per, uint64_t seed, uint32_t stream_id, hipStream_t st,
                    const int64_t* chain_state, int64_t* dec_counter) {
    DDK_REQUIRE(x && eps_hat && t && c_recip && c_recipm1 && c1 && c2 && sigma, "p_sample_update: null pointer");
    DDK_REQUIRE(B > 0 && per > 0 && per % 4 == 0, "p_sample_update: per-sample element count must be a multiple of 4");
    DDK_REQUIRE(aligned16(x) && aligned16(eps_hat) && aligned16(noise) && noise_step_stride % 4 == 0, "p_sample_update: alignment");
    const long long total4 = B * per / 4;
    hipLaunchKernelGGL(p_sample_kernel, dim3(grid1d(total4)), dim3(256), 0, st, x, eps_hat, noise, noise_step_stride, t_first, t,
                       c_recip, c_recipm1, c1, c2, sigma, per / 4, total4, seed, stream_id, chain_state, dec_counter);
    return check_launch("p_sample_kernel");
}

int randn(float* out, long long n, uint64_t seed, uint32_t step, uint32_t stream_id, hipStream_t st) {
    DDK_REQUIRE(out && n > 0 && aligned16(out), "randn: arguments");
    const long long n4 = (n + 3) / 4;
    hipLaunchKernelGGL(randn_kernel, dim3(grid1d(n4)), dim3(256), 0, st, out, n4, n, seed, step, stream_id);
    return check_launch("randn_kernel");
}

}  // namespace ddk

using namespace ddk;


// Sampler output stage (utils/eval_helpers.py:37-41 + utils/utils.py:16-24): per-image min / max over C*H*W, then
// out[b][h][w][c] = ((x[b][c][h][w] - lo) / (hi - lo)) * 255 -- the same three fp32 operations, in the same order, as
// the reference's torch expression (this file is compiled with -ffp-contract=off), written NHWC like its np.moveaxis.
// One workgroup per image: pass 1 reduces, pass 2 normalises and transposes.
__global__ __launch_bounds__(1024) void fix_samples_kernel(const float* __restrict__ x, float* __restrict__ out, int C, long long HW) {
    __shared__ float red[32];
    const long long per = (long long)C * HW;
    const float* xb = x + (long long)blockIdx.x * per;
    float* ob = out + (long long)blockIdx.x * per;
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = threadIdx.x; i < per; i += 1024) {
        const float v = xb[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    hi = block_max(hi, red);
    lo = -block_max(-lo, red);
    const float range = hi - lo;
    for (long long p = threadIdx.x; p < HW; p += 1024)
        for (int c = 0; c < C; ++c) ob[p * C + c] = ((xb[c * HW + p] - lo) / range) * 255.0f;
}

extern "C" {

int ddk_q_sample(const float* x, const float* eps, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* out,
                 int B, long long per, ddk_stream_t s) {
    DDK_REQUIRE(x && eps && t && sqrt_acp && sqrt_1m_acp && out, "q_sample: null pointer");
    DDK_REQUIRE(B > 0 && per > 0 && per % 4 == 0, "q_sample: per-sample element count must be a multiple of 4");
    DDK_REQUIRE(aligned16(x) && aligned16(eps) && aligned16(out), "q_sample: alignment");
    const long long total4 = B * per / 4;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid1d(total4)), dim3(256), 0, as_stream(s), x, eps, t, sqrt_acp, sqrt_1m_acp, out,
                       per / 4, total4);
    return check_launch("q_sample_kernel");
}

int ddk_p_sample_update(float* x, const float* eps_hat, const float* noise, const int64_t* t, const float* c_recip,
                        const float* c_recipm1, const float* c1, const float* c2, const float* sigma, int B, long long per,
                        uint64_t seed, uint32_t stream_id, ddk_stream_t s) {
    return p_sample_update(x, eps_hat, noise, 0, 0, t, c_recip, c_recipm1, c1, c2, sigma, B, per, seed, stream_id, as_stream(s));
}

int ddk_final_tail(const float* raw, const float* partials, int tiles_per_image, const float* gamma, const float* beta, float eps,
                   const float* w, const float* bias, int n_out, float* eps_out, float* x, const float* noise, const int64_t* t,
                   const float* c_recip, const float* c_recipm1, const float* c1, const float* c2, const float* sigma, uint64_t seed,
                   uint32_t stream_id, int B, int HW, int C, int groups, ddk_stream_t s) {
    return final_tail(raw, partials, tiles_per_image, gamma, beta, eps, w, bias, n_out, eps_out, x, noise, 0, 0, t, c_recip, c_recipm1, c1,
                      c2, sigma, nullptr, seed, stream_id, B, HW, C, groups, as_stream(s));
}

int ddk_randn(float* out, long long n, uint64_t seed, uint32_t step, uint32_t stream_id, ddk_stream_t s) {
    return randn(out, n, seed, step, stream_id, as_stream(s));
}

int ddk_fix_samples(const float* x_nchw, float* out_nhwc, int B, int C, int H, int W, ddk_stream_t s) {
    DDK_REQUIRE(x_nchw && out_nhwc && B > 0 && C > 0 && H > 0 && W > 0, "fix_samples: arguments");
    hipLaunchKernelGGL(fix_samples_kernel, dim3(B), dim3(1024), 0, as_stream(s), x_nchw, out_nhwc, C, (long long)H * W);
    return check_launch("fix_samples_kernel");
}

size_t ddk_vlb_terms_workspace_bytes(int B, long long per) {
    if (B <= 0 || per <= 0) return 0;
    return ((size_t)B + (size_t)B * vlb_slices(B, per) * 2) * sizeof(float);
}

int ddk_vlb_terms(const float* x, const float* x_t, const float* eps_hat, const float* eps, const int64_t* t, const float* c_recip,
                  const float* c_recipm1, const float* c1, const float* c2, const float* post_logvar, float* vlb, float* sqerr, int B,
                  long long per, void* workspace, size_t workspace_bytes, ddk_stream_t s) {
    DDK_REQUIRE(x && x_t && eps_hat && t && c_recip && c_recipm1 && c1 && c2 && post_logvar && vlb, "vlb_terms: null pointer");
    DDK_REQUIRE((eps == nullptr) == (sqerr == nullptr), "vlb_terms: eps and sqerr go together");
    DDK_REQUIRE(B > 0 && per > 0, "vlb_terms: B / per");
    DDK_REQUIRE(workspace && workspace_bytes >= ddk_vlb_terms_workspace_bytes(B, per) && (reinterpret_cast<uintptr_t>(workspace) & 3u) == 0,
                "vlb_terms: workspace (ddk_vlb_terms_workspace_bytes)");
    const int ns = vlb_slices(B, per);
    unsigned* counters = static_cast<unsigned*>(workspace);
    float* partials = static_cast<float*>(workspace) + B;
    DDK_HIP(hipMemsetAsync(counters, 0, (size_t)B * sizeof(unsigned), as_stream(s)));      // they re-arm themselves; a fresh workspace starts at 0
    hipLaunchKernelGGL(vlb_terms_kernel, dim3(ns, B), dim3(1024), 0, as_stream(s), x, x_t, eps_hat, eps, t, c_recip, c_recipm1, c1, c2,
                       post_logvar, vlb, sqerr, per, ns, counters, partials);
    return check_launch("vlb_terms_kernel");
}

int ddk_sq_err_sum(const float* a, const float* b, float* per_sample, int B, long long per, ddk_stream_t s) {
    DDK_REQUIRE(a && b && per_sample && B > 0 && per > 0 && per % 4 == 0, "sq_err_sum: arguments (per % 4 == 0)");
    DDK_REQUIRE(aligned16(a) && aligned16(b), "sq_err_sum: alignment");
    hipLaunchKernelGGL(sq_err_sum_kernel, dim3(B), dim3(1024), 0, as_stream(s), a, b, per_sample, per / 4);
    return check_launch("sq_err_sum_kernel");
}
}
