// norm_act.hip -- GroupNorm+Mish(+time shift)(+residual), channel LayerNorm, and the small elementwise /
// resampling kernels of the dDDPM ConvResNet.  All HBM/L2-bound: 16-byte accesses, wavefront-shuffle
// reductions, one read + one write per element.
//
// Reference sites: models/unet/blocks.py:57-60 (LayerNorm), :79-80 (GroupNorm, Mish), :106-115 (time
// shift + residual), models/downsampled/convblocks.py:110-130 (Mish, avg_pool2d, nearest x2),
// models/diffusion/dddpm.py:99,110 (tanh).
#include "ddk_internal.h"

namespace ddk {

// ------------------------------------------------------------------------------------------------
// GroupNorm(groups, eps, affine) -> Mish -> (+ temb[b][c]) -> (+ addend), NHWC.
// One workgroup of NT threads per (b, group): the group's HW x cpg slab (<= NT * 4 * VPT floats) is read once
// into registers, mean and biased variance are two block reductions over the registers (two-pass, like
// torch's native_group_norm), and the result is written once.
// The input may still be in split-K form: `nslab` fp32 slabs (stride `slab_stride`) whose fixed-order sum plus
// the conv bias `cbias` is the tensor -- the conv's reduce pass is then folded into this load (no extra
// kernel, no extra round trip).
template <int VPT, int NT>
__global__ __launch_bounds__(NT) void gn_mish_resident_kernel(const float* __restrict__ x, int nslab, long long slab_stride,
                                                              const float* __restrict__ cbias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ temb,
                                                              int temb_stride, const long long* __restrict__ temb_rows,
                                                              const float* __restrict__ addend,
                                                              float* __restrict__ out, int HW, int C, int groups, float eps) {
    __shared__ float red[32];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const int upr = cpg >> 2;                 // float4 units per pixel row of this group
    const int units = HW * upr;
    const long long base = (long long)b * HW * C + g * cpg;

    float4 v[VPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u < units) {
            const int row = div_upr(u, upr), cu = u - row * upr;
            const long long o = base + (long long)row * C + cu * 4;
            float4 a = *reinterpret_cast<const float4*>(x + o);
            for (int k = 1; k < nslab; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(x + k * slab_stride + o);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            if (cbias) {
                const float4 t = *reinterpret_cast<const float4*>(cbias + g * cpg + cu * 4);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            v[i] = a;
            s += (a.x + a.y) + (a.z + a.w);
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
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u < units) {
            const int row = div_upr(u, upr), cu = u - row * upr;
            const int c0 = g * cpg + cu * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
            const float4 be = *reinterpret_cast<const float4*>(beta + c0);
            float4 y;
            y.x = mish_f((v[i].x - mean) * rstd * ga.x + be.x);
            y.y = mish_f((v[i].y - mean) * rstd * ga.y + be.y);
            y.z = mish_f((v[i].z - mean) * rstd * ga.z + be.z);
            y.w = mish_f((v[i].w - mean) * rstd * ga.w + be.w);
            if (temb) {
                const long long tr = temb_rows ? temb_rows[b] : b;   // sampler: row = timestep of sample b in the precomputed table
                const float4 t = *reinterpret_cast<const float4*>(temb + tr * temb_stride + c0);
                y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
            }
            const long long o = base + (long long)row * C + cu * 4;
            if (addend) {
                const float4 r = *reinterpret_cast<const float4*>(addend + o);
                y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
            }
            *reinterpret_cast<float4*>(out + o) = y;
        }
    }
}

// GroupNorm+Mish when the producing conv already left per-tile statistics (conv3x3_wino_kernel, gn_part): `np` tiles of 128
// pixels per image, each {mean, M2 about that mean} per group.  Every workgroup first merges the np partials of its image's
// groups in fixed order (Chan et al.: equal counts), then streams its share of the image: one read, one write.
//
// RC: the addend is a 1x1 conv of a narrow tensor (the first ResnetBlock's res_conv, blocks.py:103,115: C_in = image / latent
// channels <= 8) evaluated on the fly -- addend[pix][c] = rb[c] + sum_k rx[pix][k] rw[c][k] -- instead of a [M][C] tensor that
// a separate launch would write and this one read back (2 x 16.8 MB at cfg4).  A thread's channel quad is the same for every
// unit it touches (host: 256 % (C/4) == 0), so its 4 x C_in weights live in registers.
struct Res1x1 {
    const float* x;     // [B*HW][cin]
    const float* w;     // [C][ld], row c = output channel, first cin entries used
    const float* b;     // [C] or nullptr
    int cin, ld;
};

template <bool RC>
__global__ __launch_bounds__(256) void gn_apply_parts_kernel(const float* __restrict__ x, const float2* __restrict__ part, int np,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ temb, int temb_stride,
                                                             const long long* __restrict__ temb_rows, const float* __restrict__ addend,
                                                             float* __restrict__ out, int HW, int C, int cpg, float eps, int wg_per_image,
                                                             int units_per_wg, const Res1x1 rc) {
    __shared__ float2 mr[128];
    __shared__ float2 sp[1024];
    const int b = blockIdx.x / wg_per_image, chunk = blockIdx.x - b * wg_per_image;
    const long long tr = temb ? (temb_rows ? temb_rows[b] : b) : 0;          // first: the shift row's load depends on it
    const int G = C / cpg;
    const int upr = C >> 2;                      // float4 units per pixel
    const int units = HW * upr;
    const int u_begin = chunk * units_per_wg, u_end = min(units, (chunk + 1) * units_per_wg);
    const long long base = (long long)b * HW * C;
    // Everything that does not depend on the statistics is requested BEFORE they are merged, so the launch pays one memory
    // latency, not a chain of them: the first batch of x, the per-tile partials, and -- when every unit of this thread has the
    // same channel quad (256 % (C/4) == 0: all reference widths) -- gamma, beta and the time shift.
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = u_begin + threadIdx.x + k * 256;
        if (u < u_end) v[k] = *reinterpret_cast<const float4*>(x + base + (long long)u * 4);
    }
    const float2* pb = part + (long long)b * np * G;
    for (int i = threadIdx.x; i < np * G; i += 256) sp[i] = pb[i];          // [tile][group], host guarantees np * G <= 1024
    const bool same_c0 = (256 % upr) == 0 && (units_per_wg % upr) == 0;      // wave-uniform
    const int c0_fixed = ((u_begin + (int)threadIdx.x) - div_upr(u_begin + (int)threadIdx.x, upr) * upr) << 2;
    float4 ga_f = make_float4(0.f, 0.f, 0.f, 0.f), be_f = ga_f, t_f = ga_f;
    if (same_c0) {
        ga_f = *reinterpret_cast<const float4*>(gamma + c0_fixed);
        be_f = *reinterpret_cast<const float4*>(beta + c0_fixed);
        if (temb) t_f = *reinterpret_cast<const float4*>(temb + tr * temb_stride + c0_fixed);
    }
    float rw[4][8], rb[4];
    float xr[4][8];         // RC: the narrow input rows of this iteration's four pixels, requested together with x
    auto load_rows = [&](int u0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * 256;
            const float* xs = rc.x + ((long long)b * HW + div_upr(u < u_end ? u : u_begin, upr)) * rc.cin;
            if (rc.cin == 8) {      // 32-byte rows: two 16-byte loads (rc.x is 16-byte aligned, host-checked)
                const float4 a = *reinterpret_cast<const float4*>(xs), c = *reinterpret_cast<const float4*>(xs + 4);
                xr[k][0] = a.x; xr[k][1] = a.y; xr[k][2] = a.z; xr[k][3] = a.w;
                xr[k][4] = c.x; xr[k][5] = c.y; xr[k][6] = c.z; xr[k][7] = c.w;
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) xr[k][kk] = kk < rc.cin ? xs[kk] : 0.f;
            }
        }
    };
    __shared__ __attribute__((aligned(16))) float rws[RC ? 256 * 8 : 4];     // RC: the [C][8] 1x1 weights (host: C <= 256)
    if constexpr (RC) {
        // through LDS (a few coalesced loads per thread): a workgroup handles only 4-8 outputs per thread, so 32 scattered weight
        // loads per thread would be most of its memory instructions
        for (int i = threadIdx.x; i < C * 8; i += 256) rws[i] = (i & 7) < rc.cin ? rc.w[(long long)(i >> 3) * rc.ld + (i & 7)] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = rc.b ? rc.b[c0_fixed + j] : 0.f;
        load_rows(u_begin + threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float ms = 0.f;
        for (int i = 0; i < np; ++i) ms += sp[i * G + threadIdx.x].x;
        const float mean = ms / (float)np;
        float m2 = 0.f, d2 = 0.f;
        for (int i = 0; i < np; ++i) {
            const float2 t = sp[i * G + threadIdx.x];
            m2 += t.y;
            d2 += (t.x - mean) * (t.x - mean);
        }
        const float n_i = 128.0f * (float)cpg;
        const float var = (m2 + n_i * d2) / ((float)np * n_i);
        mr[threadIdx.x] = make_float2(mean, 1.0f / sqrtf(var + eps));
    }
    if constexpr (RC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(rws + (c0_fixed + j) * 8), c = *reinterpret_cast<const float4*>(rws + (c0_fixed + j) * 8 + 4);
            rw[j][0] = a.x; rw[j][1] = a.y; rw[j][2] = a.z; rw[j][3] = a.w;
            rw[j][4] = c.x; rw[j][5] = c.y; rw[j][6] = c.z; rw[j][7] = c.w;
        }
    }
    __syncthreads();
    for (int u0 = u_begin + threadIdx.x; u0 < u_end; u0 += 1024) {
        if (u0 != u_begin + (int)threadIdx.x) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * 256;
                if (u < u_end) v[k] = *reinterpret_cast<const float4*>(x + base + (long long)u * 4);
            }
            if (RC) load_rows(u0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * 256;
            if (u >= u_end) continue;
            const int c0 = same_c0 ? c0_fixed : (u - div_upr(u, upr) * upr) << 2;
            const float2 st = mr[c0 / cpg];
            float4 ga = ga_f, be = be_f, ts = t_f;
            if (!same_c0) {
                ga = *reinterpret_cast<const float4*>(gamma + c0);
                be = *reinterpret_cast<const float4*>(beta + c0);
                if (temb) ts = *reinterpret_cast<const float4*>(temb + tr * temb_stride + c0);
            }
            float4 y;
            y.x = mish_f((v[k].x - st.x) * st.y * ga.x + be.x) + ts.x;
            y.y = mish_f((v[k].y - st.x) * st.y * ga.y + be.y) + ts.y;
            y.z = mish_f((v[k].z - st.x) * st.y * ga.z + be.z) + ts.z;
            y.w = mish_f((v[k].w - st.x) * st.y * ga.w + be.w) + ts.w;
            const long long o = base + (long long)u * 4;
            if (RC) {
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = rb[j];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) a += xr[k][kk] * rw[j][kk];
                    r[j] = a;
                }
                y.x += r[0]; y.y += r[1]; y.z += r[2]; y.w += r[3];
            } else if (addend) {
                const float4 r = *reinterpret_cast<const float4*>(addend + o);
                y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
            }
            *reinterpret_cast<float4*>(out + o) = y;
        }
    }
}

int groupnorm_mish_parts(const float* x, const float* part, int np, const float* gamma, const float* beta, const float* temb,
                         int temb_stride, const float* addend, float* out, int B, int HW, int C, int groups, float eps, hipStream_t st,
                         const long long* temb_rows, const float* rc_x, const float* rc_w, const float* rc_b, int rc_cin, int rc_ld) {
    DDK_REQUIRE(x && part && gamma && beta && out, "groupnorm_mish_partials: null pointer");
    DDK_REQUIRE(B > 0 && HW > 0 && groups > 0 && groups <= 128 && C % groups == 0 && (C / groups) % 4 == 0,
                "groupnorm_mish_partials: C must split into <= 128 groups of a multiple of 4 channels");
    DDK_REQUIRE(np > 0 && HW == np * 128 && np * groups <= 1024, "groupnorm_mish_partials: tiles_per_image * 128 must equal H*W "
                "(and tiles_per_image * groups <= 1024)");
#ifdef DDK_HOST_SANITIZE
    {
        const long long e = (long long)B * HW * C * 4;
        san::extent("groupnorm_parts x", x, e); san::extent("groupnorm_parts out", out, e); san::extent("groupnorm_parts addend", addend, e);
        san::extent("groupnorm_parts partials", part, (long long)B * np * groups * 8);
        san::extent("groupnorm_parts gamma", gamma, (long long)C * 4); san::extent("groupnorm_parts beta", beta, (long long)C * 4);
        san::extent("groupnorm_parts res x", rc_x, (long long)B * HW * rc_cin * 4);
    }
#endif
    DDK_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta) && aligned16(addend) && aligned16(temb) &&
                    temb_stride % 4 == 0, "groupnorm_mish_partials: pointers must be 16-byte aligned");
    const int units = HW * (C / 4);
    int wpi = (int)ceil_div(units, 1024);                       // 1024 float4 per workgroup ...
    while ((long long)wpi * B > 2048 && wpi > 1) wpi = (wpi + 1) / 2;   // ... unless that makes more than ~8 workgroups per CU
    if (rc_x && (long long)wpi * B >= 1024 && wpi > 1) wpi = (wpi + 1) / 2;    // the 1x1 addend has a per-workgroup prologue: 2 rounds each
    const int upw = (int)(ceil_div(ceil_div(units, wpi), 256) * 256);
    wpi = (int)ceil_div(units, upw);
    Res1x1 rc{rc_x, rc_w, rc_b, rc_cin, rc_ld};
    if (rc_x) {
        const int upr = C / 4;
        DDK_REQUIRE(!addend && rc_w && rc_cin >= 1 && rc_cin <= 8 && rc_ld >= rc_cin, "groupnorm_mish_partials: 1x1 addend needs "
                    "1 <= C_in <= 8, its weights, and no tensor addend");
        DDK_REQUIRE(C <= 256 && 256 % upr == 0, "groupnorm_mish_partials: 1x1 addend needs C <= 256 and C/4 to divide 256");
        DDK_REQUIRE(aligned16(rc_x) && (reinterpret_cast<uintptr_t>(rc_w) & 3u) == 0, "groupnorm_mish_partials: 1x1 addend alignment");
        hipLaunchKernelGGL(gn_apply_parts_kernel<true>, dim3((unsigned)(B * wpi)), dim3(256), 0, st, x, reinterpret_cast<const float2*>(part),
                           np, gamma, beta, temb, temb_stride, temb_rows, addend, out, HW, C, C / groups, eps, wpi, upw, rc);
    } else {
        hipLaunchKernelGGL(gn_apply_parts_kernel<false>, dim3((unsigned)(B * wpi)), dim3(256), 0, st, x, reinterpret_cast<const float2*>(part),
                           np, gamma, beta, temb, temb_stride, temb_rows, addend, out, HW, C, C / groups, eps, wpi, upw, rc);
    }
    return check_launch("gn_apply_parts_kernel");
}

// Large slabs (full-resolution DDPM, 256x256): stats by `nsplit` workgroups per (b, group) as Welford
// partials (count, mean, M2) combined in a fixed order, then a grid-wide apply pass.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                         int groups, int nsplit) {
    __shared__ float red[32];
    const int bg = blockIdx.x, sp = blockIdx.y;
    const int b = bg / groups, g = bg % groups;
    const int cpg = C / groups, upr = cpg >> 2;
    const long long units = (long long)HW * upr;
    const long long per = (units + nsplit - 1) / nsplit;
    const long long u0 = sp * per, u1 = (u0 + per < units) ? u0 + per : units;
    const long long base = (long long)b * HW * C + g * cpg;
    float s = 0.f;
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        const long long row = div_upr(u, upr);
        const int cu = (int)(u - row * upr);
        const float4 v = *reinterpret_cast<const float4*>(x + base + row * C + cu * 4);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float cnt = (float)((u1 > u0 ? u1 - u0 : 0) * 4);
    const float mean = cnt > 0 ? block_sum(s, red) / cnt : 0.f;
    float q = 0.f;
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        const long long row = div_upr(u, upr);
        const int cu = (int)(u - row * upr);
        const float4 v = *reinterpret_cast<const float4*>(x + base + row * C + cu * 4);
        const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + bb * bb) + (c * c + d * d);
    }
    const float m2 = block_sum(q, red);
    if (threadIdx.x == 0) {
        float* o = part + ((long long)bg * nsplit + sp) * 3;
        o[0] = cnt; o[1] = mean; o[2] = m2;
    }
}

// The same partials from ONE pass over full pixel rows: workgroup (b, sp) walks pixels [sp * per, (sp + 1) * per) of image b with a
// thread per channel quad (C/4 must divide 256), so every load instruction covers whole 128-byte lines (the per-group kernel
// above reads 64-byte pieces of them, twice).  A thread keeps shifted sums about its first sample, turns them into {n, mean, M2},
// and one thread per group merges the 256 / groups thread partials of its channels in a fixed order (Chan et al.).
__global__ __launch_bounds__(256) void gn_partial_rows_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                              int groups, int nsplit) {
    __shared__ float tp[256][3];
    const int b = blockIdx.x, sp = blockIdx.y;
    const int c4 = C >> 2, cpg4 = (C / groups) >> 2, R = 256 / c4;
    const int cq = threadIdx.x % c4, r = threadIdx.x / c4;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    const float4* xp = reinterpret_cast<const float4*>(x + (long long)b * HW * C) + cq;
    float K = 0.f, s1 = 0.f, s2 = 0.f;
    int n = 0, p = p0 + r;
    if (p < p1) K = xp[(long long)p * c4].x;
#pragma unroll 4
    for (; p < p1; p += R) {
        const float4 v = xp[(long long)p * c4];
        const float a = v.x - K, bb = v.y - K, c = v.z - K, d = v.w - K;
        s1 += (a + bb) + (c + d);
        s2 += (a * a + bb * bb) + (c * c + d * d);
        n += 4;
    }
    const float fn = (float)n;
    tp[threadIdx.x][0] = fn;
    tp[threadIdx.x][1] = n ? K + s1 / fn : 0.f;
    tp[threadIdx.x][2] = n ? fmaxf(s2 - s1 * s1 / fn, 0.f) : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x;
        float cn = 0.f, mean = 0.f, m2 = 0.f;
        for (int rr = 0; rr < R; ++rr)
            for (int q = 0; q < cpg4; ++q) {
                const float* t = tp[rr * c4 + g * cpg4 + q];
                const float nb = t[0];
                if (nb > 0) {
                    const float tot = cn + nb, delta = t[1] - mean;
                    mean += delta * (nb / tot);
                    m2 += t[2] + delta * delta * (cn * nb / tot);
                    cn = tot;
                }
            }
        float* o = part + ((long long)(b * groups + g) * nsplit + sp) * 3;
        o[0] = cn; o[1] = mean; o[2] = m2;
    }
}
static bool gn_partial_rows_ok(int C, int groups) {
    const int c4 = C >> 2;
    return c4 > 0 && c4 <= 256 && 256 % c4 == 0 && groups <= 256 && (C / groups) % 4 == 0;
}
static void launch_gn_partials(const float* x, float* part, int B, int HW, int C, int groups, int ns, hipStream_t st) {
    if (gn_partial_rows_ok(C, groups))
        hipLaunchKernelGGL(gn_partial_rows_kernel, dim3(B, ns), dim3(256), 0, st, x, part, HW, C, groups, ns);
    else
        hipLaunchKernelGGL(gn_partial_kernel, dim3(B * groups, ns), dim3(256), 0, st, x, part, HW, C, groups, ns);
}

// The apply pass: grid (workgroups per image, B).  A workgroup first merges the Welford partials of its image's groups (thread g:
// group g, Chan et al., fixed order) into (mean, rstd) in LDS -- the first version did that merge, up to 64 partials, and two
// 64-bit divides PER float4 and ran at 0.9 TB/s (571 us for the 8 x 256 x 256 x 128 tensor, a third of the full-resolution reverse
// step).  FIXED: the grid stride is a multiple of C/4, so a thread keeps ONE channel quad for its whole walk and its affine /
// shift / statistics operands stay in registers.
template <bool FIXED>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ part, int nsplit,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ temb, int temb_stride,
                                                       const long long* __restrict__ temb_rows,
                                                       const float* __restrict__ addend, float* __restrict__ out, int HW, int C,
                                                       int groups, float eps, unsigned per_image4) {
    __shared__ float mr[64][2];
    const int b = blockIdx.y;
    if ((int)threadIdx.x < groups) {
        const float* pp = part + ((long long)(b * groups + threadIdx.x) * nsplit) * 3;
        float n = pp[0], mean = pp[1], m2 = pp[2];
        for (int s = 1; s < nsplit; ++s) {
            const float nb = pp[3 * s], mb = pp[3 * s + 1], qb = pp[3 * s + 2];
            if (nb > 0) {
                const float tot = n + nb, delta = mb - mean;
                mean += delta * (nb / tot);
                m2 += qb + delta * delta * (n * nb / tot);
                n = tot;
            }
        }
        mr[threadIdx.x][0] = mean;
        mr[threadIdx.x][1] = 1.0f / sqrtf(m2 / n + eps);
    }
    __syncthreads();
    const unsigned c4 = (unsigned)C >> 2, cpg = (unsigned)(C / groups);
    const unsigned stride = gridDim.x * 256u;
    const long long tr = temb ? (temb_rows ? temb_rows[b] : b) : 0;
    const float4* xi = reinterpret_cast<const float4*>(x) + (long long)b * per_image4;
    const float4* ai = addend ? reinterpret_cast<const float4*>(addend) + (long long)b * per_image4 : nullptr;
    float4* oi = reinterpret_cast<float4*>(out) + (long long)b * per_image4;
    unsigned idx = blockIdx.x * 256u + threadIdx.x;
    float4 sc, sh;      // y = mish((v - mean) * sc + sh) + tv  (the subtraction first: |mean| >> std must not cancel)
    float mean = 0.f;
    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
    auto operands = [&](unsigned cq) {
        const unsigned c0 = cq * 4, g = c0 / cpg;
        const float rstd = mr[g][1];
        mean = mr[g][0];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c0);
        sh = *reinterpret_cast<const float4*>(beta + c0);
        sc = make_float4(rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w);
        if (temb) tv = *reinterpret_cast<const float4*>(temb + tr * temb_stride + c0);
    };
    if (FIXED) operands(idx % c4);
#pragma unroll 4
    for (; idx < per_image4; idx += stride) {
        if (!FIXED) operands(idx % c4);
        const float4 v = xi[idx];
        float4 y;
        y.x = mish_f((v.x - mean) * sc.x + sh.x) + tv.x;
        y.y = mish_f((v.y - mean) * sc.y + sh.y) + tv.y;
        y.z = mish_f((v.z - mean) * sc.z + sh.z) + tv.z;
        y.w = mish_f((v.w - mean) * sc.w + sh.w) + tv.w;
        if (ai) {
            const float4 r = ai[idx];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        oi[idx] = y;
    }
}

static int gn_nsplit(int HW, int cpg) {
    const long long n = (long long)HW * cpg;
    if (n <= 256LL * 4 * 16) return 0;  // resident path
    long long s = n / (256LL * 4 * 16);
    return (int)(s > 64 ? 64 : (s < 2 ? 2 : s));
}

// train-mode callers (backward.hip): same split rule and the same Welford partial kernel
int gn_train_nsplit(int HW, int cpg) { return gn_nsplit(HW, cpg); }
int gn_stats_partials(const float* x, float* part, int B, int HW, int C, int groups, int ns, hipStream_t st) {
    launch_gn_partials(x, part, B, HW, C, groups, ns, st);
    return check_launch("gn_partial_kernel");
}

size_t groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
    if (groups <= 0 || C % groups) return 0;
    const int ns = gn_nsplit(HW, C / groups);
    return (size_t)B * groups * ns * 3 * sizeof(float);
}

int groupnorm_mish_ex(const float* x, int nslab, long long slab_stride, const float* cbias, const float* gamma,
                      const float* beta, const float* temb, int temb_stride, const float* addend, float* out, int B, int HW, int C,
                      int groups, float eps, void* ws, size_t ws_bytes, hipStream_t st, const long long* temb_rows) {
    DDK_REQUIRE(x && gamma && beta && out, "groupnorm: null pointer");
    DDK_REQUIRE(B > 0 && HW > 0 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0,
                "groupnorm: C/groups must be a multiple of 4");
    DDK_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta) && aligned16(temb) && aligned16(addend) &&
                    aligned16(cbias),
                "groupnorm: alignment");
    DDK_REQUIRE(temb == nullptr || temb_stride % 4 == 0, "groupnorm: temb_stride % 4");
    DDK_REQUIRE(nslab >= 1 && (nslab == 1 || slab_stride % 4 == 0), "groupnorm: slabs");
#ifdef DDK_HOST_SANITIZE
    {
        const long long e = (long long)B * HW * C * 4;
        san::extent("groupnorm x (slabs)", x, (long long)(nslab - 1) * slab_stride * 4 + e);
        san::extent("groupnorm out", out, e); san::extent("groupnorm addend", addend, e);
        san::extent("groupnorm gamma", gamma, (long long)C * 4); san::extent("groupnorm beta", beta, (long long)C * 4);
        san::extent("groupnorm conv bias", cbias, (long long)C * 4); san::extent("groupnorm workspace", ws, (long long)ws_bytes);
    }
#endif
    const int cpg = C / groups;
    const int ns = gn_nsplit(HW, cpg);
    if (ns == 0) {
        const int units = HW * (cpg / 4);
        dim3 grid(B * groups);
#define GN_CASE(V, NT)                                                                                                  \
    hipLaunchKernelGGL((gn_mish_resident_kernel<V, NT>), grid, dim3(NT), 0, st, x, nslab, slab_stride, cbias, gamma, beta, temb, \
                       temb_stride, temb_rows, addend, out, HW, C, groups, eps)
        // big slabs: 1024 threads (16 waves per CU keep enough loads in flight); small ones: 256.  (One WAVE per group with
        // shuffle-only reductions was tried for the 4x4 / 8x8 maps and lost -- 8.5 vs 5.9 us, 17 vs 7.3 us: a single wave per
        // CU has too few loads in flight for the split-K slabs.)
        if (units <= 256) GN_CASE(1, 256);
        else if (units <= 512) GN_CASE(2, 256);
        else if (units <= 1024) GN_CASE(1, 1024);
        else if (units <= 2048) GN_CASE(2, 1024);
        else GN_CASE(4, 1024);
#undef GN_CASE
        return check_launch("gn_mish_resident_kernel");
    }
    DDK_REQUIRE(nslab == 1 && cbias == nullptr, "groupnorm: split-K input is only supported on the register-resident path");
    const size_t need = (size_t)B * groups * ns * 3 * sizeof(float);
    if (!ws || ws_bytes < need) {
        set_error("groupnorm: workspace too small (%zu < %zu)", ws_bytes, need);
        return DDK_ERR_WORKSPACE;
    }
    float* part = static_cast<float*>(ws);
    launch_gn_partials(x, part, B, HW, C, groups, ns, st);
    DDK_TRY(check_launch("gn_partial_kernel"));
    DDK_REQUIRE(groups <= 64 && (long long)HW * (C / 4) < (1LL << 31), "groupnorm: large-slab path takes <= 64 groups, < 2^31 float4 per image");
    const unsigned per_image4 = (unsigned)((long long)HW * (C / 4));
    // ~4096 workgroups over the batch, at least 8 float4 per thread
    long long bpi = ceil_div(4096, B);
    const long long max_bpi = ceil_div(per_image4, 256 * 8);
    if (bpi > max_bpi) bpi = max_bpi;
    if (bpi < 1) bpi = 1;
    const dim3 grid((unsigned)bpi, (unsigned)B);
    if ((bpi * 256) % (C / 4) == 0)
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, st, x, part, ns, gamma, beta, temb, temb_stride, temb_rows, addend, out,
                           HW, C, groups, eps, per_image4);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, st, x, part, ns, gamma, beta, temb, temb_stride, temb_rows, addend, out,
                           HW, C, groups, eps, per_image4);
    return check_launch("gn_apply_kernel");
}

int groupnorm_mish(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                   const float* addend, float* out, int B, int HW, int C, int groups, float eps, void* ws, size_t ws_bytes,
                   hipStream_t st, const long long* temb_rows) {
    return groupnorm_mish_ex(x, 1, 0, nullptr, gamma, beta, temb, temb_stride, addend, out, B, HW, C, groups, eps, ws, ws_bytes, st,
                             temb_rows);
}

// ------------------------------------------------------------------------------------------------
// Generic forms for widths that are not multiples of 32 (reference blocks.py:75: GroupNorm(8, C) accepts any C % 8 == 0, e.g.
// unet_chan = 24).  Activations keep a padded pitch CP = pad32(C) whose padding is zero, so every conv kernel runs unchanged
// (zero weight rows / columns); only the normalisations see the real channel count.  Correctness first: three plain passes, no tuning.
// One workgroup per (image, group); workgroup `groups` of an image zeroes the padding channels of the output.
__global__ __launch_bounds__(256) void gn_mish_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ temb, int temb_stride,
                                                              const long long* __restrict__ temb_rows, const float* __restrict__ addend,
                                                              float* __restrict__ out, int HW, int CP, int C, int groups, float eps) {
    __shared__ float red[32];
    const int b = blockIdx.x / (groups + 1), g = blockIdx.x % (groups + 1), tid = threadIdx.x;
    const float* xb = x + (long long)b * HW * CP;
    float* ob = out + (long long)b * HW * CP;
    if (g == groups) {
        const int np = CP - C;
        for (long long e = tid; e < (long long)HW * np; e += 256) ob[(e / np) * CP + C + (int)(e % np)] = 0.f;
        return;
    }
    const int cpg = C / groups, c0 = g * cpg;
    const long long n = (long long)HW * cpg;
    float s = 0.f;
    for (long long e = tid; e < n; e += 256) s += xb[(e / cpg) * CP + c0 + (int)(e % cpg)];
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (long long e = tid; e < n; e += 256) {
        const float d = xb[(e / cpg) * CP + c0 + (int)(e % cpg)] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + eps);
    const long long trow = temb ? (temb_rows ? temb_rows[b] : b) : 0;
    for (long long e = tid; e < n; e += 256) {
        const long long pix = e / cpg;
        const int c = c0 + (int)(e % cpg);
        float v = mish_f((xb[pix * CP + c] - mean) * rstd * gamma[c] + beta[c]);
        if (temb) v += temb[trow * temb_stride + c];
        if (addend) v += addend[((long long)b * HW + pix) * CP + c];
        ob[pix * CP + c] = v;
    }
}

// per-pixel LayerNorm over the C real channels of a CP-pitched row, eps on the std (blocks.py:57-60); one wave per pixel
__global__ __launch_bounds__(256) void chan_layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                     const float* __restrict__ bta, float* __restrict__ out, long long M, int CP,
                                                                     int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (long long pix = blockIdx.x * 4LL + (threadIdx.x >> 6); pix < M; pix += (long long)gridDim.x * 4) {
        const float* xr = x + pix * CP;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
        const float inv = 1.0f / (sqrtf(wave_sum(q) / (float)C) + eps);
        for (int c = lane; c < CP; c += 64) out[pix * CP + c] = c < C ? (xr[c] - mean) * inv * g[c] + bta[c] : 0.f;
    }
}

int groupnorm_mish_generic(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride, const float* addend,
                           float* out, int B, int HW, int CP, int C, int groups, float eps, hipStream_t st, const long long* temb_rows) {
    DDK_REQUIRE(x && gamma && beta && out && B > 0 && HW > 0 && groups > 0 && C > 0 && C % groups == 0 && CP >= C,
                "groupnorm (generic): arguments");
    hipLaunchKernelGGL(gn_mish_generic_kernel, dim3((unsigned)(B * (groups + 1))), dim3(256), 0, st, x, gamma, beta, temb, temb_stride, temb_rows,
                       addend, out, HW, CP, C, groups, eps);
    return check_launch("gn_mish_generic_kernel");
}

int chan_layernorm_generic(const float* x, const float* g, const float* b, float* out, long long M, int CP, int C, float eps, hipStream_t st) {
    DDK_REQUIRE(x && g && b && out && M > 0 && C > 0 && CP >= C, "layernorm (generic): arguments");
    const long long blocks = ceil_div(M, 4);
    hipLaunchKernelGGL(chan_layernorm_generic_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, g, b, out, M, CP, C, eps);
    return check_launch("chan_layernorm_generic_kernel");
}

// ------------------------------------------------------------------------------------------------
// Channel LayerNorm: LPP lanes per pixel (C/4 capped at 64), 64/LPP pixels per wave, the pixel's C
// floats live in registers, mean and biased variance by xor-shuffles inside the lane group.
template <int LPP, int VPL>
__global__ __launch_bounds__(256) void chan_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ bta, float* __restrict__ out,
                                                             long long M, int C, float eps) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63;
    const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
    const long long pix = wave * PPW + lane / LPP;
    const int sub = lane % LPP;
    const bool ok = pix < M;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = ok ? *reinterpret_cast<const float4*>(x + pix * C + (sub + i * LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float inv = 1.0f / (sqrtf(q / (float)C) + eps);  // eps on the std (blocks.py:58-60)
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (sub + i * LPP) * 4;
        const float4 ga = *reinterpret_cast<const float4*>(g + c0);
        const float4 be = *reinterpret_cast<const float4*>(bta + c0);
        float4 y;
        y.x = (v[i].x - mean) * inv * ga.x + be.x;
        y.y = (v[i].y - mean) * inv * ga.y + be.y;
        y.z = (v[i].z - mean) * inv * ga.z + be.z;
        y.w = (v[i].w - mean) * inv * ga.w + be.w;
        *reinterpret_cast<float4*>(out + pix * C + c0) = y;
    }
}

int chan_layernorm(const float* x, const float* g, const float* b, float* out, long long M, int C, float eps, hipStream_t st) {
    DDK_REQUIRE(x && g && b && out && M > 0, "layernorm: null pointer / M");
    DDK_REQUIRE(aligned16(x) && aligned16(g) && aligned16(b) && aligned16(out), "layernorm: alignment");
#define LN_CASE(LPP, VPL)                                                                                                   \
    do {                                                                                                                    \
        const long long waves = ceil_div(M, 64 / LPP);                                                                      \
        hipLaunchKernelGGL((chan_layernorm_kernel<LPP, VPL>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, st, x, g, b, out, M, \
                           C, eps);                                                                                         \
        return check_launch("chan_layernorm_kernel");                                                                       \
    } while (0)
    switch (C) {
        case 32: LN_CASE(8, 1);
        case 64: LN_CASE(16, 1);
        case 96: LN_CASE(8, 3);
        case 128: LN_CASE(32, 1);
        case 192: LN_CASE(16, 3);
        case 256: LN_CASE(64, 1);
        case 384: LN_CASE(32, 3);
        case 512: LN_CASE(64, 2);
        case 768: LN_CASE(64, 3);
        case 1024: LN_CASE(64, 4);
        default: break;
    }
#undef LN_CASE
    return fail_arg("layernorm: unsupported channel count (supported: 32,64,96,128,192,256,384,512,768,1024)");
}

// ------------------------------------------------------------------------------------------------
template <int OP>  // 0 mish, 1 tanh
__global__ __launch_bounds__(256) void unary_kernel(const float* __restrict__ x, float* __restrict__ out, long long n4, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (OP == 0) { v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w); }
        else { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        reinterpret_cast<float4*>(out)[i] = v;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        out[i] = OP == 0 ? mish_f(x[i]) : tanhf(x[i]);
    }
}

static int grid_for(long long n4) {
    const long long b = ceil_div(n4 > 0 ? n4 : 1, 256);
    return (int)(b < 2048 ? b : 2048);
}

int unary(int op, const float* x, float* out, long long n, hipStream_t st) {
    DDK_REQUIRE(x && out && n > 0, "unary: null pointer / n");
    DDK_REQUIRE(aligned16(x) && aligned16(out), "unary: alignment");
    const long long n4 = n >> 2;
    if (op == 0) hipLaunchKernelGGL(unary_kernel<0>, dim3(grid_for(n4)), dim3(256), 0, st, x, out, n4, n);
    else hipLaunchKernelGGL(unary_kernel<1>, dim3(grid_for(n4)), dim3(256), 0, st, x, out, n4, n);
    return check_launch("unary_kernel");
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                  long long n4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

int add(const float* a, const float* b, float* out, long long n, hipStream_t st) {
    DDK_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "add: null pointer / n % 4");
    DDK_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out), "add: alignment");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, a, b, out, n / 4);
    return check_launch("add_kernel");
}

// avg_pool2d(2): out[b][y][x][c] = ((x00 + x01) + (x10 + x11)) * 0.25
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                       long long total4) {
    const int c4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        long long p = i / c4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const long long b = p / Ho;
        const float* s = x + ((b * H + 2 * yo) * W + 2 * xo) * C + cq * 4;
        const float4 a = *reinterpret_cast<const float4*>(s), bb = *reinterpret_cast<const float4*>(s + C);
        const float4 c = *reinterpret_cast<const float4*>(s + (long long)W * C), d = *reinterpret_cast<const float4*>(s + (long long)W * C + C);
        float4 r;
        r.x = ((a.x + bb.x) + (c.x + d.x)) * 0.25f;
        r.y = ((a.y + bb.y) + (c.y + d.y)) * 0.25f;
        r.z = ((a.z + bb.z) + (c.z + d.z)) * 0.25f;
        r.w = ((a.w + bb.w) + (c.w + d.w)) * 0.25f;
        reinterpret_cast<float4*>(out)[i] = r;
    }
}

int avgpool2(const float* x, float* out, int B, int H, int W, int C, hipStream_t st) {
    DDK_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "avgpool2: shape");
    DDK_REQUIRE(aligned16(x) && aligned16(out), "avgpool2: alignment");
    const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(total4)), dim3(256), 0, st, x, out, H, W, C, total4);
    return check_launch("avgpool2_kernel");
}

__global__ __launch_bounds__(256) void upnearest2_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                         long long total4) {
    const int c4 = C >> 2, Ho = H * 2, Wo = W * 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % c4);
        long long p = i / c4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const long long b = p / Ho;
        reinterpret_cast<float4*>(out)[i] =
            *reinterpret_cast<const float4*>(x + ((b * H + (yo >> 1)) * W + (xo >> 1)) * C + cq * 4);
    }
}

int upsample_nearest2(const float* x, float* out, int B, int H, int W, int C, hipStream_t st) {
    DDK_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample_nearest2: shape");
    DDK_REQUIRE(aligned16(x) && aligned16(out), "upsample_nearest2: alignment");
    const long long total4 = (long long)B * (H * 2) * (W * 2) * (C / 4);
    hipLaunchKernelGGL(upnearest2_kernel, dim3(grid_for(total4)), dim3(256), 0, st, x, out, H, W, C, total4);
    return check_launch("upnearest2_kernel");
}

}  // namespace ddk

extern "C" {
int ddk_groupnorm_mish(const float* x, const float* gamma, const float* beta, const float* temb, int temb_stride,
                       const float* addend, float* out, int B, int HW, int C, int groups, float eps, void* workspace,
                       size_t workspace_bytes, ddk_stream_t s) {
    return ddk::groupnorm_mish(x, gamma, beta, temb, temb_stride, addend, out, B, HW, C, groups, eps, workspace, workspace_bytes,
                               ddk::as_stream(s));
}
size_t ddk_groupnorm_workspace_bytes(int B, int HW, int C, int groups) { return ddk::groupnorm_workspace_bytes(B, HW, C, groups); }
int ddk_groupnorm_mish_slabs(const float* slabs, int nslab, long long slab_stride, const float* conv_bias, const float* gamma,
                             const float* beta, const float* temb, int temb_stride, const float* addend, float* out, int B, int HW,
                             int C, int groups, float eps, ddk_stream_t s) {
    return ddk::groupnorm_mish_ex(slabs, nslab, slab_stride, conv_bias, gamma, beta, temb, temb_stride, addend, out, B, HW, C, groups,
                                  eps, nullptr, 0, ddk::as_stream(s));
}
int ddk_groupnorm_mish_partials(const float* x, const float* partials, int tiles_per_image, const float* gamma, const float* beta,
                                const float* temb, int temb_stride, const float* addend, float* out, int B, int HW, int C,
                                int groups, float eps, ddk_stream_t s) {
    return ddk::groupnorm_mish_parts(x, partials, tiles_per_image, gamma, beta, temb, temb_stride, addend, out, B, HW, C, groups, eps,
                                     ddk::as_stream(s));
}
int ddk_groupnorm_mish_partials_res1x1(const float* x, const float* partials, int tiles_per_image, const float* gamma, const float* beta,
                                       const float* temb, int temb_stride, const float* res_x, const float* res_w, const float* res_b,
                                       int res_cin, float* out, int B, int HW, int C, int groups, float eps, ddk_stream_t s) {
    return ddk::groupnorm_mish_parts(x, partials, tiles_per_image, gamma, beta, temb, temb_stride, nullptr, out, B, HW, C, groups, eps,
                                     ddk::as_stream(s), nullptr, res_x, res_w, res_b, res_cin, res_cin);
}
int ddk_chan_layernorm(const float* x, const float* g, const float* b, float* out, long long M, int C, float eps, ddk_stream_t s) {
    return ddk::chan_layernorm(x, g, b, out, M, C, eps, ddk::as_stream(s));
}
int ddk_mish(const float* x, float* out, long long n, ddk_stream_t s) { return ddk::unary(0, x, out, n, ddk::as_stream(s)); }
int ddk_tanh(const float* x, float* out, long long n, ddk_stream_t s) { return ddk::unary(1, x, out, n, ddk::as_stream(s)); }
int ddk_add(const float* a, const float* b, float* out, long long n, ddk_stream_t s) { return ddk::add(a, b, out, n, ddk::as_stream(s)); }
int ddk_avgpool2(const float* x, float* out, int B, int H, int W, int C, ddk_stream_t s) {
    return ddk::avgpool2(x, out, B, H, W, C, ddk::as_stream(s));
}
int ddk_upsample_nearest2(const float* x, float* out, int B, int H, int W, int C, ddk_stream_t s) {
    return ddk::upsample_nearest2(x, out, B, H, W, C, ddk::as_stream(s));
}
}
