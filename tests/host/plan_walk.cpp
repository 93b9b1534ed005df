// plan_walk.cpp -- driver of the host-sanitizer build (csrc/host_sanitize.h, `make -C downsampled-diffusion_amd/csrc asan`).
// No GPU: libddk's host half runs under ASan + UBSan with every kernel launch replaced by a checker that validates the launch
// geometry and that every pointer it would hand to the device lies inside an arena registered here.  For each BASELINE.json
// configuration (reference models/unet/unet.py:19-72 shapes): create the plan, pack every slot, run one ddk_unet_forward and a
// three-step eager ddk_sampler_run, and check the size queries against the arenas they size.
#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ddk.h"

extern "C" void ddk_san_register(const void* base, size_t bytes, const char* name);
extern "C" void ddk_san_clear(void);
extern "C" void ddk_san_stats(long* launches, long* errors);
extern "C" const char* ddk_san_first_error(void);

struct Arena {
    void* p = nullptr;
    size_t bytes = 0;
    Arena(size_t n, const char* name) : bytes(n ? n : 16) {
        // untouched pages cost nothing: cfg5's workspace is several GiB of address space, a few KiB of it are ever written
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { std::perror("mmap"); std::exit(2); }
        ddk_san_register(p, bytes, name);
    }
    ~Arena() { munmap(p, bytes); }
    float* f() const { return static_cast<float*>(p); }
};

static int failures = 0;
static long g_expected_errors = 0;      // errors provoked on purpose
#define CHECK(cond, ...)                                      \
    do {                                                      \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } \
    } while (0)

static void walk(const char* name, int in_ch, int chan, std::vector<int> mults, int B, int H, int W, int T) {
    ddk_san_clear();
    ddk_unet_config cfg{};
    cfg.in_ch = in_ch; cfg.chan = chan; cfg.n_levels = (int)mults.size();
    for (size_t i = 0; i < mults.size(); ++i) cfg.mults[i] = mults[i];
    ddk_unet* u = ddk_unet_create(&cfg);
    CHECK(u != nullptr, "%s: ddk_unet_create: %s", name, ddk_last_error());
    if (!u) return;
    const size_t packed_bytes = ddk_unet_packed_bytes(u);
    const size_t ws_bytes = ddk_unet_workspace_bytes(u, B, H, W);
    const size_t smp_bytes = ddk_sampler_workspace_bytes(u, B, H, W, T - 1);
    CHECK(packed_bytes > 0 && ws_bytes > 0 && smp_bytes >= ws_bytes, "%s: size queries %zu %zu %zu", name, packed_bytes, ws_bytes, smp_bytes);
    Arena packed(packed_bytes, "packed"), ws(ws_bytes, "unet workspace"), smp(smp_bytes, "sampler workspace");
    long long max_slot = 0;
    const int n_slots = ddk_unet_num_slots(u);
    for (int i = 0; i < n_slots; ++i) max_slot = std::max(max_slot, ddk_unet_slot_numel(u, i));
    Arena canonical((size_t)max_slot * 4, "canonical weight");
    for (int i = 0; i < n_slots; ++i) {
        const int rc = ddk_unet_pack_slot(u, i, canonical.f(), packed.p, nullptr);
        CHECK(rc == DDK_OK, "%s: pack slot %d (%s): %s", name, i, ddk_unet_slot_name(u, i), ddk_last_error());
    }
    CHECK(ddk_unet_finalize_pack(u, packed.p, nullptr) == DDK_OK, "%s: finalize_pack: %s", name, ddk_last_error());
    const size_t xb = (size_t)B * H * W * in_ch * 4;
    Arena x(xb, "x"), out(xb, "eps_hat"), t((size_t)B * 8, "t"), tables((size_t)5 * T * 4, "schedule tables");
    int rc = ddk_unet_forward(u, packed.p, x.f(), static_cast<const int64_t*>(t.p), out.f(), B, H, W, ws.p, ws_bytes, nullptr);
    CHECK(rc == DDK_OK, "%s: unet_forward: %s", name, ddk_last_error());
    // a workspace one float short must be refused, not overrun
    rc = ddk_unet_forward(u, packed.p, x.f(), static_cast<const int64_t*>(t.p), out.f(), B, H, W, ws.p, ws_bytes - 4, nullptr);
    CHECK(rc == DDK_ERR_WORKSPACE, "%s: short workspace accepted (%d)", name, rc);
    for (int opt = 0; opt <= 2; opt += 2) {          // conv + apply pairs, then the in-launch GroupNorm everywhere it is eligible
        CHECK(ddk_unet_set_option(u, DDK_OPT_CLUSTER_GROUPNORM, opt) == DDK_OK, "%s: set_option", name);
        rc = ddk_unet_forward(u, packed.p, x.f(), static_cast<const int64_t*>(t.p), out.f(), B, H, W, ws.p, ws_bytes, nullptr);
        CHECK(rc == DDK_OK, "%s: unet_forward (cluster option %d): %s", name, opt, ddk_last_error());
        CHECK(ddk_unet_cluster_check(u, ws.p, B, H, W, nullptr) == DDK_OK, "%s: cluster_check: %s", name, ddk_last_error());
    }
    CHECK(ddk_unet_set_option(u, DDK_OPT_CLUSTER_GROUPNORM, 1) == DDK_OK, "%s: set_option", name);
    ddk_sampler_args a{};
    a.unet = u; a.packed = packed.p; a.x = x.f(); a.noise = nullptr;
    a.c_recip = tables.f(); a.c_recipm1 = tables.f() + T; a.c1 = tables.f() + 2 * T; a.c2 = tables.f() + 3 * T; a.sigma = tables.f() + 4 * T;
    a.B = B; a.H = H; a.W = W; a.t_start = T - 1; a.t_end = T - 3; a.seed = 1; a.stream_id = 0; a.use_graph = 0;
    a.workspace = smp.p; a.workspace_bytes = smp_bytes;
    rc = ddk_sampler_run(&a, nullptr);
    CHECK(rc == DDK_OK, "%s: sampler_run: %s", name, ddk_last_error());
    a.workspace_bytes = smp_bytes - 4;
    CHECK(ddk_sampler_run(&a, nullptr) == DDK_ERR_WORKSPACE, "%s: short sampler workspace accepted", name);
    CHECK(ddk_unet_flops(u, B, H, W) > 0 && ddk_unet_flops_executed(u, B, H, W) > 0, "%s: flops", name);
    CHECK(ddk_unet_workspace_bytes(u, B, H + 1, W) == 0, "%s: indivisible map accepted", name);
    ddk_unet_destroy(u);
    long launches = 0, errors = 0;
    ddk_san_stats(&launches, &errors);
    std::printf("%-34s packed %8.1f MB  workspace %9.1f MB  sampler %9.1f MB  launches so far %ld  errors %ld\n", name, packed_bytes / 1e6,
                ws_bytes / 1e6, smp_bytes / 1e6, launches, errors);
}

// The training-side entry points added in round 4 (job-table launches, deferred reduces, the 32-channel and small-map conv kernels):
// every pointer a launch carries -- also inside the by-value job packs -- must lie in a registered arena.
static void walk_training_ops() {
    ddk_san_clear();
    struct Shape { int kind, B, H, W, cx, N; };
    const Shape shapes[] = {{DDK_CONV3X3_S1, 8, 64, 64, 32, 32}, {DDK_CONV3X3_S1, 4, 16, 16, 128, 256}, {DDK_CONV1X1, 4, 8, 8, 256, 384},
                            {DDK_CONV3X3_S2, 4, 16, 16, 64, 64}, {DDK_CONV1X1, 32, 4, 4, 128, 256}, {DDK_CONV3X3_S1, 2, 32, 32, 32, 32},
                            {DDK_CONV1X1, 8, 64, 64, 64, 32}, {DDK_CONV1X1, 16, 32, 32, 32, 64}};      // the last two: conv1x1_stream.hip
    std::vector<ddk_wgrad_reduce_job> jobs;
    std::vector<Arena*> keep;
    for (const Shape& sh : shapes) {
        const int k = sh.kind == DDK_CONV1X1 ? 1 : 3, Ho = sh.kind == DDK_CONV3X3_S2 ? sh.H / 2 : sh.H;
        const size_t xb = (size_t)sh.B * sh.H * sh.W * sh.cx * 4, yb = (size_t)sh.B * Ho * Ho * sh.N * 4, wb = (size_t)sh.N * sh.cx * k * k * 4;
        Arena* x = new Arena(xb, "x"); Arena* dy = new Arena(yb, "dy"); Arena* gw = new Arena(wb, "grad_w"); Arena* gb = new Arena((size_t)sh.N * 4, "grad_b");
        const size_t wsb = ddk_conv_wgrad_workspace_bytes(sh.kind, sh.B, sh.H, sh.W, sh.cx, sh.N);
        CHECK(wsb > 0, "wgrad workspace query");
        Arena* ws = new Arena(wsb, "wgrad slabs");
        int rc = ddk_conv_wgrad_bias(sh.kind, x->f(), dy->f(), gw->f(), gb->f(), sh.B, sh.H, sh.W, sh.cx, sh.cx, sh.cx, 0, sh.N, ws->p, wsb, nullptr);
        CHECK(rc == DDK_OK, "conv_wgrad_bias: %s", ddk_last_error());
        CHECK(ddk_conv_wgrad_bias(sh.kind, x->f(), dy->f(), gw->f(), gb->f(), sh.B, sh.H, sh.W, sh.cx, sh.cx, sh.cx, 0, sh.N, ws->p, wsb - 4, nullptr) ==
                  DDK_ERR_WORKSPACE, "short wgrad workspace accepted");
        ddk_wgrad_reduce_job j{};
        rc = ddk_conv_wgrad_defer(sh.kind, x->f(), dy->f(), gw->f(), gb->f(), sh.B, sh.H, sh.W, sh.cx, sh.cx, sh.cx, 0, sh.N, ws->p, wsb, &j, nullptr);
        CHECK(rc == DDK_OK, "conv_wgrad_defer: %s", ddk_last_error());
        jobs.push_back(j);
        // the forward / input-gradient convs of the same shape with the epilogues of the training path
        if (sh.kind != DDK_CONV3X3_S2) {
            Arena* wp = new Arena((size_t)sh.N * k * k * sh.cx * 4, "packed weight"); Arena* o2 = new Arena(yb, "mish_out"); Arena* h = new Arena(yb, "dmish_src");
            const size_t cwb = ddk_conv_workspace_bytes(sh.kind, sh.B, sh.H, sh.W, sh.cx, sh.N);
            Arena* cws = new Arena(cwb, "conv workspace");
            ddk_conv_args a{};
            a.kind = sh.kind; a.src0 = x->f(); a.c0 = sh.cx; a.weight = wp->f(); a.bias = gb->f(); a.out = dy->f();
            a.B = sh.B; a.H = sh.H; a.W = sh.W; a.N = sh.N; a.workspace = cws->p; a.workspace_bytes = cwb;
            CHECK(ddk_conv_forward(&a, nullptr) == DDK_OK, "conv_forward: %s", ddk_last_error());
            a.mish_out = o2->f();
            CHECK(ddk_conv_forward(&a, nullptr) == DDK_OK, "conv_forward (mish_out): %s", ddk_last_error());
            a.mish_out = nullptr; a.dmish_src = h->f(); a.resid = o2->f();
            CHECK(ddk_conv_forward(&a, nullptr) == DDK_OK, "conv_forward (dmish_src, resid): %s", ddk_last_error());
            keep.insert(keep.end(), {wp, o2, h, cws});
        }
        keep.insert(keep.end(), {x, dy, gw, gb, ws});
    }
    CHECK(ddk_wgrad_reduce_jobs(jobs.data(), (int)jobs.size(), nullptr) == DDK_OK, "wgrad_reduce_jobs: %s", ddk_last_error());
    {   // a job whose slab pointer is stale (its arena gone) must be flagged by the checker, not slip through inside the by-value pack
        long l0 = 0, e0 = 0, l1 = 0, e1 = 0;
        ddk_san_stats(&l0, &e0);
        ddk_wgrad_reduce_job bad = jobs[0];
        bad.slab = jobs[0].slab + (512u << 20) / 4;          // 512 MiB past its arena: near the registered range, inside none of it
        std::fprintf(stderr, "[plan_walk] a stale job pointer on purpose: the next message is expected\n");
        (void)ddk_wgrad_reduce_jobs(&bad, 1, nullptr);
        ddk_san_stats(&l1, &e1);
        CHECK(e1 > e0, "a stale pointer inside a reduce job was not flagged");
        g_expected_errors += e1 - e0;
    }
    {   // GroupNorm / LayerNorm parameter-gradient sums as jobs
        Arena rows((size_t)3 * 8 * 256 * 4, "rows"), t0(1024, "dgamma"), t1(1024, "dbeta"), t2(1024, "dbias");
        ddk_rows_sum_job r{};
        r.rows = rows.f(); r.out[0] = t0.f(); r.out[1] = t1.f(); r.out[2] = t2.f(); r.batch_stride = 8 * 256; r.row_stride = 256;
        r.nbatch = 3; r.nrows = 8; r.n = 256;
        std::vector<ddk_rows_sum_job> rj(60, r);
        CHECK(ddk_rows_sum_jobs(rj.data(), (int)rj.size(), nullptr) == DDK_OK, "rows_sum_jobs: %s", ddk_last_error());
    }
    {   // round 4: a conv that leaves its split-K slabs to the GroupNorm (forward and backward forms), the LayerNorm backward with the
        // Residual's gradient, the attention backward that recomputes its statistics
        const int B = 4, H = 16, W = 16, C = 128, N = 128;
        const int S = ddk_conv_wino_splits(B, H, W, C, N) > 1 ? ddk_conv_wino_splits(B, H, W, C, N) : 2;
        const size_t tb = (size_t)B * H * W * N * 4;
        Arena slabs(tb * S, "slabs"), raw(tb, "raw"), y(tb, "y"), cb(N * 4, "conv bias"), ga(N * 4, "gamma"), be(N * 4, "beta"), te((size_t)B * N * 4, "temb");
        Arena dx(tb, "dx"), part((size_t)4 * B * N * 4, "gn part");
        CHECK(ddk_groupnorm_train_workspace_bytes(B, H * W, N, 8) == 0, "register-resident GroupNorm shape expected");
        CHECK(ddk_groupnorm_mish_train_fwd_slabs(slabs.f(), S, (long long)B * H * W * N, cb.f(), raw.f(), ga.f(), be.f(), te.f(), N, nullptr, 0.1f, 7, 3,
                                                 y.f(), B, H * W, N, 8, 1e-5f, nullptr) == DDK_OK, "groupnorm_train_fwd_slabs: %s", ddk_last_error());
        CHECK(ddk_groupnorm_mish_bwd_slabs(raw.f(), ga.f(), be.f(), 0.1f, 7, 3, slabs.f(), S, (long long)B * H * W * N, dx.f(), part.f(), B, H * W, N, 8,
                                           1e-5f, nullptr) == DDK_OK, "groupnorm_bwd_slabs: %s", ddk_last_error());
        CHECK(ddk_groupnorm_mish_train_fwd_slabs(slabs.f(), 1, (long long)B * H * W * N, cb.f(), raw.f(), ga.f(), be.f(), nullptr, 0, nullptr, 0.f, 0, 0,
                                                 y.f(), B, H * W, N, 8, 1e-5f, nullptr) != DDK_OK, "a single slab accepted by the slab form");
        Arena lpart((size_t)2 * 512 * N * 4, "ln part");
        int nparts = 0;
        CHECK(ddk_chan_layernorm_bwd_add(raw.f(), ga.f(), y.f(), slabs.f(), dx.f(), lpart.f(), 512, &nparts, (long long)B * H * W, N, 1e-5f, nullptr) == DDK_OK &&
                  nparts > 0, "chan_layernorm_bwd_add: %s", ddk_last_error());
        const int heads = 4;
        Arena qkv((size_t)B * H * W * 384 * 4, "qkv"), dout((size_t)B * H * W * 128 * 4, "dout"), ctx((size_t)B * heads * 1024 * 4, "ctx"),
            dctx((size_t)B * heads * 1024 * 4, "dctx"), dqkv((size_t)B * H * W * 384 * 4, "dqkv"), st((size_t)B * heads * 64 * 4, "stats");
        const size_t lw = ddk_linattn_train_workspace_bytes(B, H * W, heads);
        Arena lws(lw ? lw : 16, "linattn workspace");
        CHECK(ddk_linattn_bwd_recompute(qkv.f(), dout.f(), ctx.f(), st.f(), dctx.f(), dqkv.f(), B, H * W, heads, lws.p, lw, nullptr) == DDK_OK,
              "linattn_bwd_recompute: %s", ddk_last_error());
    }
    {   // every kernel-layout copy of two weights by one launch
        Arena w((size_t)256 * 256 * 9 * 4, "weight"), d0((size_t)256 * 9 * 256 * 4, "fwd copy"), d1((size_t)8 * 16 * 256 * 32 * 4, "wino copy"),
            d2((size_t)256 * 9 * 256 * 4, "dgrad copy");
        ddk_pack_job pj[3]{};
        pj[0].src = w.f(); pj[0].dst = d0.f(); pj[0].kind = DDK_PACK_CONV; pj[0].p[0] = 256; pj[0].p[1] = 256; pj[0].p[2] = 9; pj[0].p[3] = 256; pj[0].p[4] = 256; pj[0].p[5] = 256;
        pj[1].src = w.f(); pj[1].dst = d1.f(); pj[1].kind = DDK_PACK_WINO; pj[1].p[0] = 256; pj[1].p[1] = 256; pj[1].p[2] = 256;
        pj[2].src = w.f(); pj[2].dst = d2.f(); pj[2].kind = DDK_PACK_DGRAD; pj[2].p[0] = 256; pj[2].p[1] = 256; pj[2].p[2] = 9; pj[2].p[3] = 256; pj[2].p[4] = 256;
        const long long blocks = ddk_pack_jobs_layout(pj, 3);
        CHECK(blocks > 0, "pack_jobs_layout: %s", ddk_last_error());
        Arena table(sizeof(pj), "job table");
        std::memcpy(table.p, pj, sizeof(pj));
        CHECK(ddk_pack_jobs(static_cast<const ddk_pack_job*>(table.p), 3, blocks, nullptr) == DDK_OK, "pack_jobs: %s", ddk_last_error());
    }
    for (Arena* a : keep) delete a;
    long launches = 0, errors = 0;
    ddk_san_stats(&launches, &errors);
    std::printf("%-34s launches so far %ld  errors %ld (of which %ld seeded)\n", "training-side entry points", launches, errors, g_expected_errors);
}

int main() {
    // BASELINE.json configs (unet_chan 128, unet_dims (1,2,2,2)); cfg1 MNIST 32x32 C_in 1 T=200; cfg2 CIFAR C_in 3; cfg3 16x16 latents
    // of 8; cfg4 32x32 latents of 8; cfg5 the full-resolution UNet
    walk("cfg1 mnist 1x32x32 bs16", 1, 128, {1, 2, 2, 2}, 16, 32, 32, 200);
    walk("cfg2 cifar 3x32x32 bs64", 3, 128, {1, 2, 2, 2}, 64, 32, 32, 1000);
    walk("cfg3 latents 8x16x16 bs64", 8, 128, {1, 2, 2, 2}, 64, 16, 16, 1000);
    walk("cfg4 latents 8x32x32 bs32", 8, 128, {1, 2, 2, 2}, 32, 32, 32, 1000);
    walk("cfg5 full-res 3x256x256 bs8", 3, 128, {1, 2, 2, 2}, 8, 256, 256, 1000);
    // off-config shapes: narrow / wide widths, other depths, odd batches, non-square maps
    walk("width 32, 2 levels, 3x24x40 b5", 3, 32, {1, 2}, 5, 24, 40, 50);
    walk("width 64 (1,2,4), 4x48x16 b3", 4, 64, {1, 2, 4}, 3, 48, 16, 100);
    walk("width 256 (1,1,2,2), 8x8x8 b7", 8, 256, {1, 1, 2, 2}, 7, 8, 8, 100);
    // widths that are not multiples of 32 (GroupNorm(8, C) of the reference takes any C % 8 == 0): the generic path, padded pitches
    walk("width 24 (1,2,4), 3x16x16 b3", 3, 24, {1, 2, 4}, 3, 16, 16, 100);
    walk("width 40 (1,2,2,2), 8x32x32 b2", 8, 40, {1, 2, 2, 2}, 2, 32, 32, 100);
    walk("width 8 (1,2), 1x8x8 b5", 1, 8, {1, 2}, 5, 8, 8, 50);
    walk_training_ops();
    ddk_unet_config bad{};
    bad.in_ch = 3; bad.chan = 44; bad.n_levels = 2; bad.mults[0] = 1; bad.mults[1] = 2;
    CHECK(ddk_unet_create(&bad) == nullptr, "unet_chan 44 accepted");
    bad.chan = 64; bad.mults[0] = 2;
    CHECK(ddk_unet_create(&bad) == nullptr, "unet_dims[0] = 2 accepted");
    long launches = 0, errors = 0;
    ddk_san_stats(&launches, &errors);
    errors -= g_expected_errors;
    // the checker checks: the same forward with the workspace arena registered HALF as large as it is must be flagged
    long seeded = 0;
    {
        ddk_san_clear();
        ddk_unet_config c{};
        c.in_ch = 3; c.chan = 32; c.n_levels = 2; c.mults[0] = 1; c.mults[1] = 2;
        ddk_unet* u = ddk_unet_create(&c);
        const size_t pb = ddk_unet_packed_bytes(u), wb = ddk_unet_workspace_bytes(u, 2, 8, 8);
        Arena packed(pb, "packed"), x(2 * 8 * 8 * 3 * 4, "x"), out(2 * 8 * 8 * 3 * 4, "out"), t(16, "t");
        void* ws = mmap(nullptr, wb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        ddk_san_register(ws, wb / 2, "half of the workspace");
        std::fprintf(stderr, "[plan_walk] seeding an overrun on purpose: the next messages are expected\n");
        (void)ddk_unet_forward(u, packed.p, x.f(), static_cast<const int64_t*>(t.p), out.f(), 2, 8, 8, ws, wb, nullptr);
        long l2 = 0, e2 = 0;
        ddk_san_stats(&l2, &e2);
        seeded = e2 - errors - g_expected_errors;
        CHECK(seeded > 0, "a workspace registered half as large as it is was not flagged");
        munmap(ws, wb);
        ddk_unet_destroy(u);
        launches = l2;
    }
    std::printf("seeded overrun: %ld launches / ranges flagged (expected > 0)\n", seeded);
    std::printf("host walk: %ld launches checked, %ld pointer/geometry errors, %d failed expectations%s%s\n", launches, errors, failures,
                errors ? "; first: " : "", errors ? ddk_san_first_error() : "");
    return (errors || failures) ? 1 : 0;
}
