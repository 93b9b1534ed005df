// pack_jobs.hip -- every kernel-layout copy of a model's conv weights refreshed by ONE launch.
//
// Training changes the weights once per optimiser step (reference trainers/trainer_ddpm.py:142-144: clip, Adam.step, EMA), after
// which every conv of the next forward / backward needs its operand copies again: forward [O][taps][I], input-gradient
// [I][taps][O] with flipped taps, and the Winograd-domain forms of both.  As one launch per copy that was 285 launches of
// 3.7-4.8 us per cfg3 optimiser step (profiles/r04_train_cfg3_kernel_stats.csv: 1.2 ms of a 20.8 ms step) for 270 MB of writes that
// the chip does in ~0.1 ms.  Here the host keeps a table of jobs (source, destination, kind, dimensions) in device memory and a
// block finds its job by binary search over the jobs' first-block prefix sums; the element functions are the single-tensor
// kernels' own (pack_elems.h), so a copy holds the same bits whichever path refreshed it.
#include "ddk_internal.h"
#include "pack_elems.h"

namespace ddk {

constexpr int PJ_PER_BLOCK = 1024;        // elements (or Winograd (n, c) items) per block: 4 per thread

__global__ __launch_bounds__(256) void pack_jobs_kernel(const ddk_pack_job* __restrict__ jobs, int n) {
    const long long blk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                        // largest j with jobs[j].block0 <= blk (block-uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= blk) lo = mid;
        else hi = mid - 1;
    }
    const ddk_pack_job j = jobs[lo];
    const long long base = (blk - j.block0) * PJ_PER_BLOCK + threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < PJ_PER_BLOCK / 256; ++it) {
        const long long idx = base + it * 256;
        if (idx >= j.total) break;
        switch (j.kind) {
            case DDK_PACK_CONV: pack_conv_weight_elem(j.src, j.dst, idx, j.p[1], j.p[2], j.p[3], j.p[4], j.p[5]); break;
            case DDK_PACK_CONVT: pack_convT_weight_elem(j.src, j.dst, idx, j.p[0], j.p[1], j.p[2], j.p[3]); break;
            case DDK_PACK_DGRAD: pack_dgrad_elem(j.src, j.dst, idx, j.p[0], j.p[1], j.p[2], j.p[4]); break;
            case DDK_PACK_WINO: pack_wino_elem<false>(j.src, j.dst, idx, j.p[0], j.p[1], j.p[2], 0, j.p[1]); break;
            case DDK_PACK_WINO_DGRAD: pack_wino_elem<true>(j.src, j.dst, idx, j.p[3] - j.p[2], j.p[0], j.p[4], j.p[2], j.p[1]); break;
            default: break;
        }
    }
}

}  // namespace ddk

// Validates the jobs, fills `total` and `block0`, returns the number of blocks of the launch (< 0: error).
extern "C" long long ddk_pack_jobs_layout(ddk_pack_job* jobs, int n) {
    using namespace ddk;
    DDK_REQUIRE(jobs && n > 0, "pack_jobs_layout: arguments");
    long long blocks = 0;
    for (int k = 0; k < n; ++k) {
        ddk_pack_job& j = jobs[k];
        DDK_REQUIRE(j.src && j.dst, "pack_jobs_layout: null tensor");
        const int* p = j.p;
        switch (j.kind) {
            case DDK_PACK_CONV:          // p = O, I, taps, i_pad, split, split_pad
                DDK_REQUIRE(p[0] > 0 && p[1] > 0 && p[2] > 0 && p[3] >= p[1] && p[4] >= 0 && p[4] <= p[1] && p[5] >= p[4] && p[5] <= p[3],
                            "pack_jobs_layout: conv job");
                j.total = (long long)p[0] * p[2] * p[3];
                break;
            case DDK_PACK_CONVT:         // p = I, O, Ip, Op
                DDK_REQUIRE(p[0] > 0 && p[1] > 0 && p[2] >= p[0] && p[3] >= p[1], "pack_jobs_layout: transpose-conv job");
                j.total = 16LL * p[2] * p[3];
                break;
            case DDK_PACK_DGRAD:         // p = O, I, taps, i_pad, o_pad
                DDK_REQUIRE(p[0] > 0 && p[1] > 0 && p[2] > 0 && p[3] >= p[1] && p[4] >= p[0], "pack_jobs_layout: dgrad job");
                j.total = (long long)p[3] * p[2] * p[4];
                break;
            case DDK_PACK_WINO:          // p = O, I, i_pad
                DDK_REQUIRE(p[0] > 0 && p[1] > 0 && p[2] >= p[1] && p[2] % 32 == 0, "pack_jobs_layout: Winograd job");
                j.total = (long long)p[0] * p[2];
                break;
            case DDK_PACK_WINO_DGRAD:    // p = O, I, c_lo, c_hi, o_pad
                DDK_REQUIRE(p[0] > 0 && p[1] > 0 && p[2] >= 0 && p[3] > p[2] && p[3] <= p[1] && p[4] >= p[0] && p[4] % 32 == 0,
                            "pack_jobs_layout: Winograd dgrad job");
                j.total = (long long)(p[3] - p[2]) * p[4];
                break;
            default:
                DDK_REQUIRE(false, "pack_jobs_layout: unknown kind");
        }
        DDK_REQUIRE(j.total > 0 && j.total < (1LL << 31), "pack_jobs_layout: a copy of 2^31 elements or more");
        j.block0 = blocks;
        blocks += ceil_div(j.total, (long long)PJ_PER_BLOCK);
    }
    DDK_REQUIRE(blocks < (1LL << 31), "pack_jobs_layout: too many blocks");
    return blocks;
}

extern "C" int ddk_pack_jobs(const ddk_pack_job* jobs_dev, int n, long long blocks, ddk_stream_t s) {
    using namespace ddk;
    DDK_REQUIRE(jobs_dev && n > 0 && blocks > 0 && blocks < (1LL << 31), "pack_jobs: arguments");
    hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), jobs_dev, n);
    return check_launch("pack_jobs_kernel");
}
