// level_chain.h -- the op list of the persistent 4x4-level kernel (level_chain.hip), shared with the plan that builds it
// (unet_plan.hip).
#pragma once
#include "ddk_internal.h"

namespace ddk {

constexpr int CH_MAX_OPS = 24;
enum ChainKind {
    CH_CONV3 = 0,   // conv3x3(pad 1) + bias -> GroupNorm(32 channels = this workgroup's slice) -> Mish (+ time shift) (+ kept residual)
    CH_CONV1 = 1,   // conv1x1 + bias (+ kept residual)
    CH_ATTN = 2,    // LayerNorm-folded to_qkv of one head + softmax_n(k) + context + apply (workgroups 0..3 of the image; 4..7 pass)
    CH_UPT = 3      // ConvTranspose2d(4, stride 2, padding 1) + bias, 4x4 -> 8x8 (blocks.py:32-38): four output phases of 2 x 2 taps each
};
enum ChainFlag {
    CHF_WAIT = 1,           // the sources were written in this launch by the image's other workgroups: wait for every signal so far
    CHF_SIGNAL = 2,         // other workgroups read `out` in this launch: publish it (drained sc1 stores, then one arrival per workgroup)
    CHF_ADD_KEEP = 4,       // y += keep      (the residual stream's element of this thread, carried in a register)
    CHF_SAVE_KEEP = 8,      // keep = y
    CHF_ADD_KEEP2 = 16,     // y += keep2     (the 1x1 skip conv of a ResnetBlock whose channel count changes)
    CHF_SAVE_KEEP2 = 32,    // keep2 = y
    CHF_KEEP_FROM_SRC = 64, // keep = src0's element at this thread's output position (the chain's external input)
    CHF_NO_OUT = 128,       // nothing is stored (the result only lives in keep2)
    CHF_NO_GN = 256,        // CH_CONV3 without the GroupNorm + Mish behind it: a plain conv3x3 + bias
    CHF_DOWN = 512          // CH_CONV3 with stride 2 (blocks.py:41-47): the source is the 8x8 map [B][64][c0], the output the 4x4 one
};

struct ChainOp {
    const float* src0;      // [B][16][c0]
    const float* src1;      // [B][16][c1] or null: channel concat behind src0 (unet.py:97)
    const float* w;         // CH_CONV3 / CH_CONV1: ddk_pack_conv_weight_local layout (9 or 1 taps); CH_ATTN: qkv operand order (all heads)
    const float* bias;      // [n_out] or null
    const float* gamma;     // CH_CONV3: GroupNorm weight; CH_ATTN: LayerNorm fold vector W g  [3 * 128]
    const float* beta;      // CH_CONV3: GroupNorm bias;   CH_ATTN: LayerNorm fold vector W b  [3 * 128]
    float* out;             // [B][16][n_out]
    int c0, c1, kind, flags, temb_off, n_out;      // temb_off < 0: no time shift
};

struct ChainParams {
    ChainOp op[CH_MAX_OPS];
    const float* temb;              // [rows][temb_stride]
    const long long* temb_rows;     // row of image b, or null: row b
    unsigned* cnt;                  // [B] arrival counters, 32 words apart, zero before the launch; the launch leaves them zero
    unsigned* done;                 // [B] departure counters, same layout
    unsigned* fail;                 // sticky count of workgroups that gave up waiting (the caller's cluster check reads it)
    int n_ops, B, temb_stride;
    float gn_eps, ln_eps;
    int hw;                         // pixels per image: 16 (4x4 maps: level_chain_kernel) or 64 (8x8 maps: level8_chain_kernel)
};

bool level_chain_device_ok();
size_t level_chain_lds_bytes();
int level_chain_launch(const ChainParams& p, hipStream_t st);
int level_chain_init_device();

}  // namespace ddk
