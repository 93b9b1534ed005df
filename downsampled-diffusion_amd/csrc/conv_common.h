// conv_common.h -- small device helpers shared by the conv kernels (conv_igemm.hip, conv_wino.hip).  gfx950 only.
#pragma once
#include "ddk_internal.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// exact unsigned division by a runtime constant (Granlund-Montgomery): 3 VALU ops instead of the ~30 of a hardware-less
// integer divide; the prologue of every workgroup does several of them per DMA piece
struct FastDivU {
    unsigned mul, sh1, sh2, d;
};
static inline FastDivU make_fastdiv_u(unsigned d) {
    FastDivU f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned fdiv_u(unsigned n, const FastDivU& f) {
    const unsigned t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// one 1-KiB LDS-DMA piece: LDS[m0 + lane*16 .. +16) <- 16 bytes at this lane's global address.  Issued from inline asm on
// purpose: hipcc drains a builtin LDS-DMA in front of every ds_read; the waits are placed by hand (s_waitcnt vmcnt).
__device__ __forceinline__ void lds_dma16(const float* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g) : "memory", "m0");
}

// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit VALU address arithmetic
__device__ __forceinline__ void lds_dma16_s(const float* sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace ddk
