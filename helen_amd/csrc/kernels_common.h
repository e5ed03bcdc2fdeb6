// kernels_common.h -- types, MFMA wrapper and gate math shared by every kernel
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace helen {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // D[16x16] += A[16x4] * B[4x16], exact fp32 (k-ordered fmaf chain).
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 splat4(float v) {
    f32x4 r = {v, v, v, v};
    return r;
}

// sigmoid / tanh on the v_exp_f32 + v_rcp_f32 fast paths (each ~1 ulp); saturate correctly at
// +-inf: exp2(+big) = inf -> rcp = 0.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (1 + e^{2x})
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}

// A lane offset the optimiser cannot hoist out of the block: keeps `uniform base + zext(offset)` visible to
// instruction selection, which then uses the SGPR-base address form instead of a 64-bit VALU add per access
// (a VALU instruction between fp32 MFMAs costs 13+ cycles of matrix-pipe time on gfx950).
__device__ __forceinline__ unsigned in_block(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

// One 1 KiB row global -> LDS (global_load_lds_dwordx4): lane l copies 16 bytes from `base + voff` (voff = 16 l
// for a contiguous row) to LDS byte address `lds_row + 16 l`.  Inline asm for the SGPR-base + lane-offset form:
// the builtin adds a uniform base on the VALU, 64 bits wide.  M0 carries the LDS address.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_row_to_lds(unsigned lds_row, const void* base, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_row), "v"(voff), "s"(base) : "memory", "m0");
}
#pragma clang diagnostic pop

// Pileup rows are 90 bytes, so a position's bytes are only 2-byte aligned: gfx950 takes unaligned wide
// global loads, and these under-aligned types make hipcc emit them (one dword / qword load instead of
// 4 / 8 byte loads).
typedef uint16_t __attribute__((aligned(2))) u16_a2;
typedef uint32_t __attribute__((aligned(2))) u32_a2;
typedef uint64_t __attribute__((aligned(2))) u64_a2;

}  // namespace helen
