// kernels_bf16.h -- HELEN_PRECISION_BF16: bf16-operand projection and recurrence
#pragma once
#include "kernels_common.h"
#include "kernels_gemm.h"
#include "kernels_gru.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// bf16 variants (BASELINE.json configs[3]): gate matmuls on v_mfma_f32_16x16x16_bf16 with fp32
// accumulation, fp32 state, fp32 gate math.  The KB16 grouping k = 16m + 4q + e is exactly the A/B
// fragment of the 16x16x16 instruction (lane holds 4 consecutive k), so the buffers and layouts are
// the fp32 path's: operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) as they are loaded, and
// one MFMA replaces four.  Weights are pre-rounded and packed as 4 x bf16 (8 bytes) per lane/group.
// The heads stay fp32.
// ------------------------------------------------------------------------------------------------
typedef short bf16x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
    // fptrunc <2 x float> -> <2 x bfloat> selects v_cvt_pk_bf16_f32 (RNE) on gfx950, and unlike an
    // inline-asm cvt the compiler pads the VALU-write -> MFMA-read hazard itself
    const bf16x2_t lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2_t);
    const bf16x2_t hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2_t);
    const uint2 u = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(bf16x4, u);
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// With bf16 MFMAs the projection is memory-bound (fp32 gi out: 24.5 KB per tile/position/direction),
// so the loop is built for bytes in flight, not for MFMA issue: each wave takes P positions (2 for the
// decoder's K = 256, 4 for the encoder's K = 96) and issues ALL of their A loads up front (P x MG x 1 KiB
// per wave), then streams the packed bf16 W_ih
// from L2 one group ahead; 8+ waves per CU hide what is left.
template <int MG, bool REV_A, int P, bool UPFRONT>
__global__ __launch_bounds__(HELEN_GEMM_WAVES * 64) void gemm_gi_bf16_kernel(
    const f32x4* __restrict__ A, long a_tile_stride, const bf16x4* __restrict__ Wp,
    const float* __restrict__ bias, f32x4* __restrict__ gi, long gi_tile_stride, int npos,
    int ntiles) {
    constexpr int N = 6;
    constexpr int ZB = 8 / HELEN_GEMM_WAVES;
    const int lane = threadIdx.x & 63;
    const int bid = blockIdx.x;
    const int unit = (bid / (8 * ZB)) * 8 + (bid & 7);   // same XCD-aware enumeration as gemm_gi_kernel
    const int zb = (bid >> 3) % ZB;
    const int npg = (npos + P - 1) / P;
    const int tile = unit / npg;
    const int pos0 = (unit % npg) * P;
    if (tile >= ntiles) return;
    const int wave = (threadIdx.x >> 6) + zb * HELEN_GEMM_WAVES;
    const int dir = wave >> 2;
    const int nt0 = (wave & 3) * N;
    const bf16x4* w_base = Wp + (size_t)((dir * kNTile + nt0) * MG) * 64 + lane;

    f32x4 acc[P][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float b = bias[dir * kG + (nt0 + n) * 16 + (lane & 15)];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][n] = splat4(b);
    }
    const f32x4* fwd[P];
    const f32x4* bwd[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int pc = min(pos0 + p, npos - 1);
        fwd[p] = A + (size_t)tile * a_tile_stride + (size_t)pc * (MG * 64) + lane;
        bwd[p] = A + (size_t)tile * a_tile_stride + (size_t)(npos - 1 - pc) * (MG * 64) + lane;
    }
    if constexpr (!UPFRONT) {
        // short K (encoder): the kernel is bound by its output stream; plain per-group loads measured best
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            f32x4 am[P];
            bf16x4 bm[N];
#pragma unroll
            for (int p = 0; p < P; ++p) am[p] = (REV_A && m >= MG / 2) ? bwd[p][m * 64] : fwd[p][m * 64];
#pragma unroll
            for (int n = 0; n < N; ++n) bm[n] = w_base[(n * MG + m) * 64];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const bf16x4 ab = to_bf16x4(am[p]);
#pragma unroll
                for (int n = 0; n < N; ++n) acc[p][n] = mfma_bf16(ab, bm[n], acc[p][n]);
            }
        }
    } else {
    f32x4 a[P][MG];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int m = 0; m < MG; ++m) a[p][m] = (REV_A && m >= MG / 2) ? bwd[p][m * 64] : fwd[p][m * 64];
    bf16x4 b0[N], b1[N];
#pragma unroll
    for (int n = 0; n < N; ++n) b0[n] = w_base[(n * MG) * 64];
#pragma unroll
    for (int m = 0; m < MG; m += 2) {
#pragma unroll
        for (int n = 0; n < N; ++n) b1[n] = w_base[(n * MG + m + 1) * 64];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bf16x4 ab = to_bf16x4(a[p][m]);
#pragma unroll
            for (int n = 0; n < N; ++n) acc[p][n] = mfma_bf16(ab, b0[n], acc[p][n]);
        }
        if (m + 2 < MG) {
#pragma unroll
            for (int n = 0; n < N; ++n) b0[n] = w_base[(n * MG + m + 2) * 64];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bf16x4 ab = to_bf16x4(a[p][m + 1]);
#pragma unroll
            for (int n = 0; n < N; ++n) acc[p][n] = mfma_bf16(ab, b1[n], acc[p][n]);
        }
    }
    }  // UPFRONT
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (pos0 + p < npos) {
            const int slot = dir ? (npos - 1 - (pos0 + p)) : (pos0 + p);
            f32x4* o = gi + (size_t)tile * gi_tile_stride +
                       ((size_t)slot * 2 + dir) * (kNTile * 64) + nt0 * 64 + lane;
#pragma unroll
            for (int n = 0; n < N; ++n) o[n * 64] = acc[p][n];
        }
    }
}

// Same structure as gru_kernel; all six W_hh column tiles fit in registers as bf16 (96 VGPRs), so
// nothing is parked in LDS.  h stays fp32 in LDS (it is also the fp32 layer output) and is rounded
// to bf16 as it is read for the MFMA.
__global__ __launch_bounds__(256, 2) void gru_bf16_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int slot0_fwd, int slot0_bwd, int T,
                                                          const bf16x4* __restrict__ Whp,
                                                          const float* __restrict__ bhn,
                                                          f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                          long y_tile_stride, f32x4* __restrict__ yplane,
                                                          long yp_tile_stride) {
    // Layer output: fp32 y (KB16, for the heads) when `yplane` is null, otherwise ONE bf16 plane
    // yplane[tile][slot][dir][256 units of 16 B] = h rounded to bf16 (RNE) in the K = 32 A-fragment
    // layout gemm_dec_x3_kernel<1, .> consumes (unit (k/8)*16 + row holds 8 consecutive k of a row).
    __shared__ f32x4 smem[2 * 512 + 4 * 384];
    f32x4* const hbuf = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4* const gbuf = smem + 1024 + w * 384;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;

    bf16x4 W[6][8];
    {
        const bf16x4* wp = Whp + (size_t)((dir * 4 + w) * 48) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int m = 0; m < 8; ++m) W[n][m] = wp[(n * 8 + m) * 64];
    }
    float bn[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) bn[hh] = bhn[dir * kH + 32 * w + 16 * hh + j];

    const f32x4* gi_p = gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) +
                        (2 * w) * 64 + lane;
    constexpr long kPosStride = 2 * kNTile * 64;
    auto dma_gi = [&](int slot) {
        const f32x4* p = gi_p + (size_t)slot * kPosStride;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1)))*)(p + (g * 8 + hh) * 64),
                    (void __attribute__((address_space(3)))*)(gbuf + (g * 2 + hh) * 64), 16, 0, 0);
    };

    f32x4* hid_p = hid + ((size_t)tile * 2 + dir) * (kHidDirStride / 4);
    hbuf[tid] = hid_p[tid];
    hbuf[tid + 256] = hid_p[tid + 256];
    dma_gi(slot0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float hprev[2][4];
    int hoff[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int u = 32 * w + 16 * hh + j;
        hoff[hh] = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[hh][r] = ((const float*)hbuf)[hoff[hh] + 4 * r];
    }
    f32x4* y_p = y + (size_t)tile * y_tile_stride + (size_t)dir * (kHidDirStride / 4);

    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        const f32x4* hb = hbuf + cur * 512 + lane;
        f32x4 acc[6];
        acc[0] = splat4(0.f);
        acc[1] = splat4(0.f);
        acc[2] = splat4(0.f);
        acc[3] = splat4(0.f);
        acc[4] = splat4(bn[0]);
        acc[5] = splat4(bn[1]);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const bf16x4 a = to_bf16x4(hb[m * 64]);
#pragma unroll
            for (int n = 0; n < 6; ++n) acc[n] = mfma_bf16(a, W[n][m], acc[n]);
        }
        // gi DMA landed (see gru_kernel): behind the 6 DMAs sit this step's output stores, 2 (y) or 1 (plane)
        if (yplane != nullptr)
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        f32x4 G[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) G[n] = gbuf[n * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 1 < T) dma_gi(slot0 + s + 1);

        float* hw = (float*)(hbuf + (cur ^ 1) * 512);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hn = gru_cell(acc[hh][r], acc[2 + hh][r], acc[4 + hh][r], G[hh][r],
                                          G[2 + hh][r], G[4 + hh][r], hprev[hh][r]);
                hprev[hh][r] = hn;
                hw[hoff[hh] + 4 * r] = hn;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const f32x4* hn4 = hbuf + (cur ^ 1) * 512;
        if (yplane != nullptr) {
            // thread = unit (octet o = tid >> 4, row = tid & 15): KB16 float4s 2o and 2o+1 of the row
            const int row = tid & 15, o = tid >> 4;
            const f32x4 lo = hn4[(2 * o) * 16 + row], hi = hn4[(2 * o + 1) * 16 + row];
            const bf16x4 l4 = to_bf16x4(lo), h4 = to_bf16x4(hi);
            uint2 a = __builtin_bit_cast(uint2, l4), b = __builtin_bit_cast(uint2, h4);
            uint4 u = {a.x, a.y, b.x, b.y};
            (yplane + (size_t)tile * yp_tile_stride + ((size_t)s * 2 + dir) * 256)[tid] = __builtin_bit_cast(f32x4, u);
        } else {
            f32x4* yo = y_p + (size_t)s * (kYStride / 4);
            yo[tid] = hn4[tid];
            yo[tid + 256] = hn4[tid + 256];
        }
    }
    const f32x4* hl = hbuf + (T & 1) * 512;
    hid_p[tid] = hl[tid];
    hid_p[tid + 256] = hl[tid + 256];
}

}  // namespace helen
