// kernels_gemm.h -- fp32 input projections: gemm_gi_kernel (streaming weights; the operator entry's encoder projection and the
// decoder's fine-grained form) and gemm_dec_ws(p)_kernel (weight-stationary decoder projection).  The polish entry points
// project the encoder input with exact bf16 products instead (gemm_enc_x3_kernel, kernels_x3.h).
#pragma once
#include "kernels_common.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Input projection  gi = A . W_ih^T + bias  for both directions (the non-recurrent half of nn.GRU,
// TransducerModel.py:70,72).  A is a KB16 operand with MG = K/16 groups per (tile, position).
//   The 48 column tiles (2 directions x 24) are split over 8 "wave slots": slot v -> direction
//   v>>2, column tiles 6(v&3) .. +5; a workgroup holds HELEN_GEMM_WAVES slots (grid.z the rest) and
//   covers P = 10 positions, so each wave keeps a 10 x 6 block of 16x16 accumulators (240 AGPRs).  The
//   kernel runs at ONE wave per SIMD (the compiler spends the whole 512-register budget; constraining it
//   to two waves per SIMD is 65 % slower), so nothing hides a workgroup's first loads and its output
//   stores but its own length: 10 positions per wave instead of 4 is +2.5 %, and W_ih is re-read 2.5x
//   less often.
//   Operands come straight from global memory in a register ping-pong (group m+1 in flight while
//   group m's 240 MFMAs issue): every load is one contiguous 1 KiB per wave and the packed weights
//   (<= 786 KB) stay L2-resident; no LDS, no barriers.  Single-wave workgroups measured best.
//   bias[dir][col] = b_ih[col] + (col < 2H ? b_hh[col] : 0)   (b_hn is applied inside r*(...)).
// Output gi[tile][slot][dir][ntile 24][lane 64] float4 (FRAG layout); slot = pos for direction 0,
// npos-1-pos for direction 1.
// ------------------------------------------------------------------------------------------------
// waves per projection workgroup; 8 / HELEN_GEMM_WAVES workgroups (grid.z) cover the 48 column tiles
#define HELEN_GEMM_WAVES 1   // measured at 10 positions per wave: 1 -> 1.164 ms, 2 -> 1.180, 4 -> 1.215 per decoder launch
#define HELEN_GEMM_P 10  // positions per wave of the streaming-weights projection (4: 1.21 ms, 5: 1.19, 10: 1.18 per decoder launch)
#define HELEN_GEMM_N 6   // column tiles per wave (48 / N wave slots per position group)
#define HELEN_GEMM_DEPTH 2   // operand groups in flight, including the one being multiplied
template <int MG, bool REV_A>
__global__ __launch_bounds__(HELEN_GEMM_WAVES * 64) void gemm_gi_kernel(const f32x4* __restrict__ A, long a_tile_stride,
                                                      const f32x4* __restrict__ Wp,
                                                      const float* __restrict__ bias,
                                                      f32x4* __restrict__ gi, long gi_tile_stride,
                                                      int npos, int ntiles) {
    // Output slot: direction 0 -> position p, direction 1 -> npos-1-p (time-reversed), so the
    // recurrence reads both directions in ascending address order.
    constexpr int P = HELEN_GEMM_P, N = HELEN_GEMM_N;
    constexpr int SLOTS = 2 * kNTile / N;   // wave slots covering the 48 column tiles
    const int lane = threadIdx.x & 63;
    // grid.x enumerates (unit, z): unit = (position group, tile), z = which HELEN_GEMM_WAVES wave
    // slots.  Workgroups are dispatched round-robin over the 8 XCDs (id % 8), so the ZB = 8 /
    // HELEN_GEMM_WAVES workgroups that share one unit's A operand get ids u, u+8, u+16, ... inside a
    // block of 8*ZB ids: same XCD, same L2, adjacent in time -> A is fetched from HBM once.
    constexpr int ZB = SLOTS / HELEN_GEMM_WAVES;
    const int bid = blockIdx.x;
    const int unit = (bid / (8 * ZB)) * 8 + (bid & 7);
    const int zb = (bid >> 3) % ZB;
    const int npg = (npos + P - 1) / P;                 // position groups per tile
    const int tile = unit / npg;
    const int pos0 = (unit % npg) * P;
    if (tile >= ntiles) return;                           // grid is padded to a multiple of 8 units
    const int wave = (threadIdx.x >> 6) + zb * HELEN_GEMM_WAVES;
    const int dir = wave / (SLOTS / 2);
    const int nt0 = (wave % (SLOTS / 2)) * N;

    const f32x4* w_base = Wp + (size_t)((dir * kNTile + nt0) * MG) * 64 + lane;
    // REV_A: A is a layer output y[tile][slot][fwd | bwd]; the bwd half (groups MG/2..) of
    // position p sits in slot npos-1-p.
    const f32x4* a_ptr[P];
    const f32x4* a_ptr_b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int pc = min(pos0 + p, npos - 1);
        a_ptr[p] = A + (size_t)tile * a_tile_stride + (size_t)pc * (MG * 64) + lane;
        a_ptr_b[p] = A + (size_t)tile * a_tile_stride + (size_t)(npos - 1 - pc) * (MG * 64) + lane;
    }

    f32x4 acc[P][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float b = bias[dir * kG + (nt0 + n) * 16 + (lane & 15)];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][n] = splat4(b);
    }

    // Register pipeline, HELEN_GEMM_DEPTH groups deep: the operands of groups m+1 .. m+DEPTH-1 are in
    // flight while group m's MFMAs issue (the kernel runs at one wave per SIMD with the accumulators in
    // AGPRs, so its own prefetch distance is all the latency hiding there is).
    constexpr int D = HELEN_GEMM_DEPTH;
    f32x4 a[D][P], b[D][N];
#define HELEN_LOAD_OPS(a, b, m)                                       \
    _Pragma("unroll") for (int p = 0; p < P; ++p)                     \
        a[p] = (REV_A && (m) >= MG / 2) ? a_ptr_b[p][(m) * 64] : a_ptr[p][(m) * 64]; \
    _Pragma("unroll") for (int n = 0; n < N; ++n) b[n] = w_base[(n * MG + (m)) * 64];
#define HELEN_MMA_OPS(a, b)                                           \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                     \
    _Pragma("unroll") for (int p = 0; p < P; ++p)                     \
    _Pragma("unroll") for (int n = 0; n < N; ++n) acc[p][n] = mfma4(a[p][e], b[n][e], acc[p][n]);

#pragma unroll
    for (int m = 0; m < D - 1; ++m) { HELEN_LOAD_OPS(a[m], b[m], m) }
#pragma unroll
    for (int m = 0; m < MG; ++m) {
        if (m + D - 1 < MG) { HELEN_LOAD_OPS(a[(m + D - 1) % D], b[(m + D - 1) % D], m + D - 1) }
        __builtin_amdgcn_sched_barrier(0);
        HELEN_MMA_OPS(a[m % D], b[m % D])
        __builtin_amdgcn_sched_barrier(0);
    }
#undef HELEN_LOAD_OPS
#undef HELEN_MMA_OPS
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

// ------------------------------------------------------------------------------------------------
// Decoder input projection, weight-stationary (fp32 MFMA): gi = Y1 . W_ih^T + bias, K = 256.
//   One workgroup = one (tile, direction): 8 waves x 3 column tiles = the direction's 24 column tiles, and a
//   wave's whole slice of W_ih (3 tiles x 16 k-groups = 192 registers) is loaded once and stays.  Only the
//   activations move: a stage is PB positions (16 KiB each: the forward half of slot p and the backward half of
//   slot npos-1-p of the encoder output) brought in by LDS-DMA into a 2-deep ring; a position is three chains of
//   64 MFMAs against 16 LDS reads, its three output stores left in flight.  Two waves per SIMD: one wave's
//   LDS-DMA issue (~90 cycles a piece), barrier and stores hide behind its partner's MFMAs.
//   No per-(position group) prologue / epilogue as in gemm_gi_kernel, whose 20,480 single-wave workgroups each pay
//   ~10 k cycles of first loads, VALU address arithmetic between MFMAs and an exposed store tail on 123 k of MFMAs.
//   Same MFMA order per accumulator as gemm_gi_kernel<16, true> (m ascending, e ascending): bit-identical gi.
//   Grid: the two directions of a tile sit on one XCD (ids b, b + 8) so that y1 comes from HBM once.
// ------------------------------------------------------------------------------------------------
#define HELEN_DWS_PB 4
constexpr int kDecWsLdsF4 = 2 * HELEN_DWS_PB * 16 * 64;   // 2 x PB x 16 KiB
// (the body is a device function over one (tile, direction) and a caller-provided LDS block of kDecWsLdsF4 float4;
// gemm_dec_ws_kernel below is one call per workgroup)
__device__ __forceinline__ void gemm_dec_ws_body(f32x4* __restrict__ smem, const int tile, const int dir,
                                                 const f32x4* __restrict__ A, long a_tile_stride,
                                                 const f32x4* __restrict__ Wp,
                                                 const float* __restrict__ bias,
                                                 f32x4* __restrict__ gi, long gi_tile_stride, int npos,
                                                 int p0 = 0, int p1 = -1) {
    // positions [p0, p1) of the npos (all of them by default): gemm_dec_wsp_kernel splits a small call's positions
    if (p1 < 0) p1 = npos;
    constexpr int MG = 16, PB = HELEN_DWS_PB, N = 3;
    constexpr int ROWS = PB * MG;            // 1 KiB rows per stage
    constexpr int RPP = ROWS / 8 / PB;       // rows a wave brings in per position (2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt0 = N * v;
    f32x4 B[N][MG];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < MG; ++m) B[n][m] = Wp[(size_t)((dir * kNTile + nt0 + n) * MG + m) * 64 + lane];
    f32x4 bsv[N];   // bias: the C operand of a chain's first MFMA
#pragma unroll
    for (int n = 0; n < N; ++n) bsv[n] = splat4(bias[dir * kG + (nt0 + n) * 16 + (lane & 15)]);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)smem;
    const char* a_tile = (const char*)(A + (size_t)tile * a_tile_stride);                     // uniform
    char* o_tile = (char*)(gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) + nt0 * 64);
    // Row r = p*16 + m of stage g (positions PB g .. PB g + PB-1); wave v brings in rows v, v+8, ...: RPP per position.
    auto stage_rows = [&](int g, int b, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int r = v + 8 * i;
            const int pc = min(p0 + PB * g + r / MG, p1 - 1);
            const int m = r % MG;
            const int slot = m < MG / 2 ? pc : npos - 1 - pc;   // forward half of slot p | backward half of slot npos-1-p
            dma_row_to_lds(lds0 + (unsigned)((b * ROWS + r) * 1024), a_tile + ((size_t)slot * MG + m) * 1024, lane16);
        }
    };
    const int ng = (p1 - p0 + PB - 1) / PB;
    stage_rows(0, 0, 0, ROWS / 8);
    for (int g = 0; g < ng; ++g) {
        // VMEM queue, oldest first: ... the last DMA rows of stage g, then the N output stores of the last position
        // of stage g-1 (the next stage's rows are issued RPP per position, BEFORE that position's MFMAs: a wave that
        // issues LDS-DMA -- ~90 cycles a piece -- leaves the matrix pipe to its partner meanwhile)
        if (g == 0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const f32x4* L = smem + (g & 1) * (ROWS * 64) + lane;
#pragma unroll
        for (int p = 0; p < PB; ++p) {       // one position at a time: three chains of 64 MFMAs
            if (g + 1 < ng) stage_rows(g + 1, (g + 1) & 1, p * RPP, (p + 1) * RPP);
            f32x4 acc[N], a[2];
            a[0] = L[(p * MG) * 64];
#pragma unroll
            for (int m = 0; m < MG; ++m) {   // A fragment m+1 is in flight behind fragment m's 12 MFMAs
                if (m + 1 < MG) a[(m + 1) & 1] = L[(p * MG + m + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < N; ++n) acc[n] = mfma4(a[m & 1][e], B[n][m][e], (m | e) ? acc[n] : bsv[n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // exactly N stores per lane per position (counted above): positions past the end of the last stage
            // rewrite the last valid one with identical values
            const int pos = min(p0 + PB * g + p, p1 - 1);
            const int slot = dir ? (npos - 1 - pos) : pos;
            char* o = o_tile + (size_t)slot * (2 * kNTile * 64 * 16) + in_block(lane16);
#pragma unroll
            for (int n = 0; n < N; ++n) *(f32x4*)(o + n * 1024) = acc[n];
        }
    }
}

__global__ __launch_bounds__(512, 1) void gemm_dec_ws_kernel(const f32x4* __restrict__ A, long a_tile_stride,
                                                             const f32x4* __restrict__ Wp,
                                                             const float* __restrict__ bias,
                                                             f32x4* __restrict__ gi, long gi_tile_stride,
                                                             int npos, int ntiles) {
    __shared__ f32x4 smem[kDecWsLdsF4];
    const int local = blockIdx.x >> 3;
    const int dir = local & 1;
    const int tile = (local >> 1) * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;
    gemm_dec_ws_body(smem, tile, dir, A, a_tile_stride, Wp, bias, gi, gi_tile_stride, npos);
}

// The same for calls of fewer than half the CUs in (tile, direction) pairs: the positions of a (tile, direction) are cut
// into `parts` runs of `run` positions (a multiple of the stage), one workgroup each, so that a small call still has
// about one workgroup per CU instead of a few long ones -- or, before this kernel, gemm_gi_kernel's thousands of
// single-wave workgroups, whose first loads and store tails are most of a small launch (256 windows: 0.123 ms for
// 0.064 ms of MFMAs).  Each workgroup loads the direction's weights once more; same chains, same bits.
// grid: 8 x ceil(tiles / 8) x 2 directions x parts, tile = 8 (group) + (blockIdx.x & 7) as above.
__global__ __launch_bounds__(512, 1) void gemm_dec_wsp_kernel(const f32x4* __restrict__ A, long a_tile_stride,
                                                              const f32x4* __restrict__ Wp,
                                                              const float* __restrict__ bias,
                                                              f32x4* __restrict__ gi, long gi_tile_stride,
                                                              int npos, int ntiles, int parts, int run) {
    __shared__ f32x4 smem[kDecWsLdsF4];
    const int rest = blockIdx.x >> 3;
    const int part = rest % parts;
    const int local = rest / parts;
    const int dir = local & 1;
    const int tile = (local >> 1) * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;
    const int p0 = part * run;
    gemm_dec_ws_body(smem, tile, dir, A, a_tile_stride, Wp, bias, gi, gi_tile_stride, npos, p0, min(p0 + run, npos));
}

}  // namespace helen
