// kernels.h -- the gfx950 kernels of the polish path (fp32 arithmetic on v_mfma_f32_16x16x4_f32).
//
// Reference semantics being implemented (file:line into kishwarshafin/helen):
//   TransducerGRU.forward             helen/modules/python/models/TransducerModel.py:60-79
//   sliding window / softmax / argmax helen/modules/python/models/predict_gpu.py:97-159
// Layouts are described in layout.h; the launch sequence is in api.hip.
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

// ------------------------------------------------------------------------------------------------
// pack: uint8 pileup windows [n, 1000, F] -> KB16 fp32 operand tiles xa[tile][pos][kb 24][16][4].
// Fuses the reference's host-side `images.type(torch.FloatTensor)` (predict_gpu.py:97); rows of
// windows past n_windows and features past F are zero.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_images_kernel(const uint8_t* __restrict__ img,
                                                          int n_windows, int npos,
                                                          f32x4* __restrict__ xa) {
    const int tile = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;  // (pos, kb, i), i fastest
    const int per_pos = (kFPad / 4) * kTile;        // 384 float4 per (tile, pos)
    if (g >= npos * per_pos) return;
    const int i = g & 15;
    const int kb = (g >> 4) % (kFPad / 4);
    const int pos = g / per_pos;
    const int window = tile * kTile + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (window < n_windows) {
        const uint8_t* p = img + ((size_t)window * npos + pos) * kF + kb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kb * 4 + e < kF) v[e] = (float)p[e];
    }
    xa[((size_t)tile * npos + pos) * per_pos + kb * kTile + i] = v;
}

// Same from float32 x [B, T, F] (the operator-level boundary, TransducerModel.py:60).
__global__ __launch_bounds__(256) void pack_x_f32_kernel(const float* __restrict__ x, int n_windows,
                                                         int T, f32x4* __restrict__ xa,
                                                         long xa_tile_stride) {
    const int tile = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int per_pos = (kFPad / 4) * kTile;
    if (g >= T * per_pos) return;
    const int i = g & 15;
    const int kb = (g >> 4) % (kFPad / 4);
    const int pos = g / per_pos;
    const int window = tile * kTile + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (window < n_windows) {
        const float* p = x + ((size_t)window * T + pos) * kF + kb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kb * 4 + e < kF) v[e] = p[e];
    }
    xa[(size_t)tile * xa_tile_stride + (size_t)pos * per_pos + kb * kTile + i] = v;
}

// hidden [B, 2, H] (TransducerModel.py:68 transposes it to [2, B, H]) <-> KB16 state
// hid[tile][dir][kb 32][16][4].
__global__ __launch_bounds__(256) void pack_hidden_kernel(const float* __restrict__ h, int n_windows,
                                                          float* __restrict__ hid) {
    const int tile = blockIdx.x;
    for (int g = threadIdx.x; g < kHidStride; g += 256) {
        const int dir = g / kHidDirStride;
        const int rem = g % kHidDirStride;
        const int k = (rem >> 6) * 4 + (rem & 3);
        const int row = (rem >> 2) & 15;
        const int window = tile * kTile + row;
        hid[(size_t)tile * kHidStride + g] =
            window < n_windows ? h[((size_t)window * 2 + dir) * kH + k] : 0.f;
    }
}
__global__ __launch_bounds__(256) void unpack_hidden_kernel(const float* __restrict__ hid,
                                                            int n_windows, float* __restrict__ h) {
    const int tile = blockIdx.x;
    for (int g = threadIdx.x; g < kHidStride; g += 256) {
        const int dir = g / kHidDirStride;
        const int rem = g % kHidDirStride;
        const int k = (rem >> 6) * 4 + (rem & 3);
        const int row = (rem >> 2) & 15;
        const int window = tile * kTile + row;
        if (window < n_windows)
            h[((size_t)window * 2 + dir) * kH + k] = hid[(size_t)tile * kHidStride + g];
    }
}

// ------------------------------------------------------------------------------------------------
// Input projection  gi = A . W_ih^T + bias  for both directions (the non-recurrent half of nn.GRU,
// TransducerModel.py:70,72).  A is a KB16 operand with MG = K/16 groups per (tile, position).
//   The 48 column tiles (2 directions x 24) are split over 8 "wave slots": slot v -> direction
//   v>>2, column tiles 6(v&3) .. +5; a workgroup holds HELEN_GEMM_WAVES slots (grid.z the rest) and
//   covers 4 positions, so each wave keeps a 4 x 6 block of 16x16 accumulators.
//   Operands come straight from global memory in a register ping-pong (group m+1 in flight while
//   group m's 96 MFMAs issue): every load is one contiguous 1 KiB per wave and the packed weights
//   (<= 786 KB) stay L2-resident; no LDS, no barriers.  Small workgroups (2 waves) measured best:
//   several independent workgroups per CU overlap each other's prologue/epilogue.
//   bias[dir][col] = b_ih[col] + (col < 2H ? b_hh[col] : 0)   (b_hn is applied inside r*(...)).
// Output gi[tile][slot][dir][ntile 24][lane 64] float4 (FRAG layout); slot = pos for direction 0,
// npos-1-pos for direction 1.
// ------------------------------------------------------------------------------------------------
// waves per projection workgroup; 8 / HELEN_GEMM_WAVES workgroups (grid.z) cover the 48 column tiles
#ifndef HELEN_GEMM_WAVES
#define HELEN_GEMM_WAVES 2
#endif
template <int MG, bool REV_A>
__global__ __launch_bounds__(HELEN_GEMM_WAVES * 64) void gemm_gi_kernel(const f32x4* __restrict__ A, long a_tile_stride,
                                                      const f32x4* __restrict__ Wp,
                                                      const float* __restrict__ bias,
                                                      f32x4* __restrict__ gi, long gi_tile_stride,
                                                      int npos, int ntiles) {
    // Output slot: direction 0 -> position p, direction 1 -> npos-1-p (time-reversed), so the
    // recurrence reads both directions in ascending address order.
    static_assert(MG % 2 == 0, "operand groups are consumed in ping-pong pairs");
    constexpr int P = 4, N = 6;
    const int lane = threadIdx.x & 63;
    // grid.x enumerates (unit, z): unit = (position group, tile), z = which HELEN_GEMM_WAVES wave
    // slots.  Workgroups are dispatched round-robin over the 8 XCDs (id % 8), so the ZB = 8 /
    // HELEN_GEMM_WAVES workgroups that share one unit's A operand get ids u, u+8, u+16, ... inside a
    // block of 8*ZB ids: same XCD, same L2, adjacent in time -> A is fetched from HBM once.
    constexpr int ZB = 8 / HELEN_GEMM_WAVES;
    const int bid = blockIdx.x;
    const int unit = (bid / (8 * ZB)) * 8 + (bid & 7);
    const int zb = (bid >> 3) % ZB;
    const int npg = (npos + P - 1) / P;                 // position groups per tile
    const int tile = unit / npg;
    const int pos0 = (unit % npg) * P;
    if (tile >= ntiles) return;                           // grid is padded to a multiple of 8 units
    const int wave = (threadIdx.x >> 6) + zb * HELEN_GEMM_WAVES;
    const int dir = wave >> 2;
    const int nt0 = (wave & 3) * N;

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

    // Register ping-pong: the operands of group m+1 are in flight while group m's 96 MFMAs issue.
    f32x4 a0[P], b0[N], a1[P], b1[N];
#define HELEN_LOAD_OPS(a, b, m)                                       \
    _Pragma("unroll") for (int p = 0; p < P; ++p)                     \
        a[p] = (REV_A && (m) >= MG / 2) ? a_ptr_b[p][(m) * 64] : a_ptr[p][(m) * 64]; \
    _Pragma("unroll") for (int n = 0; n < N; ++n) b[n] = w_base[(n * MG + (m)) * 64];
#define HELEN_MMA_OPS(a, b)                                           \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                     \
    _Pragma("unroll") for (int p = 0; p < P; ++p)                     \
    _Pragma("unroll") for (int n = 0; n < N; ++n) acc[p][n] = mfma4(a[p][e], b[n][e], acc[p][n]);

    HELEN_LOAD_OPS(a0, b0, 0)
#pragma unroll
    for (int m = 0; m < MG; m += 2) {
        HELEN_LOAD_OPS(a1, b1, m + 1)
        __builtin_amdgcn_sched_barrier(0);
        HELEN_MMA_OPS(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        if (m + 2 < MG) { HELEN_LOAD_OPS(a0, b0, m + 2) }
        __builtin_amdgcn_sched_barrier(0);
        HELEN_MMA_OPS(a1, b1)
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
// Encoder input projection, weight-stationary (fp32 MFMA): gi = X . W_ih^T + bias for all `npos` positions.
//   K is only 96, so a wave can hold its whole slice of W_ih in registers: 4 column tiles x 6 groups =
//   96 registers, loaded once.  Workgroup = 4 waves = one third of the 48 column tiles; grid = 3 column
//   sets x tiles, enumerated so that the three sets of a tile run on one XCD (its xa stream comes from HBM
//   once).  Only the activations move: a stage is 4 positions (24 KiB of KB16 fragments) brought in by
//   LDS-DMA into a 2-deep ring, 384 MFMAs per wave per barrier, and the 16 output stores of a stage stay
//   in flight across the next barrier.  Two workgroups per CU.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_enc_ws_kernel(const f32x4* __restrict__ A, long a_tile_stride,
                                                             const f32x4* __restrict__ Wp,
                                                             const float* __restrict__ bias,
                                                             f32x4* __restrict__ gi, long gi_tile_stride,
                                                             int npos, int ntiles) {
    constexpr int MG = kFPad / 16;          // 6 operand groups of 16 k
    constexpr int PB = 4, N = 4;
    constexpr int ROWS = PB * MG;           // 24 rows of 1 KiB per stage
    __shared__ f32x4 smem[2 * ROWS * 64];   // 48 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int local = blockIdx.x >> 3;
    const int set = local % 3;
    const int tile = (local / 3) * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;
    const int gt0 = 16 * set + N * w;       // first of this wave's global column tiles (dir*24 + nt)
    const int dir = gt0 / kNTile;
    const int nt0 = gt0 % kNTile;
    f32x4 B[N][MG];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < MG; ++m) B[n][m] = Wp[(size_t)((gt0 + n) * MG + m) * 64 + lane];
    float bs[N];
#pragma unroll
    for (int n = 0; n < N; ++n) bs[n] = bias[dir * kG + (nt0 + n) * 16 + (lane & 15)];
    const f32x4* ap = A + (size_t)tile * a_tile_stride + lane;
    auto stage = [&](int g, int b) {        // positions 4g..4g+3: row r = p*6 + m, 6 rows per wave
        f32x4* dst = smem + b * (ROWS * 64);
#pragma unroll
        for (int i = 0; i < ROWS / 4; ++i) {
            const int r = w + 4 * i;
            const int pc = min(PB * g + r / MG, npos - 1);
            const f32x4* src = ap + (size_t)pc * (MG * 64) + (r % MG) * 64;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(dst + r * 64), 16, 0, 0);
        }
    };
    const int ng = (npos + PB - 1) / PB;
    stage(0, 0);
    for (int g = 0; g < ng; ++g) {
        // VMEM queue, oldest first: 6 DMA rows of group g, then the 16 output stores of group g-1
        if (g == 0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + 1 < ng) stage(g + 1, (g + 1) & 1);
        const f32x4* L = smem + (g & 1) * (ROWS * 64) + lane;
        f32x4 acc[PB][N];
#pragma unroll
        for (int p = 0; p < PB; ++p)
#pragma unroll
            for (int n = 0; n < N; ++n) acc[p][n] = splat4(bs[n]);
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            f32x4 a[PB];
#pragma unroll
            for (int p = 0; p < PB; ++p) a[p] = L[(p * MG + m) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int p = 0; p < PB; ++p)
#pragma unroll
                    for (int n = 0; n < N; ++n) acc[p][n] = mfma4(a[p][e], B[n][m][e], acc[p][n]);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            // exactly 16 stores per lane per stage (counted above): positions past the end of the last
            // stage rewrite the last valid one with identical values
            const int pos = min(PB * g + p, npos - 1);
            const int slot = dir ? (npos - 1 - pos) : pos;
            f32x4* o = gi + (size_t)tile * gi_tile_stride + ((size_t)slot * 2 + dir) * (kNTile * 64) + nt0 * 64 + lane;
#pragma unroll
            for (int n = 0; n < N; ++n) o[n * 64] = acc[p][n];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GRU recurrence for one direction of one layer over T dependent steps (nn.GRU cell, see
// oracle/helen_oracle.c gru_dir for the scalar statement).
//   grid (tiles, 2 directions), 4 waves per workgroup, TWO workgroups per CU (two waves per SIMD
//   from independent tiles): while one tile is in its gate math / LDS exchange / barrier, the
//   other tile's MFMAs keep the matrix pipe busy.  That needs <= 256 registers per lane, so:
//     - wave w owns hidden units 32w..32w+31 = six 16-column tiles (r, z, n gates x two halves);
//       the W_hh slices of five of them (160 floats per lane) stay in registers for the whole
//       launch, the sixth is parked in LDS and streamed as a B operand each step;
//     - h lives in LDS in KB16 layout (double-buffered, ONE barrier per step) and is the MFMA A
//       operand of the next step;
//     - the gate pre-activations gi are DMA'd global->LDS (global_load_lds: no registers) one
//       step ahead into a per-wave, single-buffered slot that is refilled as soon as it is read.
//   Each step's h is streamed out as y[tile][slot][dir] (KB16) for the next projection, slot =
//   step index (t for direction 0, T-1-t for direction 1).
//   Direction 1 walks t = T-1 .. 0 (the `_reverse` weights); its h_n is the state after t = 0.
//   gi and y are indexed by SLOT = step order for both directions (the reverse direction is
//   stored time-reversed) so both directions walk memory upwards: descending DMA/store addresses
//   cost 1000+ cycles of VMEM issue stall per step on gfx950.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gru_cell(float ar, float az, float an, float gr, float gz, float gn,
                                          float hp) {
    const float rg = fast_sigmoid(ar + gr);
    const float zg = fast_sigmoid(az + gz);
    const float ng = fast_tanh(gn + rg * an);
    return ng + zg * (hp - ng);  // (1-z)*n + z*h
}

constexpr int kGruLdsF4 = 2 * 512 + 4 * 384 + 4 * 512;  // h[2] | gi[4 waves] | W tile 5[4 waves]

__global__ __launch_bounds__(256, 2) void gru_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                     int slot0_fwd, int slot0_bwd, int T,
                                                     const f32x4* __restrict__ Whp,
                                                     const float* __restrict__ bhn,
                                                     f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                     long y_tile_stride) {
    __shared__ f32x4 smem[kGruLdsF4];  // 72 KiB, one object (two workgroups fit in 160 KiB)
    f32x4* const hbuf = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4* const gbuf = smem + 1024 + w * 384;         // this wave's gi slot: [6][64]
    f32x4* const w5buf = smem + 1024 + 1536 + w * 512; // this wave's parked W tile: [8][64]
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;

    // W_hh slice: W[n = gate*2 + half][m] holds k = 16m + 4q + e, col = unit(half, j)
    f32x4 W[5][8];
    {
        const f32x4* wp = Whp + (size_t)((dir * 4 + w) * 48) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int m = 0; m < 8; ++m) W[n][m] = wp[(n * 8 + m) * 64];
#pragma unroll
        for (int m = 0; m < 8; ++m) w5buf[m * 64 + lane] = wp[(5 * 8 + m) * 64];
    }
    float bn[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) bn[hh] = bhn[dir * kH + 32 * w + 16 * hh + j];

    // gi fragments of this wave: column tile of (gate g, half hh) is g*8 + 2w + hh
    const f32x4* gi_p = gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) +
                        (2 * w) * 64 + lane;
    constexpr long kPosStride = 2 * kNTile * 64;  // float4 per slot
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
    int hoff[2];  // float offset of (row 4q, unit) inside an h buffer; rows r add 4r
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int u = 32 * w + 16 * hh + j;
        hoff[hh] = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[hh][r] = ((const float*)hbuf)[hoff[hh] + 4 * r];
    }
    f32x4* y_p = y + (size_t)tile * y_tile_stride + (size_t)dir * (kHidDirStride / 4);

#ifdef HELEN_GRU_TIMING
    long long tk[7] = {0, 0, 0, 0, 0, 0, 0};
#define HELEN_TICK(i) { __builtin_amdgcn_sched_barrier(0); long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
    long long tlast = __builtin_readcyclecounter();
#else
#define HELEN_TICK(i)
#endif
    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        const f32x4* hb = hbuf + cur * 512 + lane;
        const f32x4* wb = w5buf + lane;

        f32x4 acc[6];
        acc[0] = splat4(0.f);
        acc[1] = splat4(0.f);
        acc[2] = splat4(0.f);
        acc[3] = splat4(0.f);
        acc[4] = splat4(bn[0]);
        acc[5] = splat4(bn[1]);
        // LDS operand ping-pong: group m+1's A / parked-W reads are in flight behind group m's 24
        // MFMAs (pinned with sched_barriers; left alone, hipcc issues the reads right before use
        // and exposes the LDS latency four times per step).
#define HELEN_GRU_MMA(a, b5, m)                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                 \
        _Pragma("unroll") for (int n = 0; n < 5; ++n) acc[n] = mfma4(a[e], W[n][m][e], acc[n]); \
        acc[5] = mfma4(a[e], b5[e], acc[5]);                                        \
    }
        f32x4 a0 = hb[0], b0 = wb[0], a1, b1;
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
            a1 = hb[(m + 1) * 64];
            b1 = wb[(m + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
            HELEN_GRU_MMA(a0, b0, m)
            __builtin_amdgcn_sched_barrier(0);
            if (m + 2 < 8) {
                a0 = hb[(m + 2) * 64];
                b0 = wb[(m + 2) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            HELEN_GRU_MMA(a1, b1, m + 1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef HELEN_GRU_MMA
        HELEN_TICK(0)
        // gate pre-activations of this step (DMA'd during the previous step), then refill the slot
        // VMEM queue of this wave, oldest first: 6 gi DMAs (issued last step), 2 y stores (issued
        // after last step's barrier).  vmcnt(2) = the DMAs have landed; the stores may still fly.
        // (hipcc does not order these LDS reads behind the DMA by itself.)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        HELEN_TICK(5)
        f32x4 G[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) G[n] = gbuf[n * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HELEN_TICK(6)
        if (s + 1 < T) dma_gi(slot0 + s + 1);
        HELEN_TICK(1)

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
        HELEN_TICK(2)
        // raw barrier: only LDS traffic has to be drained, the gi DMA stays in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        HELEN_TICK(3)
        // stream h(t) out as the layer output
        f32x4* yo = y_p + (size_t)s * (kYStride / 4);  // slot s: t for dir 0, T-1-t for dir 1
        const f32x4* hn4 = hbuf + (cur ^ 1) * 512;
        yo[tid] = hn4[tid];
        yo[tid + 256] = hn4[tid + 256];
        HELEN_TICK(4)
    }
#ifdef HELEN_GRU_TIMING
    if (tile == 0 && lane == 0) {
        printf("gru dir %d wave %d: cycles/step  mfma %lld  vmwait %lld  Gread %lld  dma-issue %lld  gates %lld  barrier %lld  ycopy %lld\n",
               dir, w, tk[0] / T, tk[5] / T, tk[6] / T, tk[1] / T, tk[2] / T, tk[3] / T, tk[4] / T);
    }
#endif
    const f32x4* hl = hbuf + (T & 1) * 512;
    hid_p[tid] = hl[tid];
    hid_p[tid + 256] = hl[tid + 256];
}

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

// ------------------------------------------------------------------------------------------------
// fp32x3 recurrence (HELEN_PRECISION_FP32X3, opt-in): fp32-class h . W_hh^T on the bf16 matrix cores.
//   Every fp32 value is the exact sum of three bf16 terms (3 x 8 significand bits): h = h1 + h2 + h3,
//   w = w1 + w2 + w3.  Each partial product hi*wj is exact in fp32, and the six leading ones
//   (i + j <= 4) reproduce h*w to ~2^-26 relative (RNE splits: |h2| <= 2^-9 |h|, |h3| <= 2^-18 |h|; the
//   dropped h2*w3, h3*w2, h3*w3 are <= 2 * 2^-27) -- a quarter of fp32's own rounding unit -- so
//       sum_k h_k w_k = sum over the 6 products of (bf16 MFMA, fp32 accumulate)
//   is an fp32 dot product up to summation order, at 6 x 16.7 cycles per 32 k on
//   v_mfma_f32_16x16x32_bf16 instead of 8 x 32 cycles on v_mfma_f32_16x16x4_f32.
//   W_hh's three terms for a wave's columns must stay in registers (3 x the bf16 kernel's), so the
//   workgroup is 8 waves, wave v owning hidden units 16v..16v+15 (one 16-column tile per gate).
//   The new h is split once, by the lane that produced it, into three bf16 planes in LDS laid out as
//   the A fragment of the K = 32 MFMA (unit (k/8, row) of 8 bf16 = 16 bytes; group M of lane (row, q)
//   is unit 4M + q); an fp32 copy feeds the layer output y and the carried state, which keep the
//   fp32 path's layouts -- only this kernel changes, the projections stay on fp32 MFMAs.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float f) {   // RNE, via the hardware convert
    const bf16x2_t p = __builtin_convertvector((f32x2){f, 0.f}, bf16x2_t);
    return (unsigned short)(__builtin_bit_cast(unsigned, p) & 0xffffu);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}

template <int NT>
__global__ __launch_bounds__(512) void gru_x3_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                     int slot0_fwd, int slot0_bwd, int T,
                                                     const bf16x8* __restrict__ W3,
                                                     const float* __restrict__ bhn,
                                                     f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                     long y_tile_stride, f32x4* __restrict__ yplanes,
                                                     long yp_tile_stride, int ntiles) {
    // NT window tiles per workgroup share the resident W_hh terms and one barrier per step.  Measured:
    // NT = 2 is no faster than NT = 1 (0.459 vs 0.449 ms) -- a step is 2 x 1200 cycles of MFMA issue plus
    // 2 x 940 cycles of gate/split VALU work per SIMD, which do not overlap, not barrier latency -- so
    // NT = 1 (more workgroups, half the LDS) is what is launched.  A workgroup past the last tile
    // recomputes the last one (identical stores).
    // Layer output: fp32 y[tile][slot][dir] (KB16, for the heads) when `y` is given, and/or the three
    // bf16 planes yplanes[tile][slot][dir][plane][256 units] (for gemm_dec_x3_kernel) when given.
    // LDS per tile: fp32 h [2][512 f4] | bf16 planes [2 buffers][3 terms][256 units of 16 B] |
    // gi slots [8 waves][3][64 f4]
    constexpr int kPerTile = 2 * 512 + 2 * 3 * 256 + 8 * 192;   // 4096 f4 = 64 KiB
    __shared__ f32x4 smem[NT * kPerTile];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: hidden units 16v..16v+15
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int u = 16 * v + j;                                  // this lane's hidden unit
    int tile[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) tile[n] = min((int)blockIdx.x * NT + n, ntiles - 1);

    // W[g][M][t]: term t of W_hh[row g*128 + u][k = 32M + 8q + e], e = 0..7
    bf16x8 W[3][4][3];
    {
        const bf16x8* wp = W3 + (size_t)((dir * 8 + v) * 36) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int M = 0; M < 4; ++M)
#pragma unroll
                for (int t = 0; t < 3; ++t) W[g][M][t] = wp[((g * 4 + M) * 3 + t) * 64];
    }
    const float bn = bhn[dir * kH + u];

    constexpr long kPosStride = 2 * kNTile * 64;
    auto hbuf = [&](int n) { return smem + n * kPerTile; };
    auto planes = [&](int n) { return smem + n * kPerTile + 1024; };
    auto gbuf = [&](int n) { return smem + n * kPerTile + 1024 + 1536 + v * 192; };
    auto dma_gi = [&](int n, int slot) {
        const f32x4* p = gi + (size_t)tile[n] * gi_tile_stride + (size_t)dir * (kNTile * 64) + v * 64 + lane +
                         (size_t)slot * kPosStride;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1)))*)(p + (g * 8) * 64),
                (void __attribute__((address_space(3)))*)(gbuf(n) + g * 64), 16, 0, 0);
    };
    // this lane's 4 values: rows 4q + r of unit u.  fp32 h: float index ((u>>2)*16 + 4q + r)*4 + (u&3);
    // planes: bf16 index ((u>>3)*16 + 4q + r)*8 + (u&7) inside a 256-unit plane
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);
    auto store_h = [&](int n, int buf, int r, float h) {
        ((float*)(hbuf(n) + buf * 512))[hoff + 4 * r] = h;
        unsigned short* pl = (unsigned short*)(planes(n) + buf * 768);
        const unsigned short t1 = bf16_bits(h);
        const float r1 = h - bf16_to_f32(t1);
        const unsigned short t2 = bf16_bits(r1);
        const float r2 = r1 - bf16_to_f32(t2);
        const unsigned short t3 = bf16_bits(r2);
        pl[0 * 2048 + poff + 8 * r] = t1;
        pl[1 * 2048 + poff + 8 * r] = t2;
        pl[2 * 2048 + poff + 8 * r] = t3;
    };

#pragma unroll
    for (int n = 0; n < NT; ++n) {
        hbuf(n)[tid] = (hid + ((size_t)tile[n] * 2 + dir) * (kHidDirStride / 4))[tid];
        dma_gi(n, slot0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float hprev[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[n][r] = ((const float*)hbuf(n))[hoff + 4 * r];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) store_h(n, 0, r, hprev[n][r]);   // planes of h0 (fp32 copy rewritten in place)
    __syncthreads();
#ifdef HELEN_GRU_TIMING
    long long tk[7] = {0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        f32x4 acc[NT][3];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8* pa = (const bf16x8*)(planes(n) + cur * 768) + lane;
            acc[n][0] = splat4(0.f);
            acc[n][1] = splat4(0.f);
            acc[n][2] = splat4(bn);
#pragma unroll
            for (int M = 0; M < 4; ++M) {
                const bf16x8 a1 = pa[0 * 256 + M * 64], a2 = pa[1 * 256 + M * 64], a3 = pa[2 * 256 + M * 64];
                const bf16x8 at[3] = {a1, a2, a3};
                constexpr int TA[6] = {0, 2, 1, 0, 1, 0};   // six leading products, smallest first;
                constexpr int TB[6] = {2, 0, 1, 1, 0, 0};   // product index outermost: 3 accumulators rotate
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[n][g] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[TA[k]], W[g][M][TB[k]], acc[n][g], 0, 0, 0);
            }
        }
        HELEN_TICK(0)
#ifdef HELEN_GRU_TIMING
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]));
        HELEN_TICK(1)
#endif
        // VMEM queue, oldest first: 3 gi DMAs per tile, then the previous step's output stores (at least
        // one per tile): the DMAs have landed once no more than NT operations are outstanding
        if (NT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        HELEN_TICK(2)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 G[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) G[g] = gbuf(n)[g * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (s + 1 < T) dma_gi(n, slot0 + s + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hn = gru_cell(acc[n][0][r], acc[n][1][r], acc[n][2][r], G[0][r], G[1][r], G[2][r],
                                          hprev[n][r]);
                hprev[n][r] = hn;
                store_h(n, cur ^ 1, r, hn);
            }
        }
        HELEN_TICK(3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HELEN_TICK(4)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        HELEN_TICK(5)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (y != nullptr) {
                f32x4* yo = y + (size_t)tile[n] * y_tile_stride + (size_t)dir * (kHidDirStride / 4) +
                            (size_t)s * (kYStride / 4);
                yo[tid] = (hbuf(n) + (cur ^ 1) * 512)[tid];
            }
            if (yplanes != nullptr) {   // 768 units of 16 B per (tile, slot, dir)
                f32x4* po = yplanes + (size_t)tile[n] * yp_tile_stride + ((size_t)s * 2 + dir) * 768;
                const f32x4* ps = planes(n) + (cur ^ 1) * 768;
                po[tid] = ps[tid];
                if (tid < 256) po[512 + tid] = ps[512 + tid];
            }
        }
    }
#ifdef HELEN_GRU_TIMING
    if (blockIdx.x == 0 && lane == 0 && (v == 0 || v == 5))
        printf("gru_x3 dir %d wave %d: cycles/step  mfma-issue %lld  mfma-drain %lld  vmwait %lld  G+gates+stores %lld  lgkm %lld  barrier %lld  (ycopy in mfma-issue)\n",
               dir, v, tk[0] / T, tk[1] / T, tk[2] / T, tk[3] / T, tk[4] / T, tk[5] / T);
#endif
#pragma unroll
    for (int n = 0; n < NT; ++n)
        (hid + ((size_t)tile[n] * 2 + dir) * (kHidDirStride / 4))[tid] = (hbuf(n) + (T & 1) * 512)[tid];
}

// ------------------------------------------------------------------------------------------------
// fp32x3 decoder projection: gi = Y1 . W_ih^T + b with both operands as three bf16 terms (six exact
// partial products per term pair, fp32 accumulate; see gru_x3_kernel).  Y1 arrives already split
// (the encoder recurrence wrote the planes), W_ih was split on the host.
//   With MFMAs this cheap the kernel lives or dies by operand traffic, so it is WEIGHT-STATIONARY:
//   a workgroup (8 waves) owns 16 of the 48 column tiles (2 per wave) and keeps all three terms of
//   their W_ih slice -- 2 tiles x 8 groups x 3 terms = 192 registers per lane -- for its whole life,
//   walking the 100 positions of one window tile two at a time.  Per stage only A moves: 2 positions
//   x 3 planes x 8 groups = 48 rows of 1 KiB, DMA'd global->LDS into a 2-deep ring while the previous
//   stage is multiplied (192 MFMAs per wave and stage, one barrier per stage): 32 MFMAs per KiB
//   staged instead of 8-11 for a block-tiled kernel that also stages the weights.
//   k < 128 comes from the forward encoder direction at slot p, k >= 128 from the backward one at
//   slot npos-1-p; output slot order as gemm_gi_kernel.  grid (3 column sets, window tiles).
//   NP = 3 planes / weight terms (fp32x3: six products) or 1 (HELEN_PRECISION_BF16: Y1 and W_ih rounded to
//   bf16, one product); PB = positions per stage (NP * PB * 8 rows of 1 KiB).
// ------------------------------------------------------------------------------------------------
template <int NP, int PB>
__global__ __launch_bounds__(512) void gemm_dec_x3_kernel(const f32x4* __restrict__ yplanes,
                                                          long yp_tile_stride,
                                                          const f32x4* __restrict__ W3d,
                                                          const float* __restrict__ bias,
                                                          f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int npos, int ntiles) {
    static_assert(NP == 1 || NP == 3, "one bf16 plane or the three-term split");
    constexpr int ROWS = PB * NP * 8;       // rows of 1 KiB per stage: (position, plane, group)
    static_assert(ROWS % 8 == 0 && 2 * ROWS <= 96, "two stages must fit 96 KiB");
    __shared__ f32x4 smem[2 * ROWS * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of 3 * roundup8(ntiles) ids.  Workgroups go round-robin over the 8 XCDs (id % 8): the three
    // column sets of a tile get consecutive local slots of ONE XCD, so its A stream is fetched from HBM
    // once and served from that XCD's L2 to the other two.
    const int local = blockIdx.x >> 3;
    const int set = local % 3;
    const int tile = (local / 3) * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;
    const int gt0 = 16 * set + 2 * w;          // first of this wave's two global column tiles (dir*24 + nt)
    const int dir = gt0 / kNTile;
    const int nt = gt0 % kNTile;

    // weight terms -> registers: B[ti][M][t]  (W3d always holds three terms; term 0 = RNE(w))
    bf16x8 B[2][8][NP];
    {
        const bf16x8* wp = (const bf16x8*)W3d + lane;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int M = 0; M < 8; ++M)
#pragma unroll
                for (int t = 0; t < NP; ++t) B[ti][M][t] = wp[((size_t)((gt0 + ti) * 8 + M) * 3 + t) * 64];
    }
    float bs[2];
    bs[0] = bias[dir * kG + nt * 16 + (lane & 15)];
    bs[1] = bias[dir * kG + (nt + 1) * 16 + (lane & 15)];

    const f32x4* yp = yplanes + (size_t)tile * yp_tile_stride + lane;
    // DMA of position group g into buffer b; row r = (p*NP + plane)*8 + M is copied by wave r % 8
    auto stage = [&](int g, int b) {
        f32x4* dst = smem + b * (ROWS * 64);
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int r = w + 8 * i;
            const int M = r & 7, plane = (r >> 3) % NP, p = r / (8 * NP);
            const int part = M >> 2;
            const int pc = min(PB * g + p, npos - 1);
            const int slot = part ? (npos - 1 - pc) : pc;
            const f32x4* src = yp + ((size_t)slot * 2 + part) * (NP * 256) + plane * 256 + (M & 3) * 64;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(dst + r * 64), 16, 0, 0);
        }
    };
    constexpr int NPROD = NP == 3 ? 6 : 1;
    constexpr int TA[6] = {NP == 3 ? 0 : 0, 2, 1, 0, 1, 0};   // leading products, smallest first: term of A
    constexpr int TB[6] = {NP == 3 ? 2 : 0, 0, 1, 1, 0, 0};   //                                   term of B
    const int ng = (npos + PB - 1) / PB;
    stage(0, 0);
    for (int g = 0; g < ng; ++g) {
        // this wave's rows of group g have landed; after the barrier everybody's have, and the
        // other buffer (read during group g-1) is free for group g+1.  VMEM queue, oldest first: the
        // DMA rows of group g, then the 2 * PB output stores of group g-1 -- which may stay in flight.
        if (g == 0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PB) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + 1 < ng) stage(g + 1, (g + 1) & 1);
        const bf16x8* L = (const bf16x8*)(smem + (g & 1) * (ROWS * 64)) + lane;
        f32x4 acc[PB][2];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            acc[p][0] = splat4(bs[0]);
            acc[p][1] = splat4(bs[1]);
        }
#pragma unroll
        for (int M = 0; M < 8; ++M) {
            bf16x8 a[PB][NP];
#pragma unroll
            for (int p = 0; p < PB; ++p)
#pragma unroll
                for (int t = 0; t < NP; ++t) a[p][t] = L[((p * NP + t) * 8 + M) * 64];
#pragma unroll
            for (int k = 0; k < NPROD; ++k)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int p = 0; p < PB; ++p)
                        acc[p][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][TA[k]], B[ti][M][TB[k]],
                                                                            acc[p][ti], 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            // exactly 2 * PB stores per lane per stage (counted above): positions past the end of the last
            // stage rewrite the last valid one with identical values
            const int pos = min(PB * g + p, npos - 1);
            const int slot = dir ? (npos - 1 - pos) : pos;
            f32x4* o = gi + (size_t)tile * gi_tile_stride + ((size_t)slot * 2 + dir) * (kNTile * 64) + nt * 64 + lane;
            o[0] = acc[p][0];
            o[64] = acc[p][1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32x3 encoder projection.  The encoder input is raw pileup counts 0..255 (predict_gpu.py:97): every x
// is EXACTLY one bf16 term, so x*w = x*w1 + x*w2 + x*w3 with exact partial products -- three bf16 MFMAs
// per 32 k.  pack_images_x3_kernel writes the counts straight as bf16 A fragments (K padded 90 -> 96 =
// 3 groups), gemm_enc_x3_kernel is weight-stationary like gemm_dec_x3_kernel: 2 column tiles per
// wave (72 registers of weight terms), 8 positions per stage (24 KiB of A), and it runs at the speed
// of its fp32 output stream (3 MB per window).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_images_x3_kernel(const uint8_t* __restrict__ img, int n_windows,
                                                             int npos, f32x4* __restrict__ xb) {
    // one 16-byte unit (8 bf16) per thread: unit index within (tile, pos) = M*64 + q*16 + row
    const int tile = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= npos * 192) return;
    const int row = g & 15;
    const int o = (g >> 4) % 12;          // octet of k: k = 8*o + e
    const int pos = g / 192;
    const int window = tile * kTile + row;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0;
    if (window < n_windows) {
        const uint8_t* p = img + ((size_t)window * npos + pos) * kF + o * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (o * 8 + e < kF) v[e] = bf16_bits((float)p[e]);   // exact: integers <= 255
    }
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16);
    u.y = v[2] | ((unsigned)v[3] << 16);
    u.z = v[4] | ((unsigned)v[5] << 16);
    u.w = v[6] | ((unsigned)v[7] << 16);
    xb[((size_t)tile * npos + pos) * 192 + o * 16 + row] = __builtin_bit_cast(f32x4, u);
}

//   TERMS = 3 (fp32x3) or 1 (HELEN_PRECISION_BF16: W_ih rounded to bf16, i.e. the first term only).
template <int TERMS>
__global__ __launch_bounds__(512) void gemm_enc_x3_kernel(const f32x4* __restrict__ xb, long xb_tile_stride,
                                                          const f32x4* __restrict__ W3e,
                                                          const float* __restrict__ bias,
                                                          f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int npos, int ntiles) {
    constexpr int PB = 8, ROWS = PB * 3;    // rows of 1 KiB per stage: (position, group)
    __shared__ f32x4 smem[2 * ROWS * 64];   // 48 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of 3 * roundup8(ntiles) ids.  Workgroups go round-robin over the 8 XCDs (id % 8): the three
    // column sets of a tile get consecutive local slots of ONE XCD, so its A stream is fetched from HBM
    // once and served from that XCD's L2 to the other two.
    const int local = blockIdx.x >> 3;
    const int set = local % 3;
    const int tile = (local / 3) * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;
    const int gt0 = 16 * set + 2 * w;
    const int dir = gt0 / kNTile;
    const int nt = gt0 % kNTile;
    bf16x8 B[2][3][TERMS];
    {
        const bf16x8* wp = (const bf16x8*)W3e + lane;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int M = 0; M < 3; ++M)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) B[ti][M][t] = wp[((size_t)((gt0 + ti) * 3 + M) * 3 + t) * 64];
    }
    float bs[2];
    bs[0] = bias[dir * kG + nt * 16 + (lane & 15)];
    bs[1] = bias[dir * kG + (nt + 1) * 16 + (lane & 15)];
    const f32x4* xp = xb + (size_t)tile * xb_tile_stride + lane;
    auto stage = [&](int g, int b) {    // positions 8g..8g+7: 24 rows, 3 per wave; row r = p*3 + M
        f32x4* dst = smem + b * (ROWS * 64);
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int r = w + 8 * i;
            const int pc = min(PB * g + r / 3, npos - 1);
            const f32x4* src = xp + (size_t)pc * 192 + (r % 3) * 64;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(dst + r * 64), 16, 0, 0);
        }
    };
    const int ng = (npos + PB - 1) / PB;
    stage(0, 0);
    for (int g = 0; g < ng; ++g) {
        // VMEM queue, oldest first: 3 DMA rows of group g, then 16 output stores of group g-1
        if (g == 0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + 1 < ng) stage(g + 1, (g + 1) & 1);
        const bf16x8* L = (const bf16x8*)(smem + (g & 1) * (ROWS * 64)) + lane;
        f32x4 acc[PB][2];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            acc[p][0] = splat4(bs[0]);
            acc[p][1] = splat4(bs[1]);
        }
#pragma unroll
        for (int M = 0; M < 3; ++M) {
            bf16x8 a[PB];
#pragma unroll
            for (int p = 0; p < PB; ++p) a[p] = L[(p * 3 + M) * 64];
#pragma unroll
            for (int t = TERMS - 1; t >= 0; --t)   // smallest term first
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int p = 0; p < PB; ++p)
                        acc[p][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], B[ti][M][t], acc[p][ti], 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            // every stage issues exactly 16 stores per lane (counted above): out-of-range positions of the
            // last stage rewrite the last valid one with identical values
            const int pos = min(PB * g + p, npos - 1);
            const int slot = dir ? (npos - 1 - pos) : pos;
            f32x4* o = gi + (size_t)tile * gi_tile_stride + ((size_t)slot * 2 + dir) * (kNTile * 64) + nt * 64 + lane;
            o[0] = acc[p][0];
            o[64] = acc[p][1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Heads + softmax + accumulate + argmax (TransducerModel.py:75-76, predict_gpu.py:137-156).
//   One 16-column MFMA tile is exactly the 5 base + 11 run-length logits of 16 windows.
//   grid (tiles, groups of kHeadsSpan positions), 4 waves striding over the positions of the group
//   (many small workgroups: the kernel is latency/HBM-bound, so it wants waves in flight).
//   mode 0 (polish): positions 50c+t; the first half of chunk c receives its second (final)
//     contribution -> add the pending softmax of chunk c-1, argmax, labels; the second half is
//     parked in `pending` for chunk c+1 (or is final for the last chunk).  A position gets at most
//     two contributions, and 0 + a + b == a + b in fp32, so this equals the reference's
//     zero-pad-and-add into a [B,1000,C] accumulator.
//   mode 1 (logits): write base[B,T,5] / rle[B,T,11] logits (the operator-level boundary).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 16);
    return v;
}
// argmax with first-maximum tie-break (torch.max on CPU, predict_gpu.py:155)
__device__ __forceinline__ int group16_argmax(float v, int idx) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float ov = __shfl_xor(v, o, 16);
        const int oi = __shfl_xor(idx, o, 16);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    return idx;
}

constexpr int kHeadsSpan = 10;  // positions per workgroup; divides kJump so a group never straddles halves

__global__ __launch_bounds__(256) void heads_kernel(
    const f32x4* __restrict__ y2, long y_tile_stride, const f32x4* __restrict__ Whd,
    const float* __restrict__ bhd, int mode, int chunk, int T, int n_windows,
    f32x4* __restrict__ pending, uint8_t* __restrict__ bases, uint8_t* __restrict__ rles,
    float* __restrict__ acc_base, float* __restrict__ acc_rle, float* __restrict__ logit_base,
    float* __restrict__ logit_rle) {
    __shared__ uint8_t lab[2][kTile][kHeadsSpan];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int t0 = blockIdx.y * kHeadsSpan;
    const int t1 = min(T, t0 + kHeadsSpan);
    const int half = t0 / kJump;
    const bool isb = j < kNB;

    f32x4 B[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) B[m] = Whd[m * 64 + lane];
    const float bias = bhd[j];

    const bool park = (mode == 0) && (half == 1) && (chunk < kChunks - 1);
    const bool add_prev = (mode == 0) && (half == 0) && (chunk > 0);

    for (int t = t0 + w; t < t1; t += 4) {
        // y2[tile][slot][fwd | bwd]: the bwd half of position t sits in slot T-1-t
        const f32x4* a_p = y2 + (size_t)tile * y_tile_stride + (size_t)t * (kYStride / 4) + lane;
        const f32x4* a_pb = y2 + (size_t)tile * y_tile_stride + (size_t)(T - 1 - t) * (kYStride / 4) + lane;
        f32x4 acc0 = splat4(bias);
        f32x4 acc1 = splat4(0.f);
#pragma unroll
        for (int m = 0; m < 16; m += 2) {
            const f32x4 a0 = (m >= 8 ? a_pb : a_p)[m * 64];
            const f32x4 a1 = (m >= 8 ? a_pb : a_p)[(m + 1) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = mfma4(a0[e], B[m][e], acc0);
                acc1 = mfma4(a1[e], B[m + 1][e], acc1);
            }
        }
        const f32x4 logit = acc0 + acc1;  // row 4q+r (window), col j (class)

        if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int window = tile * kTile + 4 * q + r;
                if (window < n_windows) {
                    if (isb)
                        logit_base[((size_t)window * T + t) * kNB + j] = logit[r];
                    else
                        logit_rle[((size_t)window * T + t) * kNR + (j - kNB)] = logit[r];
                }
            }
            continue;
        }

        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = logit[r];
            const float mb = group16_max(isb ? x : -INFINITY);
            const float mr = group16_max(isb ? -INFINITY : x);
            const float e = expf(x - (isb ? mb : mr));
            const float sb = group16_sum(isb ? e : 0.f);
            const float sr = group16_sum(isb ? 0.f : e);
            p[r] = e / (isb ? sb : sr);
        }
        // `pending` is double-buffered by chunk parity: this launch's second half parks into slot
        // chunk&1 while its first half still reads what chunk-1 parked in the other slot.
        if (park) {
            pending[(((size_t)tile * 2 + (chunk & 1)) * kJump + (t - kJump)) * 64 + lane] = p;
            continue;
        }
        if (add_prev) p += pending[(((size_t)tile * 2 + ((chunk - 1) & 1)) * kJump + t) * 64 + lane];
        const int pos = chunk * kJump + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int window = tile * kTile + 4 * q + r;
            if (window < n_windows) {
                if (acc_base != nullptr && isb)
                    acc_base[((size_t)window * kSeq + pos) * kNB + j] = p[r];
                if (acc_rle != nullptr && !isb)
                    acc_rle[((size_t)window * kSeq + pos) * kNR + (j - kNB)] = p[r];
            }
            const int ib = group16_argmax(isb ? p[r] : -1.f, isb ? j : 99);
            const int ir = group16_argmax(isb ? -1.f : p[r], isb ? 99 : j);
            if (j == 0) {
                lab[0][4 * q + r][t - t0] = (uint8_t)ib;
                lab[1][4 * q + r][t - t0] = (uint8_t)(ir - kNB);
            }
        }
    }
    if (mode != 0 || park) return;
    __syncthreads();
    const int span = t1 - t0;
    for (int g = tid; g < 2 * kTile * kHeadsSpan; g += 256) {
        const int kind = g / (kTile * kHeadsSpan);
        const int rem = g % (kTile * kHeadsSpan);
        const int win = rem / kHeadsSpan;
        const int tl = rem % kHeadsSpan;
        const int window = tile * kTile + win;
        if (window < n_windows && tl < span) {
            uint8_t* out = kind ? rles : bases;
            out[(size_t)window * kSeq + chunk * kJump + t0 + tl] = lab[kind][win][tl];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Heads + cross-entropy + confusion counts: the per-chunk body of the reference's evaluation loop
// (models/test.py:104-121) for chunk `chunk` of 16-window tiles.
//   logits as in heads_kernel; per position: nll_base = logsumexp(base) - base[label_base],
//   nll_rle likewise (nn.CrossEntropyLoss = log_softmax + nll), predictions = first-maximum argmax of
//   the LOGITS (torchnet ConfusionMeter: np.argmax), confusion[target][predicted] += 1.
//   Outputs: stats[window][chunk][group of kHeadsSpan positions][3] = (sum nll_base, sum w[l]*nll_rle,
//   sum w[l]) summed over the group's positions in position order (deterministic; the host finishes
//   the per-batch means), and the two confusion matrices accumulated with integer atomics.
//   Labels outside 0..4 / 0..10 are the caller's error (torch raises); they are clamped here only
//   to keep the accesses in range.
// ------------------------------------------------------------------------------------------------
struct RleClassWeights {
    float w[kNR];
};

__global__ __launch_bounds__(256) void heads_eval_kernel(
    const f32x4* __restrict__ y2, long y_tile_stride, const f32x4* __restrict__ Whd,
    const float* __restrict__ bhd, int chunk, int T, int n_windows,
    const uint8_t* __restrict__ label_base, const uint8_t* __restrict__ label_rle, RleClassWeights cw,
    float* __restrict__ stats, unsigned long long* __restrict__ conf_base,
    unsigned long long* __restrict__ conf_rle) {
    __shared__ float vals[3][kTile][kHeadsSpan];
    __shared__ unsigned hist_b[kNB * kNB], hist_r[kNR * kNR];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int t0 = blockIdx.y * kHeadsSpan;
    const int t1 = min(T, t0 + kHeadsSpan);
    const bool isb = j < kNB;
    for (int g = tid; g < 3 * kTile * kHeadsSpan; g += 256) (&vals[0][0][0])[g] = 0.f;
    if (tid < kNB * kNB) hist_b[tid] = 0;
    if (tid < kNR * kNR) hist_r[tid] = 0;
    __syncthreads();

    f32x4 B[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) B[m] = Whd[m * 64 + lane];
    const float bias = bhd[j];

    for (int t = t0 + w; t < t1; t += 4) {
        const f32x4* a_p = y2 + (size_t)tile * y_tile_stride + (size_t)t * (kYStride / 4) + lane;
        const f32x4* a_pb = y2 + (size_t)tile * y_tile_stride + (size_t)(T - 1 - t) * (kYStride / 4) + lane;
        f32x4 acc0 = splat4(bias);
        f32x4 acc1 = splat4(0.f);
#pragma unroll
        for (int m = 0; m < 16; m += 2) {
            const f32x4 a0 = (m >= 8 ? a_pb : a_p)[m * 64];
            const f32x4 a1 = (m >= 8 ? a_pb : a_p)[(m + 1) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = mfma4(a0[e], B[m][e], acc0);
                acc1 = mfma4(a1[e], B[m + 1][e], acc1);
            }
        }
        const f32x4 logit = acc0 + acc1;  // row 4q+r (window), col j (class)
        const int pos = chunk * kJump + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int window = tile * kTile + 4 * q + r;
            const bool valid = window < n_windows;
            const int lb = valid ? min((int)label_base[(size_t)window * kSeq + pos], kNB - 1) : 0;
            const int lr = valid ? min((int)label_rle[(size_t)window * kSeq + pos], kNR - 1) : 0;
            const float x = logit[r];
            const float mb = group16_max(isb ? x : -INFINITY);
            const float mr = group16_max(isb ? -INFINITY : x);
            const float e = expf(x - (isb ? mb : mr));
            const float sb = group16_sum(isb ? e : 0.f);
            const float sr = group16_sum(isb ? 0.f : e);
            const float xb = group16_sum((isb && j == lb) ? x : 0.f);
            const float xr = group16_sum((!isb && j - kNB == lr) ? x : 0.f);
            const int pb = group16_argmax(isb ? x : -INFINITY, isb ? j : 99);
            const int pr = group16_argmax(isb ? -INFINITY : x, isb ? 99 : j) - kNB;
            if (j == 0 && valid) {
                const float wr = cw.w[lr];
                vals[0][4 * q + r][t - t0] = (mb + logf(sb)) - xb;
                vals[1][4 * q + r][t - t0] = wr * ((mr + logf(sr)) - xr);
                vals[2][4 * q + r][t - t0] = wr;
                atomicAdd(&hist_b[lb * kNB + pb], 1u);
                atomicAdd(&hist_r[lr * kNR + pr], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 3 * kTile) {
        const int k = tid / kTile, win = tid % kTile;
        const int window = tile * kTile + win;
        if (window < n_windows) {
            float sum = 0.f;
            for (int tl = 0; tl < t1 - t0; ++tl) sum += vals[k][win][tl];
            stats[(((size_t)window * kChunks + chunk) * (kWin / kHeadsSpan) + blockIdx.y) * 3 + k] = sum;
        }
    }
    if (tid < kNB * kNB && hist_b[tid]) atomicAdd(&conf_base[tid], (unsigned long long)hist_b[tid]);
    if (tid < kNR * kNR && hist_r[tid]) atomicAdd(&conf_rle[tid], (unsigned long long)hist_r[tid]);
}

}  // namespace helen
