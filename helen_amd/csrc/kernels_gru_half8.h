// kernels_gru_half8.h -- fp32 GRU recurrence, HALF a window tile (8 windows) per 4-wave workgroup: small calls
#pragma once
#include "kernels_gru_single8.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// For calls of at most a quarter of the chip's CUs in tiles (64 tiles = 1,024 windows on 256 CUs).  There even
// gru_single8_kernel leaves half the CUs idle, and a step is bound by the 192 fp32 MFMAs a SIMD issues for its 16
// windows (6,144 cycles = 2.56 us): 16 rows is the M of v_mfma_f32_16x16x4_f32, so a tile cannot be cut in two
// with that instruction.  v_mfma_f32_4x4x1_16b_f32 can: 16 independent 4x4 blocks, K = 1, at the same rate per
// flop (measured 8.68 cycles for 512 flops; scripts/ubench/mfma_4x4x1_semantics.hip), and by the ISA's own rule
// an fp32 MFMA is a k-ordered fmaf chain, so sixteen of them in the k order of four 16x16x4 give the same bits
// (same probe: 0 of 51,200 values differ).  A workgroup here takes 8 windows (two groups of 4) of a tile and one
// direction, a launch has 4 x tiles workgroups, and a step costs a SIMD 384 of those MFMAs = 3,333 cycles.
//
// Blocks of an MFMA: block b = (window group wg = b >> 3, column quad cq = b & 7): 8 windows x 32 columns per
// instruction.  Wave w (one per SIMD) owns hidden units 32w .. 32w+31 of all three gates: one MFMA per gate and k.
//   A operand (h): the broadcast control CBSZ = 3 feeds the eight blocks of a window group from ONE block's lanes,
//     chosen by ABID, so a register holds 8 different k for the 8 windows: 16 registers per step, fetched from LDS
//     as four ds_read_b128 (abuf: [k-octet group][lane] float4);
//   B operand (W_hh): both window groups use the same 32 columns; BLGP = 1 / 2 takes B from lanes 0-31 / 32-63 for
//     all lanes, so a register holds TWO k: 3 x 64 = 192 registers per wave (the wave has the SIMD's 512).
// MFMA t = 0..127 of a gate's chain is k = 16m + 4q + e with (m, e, q) = (t >> 4, (t >> 2) & 3, t & 3): the order
// in which gru_kernel / gru_pair_kernel / gru_single8_kernel walk k (instruction (m, e) of theirs adds q = 0..3).
// The lane that holds D[block (wg, cq)][row r][col j] -- window 4 wg + r, unit 32w + 4cq + j -- has the same four
// rows of one unit per gate as a lane of the 16x16x4 kernels: the same gate cell (gru_cell4), the same carried
// state in registers.  The new h goes to LDS twice: abuf (next step's A registers) and hbuf (the tile layout of
// layout.h, 8 of its 16 rows), from which the layer output / final hidden state leave as one float4 per thread
// and the decoder's head products are fed.
// Decoder heads: wave w multiplies the k-slices 2w and 2w+1 (16 k each) of h(s-1) by the head weights -- blocks
// (window group, slice, class quad), CBSZ = 2 -- sixteen MFMAs in the (e, q) order of the other kernels' four
// 16x16x4, partials to LDS, added in the same order by one wave: the same logits bit for bit.
// grid (2 x tiles, 2 directions), 256 threads.
// ------------------------------------------------------------------------------------------------
#define HELEN_H8_REPORT(name)

template <int N, typename F>
__device__ __forceinline__ void half8_for(F&& f) {   // f(integral_constant<0>) ... f(integral_constant<N-1>)
    if constexpr (N > 0) {
        half8_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int kHalf8AF4 = 2 * 4 * 64;   // abuf [buffer][group of four k-octets][lane]
constexpr int kHalf8HF4 = 2 * 256;      // hbuf [buffer][unit quad][row of the half]
constexpr int kHalf8PF4 = 2 * 8 * 32;   // head partials [parity][k-slice][window group x class]

template <bool DEC>
__global__ __launch_bounds__(256, 1) void gru_half8_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                           int slot0_fwd, int slot0_bwd, int T,
                                                           const f32x4* __restrict__ Whp,
                                                           const float* __restrict__ bhn,
                                                           f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                           long y_tile_stride, const f32x4* __restrict__ Whd,
                                                           f32x4* __restrict__ plogit, long pl_tile_stride) {
    __shared__ f32x4 smem[kHalf8AF4 + kHalf8HF4 + (DEC ? kHalf8PF4 : 0)];
    f32x4* const abuf = smem;
    f32x4* const hbuf = smem + kHalf8AF4;
    f32x4* const part = smem + kHalf8AF4 + kHalf8HF4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int wg = b >> 3, cq = b & 7;
    const int tile = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int u = 32 * w + 4 * cq + j;            // this lane's hidden unit (rows 4 wg + r of the half)

    // W_hh: Wr[gate][tt] = column (gate, u), k(t) of t = 2 tt + (lane >> 5), gathered from the 16x16x4 packing
    // (pack_w_hh: float4 ((dir*4 + (v>>1))*48 + (2 gate + (v&1))*8 + m)*64 + (j16 + 16 q), component e)
    float Wr[3][64];
    {
        const int v8 = u >> 4, j16 = u & 15, odd = lane >> 5;
        const float* wp = (const float*)(Whp + (size_t)((dir * 4 + (v8 >> 1)) * 48) * 64);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int tt = 0; tt < 64; ++tt) {
                const int m = tt >> 3, e = (tt >> 1) & 3, q = 2 * (tt & 1) + odd;
                Wr[g][tt] = wp[(((2 * g + (v8 & 1)) * 8 + m) * 64 + j16 + 16 * q) * 4 + e];
            }
    }
    const f32x4 bnv = splat4(bhn[dir * kH + u]);
    // decoder heads: blocks bH = (slice sl = bH >> 3, window group wgH = (bH >> 2) & 1, class quad pH = bH & 3)
    const int wgH = (b >> 2) & 1, slH = b >> 3, pH = b & 3;
    float HB[16];
    if (DEC) {
        const float* hw = (const float*)(Whd + (size_t)(dir * 8 + 2 * w + slH) * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) HB[e * 4 + q] = hw[(4 * pH + j + 16 * q) * 4 + e];
    }

    constexpr long kPosBytes = 2 * kNTile * 64 * 16;
    const char* gi_next = (const char*)(gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) + 32 * half) +
                          (size_t)slot0 * kPosBytes;                        // (uniform; the lane's part is gi16)
    const unsigned gi16 = (unsigned)((u >> 4) * 64 + (u & 15) + 16 * wg) * 16u;
    char* y_next = (char*)(y + (size_t)tile * y_tile_stride + (size_t)dir * (kHidDirStride / 4) + 8 * half);
    char* pl_next = (char*)(plogit + (size_t)tile * pl_tile_stride + (size_t)dir * 64 + 32 * half);
    f32x4* const hid_s = hid + ((size_t)tile * 2 + dir) * (kHidDirStride / 4) + 8 * half;
    // thread -> (unit quad, row of the half) of the tile layout: float4 (tid >> 3) * 16 + (tid & 7) of the half's 8 rows
    const unsigned tile16 = (unsigned)((tid >> 3) * 16 + (tid & 7)) * 16u;

    f32x4 G[2][3];
    auto load_gi = [&](int p) __attribute__((always_inline)) {
        const unsigned l16 = in_block(gi16);
#pragma unroll
        for (int g = 0; g < 3; ++g) G[p][g] = *(const f32x4*)(gi_next + (l16 + (unsigned)g * 8192u));
        gi_next += kPosBytes;
    };
    // where this lane's four new values (rows 4 wg + r, unit u) go: abuf float index of r = 0 (r adds 4), hbuf likewise
    int aoff, hoff;
    {
        const int m = u >> 4, e = u & 3, q = (u >> 2) & 3;
        aoff = (((m >> 1) * 64 + 4 * (8 * wg + 4 * (e & 1) + q)) * 4) + 2 * (m & 1) + (e >> 1);
        hoff = ((u >> 2) * 8 + 4 * wg) * 4 + (u & 3);
    }
    {   // initial state: the tile layout into hbuf[0], scattered into abuf[0]
        const f32x4 h0 = *(const f32x4*)((const char*)hid_s + tile16);
        hbuf[tid] = h0;
        const int uq = tid >> 3, rho = tid & 7, m = uq >> 2, q = uq & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            ((float*)abuf)[((m >> 1) * 64 + 4 * (8 * (rho >> 2) + 4 * (c & 1) + q) + (rho & 3)) * 4 + 2 * (m & 1) + (c >> 1)] = h0[c];
    }
    load_gi(0);
    __syncthreads();
    float hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[r] = ((const float*)hbuf)[hoff + 4 * r];
    f32x4 A4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) A4[g4] = abuf[g4 * 64 + lane];

    auto sum_partials = [&](int pb) __attribute__((always_inline)) {   // lanes 0..31: (window group, class)
        const f32x4* ps = part + pb * 256 + (lane & 31);
        return (((ps[0] + ps[32]) + (ps[64] + ps[96])) + (ps[128] + ps[160])) + (ps[192] + ps[224]);
    };
    // the half's 8 rows of the FRAG layout: lane16 = class + 16 (2 half + window group)
    const unsigned plo = (unsigned)((lane & 15) + 16 * ((lane >> 4) & 1)) * 16u;

    // this wave's two k-slices of the head product: the (e, q) order of the other kernels' four 16x16x4 MFMAs
    auto head_product = [&](const f32x4 hd) __attribute__((always_inline)) {
        f32x4 hp = splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 0], hp, 2, 0, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 1], hp, 2, 1, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 2], hp, 2, 2, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 3], hp, 2, 3, 0);
        }
        return hp;
    };
    auto step = [&](auto CUR, int s) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value;
        const bool has_prev = s > 0, has_prev2 = s > 1, has_next = s + 1 < T;
        f32x4 acc[3], yv = splat4(0.f), hd = splat4(0.f), hp = splat4(0.f);
        if (has_next) load_gi(cur ^ 1);
        if (!DEC && has_prev) yv = hbuf[cur * 256 + tid];
        // (DEC: the head product of h(s-1) rides in the stream, one MFMA per k of its sixteen; at s = 0 it multiplies the
        // initial state and nobody takes the result)
        if (DEC) hd = hbuf[cur * 256 + (4 * (2 * w + slH) + pH) * 8 + 4 * wgH + j];
        half8_for<128>([&](auto TT) __attribute__((always_inline)) {
            constexpr int t = decltype(TT)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4 c = t ? acc[g] : g < 2 ? splat4(0.f) : bnv;
                acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[t >> 5][(t >> 3) & 3], Wr[g][t >> 1], c, 3, t & 7, 1 + (t & 1));
            }
            if constexpr (DEC && t >= 16 && t < 32)
                hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[(t - 16) >> 2], HB[t - 16], hp, 2, (t - 16) & 3, 0);

        });
        // (no branch inside the MFMA stream: hipcc sinks the part of a chain that sits in front of one into the block
        // behind it -- that chain's MFMAs then run back to back, two wait states each)
        __builtin_amdgcn_sched_barrier(0);
        if (!DEC && has_prev) {
            *(f32x4*)(y_next + in_block(tile16)) = yv;
            y_next += kYStride * 4;
        }
        // DEC: the partials of slot s-2 were written in the gates of step s-1 and published by the barrier since
        if (DEC && has_prev2) {
            if (w == ((s - 2) & 3) && lane < 32) *(f32x4*)(pl_next + in_block(plo)) = sum_partials(s & 1);
            pl_next += 128 * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 hn = gru_cell4(acc[0], acc[1], acc[2], G[cur][0], G[cur][1], G[cur][2], hprev);
        float* aw = (float*)(abuf + (cur ^ 1) * 256);
        float* hw = (float*)(hbuf + (cur ^ 1) * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[r] = hn[r];
            aw[aoff + 4 * r] = hn[r];
            hw[hoff + 4 * r] = hn[r];
        }
        if (DEC && has_prev) part[(((s - 1) & 1) * 8 + 2 * w + slH) * 32 + wgH * 16 + 4 * pH + j] = hp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) A4[g4] = abuf[(cur ^ 1) * 256 + g4 * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int s = 0;
    for (; s + 1 < T; s += 2) {
        step(I0{}, s);
        step(I1{}, s + 1);
    }
    if (s < T) step(I0{}, s);
    HELEN_H8_REPORT("half8")
    const int last = T & 1;   // buffer of h(T-1)
    if (DEC) {
        if (T >= 2) {
            if (w == ((T - 2) & 3) && lane < 32) *(f32x4*)(pl_next + plo) = sum_partials((T - 2) & 1);
            pl_next += 128 * 16;
        }
        const f32x4 hp = head_product(hbuf[last * 256 + (4 * (2 * w + slH) + pH) * 8 + 4 * wgH + j]);
        part[(((T - 1) & 1) * 8 + 2 * w + slH) * 32 + wgH * 16 + 4 * pH + j] = hp;
        __syncthreads();
        if (w == ((T - 1) & 3) && lane < 32) *(f32x4*)(pl_next + plo) = sum_partials((T - 1) & 1);
    } else {
        *(f32x4*)(y_next + tile16) = hbuf[last * 256 + tid];
    }
    *(f32x4*)((char*)hid_s + tile16) = hbuf[last * 256 + tid];
}

}  // namespace helen

namespace helen {

// ------------------------------------------------------------------------------------------------
// gru_quarter4_kernel: the same idea one step further, for calls of at most an eighth of the CUs in tiles (32 tiles =
// 512 windows on 256 CUs): FOUR windows of a tile per workgroup, 8 x tiles workgroups.  All 16 blocks of an MFMA belong
// to the one window group, so an instruction covers 64 columns and A is broadcast from one block to all sixteen
// (CBSZ = 4: a register holds 16 k for the 4 windows, 8 registers per step).  A wave still owns 32 hidden units of all
// three gates = 96 columns = one and a half instructions per k:
//     MFMA 1: blocks 0-7 the r columns, blocks 8-15 the z columns of its units (one W register per k),
//     MFMA 2: the n columns in both halves (BLGP: two k per W register; the second copy is idle work: 2 MFMAs per k
//             where 1.5 would do, 256 per step = 2,222 cycles against gru_half8_kernel's 3,333).
// Lane l < 32 then holds r and n of (4 windows, unit u), lane l + 32 holds z and n of the same unit: one exchange of
// the r | z accumulator between the two halves (v_permlane32_swap_b32), both halves do the same gate math, the lower one
// writes.  Chains, gate cell and head sums as everywhere: the same bits.
// Decoder heads: blocks (k-slice, class quad), four slices per instruction: waves 0 and 1 take slices 0-3 and 4-7.
// grid (4 x tiles, 2 directions), 256 threads.
// ------------------------------------------------------------------------------------------------
constexpr int kQuarter4AF4 = 2 * 2 * 64;   // abuf [buffer][group of four k-sixteens][lane]
constexpr int kQuarter4HF4 = 2 * 128;      // hbuf [buffer][unit quad][row of the quarter]
constexpr int kQuarter4PF4 = 2 * 8 * 16;   // head partials [parity][k-slice][class]

template <bool DEC>
__global__ __launch_bounds__(256, 1) void gru_quarter4_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                              int slot0_fwd, int slot0_bwd, int T,
                                                              const f32x4* __restrict__ Whp,
                                                              const float* __restrict__ bhn,
                                                              f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                              long y_tile_stride, const f32x4* __restrict__ Whd,
                                                              f32x4* __restrict__ plogit, long pl_tile_stride) {
    __shared__ f32x4 smem[kQuarter4AF4 + kQuarter4HF4 + (DEC ? kQuarter4PF4 : 0)];
    f32x4* const abuf = smem;
    f32x4* const hbuf = smem + kQuarter4AF4;
    f32x4* const part = smem + kQuarter4AF4 + kQuarter4HF4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int upper = b >> 3, cq = b & 7;         // upper half of the wave: the z columns in MFMA 1
    const int tile = blockIdx.x >> 2, qt = blockIdx.x & 3;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int u = 32 * w + 4 * cq + j;            // this lane's hidden unit (rows r of the quarter)

    // W_hh from the 16x16x4 packing (see gru_half8_kernel): Wrz[t] = column (upper ? z : r, u), k(t);
    // Wn[tt] = column (n, u), k(t) of t = 2 tt + upper
    float Wrz[128], Wn[64];
    {
        const int v8 = u >> 4, j16 = u & 15;
        const float* wp = (const float*)(Whp + (size_t)((dir * 4 + (v8 >> 1)) * 48) * 64);
#pragma unroll
        for (int t = 0; t < 128; ++t) {
            const int m = t >> 4, e = (t >> 2) & 3, q = t & 3;
            Wrz[t] = wp[(((2 * upper + (v8 & 1)) * 8 + m) * 64 + j16 + 16 * q) * 4 + e];
        }
#pragma unroll
        for (int tt = 0; tt < 64; ++tt) {
            const int m = tt >> 3, e = (tt >> 1) & 3, q = 2 * (tt & 1) + upper;
            Wn[tt] = wp[(((4 + (v8 & 1)) * 8 + m) * 64 + j16 + 16 * q) * 4 + e];
        }
    }
    const f32x4 bnv = splat4(bhn[dir * kH + u]);
    // decoder heads (waves 0 and 1): blocks bH = (slice sl = bH >> 2, class quad pH = bH & 3)
    const int slH = b >> 2, pH = b & 3;
    float HB[16];
    if (DEC) {   // (waves 2 and 3 repeat the products of waves 0 and 1 and drop them: no branch in the MFMA stream)
        const float* hw = (const float*)(Whd + (size_t)(dir * 8 + 4 * (w & 1) + slH) * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) HB[e * 4 + q] = hw[(4 * pH + j + 16 * q) * 4 + e];
    }

    constexpr long kPosBytes = 2 * kNTile * 64 * 16;
    const char* gi_next = (const char*)(gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) + 16 * qt) +
                          (size_t)slot0 * kPosBytes;
    const unsigned gi16 = (unsigned)((u >> 4) * 64 + (u & 15)) * 16u;
    char* y_next = (char*)(y + (size_t)tile * y_tile_stride + (size_t)dir * (kHidDirStride / 4) + 4 * qt);
    char* pl_next = (char*)(plogit + (size_t)tile * pl_tile_stride + (size_t)dir * 64 + 16 * qt);
    f32x4* const hid_s = hid + ((size_t)tile * 2 + dir) * (kHidDirStride / 4) + 4 * qt;
    // thread < 128 -> (unit quad, row of the quarter) of the tile layout
    const unsigned tile16 = (unsigned)(((tid & 127) >> 2) * 16 + (tid & 3)) * 16u;
    const bool mover = tid < 128;

    f32x4 G[2][3];
    auto load_gi = [&](int p) __attribute__((always_inline)) {
        const unsigned l16 = in_block(gi16);
#pragma unroll
        for (int g = 0; g < 3; ++g) G[p][g] = *(const f32x4*)(gi_next + (l16 + (unsigned)g * 8192u));
        gi_next += kPosBytes;
    };
    int aoff, hoff;
    {
        const int m = u >> 4, e = u & 3, q = (u >> 2) & 3;
        aoff = ((m >> 2) * 64 + 4 * (4 * e + q)) * 4 + (m & 3);
        hoff = (u >> 2) * 16 + (u & 3);
    }
    if (mover) {   // initial state: the tile layout into hbuf[0], scattered into abuf[0]
        const f32x4 h0 = *(const f32x4*)((const char*)hid_s + tile16);
        hbuf[tid] = h0;
        const int uq = tid >> 2, rho = tid & 3, m = uq >> 2, q = uq & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) ((float*)abuf)[((m >> 2) * 64 + 4 * (4 * c + q) + rho) * 4 + (m & 3)] = h0[c];
    }
    load_gi(0);
    __syncthreads();
    float hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[r] = ((const float*)hbuf)[hoff + 4 * r];
    f32x4 A4[2];
    A4[0] = abuf[lane];
    A4[1] = abuf[64 + lane];

    auto sum_partials = [&](int pb) __attribute__((always_inline)) {   // lanes 0..15: class
        const f32x4* ps = part + pb * 128 + (lane & 15);
        return (((ps[0] + ps[16]) + (ps[32] + ps[48])) + (ps[64] + ps[80])) + (ps[96] + ps[112]);
    };
    const unsigned plo = (unsigned)(lane & 15) * 16u;
    auto head_product = [&](const f32x4 hd) __attribute__((always_inline)) {
        f32x4 hp = splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 0], hp, 2, 0, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 1], hp, 2, 1, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 2], hp, 2, 2, 0);
            hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[e], HB[e * 4 + 3], hp, 2, 3, 0);
        }
        return hp;
    };
    auto step = [&](auto CUR, int s) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value;
        const bool has_prev = s > 0, has_prev2 = s > 1, has_next = s + 1 < T;
        f32x4 arz = splat4(0.f), an = bnv, yv = splat4(0.f), hd = splat4(0.f), hp = splat4(0.f);
        if (has_next) load_gi(cur ^ 1);
        if (!DEC && has_prev && mover) yv = hbuf[cur * 128 + tid];
        if (DEC) hd = hbuf[cur * 128 + (4 * (4 * (w & 1) + slH) + pH) * 4 + j];
        half8_for<128>([&](auto TT) __attribute__((always_inline)) {
            constexpr int t = decltype(TT)::value;
            __builtin_amdgcn_sched_barrier(0);
            arz = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[t >> 6][(t >> 4) & 3], Wrz[t], arz, 4, t & 15, 0);
            an = __builtin_amdgcn_mfma_f32_4x4x1f32(A4[t >> 6][(t >> 4) & 3], Wn[t >> 1], an, 4, t & 15, 1 + (t & 1));
            if constexpr (DEC && t >= 16 && t < 32)
                hp = __builtin_amdgcn_mfma_f32_4x4x1f32(hd[(t - 16) >> 2], HB[t - 16], hp, 2, (t - 16) & 3, 0);
        });
        __builtin_amdgcn_sched_barrier(0);   // (no branch inside the MFMA stream: see gru_half8_kernel)
        if (!DEC && has_prev) {
            if (mover) *(f32x4*)(y_next + in_block(tile16)) = yv;
            y_next += kYStride * 4;
        }
        if (DEC && has_prev2) {
            if (w == ((s - 2) & 3) && lane < 16) *(f32x4*)(pl_next + in_block(plo)) = sum_partials(s & 1);
            pl_next += 128 * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
        // r | z: the lower half holds r and receives z, the upper half holds z and receives r
        // v_permlane32_swap_b32 with the accumulator as both operands: result 0 = the lower half's values in both halves
        // (r), result 1 = the upper half's (z).  (__builtin_bit_cast applied directly to `arz[r]` in this unrolled loop came out
        // of hipcc 7.2 reading element 0 for every r -- scripts/dev/ab_part_tiles.py showed rows 1-3 wrong --, hence the scalar.)
        f32x4 ar, az;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float f = arz[r];      // (a scalar first: see above)
            const unsigned x = __builtin_bit_cast(unsigned, f);
            const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            ar[r] = __builtin_bit_cast(float, (unsigned)sw[0]);
            az[r] = __builtin_bit_cast(float, (unsigned)sw[1]);
        }
        const f32x4 hn = gru_cell4(ar, az, an, G[cur][0], G[cur][1], G[cur][2], hprev);
        float* aw = (float*)(abuf + (cur ^ 1) * 128);
        float* hw = (float*)(hbuf + (cur ^ 1) * 128);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[r] = hn[r];
            if (!upper) {
                aw[aoff + 4 * r] = hn[r];
                hw[hoff + 4 * r] = hn[r];
            }
        }
        if (DEC && has_prev && w < 2) part[(((s - 1) & 1) * 8 + 4 * w + slH) * 16 + 4 * pH + j] = hp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        A4[0] = abuf[(cur ^ 1) * 128 + lane];
        A4[1] = abuf[(cur ^ 1) * 128 + 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int s = 0;
    for (; s + 1 < T; s += 2) {
        step(I0{}, s);
        step(I1{}, s + 1);
    }
    if (s < T) step(I0{}, s);
    HELEN_H8_REPORT("quarter4")
    const int last = T & 1;   // buffer of h(T-1)
    if (DEC) {
        if (T >= 2) {
            if (w == ((T - 2) & 3) && lane < 16) *(f32x4*)(pl_next + plo) = sum_partials((T - 2) & 1);
            pl_next += 128 * 16;
        }
        if (w < 2) {
            const f32x4 hp = head_product(hbuf[last * 128 + (4 * (4 * w + slH) + pH) * 4 + j]);   // (w < 2: w & 1 = w)
            part[(((T - 1) & 1) * 8 + 4 * w + slH) * 16 + 4 * pH + j] = hp;
        }
        __syncthreads();
        if (w == ((T - 1) & 3) && lane < 16) *(f32x4*)(pl_next + plo) = sum_partials((T - 1) & 1);
    } else if (mover) {
        *(f32x4*)(y_next + tile16) = hbuf[last * 128 + tid];
    }
    if (mover) *(f32x4*)((char*)hid_s + tile16) = hbuf[last * 128 + tid];
}

}  // namespace helen
