// kernels_x3.h -- bf16 matrix-core kernels on split operands: fp32x3 recurrence, weight-stationary projections (three terms or one)
#pragma once
#include "kernels_common.h"
#include "kernels_gru.h"

namespace helen {

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
//   is unit 4M + q); an fp32 copy feeds the carried state and the decoder's head partials.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// fptrunc <2 x float> -> <2 x bfloat> selects v_cvt_pk_bf16_f32 (RNE) on gfx950; unlike an inline-asm cvt
// the compiler pads the VALU-write -> MFMA-read hazard itself (the asm form produced NaNs)
__device__ __forceinline__ unsigned short bf16_bits(float f) {
    const bf16x2_t p = __builtin_convertvector((f32x2){f, 0.f}, bf16x2_t);
    return (unsigned short)(__builtin_bit_cast(unsigned, p) & 0xffffu);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}

// The decoder's head slice runs on the bf16 pipe (A/B record: profiles/r05_fp32x3_levers.txt).  The three planes of h(s-1) a wave has in registers for the recurrence are the
// A operand, the head weights of its K32 group are split in three bf16 terms once per kernel, and the six leading products
// are shared by the two waves of a group: three bf16 MFMAs per wave and step instead of four v_mfma_f32_16x16x4_f32, which
// hold the SIMD's VALU for 32 cycles each -- the same fp32-class product as the recurrence's own.  Decoder launch 0.475 ->
// 0.455 ms (4,096 windows).  Also measured there and NOT taken: storing the new h into the planes as 32-bit words of two
// neighbouring units (one DPP swap per plane, half the LDS stores, none of the 2-byte stores' 15.6 M bank-conflict
// cycles per launch): 0.452 / 0.474 ms against 0.452 / 0.474 -- the plane stores are not on the step's critical path.

__global__ __launch_bounds__(512) void gru_x3_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                     int slot0_fwd, int slot0_bwd, int T,
                                                     const bf16x8* __restrict__ W3,
                                                     const float* __restrict__ bhn,
                                                     f32x4* __restrict__ hid, f32x4* __restrict__ yplanes,
                                                     long yp_tile_stride, const f32x4* __restrict__ Whd,
                                                     f32x4* __restrict__ plogit, long pl_tile_stride) {
    // Output.  Encoder launch (`yplanes`): the three bf16 planes yplanes[tile][slot][dir][plane][256 units]
    // for gemm_dec_x3_kernel.  Decoder launch (`plogit`): no layer output at all -- the heads are linear in
    // [h_fwd | h_bwd], so each direction contributes its half of the 16 logits: wave v owns the k-slice
    // 16v..16v+15 of h (one fp32 MFMA A fragment in the LDS copy of h); at step s+1 it multiplies the slice
    // of h(s) by its slice of the head weights (4 fp32 MFMAs) and parks the 16x16 partial in LDS; after that
    // step's barrier one wave (s mod 8) adds the eight partials in wave order and stores 1 KiB
    // plogit[tile][slot = s][dir][64 lanes] (FRAG layout) instead of 8 KiB of y2 (heads_kernel<true>).
    // (Two window tiles per workgroup sharing W_hh and the barrier were measured: no faster, 0.459 vs
    // 0.449 ms -- a step is 2 x 1200 cycles of MFMA issue plus 2 x 940 cycles of gate/split VALU work per
    // SIMD, which do not overlap, not barrier latency.)
    // LDS: fp32 h [2][512 f4] | bf16 planes [2 buffers][3 terms][256 units of 16 B] |
    // gi slots [8 waves][3][64 f4] | head partials [2][8 waves][64 f4]
    __shared__ f32x4 smem[2 * 512 + 2 * 3 * 256 + 8 * 192 + 2 * 8 * 64];   // 80 KiB
    f32x4* const hbuf = smem;
    f32x4* const planes = smem + 1024;
    f32x4* const part = smem + 1024 + 1536 + 8 * 192;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: hidden units 16v..16v+15
    f32x4* const gbuf = smem + 1024 + 1536 + v * 192;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int u = 16 * v + j;                                  // this lane's hidden unit
    const bool dec = plogit != nullptr;

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
    // decoder: the head weights of K32 group Mv = v & 3 as a B operand in three bf16 terms: k = dir*128 + 32 Mv + 8q + e,
    // class j.  Waves v and v + 4 share the group: v < 4 takes the three small products, v >= 4 the three large ones.
    const int Mv = v & 3;
    bf16x8 Bh3[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) Bh3[t] = __builtin_bit_cast(bf16x8, uint4{0u, 0u, 0u, 0u});
    if (dec) {
        const f32x4* ws = Whd + (size_t)(dir * 8 + 2 * Mv + (q >> 1)) * 64 + (2 * (q & 1)) * 16 + j;
        const f32x4 w0 = ws[0], w1 = ws[16];
        unsigned short tb[3][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? w0[e & 3] : w1[e & 3];
            tb[0][e] = bf16_bits(x);
            const float r1 = x - bf16_to_f32(tb[0][e]);
            tb[1][e] = bf16_bits(r1);
            tb[2][e] = bf16_bits(r1 - bf16_to_f32(tb[1][e]));
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            Bh3[t] = __builtin_bit_cast(bf16x8, uint4{tb[t][0] | (unsigned)tb[t][1] << 16, tb[t][2] | (unsigned)tb[t][3] << 16,
                                                       tb[t][4] | (unsigned)tb[t][5] << 16, tb[t][6] | (unsigned)tb[t][7] << 16});
    }

    const f32x4* gi_p = gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) + v * 64 + lane;
    constexpr long kPosStride = 2 * kNTile * 64;
    auto dma_gi = [&](int slot) {
        const f32x4* p = gi_p + (size_t)slot * kPosStride;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1)))*)(p + (g * 8) * 64),
                (void __attribute__((address_space(3)))*)(gbuf + g * 64), 16, 0, 0);
    };
    // this lane's 4 values: rows 4q + r of unit u.  fp32 h: float index ((u>>2)*16 + 4q + r)*4 + (u&3);
    // planes: bf16 index ((u>>3)*16 + 4q + r)*8 + (u&7) inside a 256-unit plane
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);
    auto store_h = [&](int buf, int r, float h) {
        ((float*)(hbuf + buf * 512))[hoff + 4 * r] = h;
        unsigned short* pl = (unsigned short*)(planes + buf * 768);
        const unsigned short t1 = bf16_bits(h);
        const float r1 = h - bf16_to_f32(t1);
        const unsigned short t2 = bf16_bits(r1);
        const float r2 = r1 - bf16_to_f32(t2);
        const unsigned short t3 = bf16_bits(r2);
        pl[0 * 2048 + poff + 8 * r] = t1;
        pl[1 * 2048 + poff + 8 * r] = t2;
        pl[2 * 2048 + poff + 8 * r] = t3;
    };
    auto store_h4 = [&](int buf, const f32x4 h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) store_h(buf, r, h[r]);
    };
    // three of the six leading products of h . W_head^T over this wave's K32 group, from the planes `at` of h
    auto head_partial = [&](const bf16x8 (&at)[3], int pb) {
        f32x4 pl = splat4(0.f);
        if (v < 4) {             // smallest first: h1 w3, h3 w1, h2 w2
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[0], Bh3[2], pl, 0, 0, 0);
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[2], Bh3[0], pl, 0, 0, 0);
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[1], Bh3[1], pl, 0, 0, 0);
        } else {                 // h1 w2, h2 w1, h1 w1
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[0], Bh3[1], pl, 0, 0, 0);
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[1], Bh3[0], pl, 0, 0, 0);
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[0], Bh3[0], pl, 0, 0, 0);
        }
        (part + (pb * 8 + v) * 64)[lane] = pl;
    };
    auto head_store = [&](int slot) {           // one wave adds the eight slices in wave order
        if (v != (slot & 7)) return;
        const f32x4* pp = part + (slot & 1) * 8 * 64 + lane;
        f32x4 sum = pp[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += pp[k * 64];
        (plogit + (size_t)tile * pl_tile_stride + ((size_t)slot * 2 + dir) * 64)[lane] = sum;
    };

    f32x4* hid_p = hid + ((size_t)tile * 2 + dir) * (kHidDirStride / 4);
    hbuf[tid] = hid_p[tid];
    dma_gi(slot0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[r] = ((const float*)hbuf)[hoff + 4 * r];
    store_h4(0, f32x4{hprev[0], hprev[1], hprev[2], hprev[3]});     // planes of h0 (fp32 copy rewritten in place)
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        const bf16x8* pa = (const bf16x8*)(planes + cur * 768) + lane;
        f32x4 acc[3];
        acc[0] = splat4(0.f);
        acc[1] = splat4(0.f);
        acc[2] = splat4(bn);
#pragma unroll
        for (int M = 0; M < 4; ++M) {
            const bf16x8 a1 = pa[0 * 256 + M * 64], a2 = pa[1 * 256 + M * 64], a3 = pa[2 * 256 + M * 64];
            const bf16x8 at[3] = {a1, a2, a3};
            if (dec && s > 0 && M == Mv) head_partial(at, (s - 1) & 1);   // the planes of h(s-1) are this step's A operand
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0};   // six leading products, smallest first;
            constexpr int TB[6] = {2, 0, 1, 1, 0, 0};   // product index outermost: 3 accumulators rotate
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[TA[k]], W[g][M][TB[k]], acc[g], 0, 0, 0);
        }
        // VMEM queue, oldest first: the 3 gi DMAs of this step (issued in the previous one), then that
        // step's output stores -- encoder: at least one per wave; decoder: one, by wave (s-2) mod 8 only
        if (!dec || (s >= 2 && v == ((s - 2) & 7)))
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 G[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) G[g] = gbuf[g * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 1 < T) dma_gi(slot0 + s + 1);
        {   // four cells as two packed pairs (gru_cell4)
            const f32x4 hn4 = gru_cell4(acc[0], acc[1], acc[2], G[0], G[1], G[2], hprev);
#pragma unroll
            for (int r = 0; r < 4; ++r) hprev[r] = hn4[r];
            store_h4(cur ^ 1, hn4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (dec) {
            if (s > 0) head_store(s - 1);
        } else {   // 768 units of 16 B per (tile, slot, dir)
            f32x4* po = yplanes + (size_t)tile * yp_tile_stride + ((size_t)s * 2 + dir) * 768;
            const f32x4* ps = planes + (cur ^ 1) * 768;
            po[tid] = ps[tid];
            if (tid < 256) po[512 + tid] = ps[512 + tid];
        }
    }
    if (dec) {   // the last step's logits
        const bf16x8* pl = (const bf16x8*)(planes + (T & 1) * 768) + lane + Mv * 64;
        const bf16x8 at[3] = {pl[0], pl[256], pl[512]};
        head_partial(at, (T - 1) & 1);
        __syncthreads();
        head_store(T - 1);
    }
    hid_p[tid] = (hbuf + (T & 1) * 512)[tid];
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
//   NP = 3 planes / weight terms (fp32x3: six products); NP = 1 is the plain bf16 projection (one plane, one
//   product; 0.27 ms per launch as <1, 6>) that HELEN_PRECISION_BF16 used before its projections were fused
//   into the recurrence (kernels_fused_bf16.h).  PB = positions per stage (NP * PB * 8 rows of 1 KiB).
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
        // the last octet of a row holds k = 88, 89 only: never read past the row (and the buffer)
        const uint64_t w = o * 8 + 8 <= kF ? *(const u64_a2*)p : (uint64_t) * (const u16_a2*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_bits((float)((w >> (8 * e)) & 0xffu));   // exact: integers <= 255
    }
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16);
    u.y = v[2] | ((unsigned)v[3] << 16);
    u.z = v[4] | ((unsigned)v[5] << 16);
    u.w = v[6] | ((unsigned)v[7] << 16);
    xb[((size_t)tile * npos + pos) * 192 + o * 16 + row] = __builtin_bit_cast(f32x4, u);
}

//   TERMS = 3 (fp32x3); TERMS = 1 is the plain bf16 projection (W_ih rounded to bf16 = the first term).
template <int TERMS>
__global__ __launch_bounds__(512) void gemm_enc_x3_kernel(const f32x4* __restrict__ xb, long xb_tile_stride,
                                                          const f32x4* __restrict__ W3e,
                                                          const float* __restrict__ bias,
                                                          f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int npos, int ntiles, int parts, int run) {
    constexpr int PB = 8, ROWS = PB * 3;    // rows of 1 KiB per stage: (position, group)
    __shared__ f32x4 smem[2 * ROWS * 64];   // 48 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of parts * 3 * roundup8(ntiles) ids.  Workgroups go round-robin over the 8 XCDs (id % 8): the three
    // column sets of a tile get consecutive local slots of ONE XCD, so its A stream is fetched from HBM
    // once and served from that XCD's L2 to the other two.  A call that does not fill the chip with (tile, column set)
    // workgroups cuts the positions into `parts` runs of `run` (whole stages): every output element is still ONE
    // accumulator chain over the same MFMAs in the same order, so the partition never changes a bit.
    const int per_part = 3 * ((ntiles + 7) / 8 * 8);
    const int part = (int)blockIdx.x / per_part;
    const int id = (int)blockIdx.x % per_part;
    const int local = id >> 3;
    const int set = local % 3;
    const int tile = (local / 3) * 8 + (id & 7);
    if (tile >= ntiles) return;
    const int g_lo = part * run / PB;                       // stages of this run: positions part*run .. min(+run, npos) - 1
    const int p_hi = min(npos, (part + 1) * run);
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
            const int pc = min(PB * g + r / 3, p_hi - 1);
            const f32x4* src = xp + (size_t)pc * 192 + (r % 3) * 64;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(dst + r * 64), 16, 0, 0);
        }
    };
    const int ng = (p_hi + PB - 1) / PB;
    stage(g_lo, g_lo & 1);
    for (int g = g_lo; g < ng; ++g) {
        // VMEM queue, oldest first: 3 DMA rows of group g, then 16 output stores of group g-1
        if (g == g_lo)
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
            const int pos = min(PB * g + p, p_hi - 1);
            const int slot = dir ? (npos - 1 - pos) : pos;
            f32x4* o = gi + (size_t)tile * gi_tile_stride + ((size_t)slot * 2 + dir) * (kNTile * 64) + nt * 64 + lane;
            o[0] = acc[p][0];
            o[64] = acc[p][1];
        }
    }
}

}  // namespace helen
