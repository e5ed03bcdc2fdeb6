// kernels_fused_bf16.h -- HELEN_PRECISION_BF16: input projection fused into the recurrence
#pragma once
#include "kernels_common.h"
#include "kernels_gru.h"
#include "kernels_x3.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// The head slice of a wave (16 rows x its 16 hidden units -> 16 logits) on the bf16 pipe, from round 4 on.  Until then
// it was four fp32 MFMAs (v_mfma_f32_16x16x4_f32) per wave and step: 32 cycles of matrix pipe each beside 16 for a bf16
// MFMA of sixteen times the work, and measured 493 cycles a region for the two waves of a SIMD -- a fifth of the decoder
// (profiles/r04_bf16_own.txt).  Now h and the head weights are split in two bf16 terms each (x = hi + lo, 16 significant
// bits) and the three products that matter ride in TWO K32 MFMAs:
//     A  = [ h_hi(4 units) | h_lo(4 units) ]  per lane (row l & 15, its unit quad 4q .. 4q+3), built in registers from the
//          fp32 slice the wave reads anyway;
//     B1 = [ W_hi | W_hi ],  B2 = [ W_lo | 0 ]   built once per kernel from the fp32 head fragment of the lane (class l & 15);
//     pl = A . B1 + A . B2 = sum_k h_hi W_hi + h_lo W_hi + h_hi W_lo          (fp32 accumulate, products exact)
// The term dropped is h_lo W_lo (2^-16 of a product); against the fp32 head a logit moves by <= 2e-4 at |logit| <= 12,
// a hundredth of what bf16 gate operands already move it.  Every bf16 kernel uses these three functions: same bits.
// ------------------------------------------------------------------------------------------------
struct HeadW {
    bf16x8 b1, b2;
};
__device__ __forceinline__ bf16x8 bf16x8_of(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(bf16x8, uint4{a, b, c, d});
}
__device__ __forceinline__ HeadW head_split_w(f32x4 w) {
    unsigned hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = bf16_bits(w[e]);
        lo[e] = bf16_bits(w[e] - bf16_to_f32((unsigned short)hi[e]));
    }
    HeadW r;
    r.b1 = bf16x8_of(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, hi[0] | hi[1] << 16, hi[2] | hi[3] << 16);
    r.b2 = bf16x8_of(lo[0] | lo[1] << 16, lo[2] | lo[3] << 16, 0u, 0u);
    return r;
}
__device__ __forceinline__ bf16x8 head_split_h(f32x4 h) {
    unsigned hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = bf16_bits(h[e]);
        lo[e] = bf16_bits(h[e] - bf16_to_f32((unsigned short)hi[e]));
    }
    return bf16x8_of(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, lo[0] | lo[1] << 16, lo[2] | lo[3] << 16);
}
__device__ __forceinline__ f32x4 head_mfma(bf16x8 a, const HeadW& w) {
    f32x4 pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w.b1, splat4(0.f), 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w.b2, pl, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// One GRU layer direction, projection AND recurrence, for bf16 gate matmuls (fp32 accumulate, fp32
// state, fp32 gates).  With bf16 operands a direction's W_ih and W_hh both fit the register file
// (8 waves, wave v owns hidden units 16v..16v+15 = one 16-column tile per gate: W_hh 3 x 4 K32-groups
// = 48 registers, W_ih 3 x MI groups = 36 (encoder, K = 96) or 96 (decoder, K = 256)), so the gate
// pre-activations never exist in memory: per step the kernel reads the layer INPUT (3 KiB of packed
// pileup counts, or 8 KiB of the encoder's bf16 output plane) instead of 24 KiB of fp32 gi, and the
// separate projection kernels and their 24 MB-per-window gi round trip disappear.
//   Per step:  A  gh = h . W_hh^T on the bf16 h plane in LDS, added onto the input part computed one
//                 step earlier (n gate kept apart: n = tanh(gi_n + r * gh_n))
//              B  gates (fp32), new h -> LDS as fp32 (state / layer output) and as a bf16 plane
//              C  this wave's input row of step s+1 has landed
//              D  one barrier
//              E  store this step's output, DMA the input of step s+2 (3-deep ring)
//              F  input part of step s+1: x . W_ih^T + bias -- independent of h, so it sits between
//                 the barrier and the next step's recurrent MFMAs
//   Input rows (1 KiB = one K32 group of 16 windows, the MFMA A fragment):
//     encoder (MI = 3): xb[tile][pos][3][64 units], pos = pos0 + s (dir 0) or pos0 + T-1-s (dir 1)
//     decoder (MI = 8): yplane[tile][slot][d][4][64 units]; time t = s (dir 0) or T-1-s (dir 1);
//                       k < 128 is the forward encoder output of time t (slot t), k >= 128 the backward
//                       one (stored time-reversed: slot T-1-t)
//   Output: the encoder writes one bf16 plane yplane_out[tile][slot = s][dir][256 units] for the decoder.
//   The decoder (DEC) writes no layer output at all: the heads are linear in [h_fwd | h_bwd], so each
//   direction contributes its half of the 16 logits.  Wave v owns exactly the k-slice 16v..16v+15 of
//   h (one fp32 MFMA A fragment in the LDS copy of h): at step s+1 it multiplies the slice of h(s) by its
//   slice of the head weights (two bf16 MFMAs on two-term splits, see above; issued with the recurrent ones) and parks the 16x16 partial
//   in LDS; after that step's barrier one wave (s mod 8) adds the eight partials in wave order and stores 1 KiB
//   plogit[tile][slot = s][dir][64 lanes] (FRAG layout) instead of 8 KiB of y2.  The heads kernel then
//   only adds two partial tiles and the bias.
//   Weights come from the three-term packings of kernels_x3.h; term 0 is RNE(w).
// ------------------------------------------------------------------------------------------------
template <int MI, bool DEC>
__global__ __launch_bounds__(512) void gru_fused_bf16_kernel(
    const f32x4* __restrict__ in, long in_tile_stride, int pos0, int T, const bf16x8* __restrict__ Wi3,
    const bf16x8* __restrict__ Wh3, const float* __restrict__ bias, const float* __restrict__ bhn,
    f32x4* __restrict__ hid, f32x4* __restrict__ yplane_out, long yp_tile_stride,
    const f32x4* __restrict__ Whd, f32x4* __restrict__ plogit, long pl_tile_stride) {
    // LDS: fp32 h [2][512 f4] | bf16 h plane [2][256 units] | input ring [3][MI * 64 units] |
    // (DEC) head partials [2][8 waves][64 f4]
    __shared__ f32x4 smem[2 * 512 + 2 * 256 + 3 * MI * 64 + (DEC ? 2 * 8 * 64 : 0)];
    f32x4* const hbuf = smem;
    f32x4* const hplane = smem + 1024;
    f32x4* const inbuf = smem + 1024 + 512;
    f32x4* const part = inbuf + 3 * MI * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int dir = blockIdx.y;
    const int u = 16 * v + j;

    bf16x8 Wh[3][4], Wi[3][MI];
    {
        const bf16x8* wh = Wh3 + (size_t)((dir * 8 + v) * 36) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int M = 0; M < 4; ++M) Wh[g][M] = wh[((g * 4 + M) * 3) * 64];
            const bf16x8* wi = Wi3 + (size_t)((dir * kNTile + g * 8 + v) * MI) * 3 * 64 + lane;
#pragma unroll
            for (int M = 0; M < MI; ++M) Wi[g][M] = wi[(M * 3) * 64];
        }
    }
    HeadW Bh = head_split_w(splat4(0.f));   // DEC: head weights for k = dir*128 + 16v + 4q + e, class j, as two bf16 terms
    if (DEC) Bh = head_split_w(Whd[(dir * 8 + v) * 64 + lane]);
    float bi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bi[g] = bias[dir * kG + g * kH + u];
    const float bn = bhn[dir * kH + u];

    const f32x4* in_p = in + (size_t)tile * in_tile_stride + lane;
    auto dma_in = [&](int s, int b) {      // wave v < MI brings row v of step s into ring buffer b
        if (v >= MI) return;
        const f32x4* src;
        if (DEC) {
            const int t = dir ? (T - 1 - s) : s;
            const int part = v >> 2;
            const int slot = part ? (T - 1 - t) : t;
            src = in_p + ((size_t)slot * 2 + part) * 256 + (v & 3) * 64;
        } else {
            const int pos = pos0 + (dir ? (T - 1 - s) : s);
            src = in_p + (size_t)pos * (MI * 64) + v * 64;
        }
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(inbuf + (b * MI + v) * 64), 16, 0,
                                         0);
    };
    auto input_part = [&](int b, f32x4* acc) {   // x . W_ih^T + bias for this wave's three column tiles
        const bf16x8* L = (const bf16x8*)(inbuf + b * (MI * 64)) + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = splat4(bi[g]);
#pragma unroll
        for (int M = 0; M < MI; ++M) {
            const bf16x8 a = L[M * 64];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wi[g][M], acc[g], 0, 0, 0);
        }
    };
    // this lane's 4 values: rows 4q + r of unit u (see gru_x3_kernel)
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);
    auto store_h = [&](int buf, int r, float h) {
        ((float*)(hbuf + buf * 512))[hoff + 4 * r] = h;
        ((unsigned short*)(hplane + buf * 256))[poff + 8 * r] = bf16_bits(h);
    };

    auto head_partial = [&](int hb, int pb) {   // h in hbuf[hb]: wave v's k-slice is one fp32 A fragment
        const f32x4 a = (hbuf + hb * 512)[v * 64 + lane];
        (part + (pb * 8 + v) * 64)[lane] = head_mfma(head_split_h(a), Bh);
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
    dma_in(0, 0);
    if (T > 1) dma_in(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[r] = ((const float*)hbuf)[hoff + 4 * r];
#pragma unroll
    for (int r = 0; r < 4; ++r) store_h(0, r, hprev[r]);
    f32x4 gin[3];
    input_part(0, gin);
    __syncthreads();

    int b_next = 1, b_dma = 2;       // ring slots of step s+1 and s+2
    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        // A: recurrent part (+ DEC: this wave's k-slice of the head product for the PREVIOUS step's h,
        // which sits in hbuf[cur] since the last barrier: its LDS read rides with the plane reads)
        f32x4 ar = gin[0], az = gin[1], ahn = splat4(bn);
        if (DEC && s > 0) head_partial(cur, (s - 1) & 1);
        {
            const bf16x8* pa = (const bf16x8*)(hplane + cur * 256) + lane;
#pragma unroll
            for (int M = 0; M < 4; ++M) {
                const bf16x8 a = pa[M * 64];
                ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[0][M], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[1][M], az, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[2][M], ahn, 0, 0, 0);
            }
        }
        // B: gates
        {   // four cells as two packed pairs (gru_cell4): the gate math is what bounds this kernel
            const f32x4 hn4 = gru_cell4_pre(ar, az, ahn, gin[2], hprev);     // (weights and biases prescaled: kernels_gru.h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hprev[r] = hn4[r];
                store_h(cur ^ 1, r, hn4[r]);
            }
        }
        // C: everything this wave has in flight is older than a step except its input row of step s+1
        // (issued after the previous step's store): wait for all of it
        if (v < MI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // D
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // E: output first (DEC: the partial logits of step s-1, whose eight slices were parked before this
        // barrier), then the input of step s+2
        if (DEC) {
            if (s > 0) head_store(s - 1);
        } else if (tid < 256) {
            (yplane_out + (size_t)tile * yp_tile_stride + ((size_t)s * 2 + dir) * 256)[tid] =
                (hplane + (cur ^ 1) * 256)[tid];
        }
        if (s + 2 < T) dma_in(s + 2, b_dma);
        // F
        if (s + 1 < T) input_part(b_next, gin);
        b_next = b_next == 2 ? 0 : b_next + 1;
        b_dma = b_dma == 2 ? 0 : b_dma + 1;
    }
    if (DEC) {   // the last step's logits
        head_partial(T & 1, (T - 1) & 1);
        __syncthreads();
        head_store(T - 1);
    }
    hid_p[tid] = (hbuf + (T & 1) * 512)[tid];
}

// float32 x [B, T, F] (operator-level boundary) -> bf16 A fragments xb[tile][pos][3][64 units], RNE.
__global__ __launch_bounds__(256) void pack_x_bf16_kernel(const float* __restrict__ x, int n_windows, int T,
                                                          f32x4* __restrict__ xb, long xb_tile_stride) {
    const int tile = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= T * 192) return;
    const int row = g & 15;
    const int o = (g >> 4) % 12;
    const int pos = g / 192;
    const int window = tile * kTile + row;
    unsigned short b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = 0;
    if (window < n_windows) {
        const float* p = x + ((size_t)window * T + pos) * kF + o * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (o * 8 + e < kF) b[e] = bf16_bits(p[e]);
    }
    uint4 w;
    w.x = b[0] | ((unsigned)b[1] << 16);
    w.y = b[2] | ((unsigned)b[3] << 16);
    w.z = b[4] | ((unsigned)b[5] << 16);
    w.w = b[6] | ((unsigned)b[7] << 16);
    xb[(size_t)tile * xb_tile_stride + (size_t)pos * 192 + o * 16 + row] = __builtin_bit_cast(f32x4, w);
}

}  // namespace helen
