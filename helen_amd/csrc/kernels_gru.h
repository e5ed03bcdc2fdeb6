// kernels_gru.h -- fp32 GRU recurrence (gru_kernel) and the gate cell
#pragma once
#include "kernels_common.h"

namespace helen {

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
//   Encoder launch (DEC = false): each step's h is streamed out as y[tile][slot][dir] (KB16) for the
//   decoder projection, slot = step index (t for direction 0, T-1-t for direction 1).
//   Decoder launch (DEC = true): no layer output.  The heads are linear in [h_fwd | h_bwd], so each
//   direction's workgroup contributes its half of the 16 logits: wave w owns the k-slice 32w..32w+31
//   of h (two KB16 groups = two fp32 MFMA A fragments in the LDS copy of h); at step s+1 it multiplies
//   the slice of h(s) by its slice of the head weights (8 MFMAs beside the 192 recurrent ones), parks
//   the 16x16 partial in LDS, and after that step's barrier wave s mod 4 adds the four partials in
//   wave order and stores 1 KiB plogit[tile][slot = s][dir][64 lanes] (FRAG layout) instead of 8 KiB of
//   y2: the heads kernel shrinks from 0.12 to 0.035 ms per launch for +0.03 ms here.
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

// Gate math of four cells at once, as two packed pairs: measured alone on a gfx950 SIMD a v_pk_{add,mul,fma}_f32
// costs 5.2 cycles against 4.8 for ONE scalar add / mul / fma (v_exp_f32 / v_rcp_f32: 8.6 each, no packed form),
// and none of it overlaps with the fp32 MFMAs (scripts/ubench/f32_mfma_valu_overlap.hip).  Per component this is
// exactly:  r = sigmoid(ar + gr), z = sigmoid(az + gz), n = tanh(fma(r, an, gn)), h' = fma(z, h - n, n)
// with sigmoid(x) = rcp(1 + exp2(-log2e x)), tanh(x) = fma(-2, rcp(1 + exp2(2 log2e x)), 1).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp2_pair(f32x2 v) {
    return f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)};
}
__device__ __forceinline__ f32x2 rcp_pair(f32x2 v) {
    return f32x2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)};
}
// (pre-activations of r and z complete: the kernels whose accumulators start from the input part -- hipcc does not
// fold an `x + 0`, which differs from x for -0)
__device__ __forceinline__ f32x2 gru_cell2(f32x2 sr, f32x2 sz, f32x2 an, f32x2 gn, f32x2 hp) {
    const f32x2 one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
    const f32x2 rg = rcp_pair(one + exp2_pair(sr * -1.4426950408889634f));
    const f32x2 zg = rcp_pair(one + exp2_pair(sz * -1.4426950408889634f));
    const f32x2 pre = __builtin_elementwise_fma(rg, an, gn);
    const f32x2 ng = __builtin_elementwise_fma(m2, rcp_pair(one + exp2_pair(pre * 2.8853900817779268f)), one);
    return __builtin_elementwise_fma(zg, hp - ng, ng);  // (1-z)*n + z*h
}
__device__ __forceinline__ f32x2 gru_cell2(f32x2 ar, f32x2 az, f32x2 an, f32x2 gr, f32x2 gz, f32x2 gn, f32x2 hp) {
    return gru_cell2(ar + gr, az + gz, an, gn, hp);
}
// The same cell for PRESCALED pre-activations (bf16 mode, round 4): the rows of W_ih, W_hh and the biases of the r and z
// gates are multiplied by -log2(e) and those of the n gate by 2 log2(e) before they are rounded to bf16 (api.hip), so the
// accumulators arrive as the arguments of exp2 and three multiplications per cell disappear:
//     r = rcp(1 + exp2(sr')), z = rcp(1 + exp2(sz')), n = fma(-2, rcp(1 + exp2(fma(r, an', gn'))), 1), h' = fma(z, h - n, n)
__device__ __forceinline__ f32x2 gru_cell2_pre(f32x2 sr, f32x2 sz, f32x2 an, f32x2 gn, f32x2 hp) {
    const f32x2 one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
    const f32x2 rg = rcp_pair(one + exp2_pair(sr));
    const f32x2 zg = rcp_pair(one + exp2_pair(sz));
    const f32x2 ng = __builtin_elementwise_fma(m2, rcp_pair(one + exp2_pair(__builtin_elementwise_fma(rg, an, gn))), one);
    return __builtin_elementwise_fma(zg, hp - ng, ng);
}
__device__ __forceinline__ f32x4 gru_cell4_pre(f32x4 sr, f32x4 sz, f32x4 an, f32x4 gn, const float (&hp)[4]) {
    const f32x2 lo = gru_cell2_pre(sr.xy, sz.xy, an.xy, gn.xy, f32x2{hp[0], hp[1]});
    const f32x2 hi = gru_cell2_pre(sr.zw, sz.zw, an.zw, gn.zw, f32x2{hp[2], hp[3]});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 gru_cell4(f32x4 sr, f32x4 sz, f32x4 an, f32x4 gn, const float (&hp)[4]) {
    const f32x2 lo = gru_cell2(sr.xy, sz.xy, an.xy, gn.xy, f32x2{hp[0], hp[1]});
    const f32x2 hi = gru_cell2(sr.zw, sz.zw, an.zw, gn.zw, f32x2{hp[2], hp[3]});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 gru_cell4(f32x4 ar, f32x4 az, f32x4 an, f32x4 gr, f32x4 gz, f32x4 gn,
                                           const float (&hp)[4]) {
    const f32x2 lo = gru_cell2(ar.xy, az.xy, an.xy, gr.xy, gz.xy, gn.xy, f32x2{hp[0], hp[1]});
    const f32x2 hi = gru_cell2(ar.zw, az.zw, an.zw, gr.zw, gz.zw, gn.zw, f32x2{hp[2], hp[3]});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

constexpr int kGruLdsF4 = 2 * 512 + 4 * 384 + 4 * 512;  // h[2] | gi[4 waves] | W tile 5[4 waves]

template <bool DEC>
__global__ __launch_bounds__(256, 2) void gru_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                     int slot0_fwd, int slot0_bwd, int T,
                                                     const f32x4* __restrict__ Whp,
                                                     const float* __restrict__ bhn,
                                                     f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                     long y_tile_stride, const f32x4* __restrict__ Whd,
                                                     f32x4* __restrict__ plogit, long pl_tile_stride) {
    // DEC (decoder launch): no layer output; each direction emits its half of the 16 logits as fp32
    // partials (see gru_x3_kernel): wave w owns the k-slice 32w..32w+31 of h = two KB16 groups.
    __shared__ f32x4 smem[kGruLdsF4 + (DEC ? 2 * 4 * 64 : 0)];  // 72 (+8) KiB: two workgroups fit in 160 KiB
    f32x4* const hbuf = smem;
    f32x4* const part = smem + kGruLdsF4;
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
    f32x4 Bh[2] = {splat4(0.f), splat4(0.f)};   // DEC: head weights of k = dir*128 + 32w + 16g + 4q + e, class j
    if (DEC) {
        Bh[0] = Whd[(dir * 8 + 2 * w) * 64 + lane];
        Bh[1] = Whd[(dir * 8 + 2 * w + 1) * 64 + lane];
    }
    auto head_partial = [&](int hb_, int pb) {   // this wave's two k-groups of h in hbuf[hb_]
        const f32x4 a0 = (hbuf + hb_ * 512)[(2 * w) * 64 + lane], a1 = (hbuf + hb_ * 512)[(2 * w + 1) * 64 + lane];
        f32x4 p0 = splat4(0.f), p1 = splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p0 = mfma4(a0[e], Bh[0][e], p0);
            p1 = mfma4(a1[e], Bh[1][e], p1);
        }
        (part + (pb * 4 + w) * 64)[lane] = p0 + p1;
    };
    auto head_store = [&](int slot) {             // one wave adds the four slices in wave order
        if (w != (slot & 3)) return;
        const f32x4* pp = part + (slot & 1) * 4 * 64 + lane;
        const f32x4 sum = ((pp[0] + pp[64]) + pp[128]) + pp[192];
        (plogit + (size_t)tile * pl_tile_stride + ((size_t)slot * 2 + dir) * 64)[lane] = sum;
    };
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
        if (DEC && s > 0) head_partial(cur, (s - 1) & 1);   // h(s-1) sits in hbuf[cur] since the last barrier
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
        // gate pre-activations of this step (DMA'd during the previous step), then refill the slot
        // VMEM queue of this wave, oldest first: 6 gi DMAs (issued last step), 2 y stores (issued
        // after last step's barrier).  vmcnt(2) = the DMAs have landed; the stores may still fly.
        // (hipcc does not order these LDS reads behind the DMA by itself.)
        // DEC: the only store of the previous step is the partial-logit tile, by wave (s-2) mod 4.
        if (!DEC)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (s >= 2 && w == ((s - 2) & 3))
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 G[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) G[n] = gbuf[n * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + 1 < T) dma_gi(slot0 + s + 1);

        float* hw = (float*)(hbuf + (cur ^ 1) * 512);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 hn = gru_cell4(acc[hh], acc[2 + hh], acc[4 + hh], G[hh], G[2 + hh], G[4 + hh], hprev[hh]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hprev[hh][r] = hn[r];
                hw[hoff[hh] + 4 * r] = hn[r];
            }
        }
        // raw barrier: only LDS traffic has to be drained, the gi DMA stays in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (DEC) {
            if (s > 0) head_store(s - 1);
        } else {
            // stream h(t) out as the layer output
            f32x4* yo = y_p + (size_t)s * (kYStride / 4);  // slot s: t for dir 0, T-1-t for dir 1
            const f32x4* hn4 = hbuf + (cur ^ 1) * 512;
            yo[tid] = hn4[tid];
            yo[tid + 256] = hn4[tid + 256];
        }
    }
    if (DEC) {   // the last step's logits
        head_partial(T & 1, (T - 1) & 1);
        __syncthreads();
        head_store(T - 1);
    }
    const f32x4* hl = hbuf + (T & 1) * 512;
    hid_p[tid] = hl[tid];
    hid_p[tid + 256] = hl[tid + 256];
}

}  // namespace helen
