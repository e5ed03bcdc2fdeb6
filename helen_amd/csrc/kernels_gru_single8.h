// kernels_gru_single8.h -- fp32 GRU recurrence, ONE window tile per 8-wave workgroup, one workgroup per CU
#pragma once
#include "kernels_gru_pair.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// For calls that do not fill the chip with tile pairs (at most one (tile, direction) per CU: up to 128 tiles on 256
// CUs).  gru_kernel puts such a tile on four waves (one per SIMD, 32 hidden units each, the sixth W_hh tile parked in
// LDS, gi by LDS-DMA) and, alone on its CU, takes 0.36 ms per 100 steps: every LDS round trip, DMA issue and barrier
// of its single wave per SIMD is exposed.  Here the tile gets the pair kernel's eight waves (16 hidden units each:
// the whole W_hh slice in 96 registers, two waves per SIMD, gi straight into registers a step ahead, the swizzled h
// tile) and the simplest schedule there is:
//     M(s)  G(s) | M(s+1)  G(s+1) | ...         one barrier per step, behind the gates
// M(s) reads h(s-1) from one LDS buffer while G(s) writes h(s) into the other, so nothing separates them; the
// barrier publishes h(s).  The two waves of a SIMD share the matrix pipe during M (6144 cycles for both) and do
// their gate math one after the other's MFMAs; what stays exposed is one barrier and one LDS round trip per step.
// Same MFMA order per accumulator, same gate cell, same order of the head partial sums as gru_kernel /
// gru_pair_kernel: bit-identical (tests/test_gpu_scale.py).
// grid (tiles, 2 directions).
// ------------------------------------------------------------------------------------------------
constexpr int kSingle8HF4 = 2 * 512;          // h[buffer][512]
constexpr int kSingle8PF4 = 2 * 8 * 64;       // head partials [parity][wave][64]

template <bool DEC>
__global__ __launch_bounds__(512, 1) void gru_single8_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                             int slot0_fwd, int slot0_bwd, int T,
                                                             const f32x4* __restrict__ Whp,
                                                             const float* __restrict__ bhn,
                                                             f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                             long y_tile_stride, const f32x4* __restrict__ Whd,
                                                             f32x4* __restrict__ plogit, long pl_tile_stride) {
    __shared__ f32x4 smem[kSingle8HF4 + (DEC ? kSingle8PF4 : 0)];
    f32x4* const hbuf = smem;
    f32x4* const part = smem + kSingle8HF4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;

    f32x4 W[3][8];   // as in gru_pair_body: W[gate][m] = k 16m + 4q + e of column (gate, unit 16v + j)
    {
        const f32x4* wp = Whp + (size_t)((dir * 4 + (v >> 1)) * 48) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int m = 0; m < 8; ++m) W[g][m] = wp[((2 * g + (v & 1)) * 8 + m) * 64];
    }
    f32x4 Bh = splat4(0.f);
    if (DEC) Bh = Whd[(dir * 8 + v) * 64 + lane];
    const f32x4 bnv = splat4(bhn[dir * kH + 16 * v + j]);

    constexpr long kPosBytes = 2 * kNTile * 64 * 16;
    const char* gi_next = (const char*)(gi + (size_t)tile * gi_tile_stride + (size_t)dir * (kNTile * 64) + v * 64) +
                          (size_t)slot0 * kPosBytes;
    char* y_next = (char*)(y + (size_t)tile * y_tile_stride + (size_t)dir * (kHidDirStride / 4));
    char* pl_next = (char*)(plogit + (size_t)tile * pl_tile_stride + (size_t)dir * 64);
    char* const hid_s = (char*)(hid + ((size_t)tile * 2 + dir) * (kHidDirStride / 4));
    const unsigned lane16 = (unsigned)lane * 16u, tid16 = (unsigned)tid * 16u;
    const int slane = (lane & 48) | (j ^ q), stid = (tid & ~15) | ((tid & 15) ^ ((tid >> 4) & 3));   // swizzled h tile

    f32x4 G[2][3];   // gi fragments of step s (parity s & 1) and, in flight, of step s + 1
    auto load_gi = [&](int p) __attribute__((always_inline)) {
        const unsigned l16 = in_block(lane16);
#pragma unroll
        for (int g = 0; g < 3; ++g) G[p][g] = *(const f32x4*)(gi_next + (l16 + (unsigned)g * 8192u));
        gi_next += kPosBytes;
    };
    hbuf[stid] = *(const f32x4*)(hid_s + tid16);
    load_gi(0);
    __syncthreads();

    float hprev[4];
    const int u = 16 * v + j;
    int hoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hoff[r] = ((u >> 2) * kTile + 4 * q + (r ^ (j >> 2))) * 4 + (u & 3);
        hprev[r] = ((const float*)hbuf)[hoff[r]];
    }
    f32x4 a_pref = hbuf[slane];

    auto sum_partials = [&](int pb) __attribute__((always_inline)) {
        const f32x4* ps = part + pb * 512 + lane;
        return (((ps[0] + ps[64]) + (ps[128] + ps[192])) + (ps[256] + ps[320])) + (ps[384] + ps[448]);
    };

    auto step = [&](auto CUR, int s) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value;
        const bool has_prev = s > 0, has_prev2 = s > 1, has_next = s + 1 < T;
        const f32x4* hx = hbuf + cur * 512;              // h(s-1)
        const f32x4* hb = hx + slane;
        f32x4 acc[3], a[3], yv = splat4(0.f), hd = splat4(0.f), hp = splat4(0.f);
        a[0] = a_pref;
        a[1] = hb[1 * 64];
        if (has_next) load_gi(cur ^ 1);                  // gi(s+1): consumed by the gates of the next step
        if (!DEC && has_prev) yv = hx[stid];
        if (DEC && has_prev) hd = hb[v * 64];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = mfma4(a[m % 3][e], W[g][m][e], (m | e) ? acc[g] : g < 2 ? splat4(0.f) : bnv);
            __builtin_amdgcn_sched_barrier(0);
            if (m + 2 < 8) a[(m + 2) % 3] = hb[(m + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = mfma4(a[m % 3][e], W[g][m][e], acc[g]);
            __builtin_amdgcn_sched_barrier(0);
            if (m == 1 && DEC && has_prev) {
#pragma unroll
                for (int e = 0; e < 4; ++e) hp = mfma4(hd[e], Bh[e], hp);
            }
            if (m == 2 && !DEC && has_prev) {
                *(f32x4*)(y_next + in_block(tid16)) = yv;
                y_next += kYStride * 4;
            }
        }
        // DEC: the partials of slot s-2 were written in G(s-1) and published by the barrier since
        if (DEC && has_prev2) {
            if (v == ((s - 2) & 3)) *(f32x4*)(pl_next + in_block(lane16)) = sum_partials(s & 1);
            pl_next += 128 * 16;
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 hn = gru_cell4(acc[0], acc[1], acc[2], G[cur][0], G[cur][1], G[cur][2], hprev);
        float* hw = (float*)(hbuf + (cur ^ 1) * 512);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[r] = hn[r];
            hw[hoff[r]] = hn[r];
        }
        if (DEC && has_prev) (part + (((s - 1) & 1) * 8 + v) * 64)[lane] = hp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        a_pref = hbuf[(cur ^ 1) * 512 + slane];
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
    const int last = T & 1;   // buffer of h(T-1)
    if (DEC) {
        // after the loop slot T-2 is still to be added up, and slot T-1 has no partials yet
        if (T >= 2) {
            if (v == ((T - 2) & 3)) *(f32x4*)(pl_next + lane16) = sum_partials((T - 2) & 1);
            pl_next += 128 * 16;
        }
        const f32x4 hd = hbuf[last * 512 + v * 64 + slane];
        f32x4 hp = splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) hp = mfma4(hd[e], Bh[e], hp);
        (part + (((T - 1) & 1) * 8 + v) * 64)[lane] = hp;
        __syncthreads();
        if (v == ((T - 1) & 3)) *(f32x4*)(pl_next + lane16) = sum_partials((T - 1) & 1);
    } else {
        *(f32x4*)(y_next + tid16) = hbuf[last * 512 + stid];
    }
    *(f32x4*)(hid_s + tid16) = hbuf[last * 512 + stid];
}

}  // namespace helen
