// kernels_gru_pair.h -- fp32 GRU recurrence, two window tiles per workgroup, one 8-wave workgroup per CU
#pragma once
#include <type_traits>

#include "kernels_gru.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Same arithmetic as gru_kernel (same MFMA order per accumulator, same gate cell: results are bit-identical),
// different schedule.  Measured on gfx950 (scripts/ubench/f32_mfma_valu_overlap.hip):
//   - v_mfma_f32_16x16x4_f32 and VALU instructions never overlap on a SIMD: not from the same wave (+13 cycles
//     for the first VALU after an MFMA, +4.6 per further one), not from a co-resident wave (an MFMA-streaming
//     wave starves its partner's VALU completely: issue is arbitrated by age);
//   - v_exp_f32 / v_rcp_f32 serialise at 8.6 cycles per wave-instruction per SIMD however many waves issue them;
//     plain and packed fp32 VALU instructions at 4.9 (one wave) to 3.8 (four waves);
//   - LDS reads, SALU and (L2-warm) vector-memory instructions between MFMAs are nearly free.
// So a recurrence step of one tile costs a SIMD its 192 MFMAs (6144 cycles) PLUS ~620 cycles of gate math whatever
// the occupancy: ~0.865 of the MFMA peak is the ceiling of this recurrence in fp32.  gru_kernel (two independent
// 4-wave workgroups per CU) loses another 8 % to what its pairing cannot hide: waves of the younger workgroup
// wait behind the older one's MFMA stream even for their address arithmetic, barrier skew between eight waves
// that fight for four pipes, and a tail in which the younger workgroup runs alone.
//
// Here ONE workgroup of 8 waves per CU walks TWO tiles of 16 windows, interleaved in software:
//     M(0,s) | G(0,s)   M(1,s) | G(1,s)   M(0,s+1) | ...      M = MFMA phase, G = gate math, | = the barrier
//   - wave v owns hidden units 16v..16v+15 = three 16-column tiles (r, z, n): its W_hh slice is 96 registers,
//     everything fits 256 registers per lane and two waves share a SIMD;
//   - one barrier per half-step, between M and G.  It publishes the OTHER tile's h (written in the previous
//     half-step's G and first needed by the next half-step's M), so nobody ever waits for an LDS round trip;
//     and it lines the two waves of a SIMD up so that the older one (which gets its MFMAs out first) idles only
//     while its partner's MFMAs keep the pipe busy, and both do their gate math together;
//   - gi fragments go from global memory straight into registers (each wave reads only its own: there is
//     nothing to share through LDS), loaded at the start of the other tile's M phase, a full half-step ahead;
//   - the layer output (encoder) / the head partial products (decoder) of step s-1 leave during M(x,s); the
//     decoder's eight partial tiles of step s-2 are added up by one of the OLDER waves of a SIMD right before the
//     barrier, where it would otherwise only wait for its partner.
// grid (ceil(tiles / 2), 2 directions).  An odd tile count makes the last workgroup do its one tile twice
// (identical values are written twice).
// ------------------------------------------------------------------------------------------------
constexpr int kPairHF4 = 2 * 2 * 512;        // h[tile x][buffer][512]
constexpr int kPairPF4 = 2 * 2 * 8 * 64;     // head partials [tile x][parity][wave][64]

// The body is a device function over (tile pair, direction) and a caller-provided LDS block of kPairHF4 (+ kPairPF4)
// float4; gru_pair_kernel below is one call per workgroup.
template <bool DEC>
__device__ __forceinline__ void gru_pair_body(f32x4* __restrict__ smem, const int pair_index, const int dir,
                                              const f32x4* __restrict__ gi, long gi_tile_stride,
                                              int slot0_fwd, int slot0_bwd, int T,
                                              const f32x4* __restrict__ Whp,
                                              const float* __restrict__ bhn,
                                              f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                              long y_tile_stride, const f32x4* __restrict__ Whd,
                                              f32x4* __restrict__ plogit, long pl_tile_stride,
                                              int ntiles) {
    f32x4* const hbuf = smem;
    f32x4* const part = smem + kPairHF4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int tile_of[2] = {min(2 * pair_index, ntiles - 1), min(2 * pair_index + 1, ntiles - 1)};

    // W_hh slice: W[gate][m] holds k = 16m + 4q + e of column (gate, unit 16v + j): pack_w_hh keeps it at
    // wave w = v >> 1, n = 2 gate + (v & 1)
    f32x4 W[3][8];
    {
        const f32x4* wp = Whp + (size_t)((dir * 4 + (v >> 1)) * 48) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int m = 0; m < 8; ++m) W[g][m] = wp[((2 * g + (v & 1)) * 8 + m) * 64];
    }
    f32x4 Bh = splat4(0.f);   // DEC: head weights of k = dir*128 + 16v + 4q + e, class j
    if (DEC) Bh = Whd[(dir * 8 + v) * 64 + lane];
    const f32x4 bnv = splat4(bhn[dir * kH + 16 * v + j]);   // b_hn: initial value of the n-gate accumulator

    // Uniform running byte pointers per tile -- next gi slot to fetch, next layer-output / partial-logit slot to
    // store -- advanced by SALU adds; everything per-lane is a constant 32-bit byte offset.
    constexpr long kPosBytes = 2 * kNTile * 64 * 16;  // one slot of gi (both directions)
    const char* gi_next[2];
    char* y_next[2];
    char* pl_next[2];
    char* hid_s[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        gi_next[x] = (const char*)(gi + (size_t)tile_of[x] * gi_tile_stride + (size_t)dir * (kNTile * 64) + v * 64) +
                     (size_t)slot0 * kPosBytes;
        y_next[x] = (char*)(y + (size_t)tile_of[x] * y_tile_stride + (size_t)dir * (kHidDirStride / 4));
        pl_next[x] = (char*)(plogit + (size_t)tile_of[x] * pl_tile_stride + (size_t)dir * 64);
        hid_s[x] = (char*)(hid + ((size_t)tile_of[x] * 2 + dir) * (kHidDirStride / 4));
    }
    const unsigned lane16 = (unsigned)lane * 16u, tid16 = (unsigned)tid * 16u;
    // h in LDS is the KB16 operand layout with the 16 window entries of row (m, q) XOR-ed with q: a lane's four
    // gate results (windows 4q..4q+3 of one unit) then go to 64 different banks per ds_write_b32 instead of 16
    // (unswizzled, the four lanes j, j+4, j+8, j+12 of a 16-lane group hit the same bank: 4-way conflicts on every
    // write, 9.9 M conflict cycles per launch).  Readers fetch whole 16-byte entries: they only pick another one.
    const int slane = (lane & 48) | (j ^ q), stid = (tid & ~15) | ((tid & 15) ^ ((tid >> 4) & 3));

    // this wave's gi fragments (gate g = column tile 8g + v) of each tile's next step, in registers
    f32x4 G[2][3];
    auto load_gi = [&](int x) __attribute__((always_inline)) {
        const unsigned l16 = in_block(lane16);
#pragma unroll
        for (int g = 0; g < 3; ++g) G[x][g] = *(const f32x4*)(gi_next[x] + (l16 + (unsigned)g * 8192u));
        gi_next[x] += kPosBytes;
    };
    // initial hidden state of both tiles -> buffer 0; tile 0's first gi (tile 1's is fetched during M(0,0))
#pragma unroll
    for (int x = 0; x < 2; ++x) hbuf[x * 1024 + stid] = *(const f32x4*)(hid_s[x] + tid16);
    load_gi(0);
    __syncthreads();

    float hprev[2][4];
    const int u = 16 * v + j;
    int hoff[4];   // float offset of (window 4q + r, unit u) in an h buffer
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hoff[r] = ((u >> 2) * kTile + 4 * q + (r ^ (j >> 2))) * 4 + (u & 3);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[x][r] = ((const float*)(hbuf + x * 1024))[hoff[r]];
    f32x4 a_pref = hbuf[slane];   // group 0 of h_0(-1): the A fragment the first MFMA phase starts with

    // the eight k-slices' partial logits of (tile x, parity pb) added in the order gru_kernel adds them
    auto sum_partials = [&](int x, int pb) __attribute__((always_inline)) {
        const f32x4* ps = part + (x * 2 + pb) * 512 + lane;
        return (((ps[0] + ps[64]) + (ps[128] + ps[192])) + (ps[256] + ps[320])) + (ps[384] + ps[448]);
    };

    // One half-step: MFMA phase and gate math of tile X at step s (CUR = s & 1 at compile time so that every LDS
    // address is a lane offset + immediate).  `so` = newest step of the OTHER tile o.  STEADY = the caller
    // guarantees 2 <= s and s + 2 <= T - 1... i.e. every "is there a previous / next step" question is a
    // compile-time yes (a uniform bool that lives across the phase ends up in a VGPR, and every VALU instruction
    // between MFMAs costs matrix-pipe time).
    auto half_step = [&](auto X, auto CUR, auto STEADY, int s) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value;
        constexpr bool steady = decltype(STEADY)::value;
        constexpr int ocur = x ? (cur ^ 1) : cur;                // buffer of h_o(so): (so + 1) & 1
        const int so = x ? s : s - 1;
        const bool has_next_o = steady || so + 1 < T;            // tile o still has a step so+1 to feed
        const bool has_prev = steady || s > 0;                   // h_x(s-1) is a step's output (not the initial state)
        const bool has_prev2 = steady || s > 1;
        const f32x4* hx = hbuf + (x * 2 + cur) * 512;            // h_x(s-1): A operand of this phase
        const f32x4* hb = hx + slane;
        f32x4 acc[3], a[3], yv = splat4(0.f), hd = splat4(0.f), hp = splat4(0.f);
        a[0] = a_pref;
        a[1] = hb[1 * 64];
        if (has_next_o) load_gi(o);                              // tile o's registers were consumed in G(o, so)
        if (!DEC && has_prev) yv = hx[stid];                      // h_x(s-1) is the layer output of slot s-1
        if (DEC && has_prev) hd = hb[v * 64];                    // ... or feeds the heads: this wave's k-slice
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // A fragment m+2 goes into the register fragment m-1 used, half a group after its last MFMA (straight
            // after it, hipcc pads the MFMA-read / LDS-write hazard with s_nop 6)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g)   // the first MFMA of a chain takes its initial value as the C operand
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
                *(f32x4*)(y_next[x] + in_block(tid16)) = yv;
                y_next[x] += kYStride * 4;
            }
        }
        // DEC: tile x's partials of slot s-2 were written in G(x,s-1) and published by the barrier since.  One of
        // the OLDER waves of a SIMD (v < 4) adds them up here: it is through its MFMAs long before its partner and
        // would only wait at the barrier.
        if (DEC && has_prev2) {
            if (v == ((s - 2) & 3)) *(f32x4*)(pl_next[x] + in_block(lane16)) = sum_partials(x, s & 1);
            pl_next[x] += 128 * 16;
        }
        // every wave is through M(x,s); the h_o(so) written in the previous half-step's G becomes visible
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        a_pref = hbuf[(o * 2 + ocur) * 512 + slane];              // next phase: M(o, so+1) starts on h_o(so)
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 hn = gru_cell4(acc[0], acc[1], acc[2], G[x][0], G[x][1], G[x][2], hprev[x]);
        float* hw = (float*)(hbuf + (x * 2 + (cur ^ 1)) * 512);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[x][r] = hn[r];
            hw[hoff[r]] = hn[r];
        }
        if (DEC && has_prev) (part + ((x * 2 + ((s - 1) & 1)) * 8 + v) * 64)[lane] = hp;
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using No = std::false_type;
    using Yes = std::true_type;
    auto step = [&](auto STEADY, int s_) __attribute__((always_inline)) {   // both tiles; buffer parity = s & 1
        if (s_ & 1) {
            half_step(I0{}, I1{}, STEADY, s_);
            half_step(I1{}, I1{}, STEADY, s_);
        } else {
            half_step(I0{}, I0{}, STEADY, s_);
            half_step(I1{}, I0{}, STEADY, s_);
        }
    };
    int s = 0;
    for (; s < T && s < 2; ++s) step(No{}, s);            // the first two steps: no step s-1 / s-2 yet
    for (; s + 2 < T; s += 2) {                           // steady state: steps 2 .. T-2, two per trip
        half_step(I0{}, I0{}, Yes{}, s);
        half_step(I1{}, I0{}, Yes{}, s);
        half_step(I0{}, I1{}, Yes{}, s + 1);
        half_step(I1{}, I1{}, Yes{}, s + 1);
    }
    for (; s < T; ++s) step(No{}, s);                     // the last one or two steps: no step s+1 to feed
    __syncthreads();
    const int last = T & 1;   // buffer of h(T-1)
    if (DEC) {
        // H(x, s) turns h_x(s-1) into partials and adds up tile x's slot s-2: after the loop slot T-2 of both tiles
        // is still to be added up, and slot T-1 has no partials yet.
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (T >= 2) {
                if (v == ((T - 2) & 3)) *(f32x4*)(pl_next[x] + lane16) = sum_partials(x, (T - 2) & 1);
                pl_next[x] += 128 * 16;   // at slot T-1 now
            }
            const f32x4 hd = hbuf[(x * 2 + last) * 512 + v * 64 + slane];
            f32x4 hp = splat4(0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) hp = mfma4(hd[e], Bh[e], hp);
            (part + ((x * 2 + ((T - 1) & 1)) * 8 + v) * 64)[lane] = hp;
        }
        __syncthreads();
        if (v == ((T - 1) & 3)) {
#pragma unroll
            for (int x = 0; x < 2; ++x) *(f32x4*)(pl_next[x] + lane16) = sum_partials(x, (T - 1) & 1);
        }
    } else {   // the last layer outputs
#pragma unroll
        for (int x = 0; x < 2; ++x) *(f32x4*)(y_next[x] + tid16) = hbuf[(x * 2 + last) * 512 + stid];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) *(f32x4*)(hid_s[x] + tid16) = hbuf[(x * 2 + last) * 512 + stid];
}

template <bool DEC>
__global__ __launch_bounds__(512, 1) void gru_pair_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int slot0_fwd, int slot0_bwd, int T,
                                                          const f32x4* __restrict__ Whp,
                                                          const float* __restrict__ bhn,
                                                          f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                          long y_tile_stride, const f32x4* __restrict__ Whd,
                                                          f32x4* __restrict__ plogit, long pl_tile_stride,
                                                          int ntiles) {
    __shared__ f32x4 smem[kPairHF4 + (DEC ? kPairPF4 : 0)];   // 32 (+32) KiB
    gru_pair_body<DEC>(smem, (int)blockIdx.x, (int)blockIdx.y, gi, gi_tile_stride, slot0_fwd, slot0_bwd, T, Whp, bhn, hid,
                       y, y_tile_stride, Whd, plogit, pl_tile_stride, ntiles);
}

}  // namespace helen
