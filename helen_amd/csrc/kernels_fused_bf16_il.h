// kernels_fused_bf16_il.h -- HELEN_PRECISION_BF16: fused projection + recurrence, two window tiles per workgroup,
// gate math INTERLEAVED with the other tile's MFMAs in one instruction stream
#pragma once
#include <type_traits>

#include "kernels_fused_bf16.h"

#define HELEN_BF16_IL_PARKED 0
#define HELEN_BF16_IL_ADEPTH(dec) ((dec) ? 5 : 7)
#define HELEN_BF16_IL_LEAD(dec) ((dec) ? 6 : 1)

#define HELEN_PIN(x) asm volatile("" : "+v"(x))

namespace helen {

// ------------------------------------------------------------------------------------------------
// Same arithmetic as gru_fused_bf16_kernel (same MFMA order per accumulator, the same IEEE operations per gate
// component, same order of the head partial sums: results are bit-identical).  With bf16 operands a tile-step is only
// 21 (encoder) / 36 (decoder) bf16 MFMAs per wave -- 700 / 1200 cycles of a SIMD's matrix pipe -- plus the same ~620
// cycles of gate math as in fp32, so what one tile per workgroup leaves exposed per step (a barrier, an LDS round trip
// for the new h, the tail of the MFMA pipe before the gates) costs as much as the work itself: gru_fused_bf16_kernel
// measures 2070 / 3140 cycles per tile-step.  Here ONE workgroup of 8 waves walks TWO tiles of 16 windows:
//     M(0,s) | G(0,s) M(1,s) | G(1,s) M(0,s+1) | ...        M = MFMA phase, G = gate math, | = the barrier
//   M(x,s): tile x's whole step s -- bias, input part x . W_ih^T from the LDS ring, recurrent part on the bf16 plane of
//           h_x(s-1) -- and this wave's k-slice of the head product of h_x(s-1) (decoder);
//   the barrier publishes the OTHER tile's h (written in the previous half-step's G) and the input rows an earlier
//           half-step's LDS-DMA brought in (the issuing wave waits for them with a COUNTED vmcnt at the barrier);
//   G(x,s): gates, new h -> LDS (fp32 + bf16 plane), head partials -> LDS.
// The weights (W_hh 48 + W_ih 36 / 96 registers) are shared by both tiles; each tile has its own h buffers, input ring
// and partial-logit slots.  grid (ceil(tiles / 2), 2 directions); an odd tile count makes the last workgroup do its one
// tile twice.
// What matters is the order of instructions BETWEEN two barriers.  Round 3's form of this kernel
// (gru_fused_bf16_pair_kernel, removed in round 5) ran G(x,s) and then M(o,.) -- and measured (scripts/ubench/bf16_mfma_valu_overlap.hip, profiles/ub_bf16_overlap.txt) a SIMD then pays the MFMAs
// plus the gate math in full: v_mfma_f32_16x16x32_bf16 issues every 17 cycles, but
//   - in the SAME wave's stream one transcendental (v_exp_f32 / v_rcp_f32) or two plain fp32 VALU instructions
//     behind each MFMA are free (17.0 -> 17.7 / 17.2 cycles per MFMA; a second transcendental costs 8.2, a third
//     plain instruction 4.7);
//   - a PACKED fp32 instruction behind an MFMA costs 16 cycles (17 -> 33 per MFMA): the packed gate cell of the
//     other kernels is the wrong form beside bf16 MFMAs;
//   - a VALU-only wave beside an MFMA-only wave on the same SIMD overlaps only partly (24 MFMAs + 48 v_exp per
//     wave: phases aligned 1588 cycles, offset 1466, one interleaved stream 1291).
// So here the region between two barriers is ONE stream: MFMA i of M(x,s) followed by slot i of G(o,.) -- a slot is
// one transcendental or two plain scalar instructions of the gate math of the other tile's newest step, four cells
// per lane staggered so that no slot waits for the one before it: 40 slots per region (24 transcendentals, 16 plain
// pairs; the gates' exp2 factors are folded into the weights, kernels_gru.h gru_cell2_pre).  The decoder's 38 MFMAs per
// region cover nearly all of them; the encoder's 21 cover half and the rest follows the last MFMA.
// ------------------------------------------------------------------------------------------------
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: a step's input part is computed in the step's OWN region.  Until then the input part x(s+1) . W_ih^T + b of a
// tile's NEXT step was computed in the other tile's MFMA phase and carried (three accumulators, plus the three being
// filled: 36 registers a lane) to the region that adds the recurrent part.  At 256 registers that forced the decoder to
// park K32 groups of W_ih in LDS -- three of eight: nine more fragment reads per wave and region (72 KiB per workgroup
// beside the 96 KiB of A fragments) -- and kept it slower than the pair kernel.  Now region (x, s) runs tile x's WHOLE
// step: bias -> input part -> recurrent part in one accumulator chain per gate (the same order of MFMAs per accumulator
// as every other bf16 kernel: bit-identical), the n gate's input part in its own chain.  Nothing is carried but the
// finished accumulators awaiting their gate math, all 36 weight fragments of a decoder wave are resident, and the row a
// region needs was DMA'd a whole step earlier (lookahead one step, ring of two).  Decoder launch of 8,192 windows:
// 0.485 ms against the pair kernel's 0.500 when this form came in; with the bf16 head and the prescaled gates 0.447 against
// 0.486 (profiles/r04_bf16_own.txt); encoder 0.317 against 0.321.
// ------------------------------------------------------------------------------------------------
template <int MI, bool DEC>
__global__ __launch_bounds__(512, 1) void gru_fused_bf16_il_kernel(
    const f32x4* __restrict__ in, long in_tile_stride, int pos0, int T, const bf16x8* __restrict__ Wi3,
    const bf16x8* __restrict__ Wh3, const float* __restrict__ bias, const float* __restrict__ bhn,
    f32x4* __restrict__ hid, f32x4* __restrict__ yplane_out, long yp_tile_stride,
    const f32x4* __restrict__ Whd, f32x4* __restrict__ plogit, long pl_tile_stride, int ntiles) {
    // LDS per tile: fp32 h [2][512 f4] | bf16 h plane [2][256] | input ring [RD][MI * 64] | (DEC) head partials [2][8][64]
    // (DEC) after both tiles: the parked K32 groups of W_ih of every wave [8][kParked][3][64] (none by default).
    constexpr int RD = 2;
    constexpr int kPlane = 2 * 512, kRing = kPlane + 2 * 256, kPart = kRing + RD * MI * 64, kPerTile = kPart + (DEC ? 2 * 8 * 64 : 0);
    auto hsel = [](int b) __attribute__((always_inline)) { return b * 512; };                 // fp32 h buffer b (f4 offset)
    auto psel = [=](int b) __attribute__((always_inline)) { return kPlane + b * 256; };       // bf16 plane b
    constexpr int kParked = DEC ? HELEN_BF16_IL_PARKED : 0, MR = MI - kParked;
    __shared__ f32x4 smem[2 * kPerTile + kParked * 8 * 3 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int u = 16 * v + j;
    const int tile_of[2] = {min(2 * (int)blockIdx.x, ntiles - 1), min(2 * (int)blockIdx.x + 1, ntiles - 1)};

    bf16x8 Wh[3][4], Wi[3][MR > 0 ? MR : 1];
    bf16x8* const wpark = (bf16x8*)(smem + 2 * kPerTile) + v * (kParked * 3 * 64) + lane;
    {
        const bf16x8* wh = Wh3 + (size_t)((dir * 8 + v) * 36) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int M = 0; M < 4; ++M) Wh[g][M] = wh[((g * 4 + M) * 3) * 64];
            const bf16x8* wi = Wi3 + (size_t)((dir * kNTile + g * 8 + v) * MI) * 3 * 64 + lane;
#pragma unroll
            for (int M = 0; M < MR; ++M) Wi[g][M] = wi[(M * 3) * 64];
#pragma unroll
            for (int M = MR; M < MI; ++M) wpark[((M - MR) * 3 + g) * 64] = wi[(M * 3) * 64];
        }
    }
    HeadW Bh = head_split_w(splat4(0.f));   // DEC: head weights for k = dir*128 + 16v + 4q + e, class j, as two bf16 terms
    if (DEC) Bh = head_split_w(Whd[(dir * 8 + v) * 64 + lane]);
    float bi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bi[g] = bias[dir * kG + g * kH + u];
    const float bn = bhn[dir * kH + u];

    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)smem;
    const char* in_next[2];
    long in_step;
    {
        long first;
        if (DEC) {
            const int p = v >> 2;
            const bool up = p == dir;
            first = ((long)(up ? 0 : T - 1) * 2 + p) * 256 + (v & 3) * 64;
            in_step = (up ? 1 : -1) * 512L * 16;
        } else {
            first = (long)(pos0 + (dir ? T - 1 : 0)) * (MI * 64) + (v < MI ? v : 0) * 64;
            in_step = (dir ? -1 : 1) * (long)(MI * 64) * 16;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) in_next[x] = (const char*)(in + (size_t)tile_of[x] * in_tile_stride + first);
    }
    unsigned ring_dma[2], ring_rd[2];
    auto dma_in = [&](int x) __attribute__((always_inline)) {
        if (v < MI) dma_row_to_lds(lds0 + (unsigned)((x * kPerTile + kRing) * 16) + ring_dma[x] + (unsigned)v * 1024u,
                                   in_next[x], in_block(lane16));
        in_next[x] += in_step;
        ring_dma[x] = ring_dma[x] == (RD - 1u) * MI * 1024u ? 0u : ring_dma[x] + MI * 1024u;
    };
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);

    f32x4* hid_p[2];
    char* y_next[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        hid_p[x] = hid + ((size_t)tile_of[x] * 2 + dir) * (kHidDirStride / 4);
        y_next[x] = DEC ? (char*)(plogit + (size_t)tile_of[x] * pl_tile_stride + (size_t)dir * 64)
                        : (char*)(yplane_out + (size_t)tile_of[x] * yp_tile_stride + (size_t)dir * 256);
    }
    auto store_logits = [&](int x, int pb, unsigned voff) __attribute__((always_inline)) {
        const float* pp = (const float*)(smem + x * kPerTile + kPart + pb * 8 * 64) + tid;
        float sum = pp[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += pp[k * 256];
        *(float*)(y_next[x] + voff) = sum;
    };

    // ---- prologue: initial h and the row of step 0 -- for both tiles
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        smem[x * kPerTile + tid] = hid_p[x][tid];
        ring_dma[x] = 0;
        ring_rd[x] = 0;
        dma_in(x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            ((unsigned short*)(smem + x * kPerTile + kPlane))[poff + 8 * r] =
                bf16_bits(((const float*)(smem + x * kPerTile))[hoff + 4 * r]);
    __syncthreads();
    // The encoder keeps the previous h of a lane's four cells in registers (it has them to spare) and writes the fp32
    // state to LDS only once, at the end; the decoder's head reads that state every step, and its registers are full.
    // (Encoder launch of 8,192 windows 0.315 -> 0.291 ms.  The decoder with registers for the gates and LDS for the head
    // spills: 0.443 -> 0.465-0.501 ms, profiles/r04_bf16_own.txt.)
    constexpr bool kRegH = !DEC;
    float hprev[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) hprev[x][r] = ((const float*)(smem + x * kPerTile))[hoff + 4 * r];

    // Pending gate math of each tile: the finished accumulators of its newest step.
    f32x4 Pr[2], Pz[2], Pn[2], Pg[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) Pr[x] = Pz[x] = Pn[x] = Pg[x] = splat4(0.f);

    constexpr int NIN = 3 * MI, NHEAD = DEC ? 2 : 0, NREC = 12;
    constexpr int NM = NIN + NHEAD + NREC;            // MFMAs of one M phase: the step's input part, head slice of h(s-1), recurrent part
    constexpr int NS = 40;                            // gate slots

    // One region between two barriers: the MFMA phase of tile X at step s, and -- if `gates` -- the gate math of tile
    // O = 1 - X at its newest step so (whose accumulators are in P*[O]), slot by slot behind the MFMAs.
    // CUR = s & 1; OW = the h buffer of tile O that its gates write ((so + 1) & 1).  STEADY: s >= 2 and s + 1 < T.
    auto region = [&](auto X, auto CUR, auto OW, auto STEADY, auto GATES, int s, int so) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value, ow = decltype(OW)::value;
        constexpr bool steady = decltype(STEADY)::value, gates = decltype(GATES)::value;
        const bool has_prev = steady || s > 0;
        const bool has_prev2 = steady || s > 1;
        const bool has_next = steady || s + 1 < T;
        f32x4* const base = smem + x * kPerTile;
        f32x4* const obase = smem + o * kPerTile;
        const f32x4* hx = base + hsel(cur);
        int issued = 0;
        if (has_next) {           // the row of step s+1 into the ring slot whose row (step s-1) was read a whole step ago
            dma_in(x);
            issued += v < MI;
        }
        f32x4 ar = splat4(bi[0]), az = splat4(bi[1]), gnx = splat4(bi[2]), ahn = splat4(bn), pl = splat4(0.f);
        // gate state of tile o: four cells (rows 4q + c of unit u)
        const f32x4 gr = Pr[o], gz = Pz[o], gnn = Pn[o], ggn = Pg[o];
        float e1[4], e2[4], rg[4], zg[4], t3[4], e3[4], u3[4], qq[4], ng[4], dd[4], hn[4], hp[4];
        const float* hpo = (const float*)(obase + hsel(ow ^ 1)) + hoff;   // h_o(so - 1): fp32 buffer so & 1
        // A fragments: the MI K32 groups of tile x's input row of step s, then the four of h_x(s-1).  AD in flight, fragment
        // f + AD fetched behind the last MFMA of group f.  Inline asm loads (an ordinary LDS load is free to sink below the
        // sched_barriers), waited for by position in the in-order LDS queue: hipcc's own LDS accesses in between only make
        // a wait stricter than it has to be.
        constexpr int NF = MI + 4, AD = HELEN_BF16_IL_ADEPTH(DEC) < NF ? HELEN_BF16_IL_ADEPTH(DEC) : NF;
        bf16x8 aq[AD];
        const unsigned pa_lds = lds0 + (unsigned)((x * kPerTile + psel(cur)) * 16) + lane16;
        const unsigned in_lds = lds0 + (unsigned)((x * kPerTile + kRing) * 16) + ring_rd[x] + lane16;
        auto fetch_a = [&](auto F) __attribute__((always_inline)) {
            constexpr int f = decltype(F)::value;
            if constexpr (f < NF) {
                f32x4 t;
                if constexpr (f < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(in_lds), "n"(f * 1024));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(pa_lds), "n"((f - MI) * 1024));
                aq[f % AD] = __builtin_bit_cast(bf16x8, t);
            }
        };
        // (The head's fp32 slice of h_x(s-1) and the gate cells' previous h are ordinary loads.  As inline-asm loads in
        // front of the fragments -- so that hipcc's waits for them cannot drain the prefetch ring -- they measured SLOWER:
        // decoder 0.480 against 0.468 ms, encoder 0.345 against 0.335, profiles/r04_bf16_own.txt.)
        f32x4 hd = splat4(0.f);
        if (DEC && has_prev) hd = hx[v * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        static_for<AD>([&](auto F) __attribute__((always_inline)) { fetch_a(F); });
        // fragment f has arrived when at most min(NF - 1 - f, AD - 1) younger fetches are outstanding
        auto wait_a = [&](auto F) __attribute__((always_inline)) {
            constexpr int f = decltype(F)::value;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF - 1 - f < AD - 1 ? NF - 1 - f : AD - 1) : "memory");
        };
        auto gate_slot = [&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            if constexpr (!gates || k >= NS) {
                return;
            } else if constexpr (k < 8) {                 // T: e1, e2 (the accumulators ARE the arguments of exp2: prescaled weights)
                constexpr int c = k >> 1;
                if constexpr ((k & 1) == 0) e1[c] = __builtin_amdgcn_exp2f(gr[c]);
                else e2[c] = __builtin_amdgcn_exp2f(gz[c]);
            } else if constexpr (k < 12) {                // P2
                constexpr int c = k - 8;
                e1[c] = 1.0f + e1[c];
                e2[c] = 1.0f + e2[c];
            } else if constexpr (k < 20) {                // T: r, z
                constexpr int c = (k - 12) >> 1;
                if constexpr (((k - 12) & 1) == 0) rg[c] = __builtin_amdgcn_rcpf(e1[c]);
                else zg[c] = __builtin_amdgcn_rcpf(e2[c]);
            } else if constexpr (k < 24) {                // P3 (and this cell's previous h on its way from LDS)
                constexpr int c = k - 20;
                t3[c] = __builtin_fmaf(rg[c], gnn[c], ggn[c]);
                hp[c] = kRegH ? hprev[o][c] : hpo[4 * c];
            } else if constexpr (k < 28) {                // T: e3
                constexpr int c = k - 24;
                e3[c] = __builtin_amdgcn_exp2f(t3[c]);
            } else if constexpr (k < 30) {                // P4
                constexpr int c = 2 * (k - 28);
                u3[c] = 1.0f + e3[c];
                u3[c + 1] = 1.0f + e3[c + 1];
            } else if constexpr (k < 34) {                // T: 1 / (1 + e3)
                constexpr int c = k - 30;
                qq[c] = __builtin_amdgcn_rcpf(u3[c]);
            } else if constexpr (k < 38) {                // P5
                constexpr int c = k - 34;
                ng[c] = __builtin_fmaf(-2.0f, qq[c], 1.0f);
                dd[c] = hp[c] - ng[c];
            } else {                                      // P6
                constexpr int c = 2 * (k - 38);
                hn[c] = __builtin_fmaf(zg[c], dd[c], ng[c]);
                hn[c + 1] = __builtin_fmaf(zg[c + 1], dd[c + 1], ng[c + 1]);
            }
        };
        // The head slice of h_x(s-1): two bf16 MFMAs on two-term splits (kernels_fused_bf16.h) between the input and the
        // recurrent part; the split of h (twelve VALU instructions) rides in front of the first.  (As four fp32 MFMAs it cost
        // the two waves of a SIMD 493 cycles of a 2,530-cycle region: profiles/r04_bf16_own.txt.)
        bf16x8 ha;
        auto head_item = [&](auto E) __attribute__((always_inline)) {
            constexpr int e = decltype(E)::value;
            if (has_prev) {
                if constexpr (e == 0) {
                    ha = head_split_h(hd);
                    pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, Bh.b1, splat4(0.f), 0, 0, 0);
                } else {
                    pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, Bh.b2, pl, 0, 0, 0);
                }
                HELEN_PIN(pl);
            }
        };
        auto mfma_item = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (i < NIN) {                                  // input part: fragments 0 .. MI-1
                constexpr int M = i / 3, g = i % 3;
                if constexpr (g == 0) wait_a(std::integral_constant<int, M>{});
                const bf16x8 a_cur = aq[M % AD];
                bf16x8 b;
                if constexpr (M < MR) b = Wi[g][M < MR ? M : 0];
                else b = wpark[((M < MR ? 0 : M - MR) * 3 + g) * 64];
                // (PIN: LLVM sinks a pure MFMA chain whose result is only needed at the end of the block below every
                // sched_barrier in between; an empty volatile asm that "modifies" the accumulator keeps each MFMA in its slot)
                if constexpr (g == 0) { ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, b, ar, 0, 0, 0); HELEN_PIN(ar); }
                if constexpr (g == 1) { az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, b, az, 0, 0, 0); HELEN_PIN(az); }
                if constexpr (g == 2) {
                    gnx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, b, gnx, 0, 0, 0);
                    HELEN_PIN(gnx);
                    fetch_a(std::integral_constant<int, M + AD>{});      // the slot of group M is free again
                }
            } else if constexpr (i < NIN + NHEAD) {
                head_item(std::integral_constant<int, i - NIN>{});
            } else {                                                  // recurrent part: fragments MI .. MI+3
                constexpr int ii = i - NIN - NHEAD, M = ii / 3, g = ii % 3;
                // the head slice of h_x(s-1) (slot s-1) is parked while the partials of slot s-2 (the other parity) are
                // still to be read at the end of this region
                if constexpr (DEC && ii == 6)
                    if (has_prev) (base + kPart + (((s - 1) & 1) * 8 + v) * 64)[lane] = pl;
                if constexpr (g == 0) wait_a(std::integral_constant<int, MI + M>{});
                const bf16x8 a_cur = aq[(MI + M) % AD];
                if constexpr (g == 0) { ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, Wh[0][M], ar, 0, 0, 0); HELEN_PIN(ar); }
                if constexpr (g == 1) { az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, Wh[1][M], az, 0, 0, 0); HELEN_PIN(az); }
                if constexpr (g == 2) {
                    ahn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, Wh[2][M], ahn, 0, 0, 0);
                    HELEN_PIN(ahn);
                    fetch_a(std::integral_constant<int, MI + M + AD>{});
                }
            }
        };
        // kLead gate slots go first: they cover the LDS latency of the first A fragment
        constexpr int kLead = HELEN_BF16_IL_LEAD(DEC);
        static_for<(NM + kLead > NS ? NM + kLead : NS)>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i >= kLead && i - kLead < NM) mfma_item(std::integral_constant<int, (i >= kLead ? i - kLead : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < NS) gate_slot(I);
        });
        __builtin_amdgcn_sched_barrier(0);
        ring_rd[x] = ring_rd[x] == (RD - 1u) * MI * 1024u ? 0u : ring_rd[x] + MI * 1024u;
        if constexpr (gates) {     // new h of tile o -> LDS (fp32 state / layer output, bf16 plane)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (kRegH) hprev[o][r] = hn[r];
                else ((float*)(obase + hsel(ow)))[hoff + 4 * r] = hn[r];
                ((unsigned short*)(obase + psel(ow)))[poff + 8 * r] = bf16_bits(hn[r]);
            }
        }
        // this phase's results become tile x's pending gate math
        Pr[x] = ar;
        Pz[x] = az;
        Pn[x] = ahn;
        Pg[x] = gnx;
        if (!DEC && has_prev) {                                  // h_x(s-1) as a bf16 plane = the layer output of slot s-1
            *(uint2*)(y_next[x] + in_block((unsigned)tid * 8u)) = ((const uint2*)(base + psel(cur)))[tid];
            y_next[x] += 512 * 16;
            issued += 1;
        }
        if (DEC && has_prev2) {                                  // slot s-2: partials parked in tile x's region of step s-1
            if (v < 4) {
                store_logits(x, s & 1, in_block((unsigned)tid * 4u));
                issued += 1;
            }
            y_next[x] += 128 * 16;
        }
        if (issued == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (issued == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using No = std::false_type;
    using Yes = std::true_type;
    // Regions in order: R(0,0) [no gates] | R(1,0) [G(0,0)] | R(0,1) [G(1,0)] | R(1,1) [G(0,1)] | ... | final G(1,T-1).
    // R(0,s): gates of tile 1 at step s-1, written to its buffer (s-1+1)&1 = s&1;  R(1,s): gates of tile 0 at step s,
    // written to buffer (s+1)&1.
    auto step = [&](auto STEADY, int s_) __attribute__((always_inline)) {
        if (s_ & 1) {
            region(I0{}, I1{}, I1{}, STEADY, Yes{}, s_, s_ - 1);
            region(I1{}, I1{}, I0{}, STEADY, Yes{}, s_, s_);
        } else {
            if (s_ == 0) region(I0{}, I0{}, I0{}, STEADY, No{}, 0, -1);
            else region(I0{}, I0{}, I0{}, STEADY, Yes{}, s_, s_ - 1);
            region(I1{}, I0{}, I1{}, STEADY, Yes{}, s_, s_);
        }
    };
    int s = 0;
    for (; s < T && s < 2; ++s) step(No{}, s);
    for (; s + 2 < T; s += 2) {                           // steady state: s >= 2 and (s + 1) + 1 < T
        region(I0{}, I0{}, I0{}, Yes{}, Yes{}, s, s - 1);
        region(I1{}, I0{}, I1{}, Yes{}, Yes{}, s, s);
        region(I0{}, I1{}, I1{}, Yes{}, Yes{}, s + 1, s);
        region(I1{}, I1{}, I0{}, Yes{}, Yes{}, s + 1, s + 1);
    }
    for (; s < T; ++s) step(No{}, s);
    // the gates of tile 1's last step (nothing left to interleave them with), into buffer T & 1
    {
        const int last = T & 1;
        f32x4* const obase = smem + kPerTile;
        float hp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = kRegH ? hprev[1][r] : ((const float*)(obase + hsel(last ^ 1)))[hoff + 4 * r];
        const f32x4 hn4 = gru_cell4_pre(Pr[1], Pz[1], Pn[1], Pg[1], hp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ((float*)(obase + hsel(last)))[hoff + 4 * r] = hn4[r];
            ((unsigned short*)(obase + psel(last)))[poff + 8 * r] = bf16_bits(hn4[r]);
        }
        if constexpr (kRegH) {      // tile 0's final state (its last gates ran in region (1, T-1), into buffer T & 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) ((float*)(smem + hsel(last)))[hoff + 4 * r] = hprev[0][r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int last = T & 1;   // buffers of h(T-1)
    if (DEC) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (T >= 2) {
                if (v < 4) store_logits(x, (T - 2) & 1, (unsigned)tid * 4u);
                y_next[x] += 128 * 16;
            }
            const f32x4 a = (smem + x * kPerTile + hsel(last))[v * 64 + lane];
            (smem + x * kPerTile + kPart + (((T - 1) & 1) * 8 + v) * 64)[lane] = head_mfma(head_split_h(a), Bh);
        }
        __syncthreads();
        if (v < 4) {
#pragma unroll
            for (int x = 0; x < 2; ++x) store_logits(x, (T - 1) & 1, (unsigned)tid * 4u);
        }
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
            *(uint2*)(y_next[x] + (unsigned)tid * 8u) = ((const uint2*)(smem + x * kPerTile + psel(last)))[tid];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) hid_p[x][tid] = (smem + x * kPerTile + hsel(last))[tid];
}

}  // namespace helen
