// kernels_x3_il.h -- HELEN_PRECISION_FP32X3: the recurrence with TWO window tiles per workgroup, the gate math and the
// three-term split of one tile INTERLEAVED with the other tile's MFMAs in one instruction stream
#pragma once
#include <type_traits>

#include "kernels_fused_bf16_il.h"
#include "kernels_x3.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Same arithmetic as gru_x3_kernel (the same MFMA order per accumulator -- K32 group, then the six leading products
// smallest first, then gate; the same IEEE operations per gate component; the same split of the new h into three bf16
// terms; the same order of the head's partial sums): results are bit-identical, so which of the two a call takes is
// scheduling (dispatch.h).
// gru_x3_kernel spends a tile-step, per SIMD, 2 x 1200 cycles on its 72 bf16 MFMAs per wave and 2 x 940 on the gate math,
// the split and the LDS round trip of the new h -- one after the other, because every wave of the workgroup is in the
// same phase (profiles/r05_fp32x3_levers.txt).  Beside a v_mfma_f32_16x16x32_bf16 one transcendental or two plain VALU
// instructions are free (profiles/ub_bf16_overlap.txt), and 72 MFMAs offer more such slots than a tile's gate math and
// split need (62).  So, as gru_fused_bf16_il_kernel does for the bf16 mode, ONE workgroup of 8 waves walks TWO tiles:
//     M(0,s) | G(0,s) M(1,s) | G(1,s) M(0,s+1) | ...        M = MFMA phase, G = gate math + split, | = the barrier
// and the region between two barriers is ONE stream: MFMA i of M(x,s), then slot i of G(o,.).
//   W_hh's three terms (144 registers) are shared by both tiles; a tile's h lives as three bf16 planes in LDS (two
//   buffers), its carried fp32 state in registers; a wave's gi fragments are DMA'd into its own LDS slot one region ahead
//   and read back when the gates need them; the decoder's head-weight fragments wait in LDS too (the register file holds
//   W_hh, two tiles' accumulators and the gate math: 251 / 256 registers).
//   Encoder launch: the planes of h_x(s-1) leave for gemm_dec_x3_kernel at the start of region (x, s); decoder launch: the
//   head slice of h_x(s-1) is three bf16 MFMAs on the planes of the wave's K32 group behind the region's last MFMA (their
//   fragments fetched behind MFMA 60), parked in LDS, summed over the eight waves (in wave order) by waves 0-3 at the end of
//   the tile's next region.
//   NO run-time branch between the first and the last MFMA of a region: one splits the region's basic block, and the
//   compiler then sinks the gate math out of its slots (sched_barrier binds the machine scheduler, not the IR passes) --
//   profiles/r06_region_anatomy.txt.  The VALU-between-MFMAs histogram of the disassembly is the check.
// grid (ceil(tiles / 2), 2 directions); an odd tile count makes the last workgroup walk its one tile twice.
// ------------------------------------------------------------------------------------------------
template <bool DEC>
__global__ __launch_bounds__(512, 1) void gru_x3_il_kernel(
    const f32x4* __restrict__ gi, long gi_tile_stride, int slot0_fwd, int slot0_bwd, int T, const bf16x8* __restrict__ W3,
    const float* __restrict__ bhn, f32x4* __restrict__ hid, f32x4* __restrict__ yplanes, long yp_tile_stride,
    const f32x4* __restrict__ Whd, f32x4* __restrict__ plogit, long pl_tile_stride, int ntiles) {
    // LDS per tile: bf16 planes [2 buffers][3 terms][256 units of 16 B] | gi fragments [8 waves][3 gates][64 f4] (each wave's
    // own: DMA'd one region ahead, read back by the same wave) | (DEC) head partials [2][8 waves][64 f4];
    // (DEC) after both tiles: every wave's three head-weight fragments [8][3][64]
    constexpr int kGi = 2 * 768, kPart = kGi + 8 * 192, kPerTile = kPart + (DEC ? 2 * 8 * 64 : 0);
    __shared__ f32x4 smem[2 * kPerTile + (DEC ? 8 * 192 : 0)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: hidden units 16v..16v+15
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int u = 16 * v + j;
    const int tile_of[2] = {min(2 * (int)blockIdx.x, ntiles - 1), min(2 * (int)blockIdx.x + 1, ntiles - 1)};

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
    // decoder: the head weights of K32 group Mv = v & 3 in three bf16 terms (gru_x3_kernel): waves v and v + 4 share the
    // group, v < 4 takes the three small products (h1 w3, h3 w1, h2 w2), v >= 4 the three large ones (h1 w2, h2 w1, h1 w1).
    // Which plane of h and which term of the weights a product takes are per-wave BYTE OFFSETS, not branches: a branch
    // inside a region splits its one basic block, and the compiler then sinks the gate math out of its slots into the
    // block that stores the result (the round-6 build of this kernel had 34 + 9 + 54 VALU instructions in three lumps
    // between its MFMAs instead of one or two behind each: profiles/r06_region_anatomy.txt).
    const int Mv = v & 3;
    bf16x8* const bh_lds = (bf16x8*)(smem + 2 * kPerTile) + v * 192 + lane;     // (DEC) this wave's Bh3[t] at [t * 64]
    const unsigned head_a[3] = {(unsigned)(0 * 256 + Mv * 64) * 16u, (unsigned)((v < 4 ? 2 : 1) * 256 + Mv * 64) * 16u,
                                (unsigned)((v < 4 ? 1 : 0) * 256 + Mv * 64) * 16u};
    const unsigned head_b[3] = {(unsigned)((v < 4 ? 2 : 1) * 64) * 16u, 0u, (unsigned)((v < 4 ? 1 : 0) * 64) * 16u};
    if (DEC) {
        bf16x8 Bh3[3];
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
        for (int t = 0; t < 3; ++t) {
            Bh3[t] = __builtin_bit_cast(bf16x8, uint4{tb[t][0] | (unsigned)tb[t][1] << 16, tb[t][2] | (unsigned)tb[t][3] << 16,
                                                       tb[t][4] | (unsigned)tb[t][5] << 16, tb[t][6] | (unsigned)tb[t][7] << 16});
            bh_lds[t * 64] = Bh3[t];      // (read back by this wave only: 256 registers hold W_hh and two tiles' state)
        }
    }

    constexpr long kPosStride = 2 * kNTile * 64;
    const f32x4* gi_p[2];
    f32x4* hid_p[2];
    char* y_next[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        gi_p[x] = gi + (size_t)tile_of[x] * gi_tile_stride + (size_t)slot0 * kPosStride + (size_t)dir * (kNTile * 64) + v * 64;
        hid_p[x] = hid + ((size_t)tile_of[x] * 2 + dir) * (kHidDirStride / 4);
        y_next[x] = DEC ? (char*)(plogit + (size_t)tile_of[x] * pl_tile_stride + (size_t)dir * 64)
                        : (char*)(yplanes + (size_t)tile_of[x] * yp_tile_stride + (size_t)dir * 768);
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)smem;
    // this lane's four cells: rows 4q + r of unit u.  fp32 state (hid): float index ((u>>2)*16 + 4q + r)*4 + (u&3);
    // planes: bf16 index ((u>>3)*16 + 4q + r)*8 + (u&7) inside a 256-unit plane
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);
    auto planes_of = [&](int x, int buf) __attribute__((always_inline)) { return smem + x * kPerTile + buf * 768; };
    auto store_planes = [&](int x, int buf, int r, unsigned short t1, unsigned short t2, unsigned short t3) __attribute__((always_inline)) {
        unsigned short* pl = (unsigned short*)planes_of(x, buf);
        pl[0 * 2048 + poff + 8 * r] = t1;
        pl[1 * 2048 + poff + 8 * r] = t2;
        pl[2 * 2048 + poff + 8 * r] = t3;
    };
    auto store_logits = [&](int x, int pb, unsigned voff) __attribute__((always_inline)) {   // 256 threads: eight partials in wave order
        const float* pp = (const float*)(smem + x * kPerTile + kPart + pb * 8 * 64) + tid;
        float sum = pp[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += pp[k * 256];
        *(float*)(y_next[x] + voff) = sum;
    };

    // ---- prologue: the carried state into registers, its three planes into buffer 0 -- for both tiles
    float hprev[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h = ((const float*)hid_p[x])[hoff + 4 * r];
            hprev[x][r] = h;
            const unsigned short t1 = bf16_bits(h);
            const float r1 = h - bf16_to_f32(t1);
            const unsigned short t2 = bf16_bits(r1);
            store_planes(x, 0, r, t1, t2, bf16_bits(r1 - bf16_to_f32(t2)));
        }
    __syncthreads();

    // Pending gate math of each tile: the finished accumulators of its newest step (that step's gi fragments wait in LDS)
    f32x4 Pr[2], Pz[2], Pn[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) Pr[x] = Pz[x] = Pn[x] = splat4(0.f);
    auto gi_slot = [&](int x) __attribute__((always_inline)) { return smem + x * kPerTile + kGi + v * 192; };
    auto dma_gi = [&](int x, int s_) __attribute__((always_inline)) {      // the three gate fragments of tile x's step s_
        const f32x4* p = gi_p[x] + (size_t)s_ * kPosStride + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p + (g * 8) * 64),
                                             (void __attribute__((address_space(3)))*)(gi_slot(x) + g * 64), 16, 0, 0);
    };

    constexpr int NM = 72;          // MFMAs of one M phase: 4 K32 groups x 6 products x 3 gates
    constexpr int NS = 62;          // gate + split slots
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0};   // six leading products, smallest first (gru_x3_kernel)
    constexpr int TB[6] = {2, 0, 1, 1, 0, 0};

    // One region between two barriers: the MFMA phase of tile X at step s, and -- if GATES -- the gate math and split of tile
    // O = 1 - X at its newest step so (accumulators in P*[O], gi in G[O]), slot by slot behind the MFMAs.
    // CUR = s & 1 = the plane buffer of h_x(s-1); OW = the buffer tile O's new h goes to ((so + 1) & 1).
    auto region = [&](auto X, auto CUR, auto OW, auto STEADY, auto GATES, int s, int so) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value, ow = decltype(OW)::value;
        constexpr bool steady = decltype(STEADY)::value, gates = decltype(GATES)::value;
        const bool has_prev = steady || s > 0;
        const bool has_prev2 = steady || s > 1;
        f32x4* const base = smem + x * kPerTile;
        // the gi fragments of tile o's pending step were DMA'd a region ago.  (vmcnt(0), not a counted wait: the one younger
        // entry a decoder wave may have in the VMEM queue is the logit store of the last region's end, and loads and stores
        // do not retire in order with each other; that store has had the wait at the barrier to complete.)
        if (gates) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 gr = Pr[o], gz = Pz[o], gnn = Pn[o];
        f32x4 ir = splat4(0.f), iz = splat4(0.f), in_ = splat4(0.f);
        if (gates) {
            const f32x4* gs = gi_slot(o) + lane;
            ir = gs[0];
            iz = gs[64];
            in_ = gs[128];
        }
        // The encoder's layer output leaves at the region's start (every wave has a share of it).
        if (!DEC && has_prev) {              // the planes of h_x(s-1) = the layer output of slot s-1: 768 units of 16 B
            const f32x4* ps = base + cur * 768;
            f32x4* po = (f32x4*)y_next[x];
            po[in_block((unsigned)tid)] = ps[tid];
            if (tid < 256) po[512 + in_block((unsigned)tid)] = ps[512 + tid];
            y_next[x] += 2 * 768 * 16;
        }
        dma_gi(x, s);               // this tile's gi of step s, for the gates one region on
        f32x4 ar = splat4(0.f), az = splat4(0.f), ahn = splat4(bn), pl = splat4(0.f);
        float sr[4], sz[4], e1[4], e2[4], rg[4], zg[4], t3[4], e3[4], u3[4], qq[4], ng[4], dd[4], hn[4], r1[4], r2[4];
        unsigned short b1[4], b2[4], b3[4];
        // A fragments: plane t of K32 group M.  A group's 18 MFMAs take plane 0 (products 0, 3, 5), plane 2 (product 1) and
        // plane 1 (products 2, 4), so FOUR registers of fragments are enough: plane 1 in slot 1, plane 2 in slot 2, plane 0
        // in slot 0 (even M) or 3 (odd M); (M + 1, 0) is fetched in front of group M, (M + 1, 2) behind product 1 of group M,
        // (M + 1, 1) behind product 4.  Inline-asm loads, waited for by position in the in-order LDS queue: each of the
        // three is followed by exactly two younger fetches when it is needed (fewer in the last group).
        bf16x8 aq[4];
        const unsigned pa_lds = lds0 + (unsigned)((x * kPerTile + cur * 768) * 16) + lane16;
        auto fetch = [&](auto MM, auto TT) __attribute__((always_inline)) {
            constexpr int M = decltype(MM)::value, t = decltype(TT)::value;
            if constexpr (M < 4) {
                f32x4 tmp;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tmp) : "v"(pa_lds), "n"((t * 256 + M * 64) * 16));
                aq[t == 0 ? (M & 1 ? 3 : 0) : t] = __builtin_bit_cast(bf16x8, tmp);
            }
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        __builtin_amdgcn_sched_barrier(0);
        fetch(T0{}, T0{});
        fetch(T0{}, T2{});
        fetch(T0{}, T1{});
        auto gate_slot = [&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            if constexpr (!gates || k >= NS) {
                return;
            } else if constexpr (k < 4) {                 // P: the pre-activations of r and z complete
                constexpr int c = k;
                sr[c] = gr[c] + ir[c];
                sz[c] = gz[c] + iz[c];
            } else if constexpr (k < 8) {                 // P
                constexpr int c = k - 4;
                sr[c] = sr[c] * -1.4426950408889634f;
                sz[c] = sz[c] * -1.4426950408889634f;
            } else if constexpr (k < 16) {                // T
                constexpr int c = (k - 8) >> 1;
                if constexpr (((k - 8) & 1) == 0) e1[c] = __builtin_amdgcn_exp2f(sr[c]);
                else e2[c] = __builtin_amdgcn_exp2f(sz[c]);
            } else if constexpr (k < 20) {                // P
                constexpr int c = k - 16;
                e1[c] = 1.0f + e1[c];
                e2[c] = 1.0f + e2[c];
            } else if constexpr (k < 28) {                // T: r, z
                constexpr int c = (k - 20) >> 1;
                if constexpr (((k - 20) & 1) == 0) rg[c] = __builtin_amdgcn_rcpf(e1[c]);
                else zg[c] = __builtin_amdgcn_rcpf(e2[c]);
            } else if constexpr (k < 32) {                // P
                constexpr int c = k - 28;
                t3[c] = __builtin_fmaf(rg[c], gnn[c], in_[c]);
                t3[c] = t3[c] * 2.8853900817779268f;
            } else if constexpr (k < 36) {                // T
                constexpr int c = k - 32;
                e3[c] = __builtin_amdgcn_exp2f(t3[c]);
            } else if constexpr (k < 38) {                // P
                constexpr int c = 2 * (k - 36);
                u3[c] = 1.0f + e3[c];
                u3[c + 1] = 1.0f + e3[c + 1];
            } else if constexpr (k < 42) {                // T
                constexpr int c = k - 38;
                qq[c] = __builtin_amdgcn_rcpf(u3[c]);
            } else if constexpr (k < 46) {                // P
                constexpr int c = k - 42;
                ng[c] = __builtin_fmaf(-2.0f, qq[c], 1.0f);
                dd[c] = hprev[o][c] - ng[c];
            } else if constexpr (k < 48) {                // P
                constexpr int c = 2 * (k - 46);
                hn[c] = __builtin_fmaf(zg[c], dd[c], ng[c]);
                hn[c + 1] = __builtin_fmaf(zg[c + 1], dd[c + 1], ng[c + 1]);
            } else if constexpr (k < 52) {                // the split: h = b1 + b2 + b3 (RNE each), as gru_x3_kernel's store_h
                constexpr int c = k - 48;
                b1[c] = bf16_bits(hn[c]);
                r1[c] = hn[c] - bf16_to_f32(b1[c]);
            } else if constexpr (k < 56) {
                constexpr int c = k - 52;
                b2[c] = bf16_bits(r1[c]);
                r2[c] = r1[c] - bf16_to_f32(b2[c]);
            } else if constexpr (k < 58) {
                constexpr int c = 2 * (k - 56);
                b3[c] = bf16_bits(r2[c]);
                b3[c + 1] = bf16_bits(r2[c + 1]);
            } else {                                      // the new h becomes the carried state; its three planes -> LDS
                constexpr int c = k - 58;
                hprev[o][c] = hn[c];
                store_planes(o, ow, c, b1[c], b2[c], b3[c]);
            }
        };
        auto mfma_item = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            constexpr int M = i / 18, w18 = i % 18, kk = w18 / 3, g = i % 3;
            constexpr int slot0 = M & 1 ? 3 : 0;
            if constexpr (w18 == 0) {                     // plane 0: behind it (M, 2), (M, 1) -- and nothing else yet
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                fetch(std::integral_constant<int, M + 1>{}, T0{});
            }
            if constexpr (w18 == 3) {                     // plane 2: behind it (M, 1), (M + 1, 0)
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(M < 3 ? 2 : 1) : "memory");
            }
            if constexpr (w18 == 6)                       // plane 1: behind it (M + 1, 0), (M + 1, 2)
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(M < 3 ? 2 : 0) : "memory");
            const bf16x8 a_cur = aq[TA[kk] == 0 ? slot0 : TA[kk]];
            // (PIN: keeps each MFMA in its slot; LLVM otherwise sinks the chain below the sched_barriers)
            if constexpr (g == 0) { ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, W[0][M][TB[kk]], ar, 0, 0, 0); HELEN_PIN(ar); }
            if constexpr (g == 1) { az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, W[1][M][TB[kk]], az, 0, 0, 0); HELEN_PIN(az); }
            if constexpr (g == 2) { ahn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur, W[2][M][TB[kk]], ahn, 0, 0, 0); HELEN_PIN(ahn); }
            if constexpr (w18 == 5) fetch(std::integral_constant<int, M + 1>{}, T2{});      // product 1 done: slot 2 is free
            if constexpr (w18 == 14) fetch(std::integral_constant<int, M + 1>{}, T1{});     // product 4 done: slot 1 is free
        };
        // (DEC) the head slice of h_x(s-1): this wave's three products on the planes of its K32 group.  The six fragments
        // are fetched behind MFMA 60 -- the last counted wait of the stream is in front of it, the gate slots are through
        // and their registers free -- and multiplied behind the last MFMA of the region.
        bf16x8 ha[3], hb[3];
        auto head_fetch = [&]() __attribute__((always_inline)) {
            const unsigned bh0 = lds0 + (unsigned)((2 * kPerTile) * 16) + (unsigned)(v * 192 * 16) + lane16;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                f32x4 ta, tb;
                asm volatile("ds_read_b128 %0, %1" : "=v"(ta) : "v"(pa_lds + head_a[k]));
                asm volatile("ds_read_b128 %0, %1" : "=v"(tb) : "v"(bh0 + head_b[k]));
                ha[k] = __builtin_bit_cast(bf16x8, ta);
                hb[k] = __builtin_bit_cast(bf16x8, tb);
            }
        };
        constexpr int kLead = 2;             // gate slots in front of the first MFMA: they cover the LDS latency of group 0
        static_for<(NM + kLead > NS ? NM + kLead : NS)>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i >= kLead && i - kLead < NM) mfma_item(std::integral_constant<int, (i >= kLead ? i - kLead : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DEC && i - kLead == 60) {
                static_assert(60 + kLead >= NS, "the gate slots must be through when the head's fragments take their registers");
                if (has_prev) head_fetch();
            }
            if constexpr (i < NS) gate_slot(I);
        });
        __builtin_amdgcn_sched_barrier(0);
        if (DEC && has_prev) {
            // (the six fragments are operands of the wait: an MFMA that reads one cannot be scheduled in front of it)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ha[0]), "+v"(ha[1]), "+v"(ha[2]), "+v"(hb[0]), "+v"(hb[1]), "+v"(hb[2])::"memory");
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[0], hb[0], splat4(0.f), 0, 0, 0);      // smallest first (gru_x3_kernel)
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[1], hb[1], pl, 0, 0, 0);
            pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[2], hb[2], pl, 0, 0, 0);
            (base + kPart + (((s - 1) & 1) * 8 + v) * 64)[lane] = pl;     // parked; summed one region of this tile later
        }
        Pr[x] = ar;
        Pz[x] = az;
        Pn[x] = ahn;
        // The sum of slot s-2's eight partials (parked in tile x's region of step s-1) leaves from waves 0-3: the OLDER
        // wave of each SIMD is through its stream ~1,000 cycles before the younger one and would only wait at the barrier.
        if (DEC && has_prev2) {
            if (v < 4) store_logits(x, s & 1, in_block((unsigned)tid * 4u));
            y_next[x] += 128 * 16;
        }
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
    // R(0,s): gates of tile 1 at step s-1, written to its buffer s & 1;  R(1,s): gates of tile 0 at step s, written to
    // buffer (s+1) & 1.
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
    const int last = T & 1;
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4* gs = gi_slot(1) + lane;
        const f32x4 hn4 = gru_cell4(Pr[1], Pz[1], Pn[1], gs[0], gs[64], gs[128], hprev[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[1][r] = hn4[r];
            const unsigned short t1 = bf16_bits(hn4[r]);
            const float r1 = hn4[r] - bf16_to_f32(t1);
            const unsigned short t2 = bf16_bits(r1);
            store_planes(1, last, r, t1, t2, bf16_bits(r1 - bf16_to_f32(t2)));
        }
    }
    __syncthreads();
    if (DEC) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (T >= 2) {
                if (v < 4) store_logits(x, (T - 2) & 1, (unsigned)tid * 4u);
                y_next[x] += 128 * 16;
            }
            // the last step's logits: the head slice of h_x(T-1)
            const bf16x8* pa = (const bf16x8*)planes_of(x, last) + lane + Mv * 64;
            const bf16x8 a0 = pa[0], a1 = pa[256], a2 = pa[512];
            const bf16x8* bh = (const bf16x8*)bh_lds;
            f32x4 pl;
            if (v < 4) {
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh[128], splat4(0.f), 0, 0, 0);
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bh[0], pl, 0, 0, 0);
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh[64], pl, 0, 0, 0);
            } else {
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh[64], splat4(0.f), 0, 0, 0);
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh[0], pl, 0, 0, 0);
                pl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh[0], pl, 0, 0, 0);
            }
            (smem + x * kPerTile + kPart + (((T - 1) & 1) * 8 + v) * 64)[lane] = pl;
        }
        __syncthreads();
        if (v < 4) {
#pragma unroll
            for (int x = 0; x < 2; ++x) store_logits(x, (T - 1) & 1, (unsigned)tid * 4u);
        }
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x) {        // the planes of h_x(T-1): the layer output of the last slot
            const f32x4* ps = planes_of(x, last);
            f32x4* po = (f32x4*)y_next[x];
            po[tid] = ps[tid];
            if (tid < 256) po[512 + tid] = ps[512 + tid];
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) ((float*)hid_p[x])[hoff + 4 * r] = hprev[x][r];
}

}  // namespace helen
