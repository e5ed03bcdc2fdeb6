// kernels_fused_bf16_w4.h -- HELEN_PRECISION_BF16: fused projection + recurrence, two window tiles per workgroup of FOUR
// waves -- one wave per SIMD, 512 registers each, every weight fragment of 32 hidden units resident (AGPRs)
#pragma once
#include <type_traits>

#include "kernels_fused_bf16_il.h"

#ifndef HELEN_BF16_W4_ADEPTH      // A fragments in flight (4 registers each)
#define HELEN_BF16_W4_ADEPTH(dec) ((dec) ? 4 : 7)
#endif
#ifndef HELEN_BF16_W4_LEAD        // gate slots ahead of the first MFMA of a region
#define HELEN_BF16_W4_LEAD 6
#endif
#ifndef HELEN_BF16_W4_SPM         // gate slots behind each MFMA
#define HELEN_BF16_W4_SPM(dec) ((dec) ? 1 : 2)
#endif

namespace helen {

// ------------------------------------------------------------------------------------------------
// Why four waves.  The eight-wave kernels (pair, il) give every wave 16 hidden units: each wave reads ALL K32 groups of
// the tile's A operand (h plane + input row) from LDS -- the decoder 12 KiB per wave and region, 96 KiB per workgroup --
// and, at 256 registers a wave, parks one to three K32 groups of W_ih in LDS as well (24-72 KiB more per region).  At
// 128 B per cycle that is 1000-1300 cycles of LDS pipe per region beside 1360 cycles of MFMA issue: the decoder's MFMA
// stream alone (gate slots removed) measured 2036-2240 cycles per region, and neither deeper A prefetch nor deeper
// prefetch of the parked fragments moved it (round 4, profiles/r04_bf16_il_decoder.txt).
// Here a wave owns 32 hidden units (two 16-column groups) and is alone on its SIMD, so it may use all 512 registers:
// the decoder's 72 weight fragments (288 registers) sit in the 256 AGPRs + 32 VGPRs, nothing is parked, and the A
// operand is read by four waves instead of eight: 48 KiB of LDS reads per region.  The MFMAs are inline asm with the
// weight operand constrained to an AGPR -- hipcc's allocator, left to itself with builtins, shuffles accumulators and
// weights through v_accvgpr_* moves inside the loop.
// Arithmetic: the same MFMA order per accumulator, the same IEEE gate operations per component, the same eight head
// partials summed in the same order as gru_fused_bf16_{,pair,il}_kernel: bit-identical results.
// Region structure as in the il kernel:  M(0,s) | G(0,s) M(1,s) | G(1,s) M(0,s+1) | ...   with one instruction stream
// per region: MFMA i of M(x,.) followed by SPM gate slots of G(o,.) (eight cells per lane).
// Hazards: hipcc does not know the asm statements are MFMAs and inserts no wait states behind them.  The program reads
// their results only in the NEXT region, but the compiler may copy or spill an accumulator wherever it is live: (1) the
// two places where chains END (last recurrent MFMA, last input MFMA) are followed by 18 wait states; (2) no MFMA is
// conditional (a conditional one ends in a phi copy); (3) the kernel must not spill accumulators -- scripts/dev/mfma_hazards.py
// scans the built code object for any non-MFMA read of an asm MFMA's result within 12 instructions and
// tests/test_abi_and_layout.py runs it.  The first MFMA of a chain takes an accumulator a v_mov may just have written:
// s_nop 1 in front of it.
// ------------------------------------------------------------------------------------------------
template <bool AGPR>
__device__ __forceinline__ void mfma_bf16_w(f32x4& acc, const bf16x8& a, const bf16x8& w) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}

// the first MFMA of a chain: C is a loop-invariant splat of the bias, D the accumulator (no v_mov x4 per chain and step)
template <bool AGPR>
__device__ __forceinline__ void mfma_bf16_w0(f32x4& acc, const bf16x8& a, const bf16x8& w, const f32x4& c) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(w), "v"(c));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(w), "v"(c));
}

// the head's fp32 MFMA, as asm for the same reason: the builtin picks the AGPR form here and evicts weight fragments
template <bool FIRST>
__device__ __forceinline__ void mfma4_v(f32x4& acc, float a, float b) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int MI, bool DEC>
__global__ __launch_bounds__(256, 1) void gru_fused_bf16_w4_kernel(
    const f32x4* __restrict__ in, long in_tile_stride, int pos0, int T, const bf16x8* __restrict__ Wi3,
    const bf16x8* __restrict__ Wh3, const float* __restrict__ bias, const float* __restrict__ bhn,
    f32x4* __restrict__ hid, f32x4* __restrict__ yplane_out, long yp_tile_stride,
    const f32x4* __restrict__ Whd, f32x4* __restrict__ plogit, long pl_tile_stride, int ntiles) {
    // LDS per tile: fp32 h [2][512 f4] | bf16 h plane [2][256] | input ring [RD][MI * 64] | (DEC) head partials [2][8][64]
    constexpr int RD = 2;
    constexpr int kPlane = 2 * 512, kRing = kPlane + 2 * 256, kPart = kRing + RD * MI * 64, kPerTile = kPart + (DEC ? 2 * 8 * 64 : 0);
    auto hsel = [](int b) __attribute__((always_inline)) { return b * 512; };                 // fp32 h buffer b (f4 offset)
    auto psel = [=](int b) __attribute__((always_inline)) { return kPlane + b * 256; };       // bf16 plane b
    __shared__ f32x4 smem[2 * kPerTile];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int tile_of[2] = {min(2 * (int)blockIdx.x, ntiles - 1), min(2 * (int)blockIdx.x + 1, ntiles - 1)};
    // column group cg of this wave = column tile 2w + cg of the eight-wave kernels
    constexpr int NW = 4 + MI;                                // weight fragments per (column group, gate)
    constexpr int kAgprFrags = 64;
    bf16x8 Wh[2][3][4], Wi[2][3][MI];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
        const int vc = 2 * w + cg;
        const bf16x8* wh = Wh3 + (size_t)((dir * 8 + vc) * 36) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int M = 0; M < 4; ++M) Wh[cg][g][M] = wh[((g * 4 + M) * 3) * 64];
            const bf16x8* wi = Wi3 + (size_t)((dir * kNTile + g * 8 + vc) * MI) * 3 * 64 + lane;
#pragma unroll
            for (int M = 0; M < MI; ++M) Wi[cg][g][M] = wi[(M * 3) * 64];
        }
    }
    f32x4 Bh[2] = {splat4(0.f), splat4(0.f)};   // DEC: head weights for k = dir*128 + 16 vc + 4q + e, class j
    f32x4 bi[2][3], bn[2];       // bias splats: C of the first MFMA of a chain
    int hoff[2], poff[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
        const int vc = 2 * w + cg, u = 16 * vc + j;
        if (DEC) Bh[cg] = Whd[(dir * 8 + vc) * 64 + lane];
#pragma unroll
        for (int g = 0; g < 3; ++g) bi[cg][g] = splat4(bias[dir * kG + g * kH + u]);
        bn[cg] = splat4(bhn[dir * kH + u]);
        hoff[cg] = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
        poff[cg] = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);
    }

    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)smem;
    // input rows: K32 group k of a row is DMA'd by wave k & 3 (the decoder's groups 4..7 are the other direction's plane,
    // which runs the other way in time)
    constexpr int kDma = (MI + 3) / 4;                        // DMAs per wave and row (at most)
    const char* in_next[2][kDma];
    long in_step[kDma];
#pragma unroll
    for (int d = 0; d < kDma; ++d) {
        const int k = w + 4 * d;
        long first;
        if (DEC) {
            const int p = k >> 2;
            const bool up = p == dir;
            first = ((long)(up ? 0 : T - 1) * 2 + p) * 256 + (k & 3) * 64;
            in_step[d] = (up ? 1 : -1) * 512L * 16;
        } else {
            first = (long)(pos0 + (dir ? T - 1 : 0)) * (MI * 64) + (k < MI ? k : 0) * 64;
            in_step[d] = (dir ? -1 : 1) * (long)(MI * 64) * 16;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) in_next[x][d] = (const char*)(in + (size_t)tile_of[x] * in_tile_stride + first);
    }
    unsigned ring_dma[2], ring_rd[2];
    auto dma_in = [&](int x) __attribute__((always_inline)) {     // returns nothing; dma_count() VMEM operations
#pragma unroll
        for (int d = 0; d < kDma; ++d) {
            const int k = w + 4 * d;
            if (k < MI) dma_row_to_lds(lds0 + (unsigned)((x * kPerTile + kRing) * 16) + ring_dma[x] + (unsigned)k * 1024u,
                                       in_next[x][d], in_block(lane16));
            in_next[x][d] += in_step[d];
        }
        ring_dma[x] = ring_dma[x] == (RD - 1u) * MI * 1024u ? 0u : ring_dma[x] + MI * 1024u;
    };
    const int dma_count = (w < MI ? 1 : 0) + (w + 4 < MI ? 1 : 0);

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
        smem[x * kPerTile + 256 + tid] = hid_p[x][256 + tid];
        ring_dma[x] = 0;
        ring_rd[x] = 0;
        dma_in(x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ((unsigned short*)(smem + x * kPerTile + kPlane))[poff[cg] + 8 * r] =
                    bf16_bits(((const float*)(smem + x * kPerTile))[hoff[cg] + 4 * r]);
    __syncthreads();
    // the previous h of a lane's eight cells stays in registers (the eight-wave kernels re-read it from the fp32 buffer)
    float hprev[2][8];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int c = 0; c < 8; ++c) hprev[x][c] = ((const float*)(smem + x * kPerTile))[hoff[c >> 2] + 4 * (c & 3)];

    // Pending gate math of each tile: the finished accumulators of its newest step.
    f32x4 Pr[2][2], Pz[2][2], Pn[2][2], Pg[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) Pr[x][cg] = Pz[x][cg] = Pn[x][cg] = Pg[x][cg] = splat4(0.f);

#ifdef HELEN_BIL_TIMING   // developer probe: where a wave's cycles go
    long long tk[4] = {0, 0, 0, 0}, tm0 = 0;
    long long tlast = __builtin_readcyclecounter();
#endif
    constexpr int NREC = 24, NHEAD = DEC ? 8 : 0;
    constexpr int NIN = 6 * MI;
    constexpr int NM = NIN + NHEAD + NREC;            // MFMAs of one M phase: the step's input part, head slices of h(s-1), recurrent part
    constexpr int NC = 8;                             // gate cells per lane: column group c >> 2, row 4q + (c & 3)
    constexpr int NS = 11 * NC;                       // gate slots (below)

    // One region: the MFMA phase of tile X at step s and the gate math of tile O = 1 - X at its newest step so (so = -1
    // in the very first region: the slots then run on zeros and store nothing).  STEADY: s >= 2 and s + 1 < T, every
    // flag below a constant -- a wave alone on its SIMD pays an issue slot (4+ cycles) for every scalar compare and branch.
    auto region = [&](auto X, auto CUR, auto OW, auto STEADY, int s, int so) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value, ow = decltype(OW)::value;
        constexpr bool gates = true, steady = decltype(STEADY)::value;
        const bool has_prev = steady || s > 0;
        const bool has_prev2 = steady || s > 1;
        const bool has_next = steady || s + 1 < T;
        const bool has_gates = steady || so >= 0;
        f32x4* const base = smem + x * kPerTile;
        f32x4* const obase = smem + o * kPerTile;
        const f32x4* hx = base + hsel(cur);
        int issued = 0;
        if (has_next) {           // the row of step s+1 into the ring slot whose row (step s-1) was read a whole step ago
            dma_in(x);
            issued += dma_count;
        }
        // Accumulators of tile x's step s: r and z take bias, input part and recurrent part in ONE chain each (the order of
        // the other kernels, whose input part is computed a region ahead into separate registers -- 48 more of them,
        // carried across regions); n keeps its input part (gnx) and its recurrent part (ahn) apart.
        f32x4 ar[2], az[2], ahn[2], gnx[2], pl[2], hd[2];
        f32x4 gr[2], gz[2], gnn[2], ggn[2];
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            gr[cg] = Pr[o][cg], gz[cg] = Pz[o][cg], gnn[cg] = Pn[o][cg], ggn[cg] = Pg[o][cg];
            hd[cg] = splat4(0.f);
            if (DEC && has_prev) hd[cg] = hx[(2 * w + cg) * 64 + lane];
        }
        float gt[14][4];      // gate temporaries of the half in progress: t1 t2 e1 e2 r z t3 e3 u3 q n d h' h
        // A fragments: the MI K32 groups of tile x's input row of step s, then the four of h_x(s-1).  AD in flight, fragment
        // f + AD fetched behind the last MFMA of group f.  Inline asm loads, waited for by position in the in-order LDS queue.
        constexpr int NF = MI + 4, AD = HELEN_BF16_W4_ADEPTH(DEC) < NF ? HELEN_BF16_W4_ADEPTH(DEC) : NF;
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
        static_for<AD>([&](auto F) __attribute__((always_inline)) { fetch_a(F); });
        // fragment f has arrived when at most min(NF - 1 - f, AD - 1) younger fetches are outstanding
        auto wait_a = [&](auto F) __attribute__((always_inline)) {
            constexpr int f = decltype(F)::value;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF - 1 - f < AD - 1 ? NF - 1 - f : AD - 1) : "memory");
        };

        // Gate program: the il kernel's 44 slots for four cells (one column group), run twice -- column group 0, then 1.
        // A slot sits behind an MFMA (16 cycles), so four staggered cells are more than enough to keep dependent
        // instructions apart, and the temporaries of one half are dead before the other starts.  The last slot of a half
        // stores its four new h values (fp32 state and bf16 plane of buffer `ow`, which nobody reads in this region).
        auto gate_slot = [&](auto K) __attribute__((always_inline)) {
            constexpr int k0 = decltype(K)::value;
            if constexpr (!gates || k0 >= NS) {
                return;
            } else {
                constexpr int H = k0 / 44, k = k0 % 44;
                float* const t1 = gt[0], * const t2 = gt[1], * const e1 = gt[2], * const e2 = gt[3], * const rg = gt[4], * const zg = gt[5],
                     * const t3 = gt[6], * const e3 = gt[7], * const u3 = gt[8], * const qq = gt[9], * const ng = gt[10], * const dd = gt[11],
                     * const hn = gt[12], * const hp = gt[13];
                if constexpr (k < 4) {                        // P1
                    t1[k] = gr[H][k] * -1.4426950408889634f;
                    t2[k] = gz[H][k] * -1.4426950408889634f;
                } else if constexpr (k < 12) {                // T: e1, e2
                    constexpr int c = (k - 4) >> 1;
                    if constexpr (((k - 4) & 1) == 0) e1[c] = __builtin_amdgcn_exp2f(t1[c]);
                    else e2[c] = __builtin_amdgcn_exp2f(t2[c]);
                } else if constexpr (k < 16) {                // P2
                    constexpr int c = k - 12;
                    e1[c] = 1.0f + e1[c];
                    e2[c] = 1.0f + e2[c];
                } else if constexpr (k < 24) {                // T: r, z
                    constexpr int c = (k - 16) >> 1;
                    if constexpr (((k - 16) & 1) == 0) rg[c] = __builtin_amdgcn_rcpf(e1[c]);
                    else zg[c] = __builtin_amdgcn_rcpf(e2[c]);
                } else if constexpr (k < 28) {                // P3
                    constexpr int c = k - 24;
                    t3[c] = __builtin_fmaf(rg[c], gnn[H][c], ggn[H][c]) * 2.8853900817779268f;
                    hp[c] = hprev[o][4 * H + c];
                } else if constexpr (k < 32) {                // T: e3
                    constexpr int c = k - 28;
                    e3[c] = __builtin_amdgcn_exp2f(t3[c]);
                } else if constexpr (k < 34) {                // P4
                    constexpr int c = 2 * (k - 32);
                    u3[c] = 1.0f + e3[c];
                    u3[c + 1] = 1.0f + e3[c + 1];
                } else if constexpr (k < 38) {                // T: 1 / (1 + e3)
                    constexpr int c = k - 34;
                    qq[c] = __builtin_amdgcn_rcpf(u3[c]);
                } else if constexpr (k < 42) {                // P5
                    constexpr int c = k - 38;
                    ng[c] = __builtin_fmaf(-2.0f, qq[c], 1.0f);
                    dd[c] = hp[c] - ng[c];
                } else {                                      // P6
                    constexpr int c = 2 * (k - 42);
                    hn[c] = __builtin_fmaf(zg[c], dd[c], ng[c]);
                    hn[c + 1] = __builtin_fmaf(zg[c + 1], dd[c + 1], ng[c + 1]);
                    if constexpr (k == 43) if (has_gates) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            hprev[o][4 * H + r] = hn[r];
                            ((float*)(obase + hsel(ow)))[hoff[H] + 4 * r] = hn[r];
                            ((unsigned short*)(obase + psel(ow)))[poff[H] + 8 * r] = bf16_bits(hn[r]);
                        }
                    }
                }
            }
        };
        auto mfma_item = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (i < NIN) {                                  // input part: fragments 0 .. MI-1
                constexpr int M = i / 6, cg = (i % 6) / 3, g = i % 3;
                constexpr bool ag = (cg * 3 + g) * NW + 4 + M < kAgprFrags;
                if constexpr (i % 6 == 0) wait_a(std::integral_constant<int, M>{});
                const bf16x8 a_cur = aq[M % AD];
                if constexpr (M == 0) {
                    if constexpr (g == 0) mfma_bf16_w0<ag>(ar[cg], a_cur, Wi[cg][0][0], bi[cg][0]);
                    if constexpr (g == 1) mfma_bf16_w0<ag>(az[cg], a_cur, Wi[cg][1][0], bi[cg][1]);
                    if constexpr (g == 2) mfma_bf16_w0<ag>(gnx[cg], a_cur, Wi[cg][2][0], bi[cg][2]);
                } else {
                    if constexpr (g == 0) mfma_bf16_w<ag>(ar[cg], a_cur, Wi[cg][0][M]);
                    if constexpr (g == 1) mfma_bf16_w<ag>(az[cg], a_cur, Wi[cg][1][M]);
                    if constexpr (g == 2) mfma_bf16_w<ag>(gnx[cg], a_cur, Wi[cg][2][M]);
                }
                if constexpr (i % 6 == 5) fetch_a(std::integral_constant<int, M + AD>{});      // the slot of group M is free again
                if constexpr (i == NIN - 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory");   // the gnx chains end here
            } else if constexpr (i < NIN + NHEAD) {
                constexpr int cg = (i - NIN) / 4, e = (i - NIN) % 4;
                if (has_prev) mfma4_v<e == 0>(pl[cg], hd[cg][e], Bh[cg][e]);
            } else {                                                  // recurrent part: fragments MI .. MI+3
                constexpr int ii = i - NIN - NHEAD, M = ii / 6, cg = (ii % 6) / 3, g = ii % 3;
                constexpr bool ag = (cg * 3 + g) * NW + M < kAgprFrags;
                // the head slices of h_x(s-1) (slot s-1) are parked while the partials of slot s-2 (the other parity) are
                // still to be read at the end of this region
                if constexpr (DEC && ii == 6)
                    if (has_prev) {
                        (base + kPart + (((s - 1) & 1) * 8 + 2 * w) * 64)[lane] = pl[0];
                        (base + kPart + (((s - 1) & 1) * 8 + 2 * w + 1) * 64)[lane] = pl[1];
                    }
                if constexpr (ii % 6 == 0) wait_a(std::integral_constant<int, MI + M>{});
                const bf16x8 a_cur = aq[(MI + M) % AD];
                if constexpr (g == 0) mfma_bf16_w<ag>(ar[cg], a_cur, Wh[cg][0][M]);
                if constexpr (g == 1) mfma_bf16_w<ag>(az[cg], a_cur, Wh[cg][1][M]);
                if constexpr (g == 2 && M == 0) mfma_bf16_w0<ag>(ahn[cg], a_cur, Wh[cg][2][0], bn[cg]);
                if constexpr (g == 2 && M > 0) mfma_bf16_w<ag>(ahn[cg], a_cur, Wh[cg][2][M]);
                if constexpr (ii % 6 == 5) fetch_a(std::integral_constant<int, MI + M + AD>{});
                if constexpr (ii == NREC - 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory");   // all other chains end here
            }
        };
        constexpr int kLead = HELEN_BF16_W4_LEAD, SPM = HELEN_BF16_W4_SPM(DEC);
        constexpr int NIT = (NS - kLead + SPM - 1) / SPM > NM ? (NS - kLead + SPM - 1) / SPM : NM;
        static_for<kLead>([&](auto K) __attribute__((always_inline)) { gate_slot(K); });
        static_for<NIT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_sched_barrier(0);
#ifndef HELEN_BIL_NOMFMA     // (timing probes: results are garbage)
#ifdef HELEN_BIL_TIMING
            if constexpr (i == 0) tm0 = __builtin_readcyclecounter();
#endif
            if constexpr (i < NM) mfma_item(I);
#ifdef HELEN_BIL_TIMING
            if constexpr (i == NM - 1) tk[3] += __builtin_readcyclecounter() - tm0;
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef HELEN_BIL_NOGATES
            static_for<SPM>([&](auto J) __attribute__((always_inline)) {
                gate_slot(std::integral_constant<int, kLead + i * SPM + decltype(J)::value>{});
            });
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
        ring_rd[x] = ring_rd[x] == (RD - 1u) * MI * 1024u ? 0u : ring_rd[x] + MI * 1024u;
#ifdef HELEN_BIL_NOGATES
        if constexpr (gates)
            static_for<NC>([&](auto C) {
                constexpr int c = decltype(C)::value;
                ((float*)(obase + hsel(ow)))[hoff[c >> 2] + 4 * (c & 3)] = gr[c >> 2][c & 3] + gz[c >> 2][c & 3] + gnn[c >> 2][c & 3] + ggn[c >> 2][c & 3];
            });
#endif
        // this phase's results become tile x's pending gate math
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            Pr[x][cg] = ar[cg];
            Pz[x][cg] = az[cg];
            Pn[x][cg] = ahn[cg];
            Pg[x][cg] = gnx[cg];
        }
        if (!DEC && has_prev) {                                  // h_x(s-1) as a bf16 plane = the layer output of slot s-1
            *(uint4*)(y_next[x] + in_block((unsigned)tid * 16u)) = ((const uint4*)(base + psel(cur)))[tid];
            y_next[x] += 512 * 16;
            issued += 1;
        }
        if (DEC && has_prev2) {                                  // slot s-2: partials parked by this tile's region of step s-1
            store_logits(x, s & 1, in_block((unsigned)tid * 4u));
            issued += 1;
            y_next[x] += 128 * 16;
        }
#ifdef HELEN_BIL_TIMING
        { long long now_ = __builtin_readcyclecounter(); tk[0] += now_ - tlast; tlast = now_; }
#endif
        if (issued == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (issued == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (issued == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef HELEN_BIL_TIMING
        { long long now_ = __builtin_readcyclecounter(); tk[1] += now_ - tlast; tlast = now_; }
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef HELEN_BIL_TIMING
        { long long now_ = __builtin_readcyclecounter(); tk[2] += now_ - tlast; tlast = now_; }
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // Regions in order: R(0,0) [no gates] | R(1,0) [G(0,0)] | R(0,1) [G(1,0)] | R(1,1) [G(0,1)] | ... | final G(1,T-1).
    // R(0,s): gates of tile 1 at step s-1, written to its buffer (s-1+1)&1 = s&1;  R(1,s): gates of tile 0 at step s,
    // written to buffer (s+1)&1.
    using No = std::false_type;
    using Yes = std::true_type;
    auto step = [&](auto STEADY, int s_) __attribute__((always_inline)) {
        if (s_ & 1) {
            region(I0{}, I1{}, I1{}, STEADY, s_, s_ - 1);
            region(I1{}, I1{}, I0{}, STEADY, s_, s_);
        } else {
            region(I0{}, I0{}, I0{}, STEADY, s_, s_ - 1);
            region(I1{}, I0{}, I1{}, STEADY, s_, s_);
        }
    };
    int s = 0;
    for (; s < T && s < 2; ++s) step(No{}, s);
    for (; s + 2 < T; s += 2) {                           // steady state: s >= 2 and (s + 1) + 1 < T
        region(I0{}, I0{}, I0{}, Yes{}, s, s - 1);
        region(I1{}, I0{}, I1{}, Yes{}, s, s);
        region(I0{}, I1{}, I1{}, Yes{}, s + 1, s);
        region(I1{}, I1{}, I0{}, Yes{}, s + 1, s + 1);
    }
    for (; s < T; ++s) step(No{}, s);
#ifdef HELEN_BIL_TIMING
    if (blockIdx.x == 0 && lane == 0)
        printf("bf16 w4 %s dir %d wave %d: cycles per region  stream %lld  waits %lld  barrier %lld  mfmas %lld\n", DEC ? "dec" : "enc", dir, w,
               tk[0] / (2 * T), tk[1] / (2 * T), tk[2] / (2 * T), tk[3] / (2 * T));
#endif
    // the gates of tile 1's last step (nothing left to interleave them with), into buffer T & 1
    {
        const int last = T & 1;
        f32x4* const obase = smem + kPerTile;
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const float hp[4] = {hprev[1][4 * cg], hprev[1][4 * cg + 1], hprev[1][4 * cg + 2], hprev[1][4 * cg + 3]};
            const f32x4 hn4 = gru_cell4(Pr[1][cg], Pz[1][cg], Pn[1][cg], Pg[1][cg], hp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ((float*)(obase + hsel(last)))[hoff[cg] + 4 * r] = hn4[r];
                ((unsigned short*)(obase + psel(last)))[poff[cg] + 8 * r] = bf16_bits(hn4[r]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int last = T & 1;   // buffers of h(T-1)
    if (DEC) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (T >= 2) {
                store_logits(x, (T - 2) & 1, (unsigned)tid * 4u);
                y_next[x] += 128 * 16;
            }
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const f32x4 a = (smem + x * kPerTile + hsel(last))[(2 * w + cg) * 64 + lane];
                f32x4 pl = splat4(0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e) pl = mfma4(a[e], Bh[cg][e], pl);
                (smem + x * kPerTile + kPart + (((T - 1) & 1) * 8 + 2 * w + cg) * 64)[lane] = pl;
            }
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; ++x) store_logits(x, (T - 1) & 1, (unsigned)tid * 4u);
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
            *(uint4*)(y_next[x] + (unsigned)tid * 16u) = ((const uint4*)(smem + x * kPerTile + psel(last)))[tid];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        hid_p[x][tid] = (smem + x * kPerTile + hsel(last))[tid];
        hid_p[x][256 + tid] = (smem + x * kPerTile + hsel(last))[256 + tid];
    }
}

}  // namespace helen
