// kernels_fused_bf16_pair.h -- HELEN_PRECISION_BF16: fused projection + recurrence, two window tiles per workgroup
#pragma once
#include <type_traits>

#include "kernels_fused_bf16.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Same arithmetic as gru_fused_bf16_kernel (same MFMA order per accumulator, same gate cell, same order of the head
// partial sums: results are bit-identical), scheduled like gru_pair_kernel.  With bf16 operands a tile-step is
// only 21 (encoder) / 36 (decoder) bf16 MFMAs per wave -- 700 / 1200 cycles of a SIMD's matrix pipe -- plus the
// same ~620 cycles of gate math as in fp32, so what one tile per workgroup leaves exposed per step (a barrier,
// an LDS round trip for the new h, the tail of the MFMA pipe before the gates) costs as much as the work itself:
// gru_fused_bf16_kernel measures 2070 / 3140 cycles per tile-step (its 156 / 226 registers allow one workgroup per
// CU, so nothing else runs meanwhile).
//
// Here ONE workgroup of 8 waves walks TWO tiles of 16 windows, interleaved in software:
//     M(0,s) | G(0,s)   M(1,s) | G(1,s)   M(0,s+1) | ...      M = MFMA phase, G = gate math, | = the barrier
//   M(x,s): the recurrent part of step s on the bf16 plane of h_x(s-1), added onto tile x's input part; this wave's
//           k-slice of the head product of h_x(s-1) (decoder); then the input part x . W_ih^T + b of the OTHER
//           tile's next step from its LDS ring (independent of any h) -- last, so that the gates behind the
//           barrier find their accumulators finished and the next phase starts from registers;
//   the barrier publishes the OTHER tile's h (written in the previous half-step's G) and the input rows the
//           previous half-step's DMA brought in (see the counted vmcnt below);
//   G(x,s): gates, new h -> LDS (fp32 + bf16 plane), head partials -> LDS.
// The weights (W_hh 48 + W_ih 36 / 96 registers; the decoder keeps the last K32 group of W_ih in LDS) are shared by
// both tiles; each tile has its own h buffers, input ring (3 deep, LDS-DMA two steps ahead) and partial-logit
// slots: 33 / 64 KiB of LDS per tile.  Measured: 1830 / 2750 cycles per tile-step.
// Input rows arrive by LDS-DMA issued at the start of M(x,s) for step s+2; the issuing wave waits for them with a
// COUNTED vmcnt at the next half-step's barrier (everything older than that half-step's own operations), so a
// row has a half-step plus an MFMA phase (~1.3 us) to arrive, and it is first read after that barrier.
// grid (ceil(tiles / 2), 2 directions).  An odd tile count makes the last workgroup do its one tile twice.
// ------------------------------------------------------------------------------------------------
// (Round 3 also measured this kernel with waves 4-7 skewed by a phase -- barrier behind the gates instead of in front --
// so that each SIMD always had one wave in its gate math and one in its MFMAs: encoder 0.155 for 0.157 ms, decoder 0.249
// for 0.241; the variant left the tree in round 4.)
template <int MI, bool DEC>
__global__ __launch_bounds__(512, 1) void gru_fused_bf16_pair_kernel(
    const f32x4* __restrict__ in, long in_tile_stride, int pos0, int T, const bf16x8* __restrict__ Wi3,
    const bf16x8* __restrict__ Wh3, const float* __restrict__ bias, const float* __restrict__ bhn,
    f32x4* __restrict__ hid, f32x4* __restrict__ yplane_out, long yp_tile_stride,
    const f32x4* __restrict__ Whd, f32x4* __restrict__ plogit, long pl_tile_stride, int ntiles) {
    // LDS per tile: fp32 h [2][512 f4] | bf16 h plane [2][256] | input ring [3][MI * 64] | (DEC) head partials [2][8][64]
    // (DEC) after both tiles: the last K32 group of W_ih of every wave [8][3][64] -- 96 registers of W_ih beside 48 of
    // W_hh leave too few for two tiles' state; 12 of them live in LDS and are read once per half-step
    constexpr int kRing = 1024 + 512, kPart = kRing + 3 * MI * 64, kPerTile = kPart + (DEC ? 2 * 8 * 64 : 0);
    constexpr int kParked = DEC ? 1 : 0, MR = MI - kParked;
    __shared__ f32x4 smem[2 * kPerTile + kParked * 8 * 3 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int u = 16 * v + j;
    const int tile_of[2] = {min(2 * (int)blockIdx.x, ntiles - 1), min(2 * (int)blockIdx.x + 1, ntiles - 1)};

    bf16x8 Wh[3][4], Wi[3][MR];
    bf16x8* const wpark = (bf16x8*)(smem + 2 * kPerTile) + v * 3 * 64 + lane;
    {
        const bf16x8* wh = Wh3 + (size_t)((dir * 8 + v) * 36) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int M = 0; M < 4; ++M) Wh[g][M] = wh[((g * 4 + M) * 3) * 64];
            const bf16x8* wi = Wi3 + (size_t)((dir * kNTile + g * 8 + v) * MI) * 3 * 64 + lane;
#pragma unroll
            for (int M = 0; M < MR; ++M) Wi[g][M] = wi[(M * 3) * 64];
            if (kParked) wpark[g * 64] = wi[(MR * 3) * 64];
        }
    }
    HeadW Bh = head_split_w(splat4(0.f));   // DEC: head weights for k = dir*128 + 16v + 4q + e, class j, as two bf16 terms
    if (DEC) Bh = head_split_w(Whd[(dir * 8 + v) * 64 + lane]);
    float bi[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bi[g] = bias[dir * kG + g * kH + u];
    const float bn = bhn[dir * kH + u];

    // Input rows: wave v < MI brings row v of a step.  Running uniform byte pointers per tile, advanced by a signed
    // stride: the encoder walks positions pos0 + s (dir 0) or pos0 + T-1-s (dir 1); the decoder's row v belongs to
    // half p = v >> 2 of the encoder output, stored in slot s if p == dir and in slot T-1-s otherwise.
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
    unsigned ring_dma[2], ring_rd[2];   // LDS byte offsets (within a tile's ring) of the slot to fill / to read next
    auto dma_in = [&](int x) __attribute__((always_inline)) {
        if (v < MI) dma_row_to_lds(lds0 + (unsigned)((x * kPerTile + kRing) * 16) + ring_dma[x] + (unsigned)v * 1024u,
                                   in_next[x], in_block(lane16));
        in_next[x] += in_step;
        ring_dma[x] = ring_dma[x] == 2u * MI * 1024u ? 0u : ring_dma[x] + MI * 1024u;
    };
    // x . W_ih^T + bias of the step whose rows sit in the ring slot at ring_rd[x]
    auto input_part = [&](int x, f32x4* acc) __attribute__((always_inline)) {
        const bf16x8* L = (const bf16x8*)((const char*)(smem + x * kPerTile + kRing) + ring_rd[x]) + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = splat4(bi[g]);
#pragma unroll
        for (int M = 0; M < MI; ++M) {
            const bf16x8 a = L[M * 64];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, M < MR ? Wi[g][M < MR ? M : 0] : wpark[g * 64], acc[g],
                                                                 0, 0, 0);
        }
        ring_rd[x] = ring_rd[x] == 2u * MI * 1024u ? 0u : ring_rd[x] + MI * 1024u;
    };
    // this lane's 4 values: rows 4q + r of unit u (see gru_x3_kernel)
    const int hoff = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
    const int poff = ((u >> 3) * kTile + 4 * q) * 8 + (u & 7);

    f32x4* hid_p[2];
    char* y_next[2];    // next slot of the bf16 output plane (encoder) / of the partial logits (decoder)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        hid_p[x] = hid + ((size_t)tile_of[x] * 2 + dir) * (kHidDirStride / 4);
        y_next[x] = DEC ? (char*)(plogit + (size_t)tile_of[x] * pl_tile_stride + (size_t)dir * 64)
                        : (char*)(yplane_out + (size_t)tile_of[x] * yp_tile_stride + (size_t)dir * 256);
    }
    // The eight k-slices' partial logits of (tile x, parity pb), added in wave order like gru_fused_bf16_kernel --
    // one float per thread of waves 0..3 (a whole f32x4 per lane of one wave costs 32 registers in flight, which
    // this kernel does not have), stored as one 1 KiB row.
    auto store_logits = [&](int x, int pb, unsigned voff) __attribute__((always_inline)) {
        const float* pp = (const float*)(smem + x * kPerTile + kPart + pb * 8 * 64) + tid;
        float sum = pp[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += pp[k * 256];
        *(float*)(y_next[x] + voff) = sum;
    };

    // ---- prologue: initial h, the rows of steps 0 and 1, the input part of step 0 -- for both tiles
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        smem[x * kPerTile + tid] = hid_p[x][tid];
        ring_dma[x] = 0;
        ring_rd[x] = 0;
        dma_in(x);
        if (T > 1) dma_in(x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float hprev[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[x][r] = ((const float*)(smem + x * kPerTile))[hoff + 4 * r];
            ((unsigned short*)(smem + x * kPerTile + 1024))[poff + 8 * r] = bf16_bits(hprev[x][r]);
        }
    f32x4 gin[2][3];
    input_part(0, gin[0]);          // (tile 1's first input part is computed in M(0,0))
    __syncthreads();
    bf16x8 a_pref = ((const bf16x8*)(smem + 1024))[lane];   // group 0 of tile 0's h plane

#ifdef HELEN_BP_TIMING   // developer probe: where a wave's cycles go (scripts/dev/run_bp2.sh)
    long long tk[4] = {0, 0, 0, 0};
#define HELEN_BP_TICK(i) { __builtin_amdgcn_sched_barrier(0); long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
    long long tlast = __builtin_readcyclecounter();
#else
#define HELEN_BP_TICK(i)
#endif
    // One half-step of tile X at step s (CUR = s & 1).  STEADY: 2 <= s and s + 2 < T are compile-time facts.
    auto half_step = [&](auto X, auto CUR, auto STEADY, int s) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value;
        constexpr bool steady = decltype(STEADY)::value;
        constexpr int ocur = x ? (cur ^ 1) : cur;                // buffer of h_o(so): (so + 1) & 1
        const bool has_prev = steady || s > 0;
        const bool has_prev2 = steady || s > 1;
        const bool has_next = steady || s + 1 < T;
        const bool has_next2 = steady || s + 2 < T;
        f32x4* const base = smem + x * kPerTile;
        const f32x4* hx = base + cur * 512;                      // fp32 h_x(s-1)
        const bf16x8* pa = (const bf16x8*)(base + 1024 + cur * 256) + lane;
        int issued = 0;                                          // vector-memory operations of this half-step
        if (has_next2) {
            dma_in(x);
            issued += v < MI;
        }
        // recurrent part on top of the input part; the n gate's two halves stay apart
        f32x4 ar = gin[x][0], az = gin[x][1], ahn = splat4(bn), pl = splat4(0.f);
        const f32x4 gn = gin[x][2];
#pragma unroll
        for (int M = 0; M < 4; ++M) {
            const bf16x8 a = M ? pa[M * 64] : a_pref;
            ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[0][M], ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[1][M], az, 0, 0, 0);
            ahn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Wh[2][M], ahn, 0, 0, 0);
        }
        if (DEC && has_prev) {                                   // this wave's k-slice of the head product of h_x(s-1)
            pl = head_mfma(head_split_h(hx[v * 64 + lane]), Bh);
        }
        // The input part of the OTHER tile's next step (independent of any h; gin[o] was consumed by that tile's
        // gates in the previous half-step): last in this phase, so that the gates of tile x behind the barrier find
        // their accumulators finished, and the next phase -- recurrent part of tile o -- has everything in registers.
#ifndef HELEN_BP_NOINPUT
        if (x == 0 || has_next) input_part(o, gin[o]);
#endif
        if (!DEC && has_prev) {                                  // h_x(s-1) as a bf16 plane = the layer output of slot s-1
            *(uint2*)(y_next[x] + in_block((unsigned)tid * 8u)) = ((const uint2*)(base + 1024 + cur * 256))[tid];
            y_next[x] += 512 * 16;
            issued += 1;
        }
        if (DEC && has_prev2) {                                  // slot s-2: partials parked in G(x,s-1)
            if (v < 4) {
                store_logits(x, s & 1, in_block((unsigned)tid * 4u));
                issued += 1;
            }
            y_next[x] += 128 * 16;
        }
        HELEN_BP_TICK(0)
        auto publish = [&]() __attribute__((always_inline)) {
            // everything this wave issued before this half-step has landed (its input row of the other tile among it)
            if (issued == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (issued == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            HELEN_BP_TICK(1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            HELEN_BP_TICK(2)
            a_pref = ((const bf16x8*)(smem + o * kPerTile + 1024 + ocur * 256))[lane];   // next phase starts on h_o
            __builtin_amdgcn_sched_barrier(0);
        };
        publish();
#ifdef HELEN_BP_NOGATES   // timing probes: results are garbage
        const f32x4 hn4 = ar + az + ahn + gn;
#else
        const f32x4 hn4 = gru_cell4_pre(ar, az, ahn, gn, hprev[x]);       // (weights and biases prescaled: kernels_gru.h)
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hprev[x][r] = hn4[r];
            ((float*)(base + (cur ^ 1) * 512))[hoff + 4 * r] = hn4[r];
#ifndef HELEN_BP_NOPLANE
            ((unsigned short*)(base + 1024 + (cur ^ 1) * 256))[poff + 8 * r] = bf16_bits(hn4[r]);
#endif
        }
        if (DEC && has_prev) (base + kPart + (((s - 1) & 1) * 8 + v) * 64)[lane] = pl;
        __builtin_amdgcn_sched_barrier(0);
        HELEN_BP_TICK(3)
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using No = std::false_type;
    using Yes = std::true_type;
    auto step = [&](int s_) __attribute__((always_inline)) {
        if (s_ & 1) {
            half_step(I0{}, I1{}, No{}, s_);
            half_step(I1{}, I1{}, No{}, s_);
        } else {
            half_step(I0{}, I0{}, No{}, s_);
            half_step(I1{}, I0{}, No{}, s_);
        }
    };
    int s = 0;
    for (; s < T && s < 2; ++s) step(s);
    for (; s + 3 < T; s += 2) {                           // steady state: s >= 2 and (s + 1) + 2 < T
        half_step(I0{}, I0{}, Yes{}, s);
        half_step(I1{}, I0{}, Yes{}, s);
        half_step(I0{}, I1{}, Yes{}, s + 1);
        half_step(I1{}, I1{}, Yes{}, s + 1);
    }
    for (; s < T; ++s) step(s);
#ifdef HELEN_BP_TIMING
    if (blockIdx.x == 0 && lane == 0)
        printf("bf16 pair %s dir %d wave %d: cycles per half-step  mfma phase %lld  vmcnt %lld  barrier %lld  gates %lld\n",
               DEC ? "dec" : "enc", dir, v, tk[0] / (2 * T), tk[1] / (2 * T), tk[2] / (2 * T), tk[3] / (2 * T));
#endif
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
            const f32x4 a = (smem + x * kPerTile + last * 512)[v * 64 + lane];
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
            *(uint2*)(y_next[x] + (unsigned)tid * 8u) = ((const uint2*)(smem + x * kPerTile + 1024 + last * 256))[tid];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) hid_p[x][tid] = (smem + x * kPerTile + last * 512)[tid];
}

}  // namespace helen
