// kernels_gru_pair.h -- fp32 GRU recurrence, two window tiles per workgroup at ONE wave per SIMD
#pragma once
#include <type_traits>

#include "kernels_gru.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Same arithmetic as gru_kernel (same MFMA order per accumulator, same gate cell: results are bit-identical),
// different schedule.  Measured on gfx950 (scripts/ubench/f32_mfma_valu_overlap.hip): v_mfma_f32_16x16x4_f32 and
// VALU instructions never overlap on a SIMD -- not from the same wave (+13 cycles for the first VALU after an
// MFMA, +4.6 per further one) and not from a co-resident wave (an MFMA-streaming wave starves its partner's
// VALU completely) -- so a recurrence step costs its 192 MFMAs (6144 cycles) PLUS its gate math whatever the
// occupancy, while LDS reads, LDS-DMA, stores and SALU between MFMAs are (nearly) free.  gru_kernel (two
// co-resident workgroups) loses the rest of its time to what the pairing cannot hide: eight waves fighting for
// four pipes by age, VALU address arithmetic stuck behind the partner's MFMA stream, barrier skew, and a tail
// in which the younger workgroup runs alone.
//
// Here ONE workgroup per CU (4 waves, one per SIMD, the whole register budget) walks TWO tiles of 16 windows,
// interleaved in software:   M(0,s) G(0,s) M(1,s) G(1,s) M(0,s+1) ...    M = 192 MFMAs, G = gate math, and
// everything that is not gate math rides inside an M phase:
//   - all six W_hh column tiles of a wave live in AccVGPRs and feed the MFMAs directly as B operands (inline
//     asm: hipcc itself keeps MFMA sources in VGPRs and would shuttle the accumulators through AccVGPRs with
//     ~70 VALU moves per step); accumulators, gi fragments and gate math stay in ordinary VGPRs;
//   - the h of tile x written in G(x,s) is first needed by M(x,s+1), a whole half-step later: the one barrier
//     per half-step sits after the first MFMA group of the OTHER tile's M phase, and the LDS reads that depend
//     on it (layer-output copy, head partials, first A fragment of the next M phase) follow inside that phase;
//   - the gi fragments of (tile o, next step) are DMA'd (global_load_lds) from inside M(x,.) right after tile
//     o's slot was read, i.e. a full half-step ahead of their use; they are the last vector-memory operations
//     of a half-step, so `vmcnt(6)` at the point of use is exact;
//   - addresses are SGPR bases + constant per-lane offsets: no VALU besides the gate math in the loop.
// grid (ceil(tiles / 2), 2 directions).  An odd tile count makes the last workgroup do its one tile twice
// (identical values are written twice).
// ------------------------------------------------------------------------------------------------
constexpr int kPairHF4 = 2 * 2 * 512;        // h[tile x][buffer][512]
constexpr int kPairGF4 = 2 * 4 * 384;        // gi[tile x][wave][6][64]
constexpr int kPairPF4 = 2 * 2 * 4 * 64;     // head partials [tile x][parity][wave][64]

// D = A x B (+ C) with B in an AccVGPR, A / C / D in VGPRs.  `volatile`: program order is the schedule.
__device__ __forceinline__ void mfma_ab(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_ab_zero(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_ab_init(f32x4& c, float a, float b, const f32x4& c0) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(c) : "v"(a), "a"(b), "v"(c0));
}

// 16-byte store to uniform base + 32-bit lane offset (hipcc adds such a base on the VALU, 64 bits wide)
__device__ __forceinline__ void store_sv(char* base, unsigned voff, const f32x4& v) {
#ifdef HELEN_PAIR_NOSTORE   // timing probe
    return;
#endif
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(base) : "memory");
}

template <bool DEC>
__global__ __launch_bounds__(256, 1) void gru_pair_kernel(const f32x4* __restrict__ gi, long gi_tile_stride,
                                                          int slot0_fwd, int slot0_bwd, int T,
                                                          const f32x4* __restrict__ Whp,
                                                          const float* __restrict__ bhn,
                                                          f32x4* __restrict__ hid, f32x4* __restrict__ y,
                                                          long y_tile_stride, const f32x4* __restrict__ Whd,
                                                          f32x4* __restrict__ plogit, long pl_tile_stride,
                                                          int ntiles) {
    __shared__ f32x4 smem[kPairHF4 + kPairGF4 + (DEC ? kPairPF4 : 0)];   // 80 (+16) KiB
    f32x4* const hbuf = smem;
    f32x4* const gbuf = smem + kPairHF4;
    f32x4* const part = smem + kPairHF4 + kPairGF4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int q = lane >> 4;
    const int dir = blockIdx.y;
    const int slot0 = dir ? slot0_bwd : slot0_fwd;
    const int tile_of[2] = {min(2 * (int)blockIdx.x, ntiles - 1), min(2 * (int)blockIdx.x + 1, ntiles - 1)};

    // W_hh slice: W[n = gate*2 + half][m] holds k = 16m + 4q + e, col = unit(half, j)   (pack_w_hh)
    f32x4 W[6][8];
    {
        const f32x4* wp = Whp + (size_t)((dir * 4 + w) * 48) * 64 + lane;
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int m = 0; m < 8; ++m) W[n][m] = wp[(n * 8 + m) * 64];
    }
    f32x4 Bh[2] = {splat4(0.f), splat4(0.f)};   // DEC: head weights of k = dir*128 + 32w + 16g + 4q + e, class j
    if (DEC) {
        Bh[0] = Whd[(dir * 8 + 2 * w) * 64 + lane];
        Bh[1] = Whd[(dir * 8 + 2 * w + 1) * 64 + lane];
    }
    f32x4 bnv[2];   // b_hn of this lane's two units: the initial value of the n-gate accumulators
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) bnv[hh] = splat4(bhn[dir * kH + 32 * w + 16 * hh + j]);

    // Uniform (SGPR) running byte pointers per tile -- next gi slot to fetch, next layer-output / partial-logit
    // slot to store -- advanced by SALU adds; everything per-lane is a constant 32-bit byte offset.
    constexpr long kPosBytes = 2 * kNTile * 64 * 16;  // one slot of gi (both directions)
    const char* gi_next[2];
    char* y_next[2];
    char* pl_next[2];
    char* hid_s[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        gi_next[x] = (const char*)(gi + (size_t)tile_of[x] * gi_tile_stride + (size_t)dir * (kNTile * 64) +
                                   (2 * w) * 64) + (size_t)slot0 * kPosBytes;
        y_next[x] = (char*)(y + (size_t)tile_of[x] * y_tile_stride + (size_t)dir * (kHidDirStride / 4));
        pl_next[x] = (char*)(plogit + (size_t)tile_of[x] * pl_tile_stride + (size_t)dir * 64);
        hid_s[x] = (char*)(hid + ((size_t)tile_of[x] * 2 + dir) * (kHidDirStride / 4));
    }
    const unsigned lane16 = (unsigned)lane * 16u, tid16 = (unsigned)tid * 16u, tid16b = tid16 + 4096u;
    // LDS byte address of this wave's tile-0 gi slot
    const unsigned gbuf_bytes = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)smem +
                                (unsigned)(kPairHF4 + w * 384) * 16u;
    // Gate pair g (fragments 2g, 2g+1) of this wave's six gi fragments of tile x's next slot -> its LDS slot:
    // two 1 KiB rows.  Inline asm for the SGPR-base + lane-offset address form (the builtin adds the base on the
    // VALU); the instruction offset moves the global and the LDS address together.
    auto dma_gi = [&](int x, int g) {
#ifdef HELEN_PAIR_NODMA   // timing probe: no gi traffic (results are garbage)
        return;
#endif
        const unsigned m0v = gbuf_bytes + (unsigned)(x * 4 * 384 * 16 + g * 2048);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024"
                     :: "s"(m0v), "v"(lane16), "s"(gi_next[x] + g * 8192) : "memory", "m0");
    };

    // initial hidden state of both tiles -> buffer 0; tile 0's first gi slot (tile 1's is fetched by H(0,0))
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        hbuf[x * 1024 + tid] = *(const f32x4*)(hid_s[x] + tid16);
        hbuf[x * 1024 + tid + 256] = *(const f32x4*)(hid_s[x] + (tid16 + 4096u));
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) dma_gi(0, g);
    gi_next[0] += kPosBytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float hprev[2][2][4];
    int hoff[2];  // float offset of (row 4q, unit) inside an h buffer; rows r add 4r
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int u = 32 * w + 16 * hh + j;
        hoff[hh] = ((u >> 2) * kTile + 4 * q) * 4 + (u & 3);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) hprev[x][hh][r] = ((const float*)(hbuf + x * 1024))[hoff[hh] + 4 * r];
    }
    f32x4 a_pref = hbuf[lane];   // group 0 of h_0(-1): the A fragment the first MFMA phase starts with
    f32x4 hd_pref[2] = {splat4(0.f), splat4(0.f)};   // DEC: this wave's two head k-groups of the next tile's h

#ifdef HELEN_PAIR_TIMING   // developer probe: wall cycles per segment of a half-step (s_memtime drains lgkmcnt: perturbs)
    long long tk[5] = {0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#define HELEN_PAIR_TICK(i) { __builtin_amdgcn_sched_barrier(0); long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define HELEN_PAIR_TICK(i)
#endif
    // One half-step: MFMA phase and gate math of tile X at step s (CUR = s & 1 at compile time so that every LDS
    // address is lane*16 + immediate).  `so` = newest step of the OTHER tile o, whose h(so) the barrier publishes.
    auto half_step = [&](auto X, auto CUR, int s) {
        constexpr int x = decltype(X)::value, o = 1 - x, cur = decltype(CUR)::value;
        constexpr int ocur = x ? (cur ^ 1) : cur;                // buffer of h_o(so): (so + 1) & 1
        const int so = x ? s : s - 1;
        const f32x4* hb = hbuf + (x * 2 + cur) * 512 + lane;     // h_x(s-1): A operand of this phase
        const f32x4* ho = hbuf + (o * 2 + ocur) * 512;           // h_o(so)
        const f32x4* gb = gbuf + (x * 4 + w) * 384 + lane;
                // Uniform conditions are spelled out at every use (a bool carried across the phase ends up in a VGPR):
        //   store_y: !DEC and so >= 0  -- h_o(so) is a layer output
        //   have_o:  DEC and so >= 1   -- tile o's partials of slot so-1 exist; wave (so-1) & 3 adds them up
#define HELEN_STORE_Y (!DEC && (x == 1 || s > 0))
#define HELEN_HAVE_O (DEC && (x ? s > 0 : s > 1))
#define HELEN_SUM_O (HELEN_HAVE_O && ((w - so + 1) & 3) == 0)
        f32x4 acc[6], a[2], G[6], yv[2], hp[2], pp[4];
        a[0] = a_pref;
        HELEN_PAIR_TICK(4)
#define HELEN_PAIR_GROUP(m)                                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int n = 0; n < 6; ++n) { \
        if ((m) == 0 && e == 0 && n < 4)                                                           \
            mfma_ab_zero(acc[n], a[(m) & 1][e], W[n][m][e]);                                       \
        else if ((m) == 0 && e == 0)                                                               \
            mfma_ab_init(acc[n], a[(m) & 1][e], W[n][m][e], bnv[n - 4]);                           \
        else                                                                                       \
            mfma_ab(acc[n], a[(m) & 1][e], W[n][m][e]);                                            \
    }
        // group 0, then the barrier: every wave has written h_o(so) (and tile o's head partials) long ago
        a[1] = hb[1 * 64];
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(0)
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_TICK(0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        HELEN_PAIR_TICK(1)
        a[0] = hb[2 * 64];
        if (DEC && s > 0) {   // h_x(s-1) -> partial logits of slot s-1: 8 head MFMAs on the two k-groups fetched during the previous half-step
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e == 0) {
                    mfma_ab_zero(hp[0], hd_pref[0][e], Bh[0][e]);
                    mfma_ab_zero(hp[1], hd_pref[1][e], Bh[1][e]);
                } else {
                    mfma_ab(hp[0], hd_pref[0][e], Bh[0][e]);
                    mfma_ab(hp[1], hd_pref[1][e], Bh[1][e]);
                }
            }
        }
        if (HELEN_STORE_Y) yv[0] = ho[tid];
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(1)
        __builtin_amdgcn_sched_barrier(0);
        a[1] = hb[3 * 64];
        if (HELEN_STORE_Y) yv[1] = ho[tid + 256];
        if (HELEN_SUM_O) {
            const f32x4* ps = part + (o * 2 + ((so - 1) & 1)) * 256 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) pp[k] = ps[k * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(2)
        __builtin_amdgcn_sched_barrier(0);
        a[0] = hb[4 * 64];
        if (HELEN_STORE_Y) store_sv(y_next[o], tid16, yv[0]);
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(3)
        __builtin_amdgcn_sched_barrier(0);
        a[1] = hb[5 * 64];
        if (HELEN_STORE_Y) {
            store_sv(y_next[o], tid16b, yv[1]);
            y_next[o] += kYStride * 4;
        }
        if (so + 1 < T) dma_gi(o, 0);   // tile o still has a step so+1 to feed
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(4)
        __builtin_amdgcn_sched_barrier(0);
        a[0] = hb[6 * 64];
        if (so + 1 < T) dma_gi(o, 1);
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(5)
        __builtin_amdgcn_sched_barrier(0);
        a[1] = hb[7 * 64];
        if (so + 1 < T) {
            dma_gi(o, 2);
            gi_next[o] += kPosBytes;
        }
        // gi(x, s): its DMA is older than the six just issued for tile o, which are the newest VMEM operations
        if (so + 1 < T)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int n = 0; n < 6; ++n) G[n] = gb[n * 64];
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(6)
        __builtin_amdgcn_sched_barrier(0);
        a_pref = ho[lane];                                       // next phase: M(o, so+1) starts on h_o(so)
        if (DEC) {
            hd_pref[0] = ho[(2 * w) * 64 + lane];
            hd_pref[1] = ho[(2 * w + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        HELEN_PAIR_GROUP(7)
        HELEN_PAIR_TICK(2)
#undef HELEN_PAIR_GROUP
        // MFMA results -> VALU: hipcc pads no hazards around inline asm (8-pass MFMA: well over 11 wait states)
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float* hw = (float*)(hbuf + (x * 2 + (cur ^ 1)) * 512);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#ifdef HELEN_PAIR_NOGATES   // timing probe
            const f32x4 hn = acc[hh] + acc[2 + hh] + acc[4 + hh] + G[hh] + G[2 + hh] + G[4 + hh];
#else
            const f32x4 hn = gru_cell4(acc[hh], acc[2 + hh], acc[4 + hh], G[hh], G[2 + hh], G[4 + hh], hprev[x][hh]);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hprev[x][hh][r] = hn[r];
                hw[hoff[hh] + 4 * r] = hn[r];
            }
        }
        if (DEC && s > 0) (part + ((x * 2 + ((s - 1) & 1)) * 4 + w) * 64)[lane] = hp[0] + hp[1];
        if (HELEN_SUM_O)   // one wave adds the four k-slices in wave order
            store_sv(pl_next[o], lane16, ((pp[0] + pp[1]) + pp[2]) + pp[3]);
        if (HELEN_HAVE_O) pl_next[o] += 128 * 16;
        HELEN_PAIR_TICK(3)
#undef HELEN_STORE_Y
#undef HELEN_HAVE_O
#undef HELEN_SUM_O
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int s = 0;
    for (; s + 1 < T; s += 2) {
        half_step(I0{}, I0{}, s);
        half_step(I1{}, I0{}, s);
        half_step(I0{}, I1{}, s + 1);
        half_step(I1{}, I1{}, s + 1);
    }
    if (s < T) {   // odd T: its last step runs on buffer parity 0
        half_step(I0{}, I0{}, s);
        half_step(I1{}, I0{}, s);
    }
#ifdef HELEN_PAIR_TIMING
    if (blockIdx.x == 0 && lane == 0)
        printf("pair dir %d wave %d: cycles per half-step  group0 %lld  barrier %lld  groups1-7 %lld  gates %lld  loop %lld\n",
               dir, w, tk[0] / (2 * T), tk[1] / (2 * T), tk[2] / (2 * T), tk[3] / (2 * T), tk[4] / (2 * T));
#endif
    __syncthreads();
    const int last = T & 1;   // buffer of h(T-1)
    if (DEC) {
        // H(x, s) turns h_x(s-1) into partials and adds up tile o's slot so-1: after the loop tile 1's slot T-2
        // is still to be added up, and slot T-1 of both tiles has no partials yet.
        if (T >= 2 && w == ((T - 2) & 3)) {
            const f32x4* ps = part + (2 + ((T - 2) & 1)) * 256 + lane;
            *(f32x4*)(pl_next[1] + lane16) = ((ps[0] + ps[64]) + ps[128]) + ps[192];
        }
        if (T >= 2) pl_next[1] += 128 * 16;   // both tiles' pointers are at slot T-1 now
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const f32x4* hs = hbuf + (x * 2 + last) * 512 + lane;
            const f32x4 a0 = hs[(2 * w) * 64], a1 = hs[(2 * w + 1) * 64];
            f32x4 p0, p1;
            mfma_ab_zero(p0, a0[0], Bh[0][0]);
            mfma_ab_zero(p1, a1[0], Bh[1][0]);
#pragma unroll
            for (int e = 1; e < 4; ++e) {
                mfma_ab(p0, a0[e], Bh[0][e]);
                mfma_ab(p1, a1[e], Bh[1][e]);
            }
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            (part + ((x * 2 + ((T - 1) & 1)) * 4 + w) * 64)[lane] = p0 + p1;
        }
        __syncthreads();
        if (w == ((T - 1) & 3)) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f32x4* ps = part + (x * 2 + ((T - 1) & 1)) * 256 + lane;
                *(f32x4*)(pl_next[x] + lane16) = ((ps[0] + ps[64]) + ps[128]) + ps[192];
            }
        }
    } else {                  // tile 1's last layer output (tile 0's went out in the last half-step)
        const f32x4* h1 = hbuf + (2 + last) * 512;
        *(f32x4*)(y_next[1] + tid16) = h1[tid];
        *(f32x4*)(y_next[1] + (tid16 + 4096u)) = h1[tid + 256];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const f32x4* hl = hbuf + (x * 2 + last) * 512;
        *(f32x4*)(hid_s[x] + tid16) = hl[tid];
        *(f32x4*)(hid_s[x] + (tid16 + 4096u)) = hl[tid + 256];
    }
}

}  // namespace helen
