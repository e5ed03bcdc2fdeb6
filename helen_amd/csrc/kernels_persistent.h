// kernels_persistent.h -- the 19-chunk loop of one polish call as ONE launch (fp32, large calls)
#pragma once
#include "kernels_gemm.h"
#include "kernels_gru_pair.h"
#include "kernels_heads.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// polish_persistent_kernel: everything behind the encoder projection -- reference loop body
// models/predict_gpu.py:114-149 around TransducerModel.py:60-79 -- for all 19 chunks in one launch:
//     per chunk:  encoder recurrence | decoder projection | decoder recurrence (+ head partials) | heads
// A workgroup is a (tile pair, direction) for the whole call and runs the SAME device bodies the per-phase kernels
// run (gru_pair_body, gemm_dec_ws_body, heads_body: same MFMA order, same bits), so the per-phase launch sequence
// and this kernel are interchangeable (tests/test_gpu_scale.py::test_every_fp32_kernel_choice_gives_the_same_bits).
//
// What has to cross workgroups inside a chunk is exactly two hand-offs between the two direction-workgroups of a pair:
//     A  encoder output y1 (both directions' halves feed each direction's decoder projection),
//     B  the decoder's partial logits (both directions' halves of the 16 logits feed the softmax);
// the hidden state never leaves its workgroup (encoder h0 = previous chunk's decoder h_n, decoder h0 = encoder h_n,
// all per direction), gi_dec is written and read by the same workgroup.  No grid barrier, no global phase: pairs
// drift apart freely.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): plain stores -> __syncthreads -> one lane: agent-scope
// release fence, s_waitcnt vmcnt(0), relaxed agent-scope store of the progress counter; the consumer's lane polls
// the partner's counter with relaxed agent-scope loads (s_sleep between polls), then ONE agent-scope acquire fence,
// __syncthreads, plain loads.  Counters only grow (epoch base passed per launch: nothing to reset between calls).
//
// No co-residency assumption: a workgroup takes its (pair, direction) from a ticket counter in the order workgroups
// START, partners hold adjacent tickets, so at any time at most one started workgroup waits for a partner that has
// not started, and that partner starts as soon as any other workgroup (which depends on nobody else) retires.  A
// wait that still does not end within ~4 s (a wedged device) sets `*error` and gives up: the host reports it on the
// next call instead of hanging the queue.
// ------------------------------------------------------------------------------------------------
struct PolishPersistentArgs {
    const f32x4* gi_enc;
    long gi_enc_tile_stride;
    const f32x4* whp_enc;
    const float* bhn_enc;
    f32x4* hid;
    f32x4* y1;
    long y_tile_stride;
    const f32x4* wp_dec;
    const float* bias_dec;
    f32x4* gi_dec;
    long gi_dec_tile_stride;
    const f32x4* whp_dec;
    const float* bhn_dec;
    const f32x4* whd;
    f32x4* plogit;
    long pl_tile_stride;
    const float* bhd;
    f32x4* pending;
    uint8_t* bases;
    uint8_t* rles;
    float* acc_base;
    float* acc_rle;
    int n_windows;
    int ntiles;
    unsigned* ticket;        // device: workgroups of this launch draw ticket_base, ticket_base + 1, ...
    unsigned ticket_base;
    unsigned* progress;      // device: [pairs][2] hand-offs completed, ever
    unsigned epoch_base;     // ... before this launch
    unsigned* error;         // host-visible: non-zero when a wait gave up
};

constexpr unsigned kPersistentSpinLimit = 1u << 24;   // polls of ~0.25 us

// Hand-off with the partner workgroup: publish everything this workgroup has stored, wait for the partner's `epoch`.
__device__ __forceinline__ void pair_handoff(unsigned* mine, const unsigned* theirs, unsigned epoch, unsigned* error) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler may drop the fence's own wait: see the guide)
        __hip_atomic_store(mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        // (signed distance: the counters wrap after 2^32 hand-offs)
        while ((int)(__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kPersistentSpinLimit) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// Global data written and then read by THIS workgroup (gi_dec, hid): a workgroup-scope release / acquire.  All waves of
// a workgroup share their CU's vector L1, which is write-through and coherent for its own CU's stores, so nothing has
// to be invalidated (the agent-scope form, ~3 us a piece three times per chunk, measured 0.2 ms per call for nothing):
// every wave's stores have left, then the barrier.
__device__ __forceinline__ void own_data_fence() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

constexpr int kPersistentLdsF4 = kDecWsLdsF4 > kPairHF4 + kPairPF4 ? kDecWsLdsF4 : kPairHF4 + kPairPF4;

__global__ __launch_bounds__(512, 1) void polish_persistent_kernel(const PolishPersistentArgs a) {
    __shared__ f32x4 smem[kPersistentLdsF4];   // 128 KiB: the decoder projection's ring is the largest user
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = atomicAdd(a.ticket, 1u) - a.ticket_base;
    __syncthreads();
    const unsigned ticket = s_ticket;
    const int pair = (int)(ticket >> 1), dir = (int)(ticket & 1u);
    const int npairs = (a.ntiles + 1) / 2;
    if (pair >= npairs) return;
    const int tile0 = min(2 * pair, a.ntiles - 1), tile1 = min(2 * pair + 1, a.ntiles - 1);
    unsigned* const mine = a.progress + (pair * 2 + dir);
    const unsigned* const theirs = a.progress + (pair * 2 + (dir ^ 1));
    unsigned epoch = a.epoch_base;

    // zero initial hidden per batch (predict_gpu.py:99): this workgroup's direction of its two tiles
    {
        const f32x4 z = splat4(0.f);
        (a.hid + ((size_t)tile0 * 2 + dir) * (kHidDirStride / 4))[tid] = z;
        (a.hid + ((size_t)tile1 * 2 + dir) * (kHidDirStride / 4))[tid] = z;
    }
    own_data_fence();

#pragma unroll 1
    for (int c = 0; c < kChunks; ++c) {   // predict_gpu.py:114-149
        // encoder recurrence over positions [50c, 50c + 100) (the reverse direction's gi is stored time-reversed)
        gru_pair_body<false>(smem, pair, dir, a.gi_enc, a.gi_enc_tile_stride, c * kJump, kSeq - c * kJump - kWin, kWin,
                             a.whp_enc, a.bhn_enc, a.hid, a.y1, a.y_tile_stride, (const f32x4*)nullptr, (f32x4*)nullptr,
                             a.pl_tile_stride, a.ntiles);
        pair_handoff(mine, theirs, ++epoch, a.error);                  // A: y1 of both directions
        // decoder input projection of this direction, tile by tile
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && tile1 == tile0) break;
            gemm_dec_ws_body(smem, k ? tile1 : tile0, dir, a.y1, a.y_tile_stride, a.wp_dec, a.bias_dec, a.gi_dec,
                             a.gi_dec_tile_stride, kWin);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (no LDS-DMA row may land in the next user's LDS)
            __syncthreads();
        }
        own_data_fence();                                              // gi_dec (and hid from the encoder) are this workgroup's own
        gru_pair_body<true>(smem, pair, dir, a.gi_dec, a.gi_dec_tile_stride, 0, 0, kWin, a.whp_dec, a.bhn_dec, a.hid,
                            (f32x4*)nullptr, a.y_tile_stride, a.whd, a.plogit, a.pl_tile_stride, a.ntiles);
        pair_handoff(mine, theirs, ++epoch, a.error);                  // B: partial logits of both directions
        // heads: this direction takes half `dir` of the chunk's positions of both tiles, every wave its own positions
        heads_half_body((uint8_t*)smem, tid, tile0, tile1, dir, a.plogit, a.pl_tile_stride, a.bhd, c, kWin, a.n_windows,
                        a.pending, a.bases, a.rles, a.acc_base, a.acc_rle);
        own_data_fence();   // hid written by the decoder is the next chunk's encoder h0; LDS changes hands
    }
}

}  // namespace helen
