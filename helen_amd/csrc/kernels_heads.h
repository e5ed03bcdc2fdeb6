// kernels_heads.h -- heads + softmax + accumulate + argmax, and the evaluation variant
#pragma once
#include "kernels_common.h"

namespace helen {

// ------------------------------------------------------------------------------------------------
// Heads + softmax + accumulate + argmax (TransducerModel.py:75-76, predict_gpu.py:137-156).
//   One 16-column MFMA tile is exactly the 5 base + 11 run-length logits of 16 windows; the matrix product
//   itself happens inside the decoder recurrences (each direction's half, see gru_kernel<true>), this
//   kernel adds the two partial tiles and does everything after the logits.
//   grid (tiles, groups of kHeadsSpan positions), 4 waves striding over the positions of the group.
//   mode 0 (polish): positions 50c+t; the first half of chunk c receives its second (final)
//     contribution -> add the pending softmax of chunk c-1, argmax, labels; the second half is
//     parked in `pending` for chunk c+1 (or is final for the last chunk).  A position gets at most
//     two contributions, and 0 + a + b == a + b in fp32, so this equals the reference's
//     zero-pad-and-add into a [B,1000,C] accumulator.
//   mode 1 (logits): write base[B,T,5] / rle[B,T,11] logits (the operator-level boundary).
// ------------------------------------------------------------------------------------------------
// Butterfly reductions over the 16 lanes of a row (= the 16 classes of one window) as DPP row operations instead of
// ds_bpermute shuffles (128 LDS-crossbar round trips per position before): xor 1 and xor 2 are quad permutations;
// after them the four lanes of a quad agree, so mirroring within 8 lanes (lane i <-> 7 - i) pairs every quad with the
// other quad of its half exactly as xor 4 would, and mirroring the row (i <-> 15 - i) pairs the halves as xor 8 would.
// Same operands at every level (a + b == b + a, max and the first-maximum rule are symmetric): the same bits as the
// xor butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_f32<kDppXor1>(v));
    v = fmaxf(v, dpp_f32<kDppXor2>(v));
    v = fmaxf(v, dpp_f32<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f32<kDppMirror>(v));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_f32<kDppXor1>(v);
    v += dpp_f32<kDppXor2>(v);
    v += dpp_f32<kDppHalfMirror>(v);
    v += dpp_f32<kDppMirror>(v);
    return v;
}
// argmax with first-maximum tie-break (torch.max on CPU, predict_gpu.py:155)
template <int CTRL>
__device__ __forceinline__ void argmax_step(float& v, int& idx) {
    const float ov = dpp_f32<CTRL>(v);
    const int oi = dpp_i32<CTRL>(idx);
    if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
    }
}
__device__ __forceinline__ int group16_argmax(float v, int idx) {
    argmax_step<kDppXor1>(v, idx);
    argmax_step<kDppXor2>(v, idx);
    argmax_step<kDppHalfMirror>(v, idx);
    argmax_step<kDppMirror>(v, idx);
    return idx;
}

#define HELEN_HEADS_SPAN 10
constexpr int kHeadsSpan = HELEN_HEADS_SPAN;  // positions per workgroup; divides kJump so a group never straddles halves

// The 16 logits of 16 windows at chunk position t: the decoder recurrences already multiplied each
// direction's half of [h_fwd | h_bwd] by the head weights (plogit[tile][slot][dir][64 lanes], FRAG layout;
// the backward direction is stored time-reversed: position t sits in slot T-1-t) -- add the two tiles
// and the bias.
__device__ __forceinline__ f32x4 head_logits(const f32x4* __restrict__ plogit, long tile_stride, int tile, int t,
                                             int T, float bias, int lane) {
    const f32x4* p = plogit + (size_t)tile * tile_stride + lane;
    return p[((size_t)t * 2) * 64] + p[((size_t)(T - 1 - t) * 2 + 1) * 64] + splat4(bias);
}

// softmax over the 5 base classes and over the 11 run-length classes of 16 windows (one fragment: row 4q + r = window,
// column j = class), nn.Softmax(dim=2) of predict_gpu.py:137-138
__device__ __forceinline__ f32x4 heads_softmax(f32x4 logit, bool isb) {
    f32x4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = logit[r];
        const float mb = group16_max(isb ? x : -INFINITY);
        const float mr = group16_max(isb ? -INFINITY : x);
        const float e = expf(x - (isb ? mb : mr));
        const float sb = group16_sum(isb ? e : 0.f);
        const float sr = group16_sum(isb ? 0.f : e);
        p[r] = e / (isb ? sb : sr);
    }
    return p;
}

typedef uint8_t HeadsLabels[2][kTile][kHeadsSpan];   // LDS staging of one (tile, position group): labels written as rows

// The body is a device function over one (tile, group of kHeadsSpan positions) for 256 threads `tid` and their LDS
// staging block; heads_kernel below is one call per workgroup (all threads reach the one __syncthreads inside;
// `valid` = false computes nothing).
__device__ __forceinline__ void heads_body(
    HeadsLabels& lab, const int tid, const int tile, const int t0, const bool valid,
    const f32x4* __restrict__ plogit, long pl_tile_stride,
    const float* __restrict__ bhd, int mode, int chunk, int T, int n_windows,
    f32x4* __restrict__ pending, uint8_t* __restrict__ bases, uint8_t* __restrict__ rles,
    float* __restrict__ acc_base, float* __restrict__ acc_rle, float* __restrict__ logit_base,
    float* __restrict__ logit_rle) {
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int t1 = valid ? min(T, t0 + kHeadsSpan) : t0;
    const int half = t0 / kJump;
    const bool isb = j < kNB;

    const float bias = bhd[j];

    const bool park = (mode == 0) && (half == 1) && (chunk < kChunks - 1);
    const bool add_prev = (mode == 0) && (half == 0) && (chunk > 0);

    for (int t = t0 + w; t < t1; t += 4) {
        const f32x4 logit = head_logits(plogit, pl_tile_stride, tile, t, T, bias, lane);   // row 4q+r (window), col j (class)

        if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int window = tile * kTile + 4 * q + r;
                if (window < n_windows) {
                    if (isb)
                        logit_base[((size_t)window * T + t) * kNB + j] = logit[r];
                    else
                        logit_rle[((size_t)window * T + t) * kNR + (j - kNB)] = logit[r];
                }
            }
            continue;
        }

        f32x4 p = heads_softmax(logit, isb);
        // `pending` is double-buffered by chunk parity: this launch's second half parks into slot
        // chunk&1 while its first half still reads what chunk-1 parked in the other slot.
        if (park) {
            pending[(((size_t)tile * 2 + (chunk & 1)) * kJump + (t - kJump)) * 64 + lane] = p;
            continue;
        }
        if (add_prev) p += pending[(((size_t)tile * 2 + ((chunk - 1) & 1)) * kJump + t) * 64 + lane];
        const int pos = chunk * kJump + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int window = tile * kTile + 4 * q + r;
            if (window < n_windows) {
                if (acc_base != nullptr && isb)
                    acc_base[((size_t)window * kSeq + pos) * kNB + j] = p[r];
                if (acc_rle != nullptr && !isb)
                    acc_rle[((size_t)window * kSeq + pos) * kNR + (j - kNB)] = p[r];
            }
            const int ib = group16_argmax(isb ? p[r] : -1.f, isb ? j : 99);
            const int ir = group16_argmax(isb ? -1.f : p[r], isb ? 99 : j);
            if (j == 0) {
                lab[0][4 * q + r][t - t0] = (uint8_t)ib;
                lab[1][4 * q + r][t - t0] = (uint8_t)(ir - kNB);
            }
        }
    }
    if (mode != 0 || park) return;      // (uniform over the workgroup: mode, chunk and the half t0 sits in)
    __syncthreads();
    const int span = t1 - t0;
    for (int g = tid; valid && g < 2 * kTile * kHeadsSpan; g += 256) {
        const int kind = g / (kTile * kHeadsSpan);
        const int rem = g % (kTile * kHeadsSpan);
        const int win = rem / kHeadsSpan;
        const int tl = rem % kHeadsSpan;
        const int window = tile * kTile + win;
        if (window < n_windows && tl < span) {
            uint8_t* out = kind ? rles : bases;
            out[(size_t)window * kSeq + chunk * kJump + t0 + tl] = lab[kind][win][tl];
        }
    }
}

__global__ __launch_bounds__(256) void heads_kernel(
    const f32x4* __restrict__ plogit, long pl_tile_stride,
    const float* __restrict__ bhd, int mode, int chunk, int T, int n_windows,
    f32x4* __restrict__ pending, uint8_t* __restrict__ bases, uint8_t* __restrict__ rles,
    float* __restrict__ acc_base, float* __restrict__ acc_rle, float* __restrict__ logit_base,
    float* __restrict__ logit_rle) {
    __shared__ HeadsLabels lab;
    heads_body(lab, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y * kHeadsSpan, true, plogit, pl_tile_stride, bhd, mode,
               chunk, T, n_windows, pending, bases, rles, acc_base, acc_rle, logit_base, logit_rle);
}

// ------------------------------------------------------------------------------------------------
// Heads + cross-entropy + confusion counts: the per-chunk body of the reference's evaluation loop
// (models/test.py:104-121) for chunk `chunk` of 16-window tiles.
//   logits as in heads_kernel (from the decoder's partial tiles); per position: nll_base = logsumexp(base) - base[label_base],
//   nll_rle likewise (nn.CrossEntropyLoss = log_softmax + nll), predictions = first-maximum argmax of
//   the LOGITS (torchnet ConfusionMeter: np.argmax), confusion[target][predicted] += 1.
//   Outputs: stats[window][chunk][group of kHeadsSpan positions][3] = (sum nll_base, sum w[l]*nll_rle,
//   sum w[l]) summed over the group's positions in position order (deterministic; the host finishes
//   the per-batch means), and the two confusion matrices accumulated with integer atomics.
//   Labels outside 0..4 / 0..10 are the caller's error (torch raises); they are clamped here only
//   to keep the accesses in range.
// ------------------------------------------------------------------------------------------------
struct RleClassWeights {
    float w[kNR];
};

__global__ __launch_bounds__(256) void heads_eval_kernel(
    const f32x4* __restrict__ plogit, long pl_tile_stride,
    const float* __restrict__ bhd, int chunk, int T, int n_windows,
    const uint8_t* __restrict__ label_base, const uint8_t* __restrict__ label_rle, RleClassWeights cw,
    float* __restrict__ stats, unsigned long long* __restrict__ conf_base,
    unsigned long long* __restrict__ conf_rle) {
    __shared__ float vals[3][kTile][kHeadsSpan];
    __shared__ unsigned hist_b[kNB * kNB], hist_r[kNR * kNR];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int j = lane & 15;
    const int q = lane >> 4;
    const int tile = blockIdx.x;
    const int t0 = blockIdx.y * kHeadsSpan;
    const int t1 = min(T, t0 + kHeadsSpan);
    const bool isb = j < kNB;
    for (int g = tid; g < 3 * kTile * kHeadsSpan; g += 256) (&vals[0][0][0])[g] = 0.f;
    if (tid < kNB * kNB) hist_b[tid] = 0;
    if (tid < kNR * kNR) hist_r[tid] = 0;
    __syncthreads();

    const float bias = bhd[j];

    for (int t = t0 + w; t < t1; t += 4) {
        const f32x4 logit = head_logits(plogit, pl_tile_stride, tile, t, T, bias, lane);   // row 4q+r (window), col j (class)
        const int pos = chunk * kJump + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int window = tile * kTile + 4 * q + r;
            const bool valid = window < n_windows;
            const int lb = valid ? min((int)label_base[(size_t)window * kSeq + pos], kNB - 1) : 0;
            const int lr = valid ? min((int)label_rle[(size_t)window * kSeq + pos], kNR - 1) : 0;
            const float x = logit[r];
            const float mb = group16_max(isb ? x : -INFINITY);
            const float mr = group16_max(isb ? -INFINITY : x);
            const float e = expf(x - (isb ? mb : mr));
            const float sb = group16_sum(isb ? e : 0.f);
            const float sr = group16_sum(isb ? 0.f : e);
            const float xb = group16_sum((isb && j == lb) ? x : 0.f);
            const float xr = group16_sum((!isb && j - kNB == lr) ? x : 0.f);
            const int pb = group16_argmax(isb ? x : -INFINITY, isb ? j : 99);
            const int pr = group16_argmax(isb ? -INFINITY : x, isb ? 99 : j) - kNB;
            if (j == 0 && valid) {
                const float wr = cw.w[lr];
                vals[0][4 * q + r][t - t0] = (mb + logf(sb)) - xb;
                vals[1][4 * q + r][t - t0] = wr * ((mr + logf(sr)) - xr);
                vals[2][4 * q + r][t - t0] = wr;
                atomicAdd(&hist_b[lb * kNB + pb], 1u);
                atomicAdd(&hist_r[lr * kNR + pr], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 3 * kTile) {
        const int k = tid / kTile, win = tid % kTile;
        const int window = tile * kTile + win;
        if (window < n_windows) {
            float sum = 0.f;
            for (int tl = 0; tl < t1 - t0; ++tl) sum += vals[k][win][tl];
            stats[(((size_t)window * kChunks + chunk) * (kWin / kHeadsSpan) + blockIdx.y) * 3 + k] = sum;
        }
    }
    if (tid < kNB * kNB && hist_b[tid]) atomicAdd(&conf_base[tid], (unsigned long long)hist_b[tid]);
    if (tid < kNR * kNR && hist_r[tid]) atomicAdd(&conf_rle[tid], (unsigned long long)hist_r[tid]);
}

}  // namespace helen
