// cpu_path.cpp -- libhelen_cpu.so: the `helen polish` inference path on the HOST, for runs without --gpu_mode.
//
// The reference's CPU mode (models/predict_cpu.py:39-170: an ONNX Runtime session of the same TransducerGRU, driven by
// the same 19-chunk loop) exists so that a box without a GPU can still polish; this is that mode for this package: the
// product's OWN code (nothing of oracle/ -- that directory is test infrastructure), plain C++ with OpenMP over blocks of
// 16 windows, written so that gcc vectorises the matrix products (AVX2 / AVX-512 clones of the two hot loops).  Same
// arithmetic as the device path, statement for statement:
//   TransducerGRU.forward (models/TransducerModel.py:60-79): bidirectional GRU encoder 90 -> 128 (h0 = incoming hidden),
//     bidirectional GRU decoder 256 -> 128 (h0 = the encoder's h_n), two linear heads on the decoder output; gate order
//     r, z, n;  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
//   the window loop (models/predict_cpu.py:93-140 = predict_gpu.py:99-159): hidden = 0 per window; chunks of 100 positions
//     at stride 50; softmax of each chunk's logits ADDED into [1000, C] accumulators; argmax with the first maximum
// fp32 throughout; a matrix product is a k-ascending sum per output (fused multiply-adds where the CPU has them), so
// logits agree with the reference's to the fp32 tolerance of tests/golden_cases.py and labels wherever the top-1 / top-2
// margin is above fp32 resolution -- the bar the device path is held to.  C ABI: include/helen_cpu.h.
#include "../../include/helen_cpu.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_err[256] = "";
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

constexpr int kF = 90, kH = 128, kG = 3 * kH, kSeq = 1000, kWin = 100, kJump = 50, kChunks = 19, kNB = 5, kNR = 11;
constexpr int kBlock = 16;      // windows per task

#define HELEN_CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))

// C[m][n] = bias[n] + sum_k A[m][k] * Wt[k][n]      (A: M x K row-major, Wt: K x N row-major, N a multiple of 64)
HELEN_CLONES void gemm_bias(const float* __restrict__ A, int M, int K, const float* __restrict__ Wt, int N,
                            const float* __restrict__ bias, float* __restrict__ C) {
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mb = std::min(4, M - m0);
        for (int n0 = 0; n0 < N; n0 += 64) {
            float acc[4][64];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 64; ++j) acc[i][j] = bias ? bias[n0 + j] : 0.f;
            for (int k = 0; k < K; ++k) {
                const float* w = Wt + (size_t)k * N + n0;
                for (int i = 0; i < mb; ++i) {
                    const float a = A[(size_t)(m0 + i) * K + k];
                    for (int j = 0; j < 64; ++j) acc[i][j] += a * w[j];
                }
            }
            for (int i = 0; i < mb; ++i) memcpy(C + (size_t)(m0 + i) * N + n0, acc[i], 64 * sizeof(float));
        }
    }
}

inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// one GRU step for `nb` windows: gh = h . Whh^T + b_hh, the gates, h updated in place; y (optional) receives h
HELEN_CLONES void gru_step(const float* __restrict__ gi, float* __restrict__ h, const float* __restrict__ WhhT,
                           const float* __restrict__ bhh, float* __restrict__ gh, int nb, float* __restrict__ y,
                           int y_stride) {
    gemm_bias(h, nb, kH, WhhT, kG, bhh, gh);
    for (int b = 0; b < nb; ++b) {
        const float* g = gi + (size_t)b * kG;
        const float* q = gh + (size_t)b * kG;
        float* hb = h + (size_t)b * kH;
        for (int u = 0; u < kH; ++u) {
            const float r = sigmoidf(g[u] + q[u]);
            const float z = sigmoidf(g[kH + u] + q[kH + u]);
            const float n = tanhf(g[2 * kH + u] + r * q[2 * kH + u]);
            hb[u] = (1.f - z) * n + z * hb[u];
        }
        if (y) memcpy(y + (size_t)b * y_stride, hb, kH * sizeof(float));
    }
}

struct Packed {                 // the parameters as the loops want them: W^T, row-major [K][N]
    std::vector<float> enc_ih[2], enc_hh[2], dec_ih[2], dec_hh[2], heads;      // heads: [256][64] (16 used, zero padded)
    std::vector<float> enc_bih[2], enc_bhh[2], dec_bih[2], dec_bhh[2], heads_b;
};
void transpose(const float* w, int rows, int cols, std::vector<float>* out, int out_cols = 0) {   // w [rows][cols] -> [cols][rows (padded)]
    const int n = out_cols ? out_cols : rows;
    out->assign((size_t)cols * n, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) (*out)[(size_t)c * n + r] = w[(size_t)r * cols + c];
}
bool pack(const HelenWeights* w, Packed* p) {
    if (w->features != kF || w->hidden != kH || w->n_base != kNB || w->n_rle != kNR) return false;
    for (int d = 0; d < 2; ++d) {
        transpose(w->enc_w_ih[d], kG, kF, &p->enc_ih[d]);
        transpose(w->enc_w_hh[d], kG, kH, &p->enc_hh[d]);
        transpose(w->dec_w_ih[d], kG, 2 * kH, &p->dec_ih[d]);
        transpose(w->dec_w_hh[d], kG, kH, &p->dec_hh[d]);
        p->enc_bih[d].assign(w->enc_b_ih[d], w->enc_b_ih[d] + kG);
        p->enc_bhh[d].assign(w->enc_b_hh[d], w->enc_b_hh[d] + kG);
        p->dec_bih[d].assign(w->dec_b_ih[d], w->dec_b_ih[d] + kG);
        p->dec_bhh[d].assign(w->dec_b_hh[d], w->dec_b_hh[d] + kG);
    }
    std::vector<float> both((size_t)(kNB + kNR) * 2 * kH);
    memcpy(both.data(), w->base_w, (size_t)kNB * 2 * kH * sizeof(float));
    memcpy(both.data() + (size_t)kNB * 2 * kH, w->rle_w, (size_t)kNR * 2 * kH * sizeof(float));
    transpose(both.data(), kNB + kNR, 2 * kH, &p->heads, 64);
    p->heads_b.assign(64, 0.f);
    memcpy(p->heads_b.data(), w->base_b, kNB * sizeof(float));
    memcpy(p->heads_b.data() + kNB, w->rle_b, kNR * sizeof(float));
    return true;
}

struct Scratch {
    std::vector<float> x, gi, gh, y1, y2, logits, h;
    Scratch()
        : x((size_t)kWin * kBlock * kF), gi((size_t)kWin * kBlock * kG), gh((size_t)kBlock * kG), y1((size_t)kWin * kBlock * 2 * kH),
          y2((size_t)kWin * kBlock * 2 * kH), logits((size_t)kWin * kBlock * 64), h((size_t)2 * kBlock * kH) {}
};

// one bidirectional GRU layer over T steps of `nb` windows: in [T][nb][K] -> out [T][nb][256]; h [2][nb][128] in: h0, out: h_n
void bigru(const float* in, int K, int T, int nb, const std::vector<float>* WihT, const std::vector<float>* bih,
           const std::vector<float>* WhhT, const std::vector<float>* bhh, float* h, float* out, Scratch* s) {
    for (int d = 0; d < 2; ++d) {
        gemm_bias(in, T * nb, K, WihT[d].data(), kG, bih[d].data(), s->gi.data());       // every step's input part at once
        float* hd = h + (size_t)d * nb * kH;
        for (int i = 0; i < T; ++i) {
            const int t = d ? T - 1 - i : i;                                             // the reverse direction walks t = T-1 .. 0
            gru_step(s->gi.data() + (size_t)t * nb * kG, hd, WhhT[d].data(), bhh[d].data(), s->gh.data(), nb,
                     out + (size_t)t * nb * 2 * kH + (size_t)d * kH, 2 * kH);
        }
    }
}

// TransducerGRU.forward for `nb` windows and T <= 100 positions: x [T][nb][90], hidden [2][nb][128] (in / out),
// logits [T][nb][64] (16 used: 5 base, 11 run-length)
void forward(const Packed& p, int T, int nb, float* hidden, Scratch* s) {
    bigru(s->x.data(), kF, T, nb, p.enc_ih, p.enc_bih, p.enc_hh, p.enc_bhh, hidden, s->y1.data(), s);
    bigru(s->y1.data(), 2 * kH, T, nb, p.dec_ih, p.dec_bih, p.dec_hh, p.dec_bhh, hidden, s->y2.data(), s);   // h0 = the encoder's h_n
    gemm_bias(s->y2.data(), T * nb, 2 * kH, p.heads.data(), 64, p.heads_b.data(), s->logits.data());
}

void softmax_add(const float* logit, int n, float* acc) {          // torch.nn.Softmax(dim=2), then `+=` (predict_cpu.py:124-137)
    float mx = logit[0];
    for (int i = 1; i < n; ++i) mx = std::max(mx, logit[i]);
    float e[16], sum = 0.f;
    for (int i = 0; i < n; ++i) {
        e[i] = expf(logit[i] - mx);
        sum += e[i];
    }
    for (int i = 0; i < n; ++i) acc[i] += e[i] / sum;
}

}  // namespace

extern "C" {

const char* helen_cpu_last_error(void) { return g_err; }
int helen_cpu_abi_version(void) { return HELEN_CPU_ABI_VERSION; }

int helen_cpu_polish_batch(const HelenWeights* w, const uint8_t* images, int n_windows, uint8_t* bases, uint8_t* rles,
                           float* acc_base_opt, float* acc_rle_opt, int threads) {
    if (!w || !images || !bases || !rles) return fail("null argument");
    if (n_windows <= 0) return fail("n_windows must be > 0");
    Packed p;
    if (!pack(w, &p)) return fail("weights are not a TransducerGRU of 90 features, hidden 128, 5 + 11 classes");
    const int nblocks = (n_windows + kBlock - 1) / kBlock;
    const int nt = std::max(1, std::min(threads > 0 ? threads : omp_get_max_threads(), nblocks));
#pragma omp parallel num_threads(nt)
    {
        Scratch s;
        std::vector<float> ab((size_t)kBlock * kSeq * kNB), ar((size_t)kBlock * kSeq * kNR);
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblocks; ++blk) {
            const int w0 = blk * kBlock, nb = std::min(kBlock, n_windows - w0);
            std::fill(ab.begin(), ab.end(), 0.f);
            std::fill(ar.begin(), ar.end(), 0.f);
            std::fill(s.h.begin(), s.h.end(), 0.f);                                       // hidden = 0 per batch (predict_cpu.py:99)
            for (int c = 0; c < kChunks; ++c) {
                const int pos0 = c * kJump;
                for (int t = 0; t < kWin; ++t)                                            // images.type(FloatTensor) (:97)
                    for (int b = 0; b < nb; ++b) {
                        const uint8_t* src = images + ((size_t)(w0 + b) * kSeq + pos0 + t) * kF;
                        float* dst = s.x.data() + ((size_t)t * nb + b) * kF;
                        for (int f = 0; f < kF; ++f) dst[f] = (float)src[f];
                    }
                forward(p, kWin, nb, s.h.data(), &s);
                for (int t = 0; t < kWin; ++t)
                    for (int b = 0; b < nb; ++b) {
                        const float* lg = s.logits.data() + ((size_t)t * nb + b) * 64;
                        softmax_add(lg, kNB, ab.data() + ((size_t)b * kSeq + pos0 + t) * kNB);
                        softmax_add(lg + kNB, kNR, ar.data() + ((size_t)b * kSeq + pos0 + t) * kNR);
                    }
            }
            for (int b = 0; b < nb; ++b)
                for (int pos = 0; pos < kSeq; ++pos) {                                    // torch.max: the first maximum (:152-156)
                    const float* a = ab.data() + ((size_t)b * kSeq + pos) * kNB;
                    int best = 0;
                    for (int i = 1; i < kNB; ++i)
                        if (a[i] > a[best]) best = i;
                    bases[(size_t)(w0 + b) * kSeq + pos] = (uint8_t)best;
                    const float* r = ar.data() + ((size_t)b * kSeq + pos) * kNR;
                    best = 0;
                    for (int i = 1; i < kNR; ++i)
                        if (r[i] > r[best]) best = i;
                    rles[(size_t)(w0 + b) * kSeq + pos] = (uint8_t)best;
                }
            if (acc_base_opt) memcpy(acc_base_opt + (size_t)w0 * kSeq * kNB, ab.data(), (size_t)nb * kSeq * kNB * sizeof(float));
            if (acc_rle_opt) memcpy(acc_rle_opt + (size_t)w0 * kSeq * kNR, ar.data(), (size_t)nb * kSeq * kNR * sizeof(float));
        }
    }
    return 0;
}

int helen_cpu_chunk_forward(const HelenWeights* w, const float* x, const float* h_in, int B, int T, float* base, float* rle,
                            float* h_out, int threads) {
    if (!w || !x || !h_in || !base || !rle || !h_out) return fail("null argument");
    if (B <= 0) return fail("B must be > 0");
    if (T <= 0 || T > kWin) return fail("T %d outside 1..%d (TRAIN_WINDOW)", T, kWin);
    Packed p;
    if (!pack(w, &p)) return fail("weights are not a TransducerGRU of 90 features, hidden 128, 5 + 11 classes");
    const int nblocks = (B + kBlock - 1) / kBlock;
    const int nt = std::max(1, std::min(threads > 0 ? threads : omp_get_max_threads(), nblocks));
#pragma omp parallel num_threads(nt)
    {
        Scratch s;
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblocks; ++blk) {
            const int w0 = blk * kBlock, nb = std::min(kBlock, B - w0);
            for (int t = 0; t < T; ++t)
                for (int b = 0; b < nb; ++b)
                    memcpy(s.x.data() + ((size_t)t * nb + b) * kF, x + ((size_t)(w0 + b) * T + t) * kF, kF * sizeof(float));
            for (int d = 0; d < 2; ++d)                                                   // hidden.transpose(0, 1) (TransducerModel.py:68)
                for (int b = 0; b < nb; ++b)
                    memcpy(s.h.data() + ((size_t)d * nb + b) * kH, h_in + ((size_t)(w0 + b) * 2 + d) * kH, kH * sizeof(float));
            forward(p, T, nb, s.h.data(), &s);
            for (int t = 0; t < T; ++t)
                for (int b = 0; b < nb; ++b) {
                    const float* lg = s.logits.data() + ((size_t)t * nb + b) * 64;
                    memcpy(base + ((size_t)(w0 + b) * T + t) * kNB, lg, kNB * sizeof(float));
                    memcpy(rle + ((size_t)(w0 + b) * T + t) * kNR, lg + kNB, kNR * sizeof(float));
                }
            for (int d = 0; d < 2; ++d)
                for (int b = 0; b < nb; ++b)
                    memcpy(h_out + ((size_t)(w0 + b) * 2 + d) * kH, s.h.data() + ((size_t)d * nb + b) * kH, kH * sizeof(float));
        }
    }
    return 0;
}

}  // extern "C"
