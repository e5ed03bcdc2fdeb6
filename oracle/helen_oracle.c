/*
 * helen_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU, fp32 restatement of the reference's `helen polish` inference path, used only
 * as the checker in tests/, __graft_entry__.smoke() and as bench.py's cpu_baseline.  Nothing in
 * helen_amd/ may import, link or execute it.
 *
 * Parity pin: this restatement is checked against golden vectors generated in the build
 * container by importing the reference's own TransducerGRU (tests/golden/make_golden.py ->
 * tests/golden/ npz files; tests/test_oracle_golden.py).  The reference ships no tests or golden
 * vectors of its own for this path.
 *
 * What it follows (file:line into the reference):
 *   - GRU cell equations: torch.nn.GRU as used by `models/TransducerModel.py:43-58`; gate order
 *     r, z, n;  r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n), h' = (1-z)*n + z*h,
 *     with gi = x.W_ih^T + b_ih and gh = h.W_hh^T + b_hh.
 *   - data flow of one forward call: `models/TransducerModel.py:60-79`
 *     (encoder h0 = incoming hidden, decoder h0 = encoder h_n, returned hidden = decoder h_n;
 *      backward-direction h_n is the state after processing t = 0).
 *   - sliding-window driver: `models/predict_gpu.py:97-159`
 *     (u8 -> f32, zero hidden per batch, chunks i = 0,50,...,900, softmax(dim=2), zero-pad-and-add,
 *      argmax with first-maximum tie-break as torch.max on CPU).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/helen_hip.h"

#define SEQ HELEN_SEQ_LENGTH
#define WIN HELEN_TRAIN_WINDOW
#define JUMP HELEN_WINDOW_JUMP

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Transposed copy of a [rows, cols] matrix -> [cols, rows], so the inner loop over output units
 * is contiguous (vectorises without reassociating the k-sum). */
static float* transpose_(const float* w, int rows, int cols) {
    float* t = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return t;
}

typedef struct {
    int K, H;
    float* w_ih_t; /* [K, 3H] */
    float* w_hh_t; /* [H, 3H] */
    const float* b_ih;
    const float* b_hh;
} Dir;

static void dir_init(Dir* d, const float* w_ih, const float* w_hh, const float* b_ih,
                     const float* b_hh, int K, int H) {
    d->K = K;
    d->H = H;
    d->w_ih_t = transpose_(w_ih, 3 * H, K);
    d->w_hh_t = transpose_(w_hh, 3 * H, H);
    d->b_ih = b_ih;
    d->b_hh = b_hh;
}
static void dir_free(Dir* d) {
    free(d->w_ih_t);
    free(d->w_hh_t);
}

/* One GRU direction over T steps for ONE window.
 *   x: [T, K] (row stride xs), h: [H] in/out, y: [T, ys] output written at y[t*ys + yoff .. +H)
 *   reverse: process t = T-1 .. 0 (nn.GRU `_reverse` weights), output stays at index t. */
static void gru_dir(const Dir* d, const float* x, int xs, int T, int reverse, float* h, float* y,
                    int ys, int yoff, float* gi, float* gh) {
    const int H = d->H, G = 3 * d->H, K = d->K;
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? (T - 1 - s) : s;
        const float* xt = x + (size_t)t * xs;
        for (int j = 0; j < G; ++j) {
            gi[j] = 0.0f;
            gh[j] = 0.0f;
        }
        for (int k = 0; k < K; ++k) {
            const float xv = xt[k];
            const float* w = d->w_ih_t + (size_t)k * G;
            for (int j = 0; j < G; ++j) gi[j] += xv * w[j];
        }
        for (int k = 0; k < H; ++k) {
            const float hv = h[k];
            const float* w = d->w_hh_t + (size_t)k * G;
            for (int j = 0; j < G; ++j) gh[j] += hv * w[j];
        }
        for (int j = 0; j < G; ++j) {
            gi[j] += d->b_ih[j];
            gh[j] += d->b_hh[j];
        }
        for (int j = 0; j < H; ++j) {
            const float r = sigmoidf_(gi[j] + gh[j]);
            const float z = sigmoidf_(gi[H + j] + gh[H + j]);
            const float n = tanhf(gi[2 * H + j] + r * gh[2 * H + j]);
            h[j] = (1.0f - z) * n + z * h[j];
        }
        memcpy(y + (size_t)t * ys + yoff, h, sizeof(float) * H);
    }
}

typedef struct {
    int F, H, nb, nr;
    Dir enc[2], dec[2];
    const float *base_w, *base_b, *rle_w, *rle_b;
} Net;

static int net_init(Net* n, const HelenWeights* w) {
    if (!w || w->features <= 0 || w->hidden <= 0) return -1;
    n->F = w->features;
    n->H = w->hidden;
    n->nb = w->n_base;
    n->nr = w->n_rle;
    for (int d = 0; d < 2; ++d) {
        dir_init(&n->enc[d], w->enc_w_ih[d], w->enc_w_hh[d], w->enc_b_ih[d], w->enc_b_hh[d], n->F,
                 n->H);
        dir_init(&n->dec[d], w->dec_w_ih[d], w->dec_w_hh[d], w->dec_b_ih[d], w->dec_b_hh[d],
                 2 * n->H, n->H);
    }
    n->base_w = w->base_w;
    n->base_b = w->base_b;
    n->rle_w = w->rle_w;
    n->rle_b = w->rle_b;
    return 0;
}
static void net_free(Net* n) {
    for (int d = 0; d < 2; ++d) {
        dir_free(&n->enc[d]);
        dir_free(&n->dec[d]);
    }
}

/* TransducerGRU.forward for ONE window (`models/TransducerModel.py:60-79`).
 *   x [T,F], hidden [2,H] in/out, base [T,nb], rle [T,nr]; scratch y1,y2 [T,2H], gi,gh [3H]. */
static void forward_one(const Net* n, const float* x, int T, float* hidden, float* base, float* rle,
                        float* y1, float* y2, float* gi, float* gh) {
    const int H = n->H;
    /* encoder: h0 = incoming hidden (index 0 forward, 1 backward) */
    gru_dir(&n->enc[0], x, n->F, T, 0, hidden, y1, 2 * H, 0, gi, gh);
    gru_dir(&n->enc[1], x, n->F, T, 1, hidden + H, y1, 2 * H, H, gi, gh);
    /* decoder: h0 = encoder h_n, input = [h_fwd(t) | h_bwd(t)] */
    gru_dir(&n->dec[0], y1, 2 * H, T, 0, hidden, y2, 2 * H, 0, gi, gh);
    gru_dir(&n->dec[1], y1, 2 * H, T, 1, hidden + H, y2, 2 * H, H, gi, gh);
    /* heads (`TransducerModel.py:75-76`) */
    for (int t = 0; t < T; ++t) {
        const float* y = y2 + (size_t)t * 2 * H;
        for (int c = 0; c < n->nb; ++c) {
            float a = 0.0f;
            const float* w = n->base_w + (size_t)c * 2 * H;
            for (int k = 0; k < 2 * H; ++k) a += y[k] * w[k];
            base[(size_t)t * n->nb + c] = a + n->base_b[c];
        }
        for (int c = 0; c < n->nr; ++c) {
            float a = 0.0f;
            const float* w = n->rle_w + (size_t)c * 2 * H;
            for (int k = 0; k < 2 * H; ++k) a += y[k] * w[k];
            rle[(size_t)t * n->nr + c] = a + n->rle_b[c];
        }
    }
}

static void softmax_add(const float* logits, int C, float* acc) {
    float m = logits[0];
    for (int c = 1; c < C; ++c)
        if (logits[c] > m) m = logits[c];
    float e[16];
    float s = 0.0f;
    for (int c = 0; c < C; ++c) {
        e[c] = expf(logits[c] - m);
        s += e[c];
    }
    for (int c = 0; c < C; ++c) acc[c] += e[c] / s;
}

static uint8_t argmax_first(const float* v, int C) {
    int best = 0;
    for (int c = 1; c < C; ++c)
        if (v[c] > v[best]) best = c; /* strict > keeps the first maximum (torch.max on CPU) */
    return (uint8_t)best;
}

/* ---- exported ---------------------------------------------------------------------------- */

/* One TransducerGRU.forward over a batch: x [B,T,F], h_in/h_out [B,2,H], base [B,T,nb],
 * rle [B,T,nr].  Returns 0, or -1 on bad arguments. */
int oracle_gru_chunk_forward(const HelenWeights* w, const float* x, const float* h_in, int B, int T,
                             float* base, float* rle, float* h_out) {
    Net n;
    if (net_init(&n, w) != 0 || B < 0 || T <= 0) return -1;
    const int H = n.H;
#pragma omp parallel
    {
        float* y1 = (float*)malloc(sizeof(float) * (size_t)T * 2 * H);
        float* y2 = (float*)malloc(sizeof(float) * (size_t)T * 2 * H);
        float* gi = (float*)malloc(sizeof(float) * 3 * H);
        float* gh = (float*)malloc(sizeof(float) * 3 * H);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            float* hid = h_out + (size_t)b * 2 * H;
            memcpy(hid, h_in + (size_t)b * 2 * H, sizeof(float) * 2 * H);
            forward_one(&n, x + (size_t)b * T * n.F, T, hid, base + (size_t)b * T * n.nb,
                        rle + (size_t)b * T * n.nr, y1, y2, gi, gh);
        }
        free(y1);
        free(y2);
        free(gi);
        free(gh);
    }
    net_free(&n);
    return 0;
}

/* The per-batch body of the reference loop (`models/predict_gpu.py:97-159`) for B windows.
 *   images [B,1000,F] u8 -> bases, rles [B,1000] u8.
 *   Optional traces (NULL to skip):
 *     acc_base [B,1000,nb], acc_rle [B,1000,nr]   the accumulated softmax tensors
 *     hidden_trace [19,B,2,H]                     hidden returned by each chunk's forward
 *     logit_base_trace [19,B,100,nb], logit_rle_trace [19,B,100,nr]   per-chunk logits */
int oracle_polish_batch(const HelenWeights* w, const uint8_t* images, int B, uint8_t* bases,
                        uint8_t* rles, float* acc_base, float* acc_rle, float* hidden_trace,
                        float* logit_base_trace, float* logit_rle_trace) {
    Net n;
    if (net_init(&n, w) != 0 || B < 0) return -1;
    const int H = n.H, F = n.F, nb = n.nb, nr = n.nr;
    if (nb > 16 || nr > 16) {
        net_free(&n);
        return -1;
    }
#pragma omp parallel
    {
        float* xf = (float*)malloc(sizeof(float) * (size_t)SEQ * F);
        float* y1 = (float*)malloc(sizeof(float) * (size_t)WIN * 2 * H);
        float* y2 = (float*)malloc(sizeof(float) * (size_t)WIN * 2 * H);
        float* gi = (float*)malloc(sizeof(float) * 3 * H);
        float* gh = (float*)malloc(sizeof(float) * 3 * H);
        float* lb = (float*)malloc(sizeof(float) * (size_t)WIN * nb);
        float* lr = (float*)malloc(sizeof(float) * (size_t)WIN * nr);
        float* ab = (float*)malloc(sizeof(float) * (size_t)SEQ * nb);
        float* ar = (float*)malloc(sizeof(float) * (size_t)SEQ * nr);
        float* hid = (float*)malloc(sizeof(float) * 2 * H);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const uint8_t* img = images + (size_t)b * SEQ * F;
            for (size_t i = 0; i < (size_t)SEQ * F; ++i) xf[i] = (float)img[i]; /* :97 */
            memset(hid, 0, sizeof(float) * 2 * H);                              /* :99 */
            memset(ab, 0, sizeof(float) * (size_t)SEQ * nb);                    /* :105 */
            memset(ar, 0, sizeof(float) * (size_t)SEQ * nr);                    /* :106 */
            int c = 0;
            for (int i = 0; i < SEQ; i += JUMP) { /* :114-117 */
                if (i + WIN > SEQ) break;
                forward_one(&n, xf + (size_t)i * F, WIN, hid, lb, lr, y1, y2, gi, gh); /* :129 */
                for (int t = 0; t < WIN; ++t) { /* :137-149 */
                    softmax_add(lb + (size_t)t * nb, nb, ab + (size_t)(i + t) * nb);
                    softmax_add(lr + (size_t)t * nr, nr, ar + (size_t)(i + t) * nr);
                }
                if (hidden_trace)
                    memcpy(hidden_trace + ((size_t)c * B + b) * 2 * H, hid, sizeof(float) * 2 * H);
                if (logit_base_trace)
                    memcpy(logit_base_trace + ((size_t)c * B + b) * WIN * nb, lb,
                           sizeof(float) * WIN * nb);
                if (logit_rle_trace)
                    memcpy(logit_rle_trace + ((size_t)c * B + b) * WIN * nr, lr,
                           sizeof(float) * WIN * nr);
                ++c;
            }
            for (int p = 0; p < SEQ; ++p) { /* :155-156 */
                bases[(size_t)b * SEQ + p] = argmax_first(ab + (size_t)p * nb, nb);
                rles[(size_t)b * SEQ + p] = argmax_first(ar + (size_t)p * nr, nr);
            }
            if (acc_base) memcpy(acc_base + (size_t)b * SEQ * nb, ab, sizeof(float) * SEQ * nb);
            if (acc_rle) memcpy(acc_rle + (size_t)b * SEQ * nr, ar, sizeof(float) * SEQ * nr);
        }
        free(xf);
        free(y1);
        free(y2);
        free(gi);
        free(gh);
        free(lb);
        free(lr);
        free(ab);
        free(ar);
        free(hid);
    }
    net_free(&n);
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
