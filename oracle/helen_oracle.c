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

/* Arithmetic of the gate matmuls: 0 = fp32 (the reference); 1, 2 = operands rounded to bf16 (RNE), fp32
 * accumulate/state -- what the bf16 HIP variant (BASELINE.json configs[3]) is checked against.  Heads, gates and
 * softmax stay fp32 in all of them.
 *   2 = the TEXTBOOK form, written without looking at the kernels: W_ih, W_hh, x and h rounded to bf16 as they
 *       stand, fp32 accumulate, fp32 biases, sigmoid / tanh as in the reference.  This is the SPECIFICATION of
 *       "bf16 gate GEMMs" (DESIGN.md 5).
 *   1 = the same with the kernels' own operand preparation (weights and biases prescaled by -log2 e / 2 log2 e
 *       BEFORE the rounding, gates on exp2): a tight implementation check -- it follows the kernel, so agreeing
 *       with it says the kernel does what it means to, not that what it means to do is right.  That is mode 2's job. */
static int g_precision = 0;
void oracle_set_precision(int p) { g_precision = (p == 1 || p == 2) ? p : 0; }
int oracle_get_precision(void) { return g_precision; }

static inline float bf16r(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

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
    float* b_own;  /* bf16 emulation: the prescaled copies b_ih / b_hh point into */
} Dir;

/* bf16 emulation, as the HIP bf16 mode does it since round 4 (helen_amd/csrc/api.hip gate_prescale, kernels_gru.h
 * gru_cell2_pre): the rows of the r and z gates are multiplied by -log2(e) and those of the n gate by 2 log2(e) in fp32
 * BEFORE the rounding to bf16, the biases likewise, and the gates are evaluated on exp2 of the accumulators. */
static inline float gate_prescale(int row, int H) { return row < 2 * H ? -1.4426950408889634f : 2.8853900817779268f; }

static void dir_init(Dir* d, const float* w_ih, const float* w_hh, const float* b_ih,
                     const float* b_hh, int K, int H) {
    d->K = K;
    d->H = H;
    d->w_ih_t = transpose_(w_ih, 3 * H, K);
    d->w_hh_t = transpose_(w_hh, 3 * H, H);
    d->b_ih = b_ih;
    d->b_hh = b_hh;
    d->b_own = NULL;
    if (g_precision == 2) { /* textbook: the operands as they stand, rounded */
        const int G = 3 * H;
        for (size_t i = 0; i < (size_t)K * G; ++i) d->w_ih_t[i] = bf16r(d->w_ih_t[i]);
        for (size_t i = 0; i < (size_t)H * G; ++i) d->w_hh_t[i] = bf16r(d->w_hh_t[i]);
    }
    if (g_precision == 1) {
        const int G = 3 * H;
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < G; ++j) d->w_ih_t[(size_t)k * G + j] = bf16r(d->w_ih_t[(size_t)k * G + j] * gate_prescale(j, H));
        for (int k = 0; k < H; ++k)
            for (int j = 0; j < G; ++j) d->w_hh_t[(size_t)k * G + j] = bf16r(d->w_hh_t[(size_t)k * G + j] * gate_prescale(j, H));
        /* the kernels add b_ih + b_hh of the r and z gates first, then scale the sum; the n gate's two biases separately */
        d->b_own = (float*)malloc(sizeof(float) * 2 * (size_t)G);
        for (int j = 0; j < G; ++j) {
            if (j < 2 * H) {
                d->b_own[j] = (b_ih[j] + b_hh[j]) * gate_prescale(j, H);
                d->b_own[G + j] = 0.f;
            } else {
                d->b_own[j] = b_ih[j] * gate_prescale(j, H);
                d->b_own[G + j] = b_hh[j] * gate_prescale(j, H);
            }
        }
        d->b_ih = d->b_own;
        d->b_hh = d->b_own + G;
    }
}
static void dir_free(Dir* d) {
    free(d->w_ih_t);
    free(d->w_hh_t);
    free(d->b_own);
}

/* One GRU direction over T steps for a block of nb <= OB windows (windows never interact; the
 * block only lets one pass over the weights serve several windows).
 *   x[b]: [T, K] (row stride xs), h[b]: [H] in/out, y[b]: [T, ys], output at y[t*ys + yoff .. +H)
 *   reverse: process t = T-1 .. 0 (nn.GRU `_reverse` weights), output stays at index t.
 *   Per element the k-sum order is fixed (k ascending, then + bias), independent of nb. */
#define OB 8
static void gru_dir(const Dir* d, const float* const* x, int xs, int T, int reverse, float* const* h,
                    float* const* y, int ys, int yoff, int nb, float* gi, float* gh) {
    const int H = d->H, G = 3 * d->H, K = d->K;
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? (T - 1 - s) : s;
        memset(gi, 0, sizeof(float) * (size_t)nb * G);
        memset(gh, 0, sizeof(float) * (size_t)nb * G);
        for (int k = 0; k < K; ++k) {
            const float* w = d->w_ih_t + (size_t)k * G;
            for (int b = 0; b < nb; ++b) {
                const float xr = x[b][(size_t)t * xs + k];
                const float xv = g_precision ? bf16r(xr) : xr;
                float* g = gi + (size_t)b * G;
                for (int j = 0; j < G; ++j) g[j] += xv * w[j];
            }
        }
        for (int k = 0; k < H; ++k) {
            const float* w = d->w_hh_t + (size_t)k * G;
            for (int b = 0; b < nb; ++b) {
                const float hv = g_precision ? bf16r(h[b][k]) : h[b][k];
                float* g = gh + (size_t)b * G;
                for (int j = 0; j < G; ++j) g[j] += hv * w[j];
            }
        }
        for (int b = 0; b < nb; ++b) {
            float* a = gi + (size_t)b * G;
            float* c = gh + (size_t)b * G;
            for (int j = 0; j < G; ++j) {
                a[j] += d->b_ih[j];
                c[j] += d->b_hh[j];
            }
            float* hb = h[b];
            for (int j = 0; j < H; ++j) {
                if (g_precision == 1) { /* prescaled: the accumulators are the arguments of exp2 */
                    const float r = 1.0f / (1.0f + exp2f(a[j] + c[j]));
                    const float z = 1.0f / (1.0f + exp2f(a[H + j] + c[H + j]));
                    const float n = 1.0f - 2.0f / (1.0f + exp2f(a[2 * H + j] + r * c[2 * H + j]));
                    hb[j] = n + z * (hb[j] - n);
                    continue;
                }
                const float r = sigmoidf_(a[j] + c[j]);
                const float z = sigmoidf_(a[H + j] + c[H + j]);
                const float n = tanhf(a[2 * H + j] + r * c[2 * H + j]);
                hb[j] = (1.0f - z) * n + z * hb[j];
            }
            memcpy(y[b] + (size_t)t * ys + yoff, hb, sizeof(float) * H);
        }
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

/* TransducerGRU.forward for a block of nb windows (`models/TransducerModel.py:60-79`).
 *   x[b] [T,F], hidden[b] [2,H] in/out, base[b] [T,nb_], rle[b] [T,nr];
 *   scratch y1[b], y2[b] [T,2H]; gi, gh [OB*3H]. */
static void forward_block(const Net* n, const float* const* x, int T, float* const* hidden,
                          float* const* base, float* const* rle, float* const* y1,
                          float* const* y2, int nb, float* gi, float* gh) {
    const int H = n->H;
    float* hb[OB];
    const float* y1c[OB];
    for (int b = 0; b < nb; ++b) {
        hb[b] = hidden[b] + H;
        y1c[b] = y1[b];
    }
    /* encoder: h0 = incoming hidden (index 0 forward, 1 backward) */
    gru_dir(&n->enc[0], x, n->F, T, 0, hidden, y1, 2 * H, 0, nb, gi, gh);
    gru_dir(&n->enc[1], x, n->F, T, 1, hb, y1, 2 * H, H, nb, gi, gh);
    /* decoder: h0 = encoder h_n, input = [h_fwd(t) | h_bwd(t)] */
    gru_dir(&n->dec[0], y1c, 2 * H, T, 0, hidden, y2, 2 * H, 0, nb, gi, gh);
    gru_dir(&n->dec[1], y1c, 2 * H, T, 1, hb, y2, 2 * H, H, nb, gi, gh);
    /* heads (`TransducerModel.py:75-76`) */
    for (int b = 0; b < nb; ++b)
        for (int t = 0; t < T; ++t) {
            const float* y = y2[b] + (size_t)t * 2 * H;
            for (int c = 0; c < n->nb; ++c) {
                float a = 0.0f;
                const float* w = n->base_w + (size_t)c * 2 * H;
                for (int k = 0; k < 2 * H; ++k) a += y[k] * w[k];
                base[b][(size_t)t * n->nb + c] = a + n->base_b[c];
            }
            for (int c = 0; c < n->nr; ++c) {
                float a = 0.0f;
                const float* w = n->rle_w + (size_t)c * 2 * H;
                for (int k = 0; k < 2 * H; ++k) a += y[k] * w[k];
                rle[b][(size_t)t * n->nr + c] = a + n->rle_b[c];
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
    const int nblk = (B + OB - 1) / OB;
#pragma omp parallel
    {
        float* y1 = (float*)malloc(sizeof(float) * (size_t)OB * T * 2 * H);
        float* y2 = (float*)malloc(sizeof(float) * (size_t)OB * T * 2 * H);
        float* gi = (float*)malloc(sizeof(float) * OB * 3 * H);
        float* gh = (float*)malloc(sizeof(float) * OB * 3 * H);
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblk; ++blk) {
            const int b0 = blk * OB;
            const int nb = (B - b0 < OB) ? B - b0 : OB;
            const float* xp[OB];
            float *hp[OB], *bp[OB], *rp[OB], *y1p[OB], *y2p[OB];
            for (int b = 0; b < nb; ++b) {
                hp[b] = h_out + (size_t)(b0 + b) * 2 * H;
                memcpy(hp[b], h_in + (size_t)(b0 + b) * 2 * H, sizeof(float) * 2 * H);
                xp[b] = x + (size_t)(b0 + b) * T * n.F;
                bp[b] = base + (size_t)(b0 + b) * T * n.nb;
                rp[b] = rle + (size_t)(b0 + b) * T * n.nr;
                y1p[b] = y1 + (size_t)b * T * 2 * H;
                y2p[b] = y2 + (size_t)b * T * 2 * H;
            }
            forward_block(&n, xp, T, hp, bp, rp, y1p, y2p, nb, gi, gh);
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
    const int nblk = (B + OB - 1) / OB;
#pragma omp parallel
    {
        float* xf = (float*)malloc(sizeof(float) * (size_t)OB * SEQ * F);
        float* y1 = (float*)malloc(sizeof(float) * (size_t)OB * WIN * 2 * H);
        float* y2 = (float*)malloc(sizeof(float) * (size_t)OB * WIN * 2 * H);
        float* gi = (float*)malloc(sizeof(float) * OB * 3 * H);
        float* gh = (float*)malloc(sizeof(float) * OB * 3 * H);
        float* lb = (float*)malloc(sizeof(float) * (size_t)OB * WIN * nb);
        float* lr = (float*)malloc(sizeof(float) * (size_t)OB * WIN * nr);
        float* ab = (float*)malloc(sizeof(float) * (size_t)OB * SEQ * nb);
        float* ar = (float*)malloc(sizeof(float) * (size_t)OB * SEQ * nr);
        float* hid = (float*)malloc(sizeof(float) * OB * 2 * H);
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblk; ++blk) {
            const int b0 = blk * OB;
            const int m = (B - b0 < OB) ? B - b0 : OB;
            const float* xp[OB];
            float *hp[OB], *bp[OB], *rp[OB], *y1p[OB], *y2p[OB];
            for (int b = 0; b < m; ++b) {
                const uint8_t* img = images + (size_t)(b0 + b) * SEQ * F;
                float* xb = xf + (size_t)b * SEQ * F;
                for (size_t i = 0; i < (size_t)SEQ * F; ++i) xb[i] = (float)img[i]; /* :97 */
                hp[b] = hid + (size_t)b * 2 * H;
                bp[b] = lb + (size_t)b * WIN * nb;
                rp[b] = lr + (size_t)b * WIN * nr;
                y1p[b] = y1 + (size_t)b * WIN * 2 * H;
                y2p[b] = y2 + (size_t)b * WIN * 2 * H;
            }
            memset(hid, 0, sizeof(float) * OB * 2 * H);                 /* :99 */
            memset(ab, 0, sizeof(float) * (size_t)OB * SEQ * nb);       /* :105 */
            memset(ar, 0, sizeof(float) * (size_t)OB * SEQ * nr);       /* :106 */
            int c = 0;
            for (int i = 0; i < SEQ; i += JUMP) { /* :114-117 */
                if (i + WIN > SEQ) break;
                for (int b = 0; b < m; ++b) xp[b] = xf + (size_t)b * SEQ * F + (size_t)i * F;
                forward_block(&n, xp, WIN, hp, bp, rp, y1p, y2p, m, gi, gh); /* :129 */
                for (int b = 0; b < m; ++b) {
                    const size_t wb = (size_t)(b0 + b);
                    for (int t = 0; t < WIN; ++t) { /* :137-149 */
                        softmax_add(bp[b] + (size_t)t * nb, nb,
                                    ab + ((size_t)b * SEQ + i + t) * nb);
                        softmax_add(rp[b] + (size_t)t * nr, nr,
                                    ar + ((size_t)b * SEQ + i + t) * nr);
                    }
                    if (hidden_trace)
                        memcpy(hidden_trace + ((size_t)c * B + wb) * 2 * H, hp[b],
                               sizeof(float) * 2 * H);
                    if (logit_base_trace)
                        memcpy(logit_base_trace + ((size_t)c * B + wb) * WIN * nb, bp[b],
                               sizeof(float) * WIN * nb);
                    if (logit_rle_trace)
                        memcpy(logit_rle_trace + ((size_t)c * B + wb) * WIN * nr, rp[b],
                               sizeof(float) * WIN * nr);
                }
                ++c;
            }
            for (int b = 0; b < m; ++b) {
                const size_t wb = (size_t)(b0 + b);
                const float* abb = ab + (size_t)b * SEQ * nb;
                const float* arb = ar + (size_t)b * SEQ * nr;
                for (int p = 0; p < SEQ; ++p) { /* :155-156 */
                    bases[wb * SEQ + p] = argmax_first(abb + (size_t)p * nb, nb);
                    rles[wb * SEQ + p] = argmax_first(arb + (size_t)p * nr, nr);
                }
                if (acc_base) memcpy(acc_base + wb * SEQ * nb, abb, sizeof(float) * SEQ * nb);
                if (acc_rle) memcpy(acc_rle + wb * SEQ * nr, arb, sizeof(float) * SEQ * nr);
            }
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

/* ---- float64 arbiter ------------------------------------------------------------------------
 * The same path (`models/TransducerModel.py:60-79`, `models/predict_gpu.py:97-159`) evaluated in
 * double precision end to end: fp32 parameters and uint8 inputs converted exactly, every product,
 * sum, exp() and tanh() in IEEE double (libm).  Its accumulated softmax is correct to ~1e-13 (3,800
 * dependent steps x 2^-53), seven orders of magnitude below fp32's own rounding, so it can say which of
 * two fp32 implementations that disagree on an argmax label is "right" -- or that the true margin is
 * below fp32 resolution and neither is.  Used by the tests only where labels differ (a handful of
 * windows): plain loops, one window at a time, OpenMP over windows.
 *   images [B,1000,F] u8 -> acc_base [B,1000,nb] f64, acc_rle [B,1000,nr] f64 */
static void gru_dir_f64(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int K, int H,
                        const double* x, int xs, int T, int reverse, double* h, double* y, int ys, int yoff) {
    double gi[3 * 512], gh[3 * 512];
    const int G = 3 * H;
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? (T - 1 - s) : s;
        const double* xt = x + (size_t)t * xs;
        for (int j = 0; j < G; ++j) {
            double a = 0.0, c = 0.0;
            const float* wi = w_ih + (size_t)j * K;
            const float* wh = w_hh + (size_t)j * H;
            for (int k = 0; k < K; ++k) a += xt[k] * (double)wi[k];
            for (int k = 0; k < H; ++k) c += h[k] * (double)wh[k];
            gi[j] = a + (double)b_ih[j];
            gh[j] = c + (double)b_hh[j];
        }
        for (int j = 0; j < H; ++j) {
            const double r = 1.0 / (1.0 + exp(-(gi[j] + gh[j])));
            const double z = 1.0 / (1.0 + exp(-(gi[H + j] + gh[H + j])));
            const double n = tanh(gi[2 * H + j] + r * gh[2 * H + j]);
            h[j] = (1.0 - z) * n + z * h[j];
        }
        memcpy(y + (size_t)t * ys + yoff, h, sizeof(double) * H);
    }
}

static void softmax_add_f64(const double* logits, int C, double* acc) {
    double m = logits[0], e[16], s = 0.0;
    for (int c = 1; c < C; ++c)
        if (logits[c] > m) m = logits[c];
    for (int c = 0; c < C; ++c) {
        e[c] = exp(logits[c] - m);
        s += e[c];
    }
    for (int c = 0; c < C; ++c) acc[c] += e[c] / s;
}

int oracle_polish_batch_f64(const HelenWeights* w, const uint8_t* images, int B, double* acc_base, double* acc_rle) {
    if (!w || !images || !acc_base || !acc_rle || B < 0) return -1;
    const int H = w->hidden, F = w->features, nb = w->n_base, nr = w->n_rle;
    if (H <= 0 || H > 512 || F <= 0 || nb > 16 || nr > 16) return -1;
#pragma omp parallel
    {
        double* xf = (double*)malloc(sizeof(double) * (size_t)SEQ * F);
        double* y1 = (double*)malloc(sizeof(double) * (size_t)WIN * 2 * H);
        double* y2 = (double*)malloc(sizeof(double) * (size_t)WIN * 2 * H);
        double* hid = (double*)malloc(sizeof(double) * 2 * H);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const uint8_t* img = images + (size_t)b * SEQ * F;
            for (size_t i = 0; i < (size_t)SEQ * F; ++i) xf[i] = (double)img[i];
            double* ab = acc_base + (size_t)b * SEQ * nb;
            double* ar = acc_rle + (size_t)b * SEQ * nr;
            memset(hid, 0, sizeof(double) * 2 * H);
            memset(ab, 0, sizeof(double) * (size_t)SEQ * nb);
            memset(ar, 0, sizeof(double) * (size_t)SEQ * nr);
            for (int i = 0; i + WIN <= SEQ; i += JUMP) {
                const double* x = xf + (size_t)i * F;
                gru_dir_f64(w->enc_w_ih[0], w->enc_w_hh[0], w->enc_b_ih[0], w->enc_b_hh[0], F, H, x, F, WIN, 0, hid,
                            y1, 2 * H, 0);
                gru_dir_f64(w->enc_w_ih[1], w->enc_w_hh[1], w->enc_b_ih[1], w->enc_b_hh[1], F, H, x, F, WIN, 1,
                            hid + H, y1, 2 * H, H);
                gru_dir_f64(w->dec_w_ih[0], w->dec_w_hh[0], w->dec_b_ih[0], w->dec_b_hh[0], 2 * H, H, y1, 2 * H, WIN,
                            0, hid, y2, 2 * H, 0);
                gru_dir_f64(w->dec_w_ih[1], w->dec_w_hh[1], w->dec_b_ih[1], w->dec_b_hh[1], 2 * H, H, y1, 2 * H, WIN,
                            1, hid + H, y2, 2 * H, H);
                for (int t = 0; t < WIN; ++t) {
                    const double* y = y2 + (size_t)t * 2 * H;
                    double lb[16], lr[16];
                    for (int c = 0; c < nb; ++c) {
                        double a = 0.0;
                        for (int k = 0; k < 2 * H; ++k) a += y[k] * (double)w->base_w[(size_t)c * 2 * H + k];
                        lb[c] = a + (double)w->base_b[c];
                    }
                    for (int c = 0; c < nr; ++c) {
                        double a = 0.0;
                        for (int k = 0; k < 2 * H; ++k) a += y[k] * (double)w->rle_w[(size_t)c * 2 * H + k];
                        lr[c] = a + (double)w->rle_b[c];
                    }
                    softmax_add_f64(lb, nb, ab + (size_t)(i + t) * nb);
                    softmax_add_f64(lr, nr, ar + (size_t)(i + t) * nr);
                }
            }
        }
        free(xf);
        free(y1);
        free(y2);
        free(hid);
    }
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
