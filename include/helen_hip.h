/*
 * helen_hip.h -- C ABI of libhelen_hip.so, the MI355X (gfx950) implementation of the
 * `helen polish` / `call_consensus` RNN inference path of kishwarshafin/helen.
 *
 * Every entry point replaces one Python-level interface of the reference; citations are
 * file:line into the reference tree.  The library is plain HIP (no torch types): all buffers are
 * raw pointers, `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 * Conventions
 *   - every function returns HELEN_OK (0) or a negative HELEN_E* code and never throws;
 *     helen_last_error() returns a thread-local, NUL-terminated description of the last failure.
 *   - a HelenModel is bound to one device (one process per GPU, `models/predict_gpu.py:223`);
 *     calls on one handle must be serialised by the caller (one host thread / one stream at a time):
 *     a second thread entering a busy handle gets HELEN_EINVAL.
 *   - "window" = one MarginPolish pileup image, SEQ_LENGTH(1000) positions x features(90) uint8
 *     (`Options.py:13-21`); "chunk" = TRAIN_WINDOW(100) consecutive positions, stride
 *     WINDOW_JUMP(50), 19 per window (`Options.py:24-29`, `models/predict_gpu.py:114-117`).
 */
#ifndef HELEN_HIP_H
#define HELEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HELEN_ABI_VERSION 4

enum {
    HELEN_OK = 0,
    HELEN_EINVAL = -1,   /* bad argument (null pointer, size out of range, unsupported geometry) */
    HELEN_ENOMEM = -2,   /* device or host allocation failed */
    HELEN_EHIP = -3,     /* a HIP runtime call failed; see helen_last_error() */
    HELEN_ENODEV = -4    /* no usable gfx950 device / wrong architecture */
};

/* Geometry constants of the path (`Options.py:13-29`). */
#define HELEN_SEQ_LENGTH 1000
#define HELEN_TRAIN_WINDOW 100
#define HELEN_WINDOW_JUMP 50
#define HELEN_NUM_CHUNKS 19
#define HELEN_HIDDEN 128
#define HELEN_FEATURES 90
#define HELEN_BASE_LABELS 5
#define HELEN_RLE_LABELS 11

/*
 * Host-side view of the TransducerGRU parameter set, exactly the tensors of the reference
 * model's state_dict (`models/TransducerModel.py:43-58`; loaded by
 * `models/ModelHander.py:50-78`).  All pointers are HOST pointers to row-major float32.
 * Index 0 = forward direction (`*_l0`), 1 = reverse (`*_l0_reverse`).  Gate row order inside each
 * 3H block is PyTorch's r, z, n.
 */
typedef struct HelenWeights {
    int32_t features;          /* F: encoder input width, 90 */
    int32_t hidden;            /* H: GRU width, 128 */
    int32_t n_base;            /* 5 */
    int32_t n_rle;             /* 11 */
    const float* enc_w_ih[2];  /* [3H, F]   gru_encoder.weight_ih_l0{,_reverse} */
    const float* enc_w_hh[2];  /* [3H, H]   gru_encoder.weight_hh_l0{,_reverse} */
    const float* enc_b_ih[2];  /* [3H] */
    const float* enc_b_hh[2];  /* [3H] */
    const float* dec_w_ih[2];  /* [3H, 2H]  gru_decoder.weight_ih_l0{,_reverse} */
    const float* dec_w_hh[2];  /* [3H, H] */
    const float* dec_b_ih[2];  /* [3H] */
    const float* dec_b_hh[2];  /* [3H] */
    const float* base_w;       /* [n_base, 2H]  dense1_base.weight */
    const float* base_b;       /* [n_base] */
    const float* rle_w;        /* [n_rle, 2H]   dense2_rle.weight */
    const float* rle_b;        /* [n_rle] */
} HelenWeights;

typedef struct HelenModel HelenModel;

/* Arithmetic used for the GRU gate matmuls. */
enum {
    HELEN_PRECISION_FP32 = 0,  /* v_mfma_f32_16x16x4_f32, exact fp32 (BASELINE.json configs 1-3) */
    HELEN_PRECISION_BF16 = 1,  /* bf16 MFMA operands, fp32 accumulate/state (config 4); projections fused into
                                  the recurrence, no gate pre-activations in memory.  Gate weights and biases
                                  are scaled by the gates' exp2 factors before they are rounded to bf16; the
                                  heads take h and their weights as two bf16 terms each (16 significant bits) */
    HELEN_PRECISION_FP32X3 = 2 /* opt-in: gate matmuls as exact bf16 partial products (each fp32 operand =
                                  three bf16 terms, six leading products; pileup counts are one term),
                                  fp32 accumulate: fp32-class results on the bf16 matrix cores */
};

/* Kernel classes reported by helen_get_kernel_stats(). */
enum {
    HELEN_K_PACK = 0,        /* uint8 image -> fp32 MFMA operand tiles */
    HELEN_K_GEMM_ENC = 1,    /* encoder input projection  X.W_ih^T + b */
    HELEN_K_GRU_ENC = 2,     /* encoder recurrence (100 dependent steps, both directions) */
    HELEN_K_GEMM_DEC = 3,    /* decoder input projection  Y1.W_ih^T + b */
    HELEN_K_GRU_DEC = 4,     /* decoder recurrence */
    HELEN_K_HEADS = 5,       /* heads + softmax + accumulate + argmax (or + cross-entropy terms) */
    HELEN_K_COUNT = 6
};

/* ABI version of the loaded library (== HELEN_ABI_VERSION of the header it was built from). */
int helen_abi_version(void);

/* Description of the most recent failure on the calling thread ("" if none). */
const char* helen_last_error(void);

/*
 * Build a device-resident model: validates the geometry, packs the weights into the MFMA
 * fragment layouts the kernels consume and allocates scratch for up to `max_windows` windows per
 * call.  Replaces `ModelHandler.load_simple_model(...)` + `transducer_model.to(device_id)`
 * (`models/ModelHander.py:38-82`, `models/predict_gpu.py:58-69`).
 *   device       HIP device ordinal (`torch.cuda.set_device(device_id)`, predict_gpu.py:67)
 *   max_windows  capacity in windows of one helen_polish_batch / helen_gru_chunk_forward call
 *   precision    HELEN_PRECISION_*
 */
int helen_model_create(const HelenWeights* weights, int device, int max_windows, int precision,
                       HelenModel** out_model);

int helen_model_destroy(HelenModel* model);

/* Bytes of device memory the model holds (packed weights + scratch). */
int helen_model_device_bytes(const HelenModel* model, size_t* out_bytes);

/*
 * Which kernels a call takes is one table derived from the device's CU count (helen_amd/csrc/dispatch.h); every
 * choice gives the same bits.  The environment's A/B switches (HELEN_GRU_PAIR, HELEN_GRU_SINGLE8, HELEN_GRU_HALF8,
 * HELEN_GRU_QUARTER4, HELEN_DEC_WS, HELEN_DEC_WSP[_PARTS], HELEN_SPLIT[_AT],
 * HELEN_BF16_PAIR, HELEN_X3_PAIR, HELEN_HOST_LOCK, HELEN_VERBOSE = print the table) are read ONCE, when the model is
 * created; helen_reload_overrides reads them again for this model (tests and probes that flip one between calls).
 *   helen_describe_dispatch   the table for a device of `cus` compute units, as text (a dry run: no device needed)
 *   helen_plan_call           out[8] = split?, tiles of the first group, recurrence kernel, decoder projection, its
 *                             position runs, 0 (one encoder projection: gemm_enc_x3_kernel), its position runs, bf16
 *                             two-tile kernels?  (the enums of dispatch.h) for a call of `tiles` tiles on `cus` CUs
 */
int helen_reload_overrides(HelenModel* model);
int helen_describe_dispatch(int cus, char* out, size_t cap);
int helen_plan_call(int cus, int tiles, int* out);

/*
 * The whole per-batch body of the reference loop (`models/predict_gpu.py:97-159`): uint8 -> f32,
 * zero initial hidden, 19 chunks of TransducerGRU.forward with the hidden state carried
 * chunk-to-chunk, per-chunk softmax zero-padded and added into [n,1000,C] accumulators, argmax
 * (first maximum on ties, like torch.max on CPU, predict_gpu.py:155-156).
 *   images        DEVICE pointer, uint8 [n_windows, 1000, F] (what SequenceDataset yields per item,
 *                 `models/dataloader_predict.py:69`, already padded to 1000 positions)
 *   n_windows     1 .. max_windows; windows are independent, so several loader batches may be
 *                 coalesced into one call
 *   bases, rles   DEVICE pointers, uint8 [n_windows, 1000] (what DataStore stores, DataStore.py:126-133)
 *   acc_base_opt  optional DEVICE pointer float32 [n_windows, 1000, 5]  (prediction_base_tensor)
 *   acc_rle_opt   optional DEVICE pointer float32 [n_windows, 1000, 11] (prediction_rle_tensor)
 * Asynchronous with respect to the host: work is enqueued on `stream`.
 */
int helen_polish_batch(HelenModel* model, const uint8_t* images, int n_windows, uint8_t* bases,
                       uint8_t* rles, float* acc_base_opt, float* acc_rle_opt, void* stream);

/*
 * Same, from HOST memory: sub-batches of `max_windows` windows go up with hipMemcpyAsync on a copy
 * stream, overlapped with the kernels of the previous sub-batch and the label download of the one
 * before.  Page-locked caller memory (hipHostMalloc / hipHostRegister, e.g. a torch pinned tensor) is
 * the source and destination of the DMA itself (79.9 k windows/s); pageable memory goes through two pinned
 * mirrors the library owns (78.9 k) -- it is NOT page-locked in place by default: on this ROCm a registration maps the
 * caller's pages in place without a reference count, so unregistering a range takes GPU access away from every page
 * it shares with any other registration of the process (api.hip: helen_polish_host has the whole story;
 * $HELEN_HOST_LOCK=own | all at model creation restores the in-place rules of rounds 2-3.  Those two settings are
 * UNSUPPORTED, at your own risk: they re-open the exposure to the GPU memory-access fault described there, which was
 * never reproduced in isolation; accepted values are exactly none | own | all, anything else is ignored).
 * Synchronous: returns when the labels are in host memory; on an error nothing is left in flight.
 * Hand it MANY sub-batches per call: the first upload and the last download are the only exposed
 * copies.  Replaces the DataLoader -> `.to(device_id)` -> `.cpu()` hand-offs of
 * `models/predict_gpu.py:94-159`.
 */
int helen_polish_host(HelenModel* model, const uint8_t* images, int n_windows, uint8_t* bases,
                      uint8_t* rles, void* stream);

/*
 * The slot pipeline: the asynchronous form of helen_polish_host for callers whose buffers are page-locked and hold ONE device
 * call each -- the hand-off `helen polish` itself uses (helen_amd/predict.py: reader threads fill slot k+2 while slot k+1 goes
 * up, slot k is computed and slot k-1's labels come down; the "pinned hipMemcpyAsync double-buffering" of the loader -> HBM
 * step).  submit enqueues upload (copy stream), kernels (`stream`) and label download (second copy stream) of one slot and
 * returns; at most TWO slots are in flight; wait blocks until the OLDEST slot's labels are in its bases / rles.  Buffers:
 * page-locked host memory (helen_host_alloc, hipHostMalloc, a pinned torch tensor), n_windows <= max_windows.  After an
 * error nothing is in flight and the pipeline starts over.  Replaces `images.to(device_id)` ... `.cpu()` of
 * `models/predict_gpu.py:94-159` for a caller that never needs torch.
 *   helen_device_count  number of HIP devices visible (0 when there is none: never an error)
 *   helen_host_alloc    page-locked host memory of the runtime for `device` (hipHostMalloc); helen_host_free returns it
 */
int helen_device_count(int* out);
int helen_host_alloc(int device, size_t bytes, void** out);
int helen_host_free(void* p);
int helen_polish_slot_submit(HelenModel* model, const uint8_t* images, int n_windows, uint8_t* bases, uint8_t* rles,
                             void* stream);
int helen_polish_slot_wait(HelenModel* model);

/*
 * The queueing form of helen_polish_host, for a caller that holds ONE loader batch at a time (the reference's loop,
 * `models/predict_gpu.py:94-159`): a device call of 256 windows runs at 0.40 of the rate of one of 4,096 (a window is
 * 3,800 dependent GRU steps however few windows there are), so the library gathers.
 *   helen_polish_submit  copies the batch's images (host memory, pageable or not) into the library's pinned mirror and
 *                        returns; whenever `max_windows` windows have gathered, a device call goes out asynchronously
 *                        (upload, kernels on `stream`, label download; two in flight at most).  `bases` / `rles`
 *                        (host, uint8 [n_windows, 1000]) must stay valid until the flush: labels are written into them
 *                        by a LATER submit or by the flush, never before.  One stream per queue.
 *   helen_polish_flush   sends what is left and returns when the labels of every submitted batch are in place.
 * After an error nothing is in flight and nothing is pending (the queue is empty again).  helen_polish_host refuses
 * while submissions are pending (the staging ring is shared).
 */
int helen_polish_submit(HelenModel* model, const uint8_t* images, int n_windows, uint8_t* bases, uint8_t* rles,
                        void* stream);
int helen_polish_flush(HelenModel* model);

/*
 * Test hook for the error path of helen_polish_host: the NEXT call fails with HELEN_EHIP right after it has
 * enqueued sub-batch `sub_batch` (copies and kernels of that and earlier sub-batches are in flight at that
 * moment).  The call must still return with nothing in flight and the handle usable.  -1 disarms.
 * INERT IN PRODUCTION: returns HELEN_EINVAL unless the environment held HELEN_DEBUG_HOOKS=1 when the model was
 * created (the test suite sets it for the one test that needs it); refused while another thread is in a call.
 */
int helen_debug_inject_failure(HelenModel* model, int sub_batch);

/*
 * One TransducerGRU.forward call (`models/TransducerModel.py:60-79`), the operator-level
 * boundary invoked at `models/predict_gpu.py:129`:
 *   x      DEVICE float32 [B, T, F]           (T <= 100)
 *   h_in   DEVICE float32 [B, 2, H]           (index 0 forward, 1 backward)
 *   base   DEVICE float32 [B, T, 5]   logits  (dense1_base)
 *   rle    DEVICE float32 [B, T, 11]  logits  (dense2_rle)
 *   h_out  DEVICE float32 [B, 2, H]           decoder h_n
 */
int helen_gru_chunk_forward(HelenModel* model, const float* x, const float* h_in, int B, int T,
                            float* base, float* rle, float* h_out, void* stream);

/*
 * The per-batch body of the reference's evaluation loop on labeled images (`models/test.py:78-126`,
 * `helen_train test`): the same 19-chunk forward as helen_polish_batch, but each
 * chunk's logits go into nn.CrossEntropyLoss terms and torchnet-ConfusionMeter counts instead of the
 * softmax accumulators.
 *   images             DEVICE uint8 [n_windows, 1000, F]
 *   label_base         DEVICE uint8 [n_windows, 1000], values 0..4   (`models/dataloader.py:60`)
 *   label_rle          DEVICE uint8 [n_windows, 1000], values 0..10  (`models/dataloader.py:61`)
 *   rle_class_weights  HOST float[11]: TrainOptions.CLASS_WEIGHTS (`Options.py:29`, `models/test.py:55-58`)
 *   chunk_stats        DEVICE float32 [n_windows, 19, 10, 3], overwritten: for window, chunk and group of
 *                      10 consecutive chunk positions: sum of nll_base, sum of w[label]*nll_rle, sum of
 *                      w[label].  A loader batch's chunk losses (`models/test.py:108-113`) are
 *                      sum(nll_base)/(B*100) and sum(w*nll_rle)/sum(w) over its windows.
 *   base_confusion     DEVICE uint64 [5, 5],   ADDED to: [target][predicted], predicted = first-maximum
 *   rle_confusion      DEVICE uint64 [11, 11]  argmax of the chunk's logits (`models/test.py:116-119`)
 * Asynchronous on `stream`.  Labels out of range are the caller's error (torch raises on them).
 */
int helen_evaluate_batch(HelenModel* model, const uint8_t* images, const uint8_t* label_base,
                         const uint8_t* label_rle, int n_windows, const float* rle_class_weights,
                         float* chunk_stats, unsigned long long* base_confusion,
                         unsigned long long* rle_confusion, void* stream);

/*
 * Per-kernel-class timing with HIP events on the launch stream.  When enabled, each launch of a
 * class selected in `class_mask` (bit i = HELEN_K_*) is bracketed by an event pair; stats
 * accumulate until reset.  helen_get_kernel_stats synchronises the recorded events.
 */
int helen_set_profiling(HelenModel* model, unsigned class_mask);
int helen_reset_kernel_stats(HelenModel* model);
int helen_get_kernel_stats(HelenModel* model, int kernel_class, double* out_total_ms,
                           long long* out_launches);

#ifdef __cplusplus
}
#endif
#endif /* HELEN_HIP_H */
