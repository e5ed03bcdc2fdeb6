/*
 * helen_cpu.h -- C ABI of libhelen_cpu.so: the `helen polish` inference path on the host, for runs WITHOUT --gpu_mode.
 *
 * The reference's CPU mode is an ONNX Runtime session of the same TransducerGRU driven by the same 19-chunk loop
 * (`models/predict_cpu.py:39-170`, chosen by `CallConsensusInterface.py:131,152`).  This is that mode for this package:
 * plain C++ / OpenMP (helen_amd/csrc/cpu_path.cpp), the product's own code -- it shares nothing with oracle/, which is
 * test infrastructure -- and the same arithmetic as libhelen_hip.so statement for statement.  It is NOT a fallback: the
 * MI355X path never routes here; `--gpu_mode` without a GPU still fails loudly.
 *
 * Plain pointers and sizes, host memory, no torch types; 0 on success, -1 on error (helen_cpu_last_error() describes it,
 * thread-local); nothing throws across the ABI.  The weights struct is include/helen_hip.h's HelenWeights (host
 * pointers, the state_dict's own tensors).
 */
#ifndef HELEN_CPU_H
#define HELEN_CPU_H

#include "helen_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HELEN_CPU_ABI_VERSION 1

int helen_cpu_abi_version(void);
const char* helen_cpu_last_error(void);

/*
 * The per-batch body of the reference loop (`models/predict_cpu.py:93-159`) for `n_windows` windows in host memory:
 * uint8 -> float, zero hidden, 19 chunks of 100 positions at stride 50 through TransducerGRU.forward with the hidden
 * state carried, softmax of each chunk added into [1000, C] accumulators, argmax (first maximum).
 *   images  uint8 [n_windows, 1000, 90]     bases, rles  uint8 [n_windows, 1000]
 *   acc_base_opt / acc_rle_opt  optional float32 [n_windows, 1000, 5] / [n_windows, 1000, 11]
 *   threads  OpenMP threads over blocks of 16 windows (<= 0: all the runtime offers) -- `threads_per_caller` of
 *            `CallConsensusInterface.py:131`
 */
int helen_cpu_polish_batch(const HelenWeights* weights, const uint8_t* images, int n_windows, uint8_t* bases,
                           uint8_t* rles, float* acc_base_opt, float* acc_rle_opt, int threads);

/*
 * TransducerGRU.forward (`models/TransducerModel.py:60-79`; the ONNX graph of `predict_cpu.py:228-239`):
 * x float32 [B, T, 90], h_in float32 [B, 2, 128] -> base [B, T, 5], rle [B, T, 11], h_out [B, 2, 128]; T <= 100.
 */
int helen_cpu_chunk_forward(const HelenWeights* weights, const float* x, const float* h_in, int B, int T, float* base,
                            float* rle, float* h_out, int threads);

#ifdef __cplusplus
}
#endif
#endif /* HELEN_CPU_H */
