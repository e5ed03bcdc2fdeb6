// kernels.h -- the gfx950 kernels of the polish path (fp32 arithmetic on v_mfma_f32_16x16x4_f32).
//
// Reference semantics being implemented (file:line into kishwarshafin/helen):
//   TransducerGRU.forward             helen/modules/python/models/TransducerModel.py:60-79
//   sliding window / softmax / argmax helen/modules/python/models/predict_gpu.py:97-159
// Layouts are described in layout.h; the launch sequence is in api.hip.
#pragma once
#include "kernels_common.h"
#include "kernels_pack.h"
#include "kernels_gemm.h"
#include "kernels_gru.h"
#include "kernels_gru_pair.h"
#include "kernels_gru_single8.h"
#include "kernels_gru_half8.h"
#include "kernels_x3.h"
#include "kernels_fused_bf16.h"
#include "kernels_fused_bf16_il.h"
#include "kernels_x3_il.h"
#include "kernels_heads.h"
