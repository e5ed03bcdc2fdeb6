// kernels_pack.h -- uint8 windows / fp32 chunks / hidden state -> MFMA operand tiles
#pragma once
#include "kernels_common.h"

namespace helen {

// (uint8 pileup windows go straight to bf16 A fragments: pack_images_x3_kernel, kernels_x3.h -- the reference's host-side
// `images.type(torch.FloatTensor)` of predict_gpu.py:97 is exact in bf16 for counts 0..255)

// float32 x [B, T, F] (the operator-level boundary, TransducerModel.py:60) -> KB16 fp32 operand tiles
// xa[tile][pos][kb 24][16][4]; rows of windows past n_windows and features past F are zero.
__global__ __launch_bounds__(256) void pack_x_f32_kernel(const float* __restrict__ x, int n_windows,
                                                         int T, f32x4* __restrict__ xa,
                                                         long xa_tile_stride) {
    const int tile = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int per_pos = (kFPad / 4) * kTile;
    if (g >= T * per_pos) return;
    const int i = g & 15;
    const int kb = (g >> 4) % (kFPad / 4);
    const int pos = g / per_pos;
    const int window = tile * kTile + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (window < n_windows) {
        const float* p = x + ((size_t)window * T + pos) * kF + kb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kb * 4 + e < kF) v[e] = p[e];
    }
    xa[(size_t)tile * xa_tile_stride + (size_t)pos * per_pos + kb * kTile + i] = v;
}

// hidden [B, 2, H] (TransducerModel.py:68 transposes it to [2, B, H]) <-> KB16 state
// hid[tile][dir][kb 32][16][4].
__global__ __launch_bounds__(256) void pack_hidden_kernel(const float* __restrict__ h, int n_windows,
                                                          float* __restrict__ hid) {
    const int tile = blockIdx.x;
    for (int g = threadIdx.x; g < kHidStride; g += 256) {
        const int dir = g / kHidDirStride;
        const int rem = g % kHidDirStride;
        const int k = (rem >> 6) * 4 + (rem & 3);
        const int row = (rem >> 2) & 15;
        const int window = tile * kTile + row;
        hid[(size_t)tile * kHidStride + g] =
            window < n_windows ? h[((size_t)window * 2 + dir) * kH + k] : 0.f;
    }
}
__global__ __launch_bounds__(256) void unpack_hidden_kernel(const float* __restrict__ hid,
                                                            int n_windows, float* __restrict__ h) {
    const int tile = blockIdx.x;
    for (int g = threadIdx.x; g < kHidStride; g += 256) {
        const int dir = g / kHidDirStride;
        const int rem = g % kHidDirStride;
        const int k = (rem >> 6) * 4 + (rem & 3);
        const int row = (rem >> 2) & 15;
        const int window = tile * kTile + row;
        if (window < n_windows)
            h[((size_t)window * 2 + dir) * kH + k] = hid[(size_t)tile * kHidStride + g];
    }
}

}  // namespace helen
