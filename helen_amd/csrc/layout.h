// layout.h -- data layouts shared by the kernels and the host-side weight packer.
//
// Everything on the device is organised around the operand/result fragments of
// v_mfma_f32_16x16x4_f32 (wave64): a "tile" is 16 pileup windows, the M dimension of every MFMA.
//
//   lane l:  i = j = l & 15,  q = l >> 4
//   A fragment (16 rows x 4 k):  lane holds A[row i][k = q]
//   B fragment (4 k x 16 cols):  lane holds B[k = q][col j]
//   C/D fragment (16 x 16):      lane holds D[row 4q + r][col j], r = 0..3
//
// K ordering.  A dot product may visit k in any order as long as A and B agree, so K is consumed
// in groups of 16: group m, sub-step e (0..3), lane-quarter q  <->  k = 16m + 4q + e.  A lane then
// needs 4 consecutive k per group, i.e. one 16-byte load, and the 64 lanes of a wave together
// read one contiguous 1 KiB block.
//
// KB16 operand layout of a [16 rows x K] fp32 matrix (K a multiple of 16):
//     float index = ((k >> 2) * 16 + row) * 4 + (k & 3)
//   as float4:  index (k >> 2) * 16 + row;  group m of lane l is float4 index 64 m + l.
//   Used for: packed images (xa), GRU outputs (y1, y2), the hidden state, and the LDS copy of h.
//
// FRAG result layout of a [16 rows x 16 cols] fp32 block:  float4 index = lane, component r
//   (row 4q + r, col j).  Used for the gate pre-activations gi and the pending softmax values.
#pragma once
#include <stdint.h>

namespace helen {

constexpr int kTile = 16;            // windows per tile (MFMA M)
constexpr int kSeq = 1000;           // positions per window            (Options.py:16)
constexpr int kWin = 100;            // chunk width                     (Options.py:25)
constexpr int kJump = 50;            // chunk stride                    (Options.py:26)
constexpr int kChunks = 19;          // predict_gpu.py:114-117
constexpr int kH = 128;              // hidden size                     (Options.py:28)
constexpr int kG = 3 * kH;           // gate rows r,z,n
constexpr int kF = 90;               // features                        (Options.py:14)
constexpr int kFPad = 96;            // features padded to a multiple of 16
constexpr int kNB = 5;               // base labels
constexpr int kNR = 11;              // run-length labels
constexpr int kNTile = kG / 16;      // 24 column tiles of 16 per direction

// floats per (tile, position) of each buffer
constexpr int kXaStride = kFPad * kTile;        // 1536  packed image operand
constexpr int kYStride = 2 * kH * kTile;        // 4096  [fwd 128 | bwd 128] x 16 windows
constexpr int kGiDirStride = kG * kTile;        // 6144  one direction's 384 gate columns
constexpr int kGiStride = 2 * kGiDirStride;     // 12288 both directions
constexpr int kHidDirStride = kH * kTile;       // 2048  one direction's hidden state
constexpr int kHidStride = 2 * kHidDirStride;   // 4096

__host__ __device__ inline int kb16_index(int row, int k) {
    return ((k >> 2) * kTile + row) * 4 + (k & 3);
}

}  // namespace helen
