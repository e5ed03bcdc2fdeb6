// Developer microbenchmark: (1) issue cost of v_mfma_f32_16x16x16_bf16 vs v_mfma_f32_16x16x32_bf16;
// (2) does VALU / transcendental work of a co-resident wave overlap with bf16 MFMAs (it does not with
// the fp32 MFMA, see mfma_valu_split.hip)?  8 waves per workgroup: waves 0-3 MFMA only, waves 4-7 VALU only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

template <int K32, int NV>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int steps) {
    const int wave = threadIdx.x >> 6;
    float v = in[threadIdx.x];
    if (wave < 4) {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = (f32x4){v, 0, 0, 0};
        s16x4 a4 = *(const s16x4*)(in + threadIdx.x * 2), b4 = *(const s16x4*)(in + 4096 + threadIdx.x * 2);
        b16x8 a8 = *(const b16x8*)(in + threadIdx.x * 4), b8 = *(const b16x8*)(in + 8192 + threadIdx.x * 4);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < 192; ++i) {
                if (K32) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i % 6], 0, 0, 0);
                else acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i % 6], 0, 0, 0);
            }
        }
        v = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][1];
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = v + i;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                x[i % 8] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i % 8] * 1.4426f)) + 0.25f;
        }
        for (int i = 0; i < 8; ++i) v += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = v;
}

template <int K32, int NV>
void run(const float* in, float* out) {
    const int steps = 400, grid = 256;
    hipLaunchKernelGGL((k<K32, NV>), dim3(grid), dim3(512), 0, 0, in, out, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K32, NV>), dim3(grid), dim3(512), 0, 0, in, out, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * (K32 ? 32 : 16) * 192.0 * steps * 4.0 * grid;
    const double cyc = ms * 1e-3 * 2.3e9 / (192.0 * steps);
    printf("K=%d  valu pairs/step %3d: %.3f ms  %.0f TFLOP/s  ~%.1f cycles/MFMA @2.3GHz\n", K32 ? 32 : 16, NV, ms,
           flop / (ms * 1e-3) / 1e12, cyc);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24);
    hipMemset(in, 0, 1 << 24);
    run<0, 0>(in, out); run<1, 0>(in, out);
    run<0, 48>(in, out); run<0, 96>(in, out);
    run<1, 48>(in, out); run<1, 96>(in, out); run<1, 192>(in, out);
    return 0;
}
