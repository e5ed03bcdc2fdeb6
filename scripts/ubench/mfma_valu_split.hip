// Developer microbenchmark: does a VALU/transcendental-only wave slow an MFMA-only wave on the same
// SIMD?  Workgroup of 8 waves (2 per SIMD): waves 0-3 issue MFMAs only, waves 4-7 run NV dependent
// exp+rcp pairs per "step" (or nothing).  One workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int INDEP>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int steps) {
    const int wave = threadIdx.x >> 6;
    float v = in[threadIdx.x];
    if (wave < 4) {
        float b[160];
        for (int i = 0; i < 160; ++i) b[i] = in[threadIdx.x + 256 * i];
        f32x4 a4[8];
        for (int i = 0; i < 8; ++i) a4[i] = ((const f32x4*)in)[threadIdx.x + 64 * i];
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = (f32x4){v, 0, 0, 0};
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < 192; ++i)
                acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(i / 24) % 8][(i / 6) % 4], b[i % 160], acc[i % 6], 0, 0, 0);
        }
        v = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][1];
    } else {
        float x[INDEP];
        for (int i = 0; i < INDEP; ++i) x[i] = v + i;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                x[i % INDEP] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i % INDEP] * 1.4426f)) + 0.25f;
        }
        for (int i = 0; i < INDEP; ++i) v += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = v;
}

template <int NV, int INDEP>
void run(const float* in, float* out) {
    const int steps = 200, grid = 256;
    hipLaunchKernelGGL((k<NV, INDEP>), dim3(grid), dim3(512), 0, 0, in, out, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, INDEP>), dim3(grid), dim3(512), 0, 0, in, out, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = 2.0 * 16 * 16 * 4 * 192.0 * steps * 4.0 * grid / (ms * 1e-3) / 1e12;
    printf("valu-wave pairs/step %3d (indep chains %d): %.3f ms  MFMA %.1f TFLOP/s  %.1f%%\n", NV, INDEP, ms, tf, tf / 1.573);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24);
    hipMemset(in, 0, 1 << 24);
    run<0, 1>(in, out); run<48, 1>(in, out); run<96, 1>(in, out); run<192, 1>(in, out);
    run<96, 8>(in, out); run<192, 8>(in, out); run<384, 8>(in, out);
    return 0;
}
