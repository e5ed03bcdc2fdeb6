// Developer microbenchmark: v_mfma_f32_16x16x32_bf16 issue rate with the operand pattern of
// gemm_dec_x3_kernel (6 A fragments x 12 B fragments -> 8 accumulators, 48 MFMAs per group),
// one or two waves per SIMD, versus the same-operand loop of mfma_bf16_valu.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(const float* in, float* out, int steps) {
    b16x8 a[2][3], b[4][3];
    for (int p = 0; p < 2; ++p) for (int t = 0; t < 3; ++t) a[p][t] = *(const b16x8*)(in + (threadIdx.x + 64 * (p * 3 + t)) * 4);
    for (int n = 0; n < 4; ++n) for (int t = 0; t < 3; ++t) b[n][t] = *(const b16x8*)(in + 8192 + (threadIdx.x + 64 * (n * 3 + t)) * 4);
    f32x4 acc[2][4];
    for (int p = 0; p < 2; ++p) for (int n = 0; n < 4; ++n) acc[p][n] = (f32x4){0, 0, 0, 0};
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (MODE == 0) acc[p][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][TA[k6]], b[n][TB[k6]], acc[p][n], 0, 0, 0);
                    else acc[p][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], b[0][0], acc[p][n], 0, 0, 0);
                }
        // keep the operands "live and changing" so nothing is hoisted
        asm volatile("" : "+v"(a[0][0]), "+v"(b[0][0]));
    }
    f32x4 r = acc[0][0];
    for (int p = 0; p < 2; ++p) for (int n = 0; n < 4; ++n) r += acc[p][n];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

template <int WAVES, int MODE>
void run(const float* in, float* out) {
    const int steps = 2000, grid = 256;
    hipLaunchKernelGGL((k<WAVES, MODE>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, MODE>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, steps);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 48.0 * steps * (WAVES / 4);
    printf("waves/SIMD %d  %s operands: %.3f ms  %.1f cycles per MFMA per SIMD @2.3GHz  (%.0f TFLOP/s)\n", WAVES / 4,
           MODE ? "same" : "distinct", ms, ms * 1e-3 * 2.3e9 / per_simd,
           2.0 * 16 * 16 * 32 * 48.0 * steps * WAVES * grid / (ms * 1e-3) / 1e12);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24); hipMemset(in, 0, 1 << 24);
    run<4, 1>(in, out); run<4, 0>(in, out); run<8, 1>(in, out); run<8, 0>(in, out);
    return 0;
}
