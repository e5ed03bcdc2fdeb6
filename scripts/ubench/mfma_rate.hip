// Developer microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 with NACC independent
// accumulators in rotation and NB distinct B registers, one wave per SIMD (256 threads / CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NB>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, long long* cyc, int iters) {
    float b[NB];
    for (int i = 0; i < NB; ++i) b[i] = in[threadIdx.x + 256 * i];
    f32x4 a4[8];
    for (int i = 0; i < 8; ++i) a4[i] = ((const f32x4*)in)[threadIdx.x + 64 * i];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(i / 4) % 8][i % 4], b[i], acc[i % NACC], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NB>
void run(const float* in, float* out, long long* cyc, int grid) {
    const int iters = 200;
    hipLaunchKernelGGL((k<NACC, NB>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NB>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)c / (iters * (double)NB);
    double tf = 2.0 * 16 * 16 * 4 * NB * iters * 4.0 * grid / (ms * 1e-3) / 1e12;
    printf("grid %4d NACC %d NB %3d: %.2f cycles/MFMA, %.3f ms, %.1f TFLOP/s, eff clock %.2f GHz\n", grid, NACC, NB, per, ms,
           tf, c / (ms * 1e-3) / 1e9);
}

int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    hipMemset(in, 0, 1 << 24);
    for (int grid : {1, 256}) {
        run<1, 192>(in, out, cyc, grid);
        run<2, 192>(in, out, cyc, grid);
        run<3, 192>(in, out, cyc, grid);
        run<4, 192>(in, out, cyc, grid);
        run<6, 192>(in, out, cyc, grid);
        run<3, 96>(in, out, cyc, grid);
        run<6, 48>(in, out, cyc, grid);
    }
    return 0;
}
