// Developer microbenchmark: the step structure of the GRU recurrence without memory traffic.
// Each wave: [192 MFMAs] [VALU block of NV exp+rcp pairs] [optional barrier], repeated; occupancy 1 or 2
// workgroups (of 4 waves) per CU.  Reports MFMA pipe utilisation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OCC, int NV, bool BAR, int PRIO>
__global__ __launch_bounds__(256, OCC) void k(const float* in, float* out, int steps) {
    __shared__ float sh[256];
    float b[160];
    for (int i = 0; i < 160; ++i) b[i] = in[threadIdx.x + 256 * i];
    f32x4 a4[8];
    for (int i = 0; i < 8; ++i) a4[i] = ((const f32x4*)in)[threadIdx.x + 64 * i];
    float v = in[threadIdx.x];
    if (PRIO == 2 && blockIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 3 && blockIdx.x >= 256) { for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(127); }
    f32x4 acc[6];
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < 6; ++i) acc[i] = (f32x4){v, 0, 0, 0};
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < 192; ++i)
            acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[(i / 24) % 8][(i / 6) % 4], b[i % 160], acc[i % 6], 0, 0, 0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        float x = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][1];
#pragma unroll
        for (int i = 0; i < NV; ++i) x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 1.4426f)) + acc[i % 6][i % 4];
        v = x;
        if (BAR) {
            sh[threadIdx.x] = v;
            __syncthreads();
            v += sh[(threadIdx.x + 64) & 255];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int OCC, int NV, bool BAR, int PRIO>
void run(const float* in, float* out) {
    const int steps = 200, grid = 256 * OCC;
    hipLaunchKernelGGL((k<OCC, NV, BAR, PRIO>), dim3(grid), dim3(256), 0, 0, in, out, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OCC, NV, BAR, PRIO>), dim3(grid), dim3(256), 0, 0, in, out, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = 2.0 * 16 * 16 * 4 * 192.0 * steps * 4.0 * grid / (ms * 1e-3) / 1e12;
    printf("prio %d occ %d  valu-pairs %3d  barrier %d : %.3f ms  %.1f TFLOP/s  %.1f%% of 157.3\n", PRIO, OCC, NV, (int)BAR, ms, tf, tf / 1.573);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24);
    hipMemset(in, 0, 1 << 24);
    run<2, 48, true, 0>(in, out); run<2, 48, true, 2>(in, out); run<2, 48, true, 3>(in, out);
    run<2, 24, true, 0>(in, out); run<2, 24, true, 2>(in, out); run<2, 24, true, 3>(in, out);
    return 0;
}
