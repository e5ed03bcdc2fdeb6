// Developer microbenchmark (gfx950): the issue rate of v_mfma_f32_16x16x32_bf16 in ONE wave per SIMD as a function of
// where its operands come from -- the weight-stationary recurrences give every MFMA a different B fragment:
//   same      one A and one B fragment for all MFMAs (what scripts/ubench/bf16_mfma_valu_overlap.hip measures: 17 cycles)
//   B vgpr    24 different B fragments in VGPRs, one A
//   B agpr    24 different B fragments in AGPRs, one A
//   AB vgpr   24 different B fragments, 4 different A fragments (A changes every 6 MFMAs)
//   AB agpr   the same with B in AGPRs
//   C agpr    accumulators in AGPRs, B in VGPRs
// 24 MFMAs per trip over NACC accumulators (6 or 12).  Everything inline asm.  Cycles from the cycle counter of wave 0.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub this.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SAME = 0, B_VGPR = 1, B_AGPR = 2, AB_VGPR = 3, AB_AGPR = 4, C_AGPR = 5 };
static const char* kName[] = {"same A, same B", "one A, 24 B in VGPRs", "one A, 24 B in AGPRs", "4 A, 24 B in VGPRs", "4 A, 24 B in AGPRs",
                              "4 A, 24 B in VGPRs, C in AGPRs"};

// FILL: 0 none, 1 one v_exp_f32 behind every MFMA, 2 two v_fma_f32, 3 two v_exp_f32 (8 independent chains)
template <int MODE, int NACC, int THREADS, int FILL = 0>
__global__ __launch_bounds__(THREADS, 1) void stream(const float* in, float* out, int trips, long long* cyc) {
    f32x4 acc[NACC], a[4], b[24];
    const float a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{a0, b0, a0, b0};
    for (int i = 0; i < 4; ++i) a[i] = f32x4{a0 + i, b0, a0, b0 - i};
    for (int i = 0; i < 24; ++i) b[i] = f32x4{b0, a0 + i, b0 - i, a0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + i;
    const float cf = in[threadIdx.x + 1024];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            f32x4& c = acc[i % NACC];
            const f32x4& av = a[(MODE == AB_VGPR || MODE == AB_AGPR || MODE == C_AGPR) ? i / 6 : 0];
            const f32x4& bv = b[MODE == SAME ? 0 : i];
            if (MODE == B_AGPR || MODE == AB_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
            else if (MODE == C_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
            if (FILL == 1 || FILL == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(2 * i) & 7]));
            if (FILL == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(2 * i + 1) & 7]));
            if (FILL == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(2 * i) & 7]) : "v"(cf));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(2 * i + 1) & 7]) : "v"(cf));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < NACC; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) v += x[i];
    out[blockIdx.x * THREADS + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static const char* kFill[] = {"", " + 1 v_exp", " + 2 v_fma", " + 2 v_exp"};
template <int MODE, int NACC, int THREADS, int FILL = 0>
void run(const float* in, float* out, long long* cyc) {
    const int trips = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, NACC, THREADS, FILL>), dim3(256), dim3(THREADS), 0, 0, in, out, trips, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream<MODE, NACC, THREADS, FILL>), dim3(256), dim3(THREADS), 0, 0, in, out, trips, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tflops = 256.0 * (THREADS / 64) * trips * 24.0 * 16384.0 / (ms * 1e-3) / 1e12;
    long long c;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-34s%-11s %2d accumulators  %d wave(s)/SIMD: %6.2f cycles per MFMA per wave  %.3f ms = %.0f TFLOP/s on 256 CUs\n", kName[MODE], kFill[FILL], NACC, THREADS / 256, (double)c / (trips * 24.0), ms, tflops);
}

int main() {
    float *in, *out;
    long long* cyc;
    hipMalloc(&in, 4096 * sizeof(float));
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, sizeof(long long));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<SAME, 6, 256>(in, out, cyc);
    run<B_VGPR, 6, 256>(in, out, cyc);
    run<B_AGPR, 6, 256>(in, out, cyc);
    run<AB_VGPR, 6, 256>(in, out, cyc);
    run<AB_AGPR, 6, 256>(in, out, cyc);
    run<C_AGPR, 6, 256>(in, out, cyc);
    run<SAME, 12, 256>(in, out, cyc);
    run<AB_VGPR, 12, 256>(in, out, cyc);
    run<AB_AGPR, 12, 256>(in, out, cyc);
    run<SAME, 6, 512>(in, out, cyc);
    run<B_VGPR, 6, 512>(in, out, cyc);
    run<AB_VGPR, 6, 512>(in, out, cyc);
    run<AB_AGPR, 6, 512>(in, out, cyc);
    // VALU work in the shadow of the MFMAs: does it depend on where the operands come from?
    run<SAME, 6, 256, 1>(in, out, cyc);
    run<AB_VGPR, 6, 256, 1>(in, out, cyc);
    run<AB_AGPR, 6, 256, 1>(in, out, cyc);
    run<SAME, 6, 256, 2>(in, out, cyc);
    run<AB_VGPR, 6, 256, 2>(in, out, cyc);
    run<AB_AGPR, 6, 256, 2>(in, out, cyc);
    run<SAME, 6, 256, 3>(in, out, cyc);
    run<AB_VGPR, 6, 256, 3>(in, out, cyc);
    run<AB_AGPR, 6, 256, 3>(in, out, cyc);
    return 0;
}
