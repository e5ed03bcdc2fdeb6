// Developer microbenchmark (gfx950): the recurrence kernels' phase structure without any memory traffic --
// 8 waves (two per SIMD), per trip:  s_barrier | gate math (helen::gru_cell4 on this wave's accumulators) | NM fp32
// MFMAs in three chains.  Why do the two waves of a SIMD not do their gate math concurrently (as they do alone,
// gate_math_cost.hip) once MFMA phases follow?  Prints per wave the average cycles of each phase.
//   MODE 0 plain   1 s_setprio 3 during the gates (all waves)   2 ... only waves 4-7   3 second barrier behind the gates
//        4 waves 0-3 s_sleep 2 first   5 gates split: waves 4-7 first, then barrier, then waves 0-3 (serial reference)
//        6 barrier | MFMAs | gates  (the barrier behind the gates instead of in front of them: the older wave's gate
//          math, which has issue priority, then runs beside the younger wave's MFMAs)   7 = 6 with setprio 3 in the gates
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I helen_amd/csrc -o /tmp/gates_then_mfma scripts/ubench/gates_then_mfma.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#include "kernels_gru.h"
using namespace helen;

template <int MODE, int NM>
__global__ __launch_bounds__(512, 1) void phases(const float* in, float* out, int trips, long long* cyc) {
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float x = in[threadIdx.x];
    f32x4 acc[3] = {f32x4{x, x * 0.5f, -x, x + 0.1f}, f32x4{x, -x, x, 0.2f}, f32x4{0.1f, x, x, x}};
    f32x4 gn = acc[0] * 0.11f;
    float hp[4] = {x, x * 0.1f, x * 0.2f, x * 0.3f};
    const float a = x * 1e-3f, b = 1e-3f;
    long long tk[3] = {0, 0, 0};
    __syncthreads();
    long long tlast = __builtin_readcyclecounter();
#define TICK(i) { __builtin_amdgcn_sched_barrier(0); long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
    for (int t = 0; t < trips; ++t) {
        __builtin_amdgcn_s_barrier();
        TICK(0)
        if (MODE >= 6) {
#pragma unroll
            for (int i = 0; i < NM; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 3]) : "v"(a), "v"(b));
            TICK(2)
            if (MODE == 7) __builtin_amdgcn_s_setprio(3);
            const f32x4 hn6 = gru_cell4(acc[0], acc[1], acc[2], splat4(0.f), splat4(0.f), gn, hp);
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[r] = hn6[r];
            asm volatile("" : "+v"(hp[0]), "+v"(hp[1]), "+v"(hp[2]), "+v"(hp[3]));
            if (MODE == 7) __builtin_amdgcn_s_setprio(0);
            TICK(1)
            continue;
        }
        if (MODE == 1 || (MODE == 2 && v >= 4)) __builtin_amdgcn_s_setprio(3);
        if (MODE == 4 && v < 4) __builtin_amdgcn_s_sleep(2);
        if (MODE == 5 && v < 4) __builtin_amdgcn_s_barrier();
        const f32x4 hn = gru_cell4(acc[0], acc[1], acc[2], splat4(0.f), splat4(0.f), gn, hp);
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = hn[r];
        asm volatile("" : "+v"(hp[0]), "+v"(hp[1]), "+v"(hp[2]), "+v"(hp[3]));
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_setprio(0);
        if (MODE == 5 && v >= 4) __builtin_amdgcn_s_barrier();
        if (MODE == 3) __builtin_amdgcn_s_barrier();
        TICK(1)
#pragma unroll
        for (int i = 0; i < NM; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 3]) : "v"(a), "v"(b));
        TICK(2)
    }
    out[threadIdx.x] = hp[0] + hp[1] + hp[2] + hp[3] + acc[0][0] + acc[1][1] + acc[2][2];
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 3; ++i) cyc[v * 3 + i] = tk[i] / trips;
}

template <int MODE, int NM>
void run(const char* what, const float* in, float* out, long long* cyc) {
    hipLaunchKernelGGL((phases<MODE, NM>), dim3(1), dim3(512), 0, 0, in, out, 500, cyc);
    (void)hipDeviceSynchronize();
    long long c[24];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-34s %2d MFMAs  wave 0: barrier %5lld gates %5lld mfma %5lld | wave 4: barrier %5lld gates %5lld mfma %5lld | trip %lld\n",
           what, NM, c[0], c[1], c[2], c[12], c[13], c[14], c[0] + c[1] + c[2]);
}

int main() {
    float *in, *out;
    long long* cyc;
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 24 * 8);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 977) - 0.4f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 96>("plain", in, out, cyc);
    run<1, 96>("setprio 3 in the gates, all waves", in, out, cyc);
    run<2, 96>("setprio 3 in the gates, waves 4-7", in, out, cyc);
    run<3, 96>("second barrier behind the gates", in, out, cyc);
    run<4, 96>("waves 0-3 sleep first", in, out, cyc);
    run<5, 96>("gates one wave at a time", in, out, cyc);
    run<6, 96>("barrier | MFMAs | gates", in, out, cyc);
    run<7, 96>("barrier | MFMAs | gates, setprio 3", in, out, cyc);
    run<6, 24>("barrier | MFMAs | gates", in, out, cyc);
    run<7, 24>("barrier | MFMAs | gates, setprio 3", in, out, cyc);
    run<0, 24>("plain", in, out, cyc);
    run<3, 24>("second barrier behind the gates", in, out, cyc);
    run<0, 0>("plain (no MFMAs)", in, out, cyc);
    return 0;
}
