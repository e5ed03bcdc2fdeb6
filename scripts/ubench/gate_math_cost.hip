// Developer microbenchmark (gfx950): what does the GRU gate math (helen::gru_cell4, the packed version the recurrence
// kernels use) cost a SIMD?  One workgroup of W waves (W = 4: one wave per SIMD, 8: two per SIMD, 16: four) runs
// CELLS/4 independent gru_cell4 per trip on register operands (hipcc's own schedule, as in the kernels), optionally
// with the bf16 conversions + LDS stores of the fused bf16 kernels.  Cycles from s_memtime: wave 0 (the oldest wave
// of its SIMD, which wins every issue slot it can use) and the SLOWEST wave -- the second is what a SIMD pays.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I helen_amd/csrc -o /tmp/gate_math_cost scripts/ubench/gate_math_cost.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#include "kernels_gru.h"
#include "kernels_x3.h"
using namespace helen;

__shared__ float g_lds[16 * 1024];

template <int GROUPS, bool STORE>
__global__ __launch_bounds__(1024) void gates(const float* in, float* out, int trips, long long* cyc) {
    f32x4 ar[GROUPS], az[GROUPS], an[GROUPS], gn[GROUPS];
    float hp[GROUPS][4];
    const float x = in[threadIdx.x];
    for (int g = 0; g < GROUPS; ++g) {
        ar[g] = f32x4{x, x * 0.5f, -x, x + 0.1f} + (float)g;
        az[g] = ar[g] * 0.3f;
        an[g] = ar[g] * -0.7f;
        gn[g] = ar[g] * 0.11f;
        for (int r = 0; r < 4; ++r) hp[g][r] = x * 0.01f * (r + 1);
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {   // operands opaque per trip, like fresh MFMA results
            asm volatile("" : "+v"(ar[g]), "+v"(az[g]), "+v"(an[g]), "+v"(gn[g]));
        }
        f32x4 hn[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) hn[g] = gru_cell4(ar[g], az[g], an[g], splat4(0.f), splat4(0.f), gn[g], hp[g]);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hp[g][r] = hn[g][r];
                if (STORE) {
                    g_lds[(threadIdx.x * 4 + r) + g * 4096] = hn[g][r];
                    ((unsigned short*)(g_lds + 8192))[threadIdx.x * 4 + r + g * 4096] = bf16_bits(hn[g][r]);
                }
            }
        if (STORE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float v = 0.f;
    for (int g = 0; g < GROUPS; ++g)
        for (int r = 0; r < 4; ++r) v += hp[g][r];
    out[threadIdx.x] = v + g_lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int GROUPS, bool STORE>
void run(int waves, const float* in, float* out, long long* cyc) {
    const int trips = 2000;
    hipLaunchKernelGGL((gates<GROUPS, STORE>), dim3(1), dim3(64 * waves), 0, 0, in, out, trips, cyc);
    (void)hipDeviceSynchronize();
    long long c[16];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    long long worst = 0;
    for (int w = 0; w < waves; ++w) worst = c[w] > worst ? c[w] : worst;
    printf("%2d waves (%d per SIMD)  %d cells per lane%s: wave 0 %7.1f cycles per trip, slowest wave %7.1f = %6.1f per SIMD per "
           "wave's 4 cells\n", waves, waves / 4, 4 * GROUPS, STORE ? " + bf16/LDS stores" : "", (double)c[0] / trips,
           (double)worst / trips, (double)worst / trips / GROUPS / (waves / 4));
}

int main() {
    float *in, *out;
    long long* cyc;
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&out, 4096 * 4);
    (void)hipMalloc(&cyc, 16 * 8);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 977) - 0.4f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {4, 8, 16}) {
        run<1, false>(w, in, out, cyc);
        run<2, false>(w, in, out, cyc);
        run<1, true>(w, in, out, cyc);
    }
    return 0;
}
