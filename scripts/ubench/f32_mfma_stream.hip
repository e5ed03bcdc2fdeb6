// Developer microbenchmark (gfx950): what does the MFMA phase of the fp32 recurrence cost beyond 32 cycles per
// v_mfma_f32_16x16x4_f32?  Each wave: 3 accumulators x 8 k-groups x 4 = 96 MFMAs per phase with 96 different B
// registers (its W_hh slice) and A fragments read from LDS one group ahead -- the shape of gru_pair_kernel's M phase.
//   WAVES   waves per workgroup (4 = one per SIMD, 8 = two per SIMD); one workgroup per CU
//   BARRIER a workgroup barrier after every phase (as the recurrence has)
//   SAMEREG every MFMA uses the same A and B registers (the idealised stream of f32_mfma_valu_overlap.hip)
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, bool BARRIER, bool SAMEREG>
__global__ __launch_bounds__(WAVES * 64, 1) void stream(const f32x4* in, float* out, int phases, long long* cyc) {
    __shared__ f32x4 h[2 * 512];
    const int lane = threadIdx.x & 63;
    f32x4 W[3][8];
    for (int g = 0; g < 3; ++g)
        for (int m = 0; m < 8; ++m) W[g][m] = in[(g * 8 + m) * 64 + lane];
    for (int i = threadIdx.x; i < 1024; i += WAVES * 64) h[i] = in[2048 + i];
    __syncthreads();
    f32x4 acc[3] = {W[0][0], W[1][0], W[2][0]};
    long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < phases; ++p) {
        const f32x4* hb = h + (p & 1) * 512 + lane;
        f32x4 a[3];
        a[0] = hb[0];
        a[1] = hb[64];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(SAMEREG ? a[0][0] : a[m % 3][e], SAMEREG ? W[0][0][0] : W[g][m][e], acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (m + 2 < 8) a[(m + 2) % 3] = hb[(m + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(SAMEREG ? a[0][0] : a[m % 3][e], SAMEREG ? W[0][0][0] : W[g][m][e], acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int WAVES, bool BARRIER, bool SAMEREG>
void run(const f32x4* in, float* out, long long* cyc) {
    const int phases = 400;
    auto launch = [&] { hipLaunchKernelGGL((stream<WAVES, BARRIER, SAMEREG>), dim3(256), dim3(WAVES * 64), 0, 0, in, out, phases, cyc); };
    launch();
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = 96.0 * phases * (WAVES / 4);
    printf("waves/SIMD %d  barrier %d  same-registers %d : wave 0 %7.2f cycles per MFMA of its SIMD;  wall %.3f ms = %.2f cycles at 2.4 GHz, %.1f%% of the MFMA peak\n",
           WAVES / 4, (int)BARRIER, (int)SAMEREG, (double)c / mfma_per_simd, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd,
           100.0 * mfma_per_simd * 32 / (ms * 1e-3 * 2.4e9));
}

int main() {
    f32x4* in;
    float* out;
    long long* cyc;
    (void)hipMalloc(&in, 1 << 20);
    (void)hipMalloc(&out, 1 << 22);
    (void)hipMalloc(&cyc, 64);
    (void)hipMemset(in, 0, 1 << 20);
    run<4, false, true>(in, out, cyc);
    run<4, false, false>(in, out, cyc);
    run<4, true, false>(in, out, cyc);
    run<8, false, true>(in, out, cyc);
    run<8, false, false>(in, out, cyc);
    run<8, true, false>(in, out, cyc);
    return 0;
}
