// Developer microbenchmark (gfx950): how much VALU / transcendental work hides under bf16 MFMAs
// (v_mfma_f32_16x16x32_bf16, the instruction of the bf16 mode's layer kernels)?
//   A. same wave:   [MFMA + K filler instructions] x 6 accumulators; one wave per SIMD (256 threads) and two waves
//                   per SIMD (512 threads, both running the same stream)
//   B. other wave:  waves 0-3 MFMA only, waves 4-7 (same SIMDs) filler only
//   C. phases:      two waves per SIMD, per trip NM MFMAs and NF fillers each, separated by a barrier:
//                   aligned (both M then G), offset (older: M | G, younger: G | M), interleaved in one stream
// Everything is inline asm so that hipcc neither reorders nor packs anything; cycles from the cycle counter of
// wave 0 of workgroup 0, wall time from HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub this.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE = 0, F_FMA = 1, F_PKFMA = 2, F_EXP = 3, F_RCP = 4 };
static const char* kName[] = {"none", "v_fma", "v_pk_fma", "v_exp", "v_rcp"};

template <int KIND>
__device__ __forceinline__ void filler(float& x, f32x2& y, float c) {
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(f32x2{c, c}));
    if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (KIND == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
}
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// A: same wave.  Per trip 24 MFMAs over 6 accumulators, K fillers after each (8 independent chains).
template <int KIND, int K, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void same_wave(const float* in, float* out, int trips, long long* cyc) {
    f32x4 acc[6];
    float a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    const f32x4 a = {a0, b0, a0, b0}, b = {b0, a0, b0, a0};
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{a0, b0, a0, b0};
    float x[8];
    f32x2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + i; y[i] = f32x2{a0 + i, b0 + i}; }
    const float c = in[threadIdx.x + 1024];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            MFMA(acc[i % 6], a, b);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(x[(i * K + k) & 7], y[(i * K + k) & 7], c);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * THREADS + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// B / C: two waves per SIMD.  MODE 0: waves 0-3 MFMA only, waves 4-7 fillers only (no barrier).
// MODE 1 (aligned): every wave  M | G | ;  MODE 2 (offset): waves 0-3  M | G |, waves 4-7  G | M | ;
// MODE 3 (interleaved): every wave one stream of NM MFMAs with NF fillers spread evenly between them, one barrier.
template <int KIND, int NM, int NF, int MODE>
__global__ __launch_bounds__(512, 1) void two_waves(const float* in, float* out, int trips, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[6];
    float a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    const f32x4 a = {a0, b0, a0, b0}, b = {b0, a0, b0, a0};
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{a0, b0, a0, b0};
    float x[8];
    f32x2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + i; y[i] = f32x2{a0 + i, b0 + i}; }
    const float c = in[threadIdx.x + 1024];
    auto M = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NM; ++i) MFMA(acc[i % 6], a, b);
    };
    auto G = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NF; ++k) filler<KIND>(x[k & 7], y[k & 7], c);
    };
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        if (MODE == 0) {
            if (wave < 4) M(); else G();
        } else if (MODE == 1) {
            M();
            __builtin_amdgcn_s_barrier();
            G();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 2) {
            if (wave < 4) M(); else G();
            __builtin_amdgcn_s_barrier();
            if (wave < 4) G(); else M();
            __builtin_amdgcn_s_barrier();
        } else {
            constexpr int per = (NF + NM - 1) / NM;
            int k = 0;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                MFMA(acc[i % 6], a, b);
#pragma unroll
                for (int f = 0; f < per; ++f)
                    if (k < NF) { filler<KIND>(x[k & 7], y[k & 7], c); ++k; }
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = v;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

static float* g_in;
static float* g_out;
static long long* g_cyc;

template <typename F>
float timed(F launch) {
    launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND, int K, int THREADS>
void runA() {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((same_wave<KIND, K, THREADS>), dim3(256), dim3(THREADS), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("A same-wave %d wave(s)/SIMD  %-8s x%d per MFMA : %7.2f cycles per MFMA per wave   %.3f ms\n", THREADS / 256,
           kName[KIND], K, (double)c / (24.0 * trips), ms);
}
template <int KIND, int NM, int NF, int MODE>
void runC() {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((two_waves<KIND, NM, NF, MODE>), dim3(256), dim3(512), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c[2];
    hipMemcpy(c, g_cyc, 16, hipMemcpyDeviceToHost);
    static const char* mode[] = {"split (0-3 MFMA, 4-7 fillers)", "aligned M|G|", "offset M|G| beside G|M|", "interleaved stream |"};
    printf("C %-30s %2d MFMA + %3d %-8s per wave per trip: wave0 %7.1f  wave4 %7.1f cycles per trip   %.3f ms\n", mode[MODE], NM, NF,
           kName[KIND], (double)c[0] / trips, (double)c[1] / trips, ms);
}

int main() {
    hipMalloc(&g_in, 1 << 22);
    hipMalloc(&g_out, 1 << 22);
    hipMalloc(&g_cyc, 64);
    hipMemset(g_in, 0, 1 << 22);
    printf("--- A: fillers in the same wave's MFMA stream (v_mfma_f32_16x16x32_bf16) ---\n");
    runA<F_NONE, 0, 256>(); runA<F_NONE, 0, 512>();
    runA<F_FMA, 1, 256>(); runA<F_FMA, 2, 256>(); runA<F_FMA, 3, 256>(); runA<F_FMA, 4, 256>(); runA<F_FMA, 6, 256>();
    runA<F_PKFMA, 1, 256>(); runA<F_PKFMA, 2, 256>(); runA<F_PKFMA, 3, 256>();
    runA<F_EXP, 1, 256>(); runA<F_EXP, 2, 256>(); runA<F_EXP, 3, 256>();
    runA<F_RCP, 1, 256>(); runA<F_RCP, 2, 256>();
    runA<F_FMA, 1, 512>(); runA<F_FMA, 2, 512>(); runA<F_FMA, 3, 512>(); runA<F_FMA, 4, 512>();
    runA<F_PKFMA, 1, 512>(); runA<F_PKFMA, 2, 512>();
    runA<F_EXP, 1, 512>(); runA<F_EXP, 2, 512>();
    printf("--- C: two waves per SIMD; the bf16 layer kernels' half-step is ~21-36 MFMAs + 24 transcendentals + 24 packed per wave ---\n");
    runC<F_EXP, 24, 0, 1>(); runC<F_EXP, 36, 0, 1>();
    runC<F_EXP, 24, 48, 0>(); runC<F_FMA, 24, 48, 0>(); runC<F_PKFMA, 24, 48, 0>();
    runC<F_EXP, 24, 48, 1>(); runC<F_EXP, 24, 48, 2>(); runC<F_EXP, 24, 48, 3>();
    runC<F_FMA, 24, 48, 1>(); runC<F_FMA, 24, 48, 2>(); runC<F_FMA, 24, 48, 3>();
    runC<F_PKFMA, 24, 48, 1>(); runC<F_PKFMA, 24, 48, 2>(); runC<F_PKFMA, 24, 48, 3>();
    runC<F_EXP, 36, 48, 1>(); runC<F_EXP, 36, 48, 2>(); runC<F_EXP, 36, 48, 3>();
    runC<F_FMA, 36, 72, 1>(); runC<F_FMA, 36, 72, 2>(); runC<F_FMA, 36, 72, 3>();
    return 0;
}
