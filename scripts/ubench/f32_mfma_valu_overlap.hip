// Developer microbenchmark (gfx950): how much VALU / transcendental work hides under fp32 MFMAs?
//   A. same wave:   [v_mfma_f32_16x16x4_f32 + K filler instructions] x 6 accumulators, one wave per SIMD
//   B. other wave:  waves 0-3 MFMA only, waves 4-7 (same SIMDs) filler only, 8 independent chains
//   C. fillers alone: cycles per wave-instruction of v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_rcp_f32
// Everything is inline asm so that hipcc neither reorders nor packs anything; cycles from s_memtime
// of wave 0 of workgroup 0, wall time from HIP events.
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE = 0, F_FMA = 1, F_PKFMA = 2, F_EXP = 3, F_RCP = 4, F_ADD = 5, F_DSREAD = 6, F_SALU = 7, F_DMA = 8, F_STORE = 9 };
__shared__ f32x4 g_lds[2048];

template <int KIND>
__device__ __forceinline__ void filler(float& x, float2& y, float c) {
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(float2{c, c}));
    if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (KIND == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
}
// non-VALU fillers: LDS read (result unused until the end), SALU add, LDS-DMA, global store
template <int KIND>
__device__ __forceinline__ void filler2(f32x4& r, int& sacc, const f32x4* gp, f32x4* gout, int k) {
    if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(threadIdx.x & 63) * 16u), "n"(1024) : "memory");
    if (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
    if (KIND == F_DMA)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gp + (threadIdx.x & 63)),
                                         (void __attribute__((address_space(3)))*)(g_lds + 64 * (k & 7)), 16, 0, 0);
    if (KIND == F_STORE) gout[(threadIdx.x & 63) + 64 * (k & 7)] = r;
}
// A2: same wave, non-VALU fillers every EVERY-th MFMA
template <int KIND, int EVERY>
__global__ __launch_bounds__(256, 1) void same_wave2(const float* in, float* out, int trips, long long* cyc) {
    f32x4 acc[6];
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{a, b, a, b};
    f32x4 r = {a, a, b, b};
    int sacc = 0;
    g_lds[threadIdx.x] = r;
    __syncthreads();
    const f32x4* gp = (const f32x4*)in + 4096 + blockIdx.x * 1024 + (threadIdx.x >> 6) * 64;
    f32x4* gout = (f32x4*)out + 65536 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 512;
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
            if (i % EVERY == 0) filler2<KIND>(r, sacc, gp, gout, i);
        }
        if (KIND == F_DSREAD || KIND == F_DMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = r[0] + sacc + g_lds[threadIdx.x][1];
    for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// A: same wave.  Per loop trip: 24 MFMAs over 6 accumulators, K fillers after each (8 independent chains).
template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void same_wave(const float* in, float* out, int trips, long long* cyc) {
    f32x4 acc[6];
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{a, b, a, b};
    float x[8];
    float2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = float2{a + i, b + i}; }
    const float c = in[threadIdx.x + 512];
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(x[(i * K + k) & 7], y[(i * K + k) & 7], c);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// B: two waves per SIMD (512-thread workgroup): waves 0-3 issue NM MFMAs per trip, waves 4-7 NF fillers.
template <int KIND, int NM, int NF>
__global__ __launch_bounds__(512, 1) void split_wave(const float* in, float* out, int trips, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    float a = in[threadIdx.x], b = in[threadIdx.x + 512];
    float v = 0;
    long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = f32x4{a, b, a, b};
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int i = 0; i < NM; ++i)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 6]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float x[8];
        float2 y[8];
        for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = float2{a + i, b + i}; }
        const float c = in[threadIdx.x + 1024];
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int k = 0; k < NF; ++k) filler<KIND>(x[k & 7], y[k & 7], c);
        }
        for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y;
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = v;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// D: NW waves per SIMD (NW*4 waves per workgroup) ALL issue NF fillers per trip: VALU issue rate with company
template <int KIND, int NW, int NF>
__global__ __launch_bounds__(NW * 256, 1) void all_fill(const float* in, float* out, int trips, long long* cyc) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 1024];
    float x[8];
    float2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = float2{a + i, b + i}; }
    const float c = in[threadIdx.x + 2048];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < NF; ++k) filler<KIND>(x[k & 7], y[k & 7], c);
    }
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * NW * 256 + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float *g_in, *g_out;
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

template <int KIND, int K>
void run_same(const char* name) {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((same_wave<KIND, K>), dim3(256), dim3(256), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("A same-wave  %-8s K=%d : %7.2f cycles per MFMA (+%d fillers)   %.3f ms\n", name, K, (double)c / (trips * 24.0), K, ms);
}

template <int KIND, int EVERY>
void run_same2(const char* name) {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((same_wave2<KIND, EVERY>), dim3(256), dim3(256), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("A2 same-wave %-10s one per %d MFMAs : %7.2f cycles per MFMA   %.3f ms\n", name, EVERY, (double)c / (trips * 24.0), ms);
}

template <int KIND, int NW, int NF>
void run_all(const char* name) {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((all_fill<KIND, NW, NF>), dim3(256), dim3(NW * 256), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    printf("D all-fill   %-8s %d wave(s)/SIMD : %6.2f cycles per filler per wave = %5.2f per SIMD-instruction   %.3f ms\n", name, NW,
           (double)c / trips / NF, (double)c / trips / NF / NW, ms);
}

template <int KIND, int NM, int NF>
void run_split(const char* name) {
    const int trips = 2000;
    float ms = timed([&] { hipLaunchKernelGGL((split_wave<KIND, NM, NF>), dim3(256), dim3(512), 0, 0, g_in, g_out, trips, g_cyc); });
    long long c[2];
    hipMemcpy(c, g_cyc, 16, hipMemcpyDeviceToHost);
    printf("B split-wave %-8s MFMA/trip %3d fillers/trip %3d : mfma wave %8.1f cyc/trip (%.2f per MFMA)  filler wave %8.1f cyc/trip (%.2f per filler)  %.3f ms\n",
           name, NM, NF, (double)c[0] / trips, NM ? (double)c[0] / trips / NM : 0.0, (double)c[1] / trips,
           NF ? (double)c[1] / trips / NF : 0.0, ms);
}

int main() {
    hipMalloc(&g_in, 1 << 24);
    hipMalloc(&g_out, 1 << 26);
    hipMalloc(&g_cyc, 64);
    hipMemset(g_in, 0, 1 << 24);
    printf("--- A: fillers in the same wave as fp32 MFMAs (one wave per SIMD) ---\n");
    run_same<F_NONE, 0>("none");
    run_same<F_FMA, 1>("v_fma");  run_same<F_FMA, 2>("v_fma");  run_same<F_FMA, 4>("v_fma");  run_same<F_FMA, 6>("v_fma");  run_same<F_FMA, 8>("v_fma");
    run_same<F_PKFMA, 1>("v_pk_fma"); run_same<F_PKFMA, 2>("v_pk_fma"); run_same<F_PKFMA, 4>("v_pk_fma");
    run_same<F_EXP, 1>("v_exp");  run_same<F_EXP, 2>("v_exp");  run_same<F_EXP, 4>("v_exp");
    run_same<F_RCP, 1>("v_rcp");  run_same<F_RCP, 2>("v_rcp");  run_same<F_RCP, 4>("v_rcp");
    printf("--- A2: non-VALU fillers in the same wave ---\n");
    run_same2<F_DSREAD, 1>("ds_read128"); run_same2<F_DSREAD, 4>("ds_read128");
    run_same2<F_SALU, 1>("s_add"); run_same2<F_DMA, 4>("lds_dma"); run_same2<F_DMA, 12>("lds_dma");
    run_same2<F_STORE, 4>("store128"); run_same2<F_STORE, 12>("store128");
    printf("--- D: every wave issues fillers ---\n");
    run_all<F_FMA, 1, 192>("v_fma"); run_all<F_FMA, 2, 192>("v_fma"); run_all<F_FMA, 4, 192>("v_fma");
    run_all<F_PKFMA, 1, 192>("v_pk_fma"); run_all<F_PKFMA, 2, 192>("v_pk_fma"); run_all<F_PKFMA, 4, 192>("v_pk_fma");
    run_all<F_EXP, 1, 192>("v_exp"); run_all<F_EXP, 2, 192>("v_exp"); run_all<F_EXP, 4, 192>("v_exp");
    run_all<F_RCP, 1, 192>("v_rcp"); run_all<F_RCP, 2, 192>("v_rcp");
    printf("--- C: fillers alone (waves 4-7 only; waves 0-3 idle) ---\n");
    run_split<F_FMA, 0, 192>("v_fma"); run_split<F_ADD, 0, 192>("v_add"); run_split<F_PKFMA, 0, 192>("v_pk_fma");
    run_split<F_EXP, 0, 192>("v_exp"); run_split<F_RCP, 0, 192>("v_rcp");
    printf("--- B: MFMA-only waves beside filler-only waves on the same SIMDs ---\n");
    run_split<F_NONE, 192, 0>("none");
    run_split<F_FMA, 192, 96>("v_fma"); run_split<F_FMA, 192, 192>("v_fma"); run_split<F_FMA, 192, 384>("v_fma"); run_split<F_FMA, 192, 768>("v_fma");
    run_split<F_PKFMA, 192, 192>("v_pk_fma"); run_split<F_PKFMA, 192, 384>("v_pk_fma");
    run_split<F_EXP, 192, 96>("v_exp"); run_split<F_EXP, 192, 192>("v_exp"); run_split<F_EXP, 192, 384>("v_exp");
    run_split<F_RCP, 192, 192>("v_rcp");
    return 0;
}
