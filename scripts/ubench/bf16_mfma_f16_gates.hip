// Developer microbenchmark (gfx950): do HALF-PRECISION gate instructions ride free in the stream of bf16 MFMAs
// (v_mfma_f32_16x16x32_bf16), where a packed fp32 instruction costs a whole MFMA slot (bf16_mfma_valu_overlap.hip:
// v_pk_fma_f32 behind an MFMA 17 -> 33 cycles)?  The bf16 mode's layer kernels are bound by ~730 cycles of fp32 gate
// VALU per region that add to the matrix pipe's time (profiles/r04_bf16_own.txt); a half-precision gate cell (state kept in
// fp32) would halve the instruction count -- if its instructions cost what a scalar fp32 one does.
//   A. same wave: [MFMA + K fillers] x 24 per trip over 6 accumulators, one wave per SIMD (256 threads), and two (512);
//   B. the fillers alone (no MFMA): cycles per instruction.
// Everything is inline asm so that hipcc neither reorders nor packs anything; cycles from the cycle counter of wave 0 of
// workgroup 0.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_f16 scripts/ubench/bf16_mfma_f16_gates.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE = 0, F_FMA32, F_PKFMA32, F_EXP32, F_RCP32, F_EXP16, F_RCP16, F_PKFMA16, F_PKMUL16, F_PKADD16, F_FMA16, F_CVTPK16, F_CVT32, F_COUNT };
static const char* kName[] = {"none", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_pk_fma_f16",
                              "v_pk_mul_f16", "v_pk_add_f16", "v_fma_f16", "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16"};

template <int KIND>
__device__ __forceinline__ void filler(float& x, f32x2& y, unsigned& h, float c, unsigned ch) {
    if (KIND == F_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (KIND == F_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(f32x2{c, c}));
    if (KIND == F_EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (KIND == F_RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (KIND == F_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(h));
    if (KIND == F_RCP16) asm volatile("v_rcp_f16 %0, %0" : "+v"(h));
    if (KIND == F_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(h) : "v"(ch));
    if (KIND == F_PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h) : "v"(ch));
    if (KIND == F_PKADD16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h) : "v"(ch));
    if (KIND == F_FMA16) asm volatile("v_fma_f16 %0, %0, %1, %1" : "+v"(h) : "v"(ch));
    if (KIND == F_CVTPK16) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(c));
    if (KIND == F_CVT32) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x) : "v"(h));
}
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND, int K, int THREADS, bool WITH_MFMA>
__global__ __launch_bounds__(THREADS, 1) void stream(const float* in, float* out, int trips, long long* cyc) {
    f32x4 acc[6];
    float a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    const f32x4 a = {a0, b0, a0, b0}, b = {b0, a0, b0, a0};
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{a0, b0, a0, b0};
    float x[8];
    f32x2 y[8];
    unsigned h[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + i; y[i] = f32x2{a0 + i, b0 + i}; h[i] = 0x3c003c00u + i; }
    const float c = in[threadIdx.x + 1024];
    const unsigned ch = 0x38003800u;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if (WITH_MFMA) MFMA(acc[i % 6], a, b);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(x[(i * K + k) & 7], y[(i * K + k) & 7], h[(i * K + k) & 7], c, ch);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float v = 0;
    for (int i = 0; i < 6; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) v += x[i] + y[i].x + y[i].y + (float)h[i];
    out[blockIdx.x * THREADS + threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float* g_in;
static float* g_out;
static long long* g_cyc;

template <int KIND, int K, int THREADS, bool WITH_MFMA>
double run() {
    const int trips = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((stream<KIND, K, THREADS, WITH_MFMA>), dim3(256), dim3(THREADS), 0, 0, g_in, g_out, trips, g_cyc);
        hipDeviceSynchronize();
    }
    long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    return (double)c / (24.0 * trips);
}

template <int KIND>
void row() {
    const double alone1 = run<KIND, 1, 256, false>(), alone4 = run<KIND, 4, 256, false>();
    const double m1 = run<KIND, 1, 256, true>(), m2 = run<KIND, 2, 256, true>(), m3 = run<KIND, 3, 256, true>(), m4 = run<KIND, 4, 256, true>();
    const double w1 = run<KIND, 1, 512, true>(), w2 = run<KIND, 2, 512, true>();
    printf("%-20s alone %5.2f cycles each (x4: %5.2f) | behind each MFMA, 1 wave/SIMD: x1 %6.2f  x2 %6.2f  x3 %6.2f  x4 %6.2f | 2 waves/SIMD: x1 %6.2f  x2 %6.2f\n",
           kName[KIND], alone1, alone4 / 4.0, m1, m2, m3, m4, w1, w2);
}

int main() {
    hipMalloc(&g_in, 1 << 22);
    hipMalloc(&g_out, 1 << 22);
    hipMalloc(&g_cyc, 64);
    hipMemset(g_in, 0, 1 << 22);
    printf("cycles per MFMA slot (v_mfma_f32_16x16x32_bf16 alone: %.2f with one wave per SIMD, %.2f with two)\n",
           run<F_NONE, 0, 256, true>(), run<F_NONE, 0, 512, true>());
    row<F_FMA32>(); row<F_PKFMA32>(); row<F_EXP32>(); row<F_RCP32>();
    row<F_FMA16>(); row<F_PKFMA16>(); row<F_PKMUL16>(); row<F_PKADD16>(); row<F_EXP16>(); row<F_RCP16>(); row<F_CVTPK16>(); row<F_CVT32>();
    return 0;
}
