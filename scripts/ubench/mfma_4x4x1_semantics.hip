// Developer probe: semantics and rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (operand layout, A broadcast by
// CBSZ / ABID, B lane-group pattern BLGP, issue rate, bit-identity of a k-ordered chain with v_mfma_f32_16x16x4_f32).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_4x4x1_semantics.bin mfma_4x4x1_semantics.hip && ./mfma_4x4x1_semantics.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID, int BLGP>
__global__ void one(const float* a, const float* b, f32x4* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    d[l] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, BLGP);
}

// 8 windows x 32 columns x K = 16 as sixteen 4x4x1 MFMAs in the order (e, q): k = 4 q + e, against the same
// product as four 16x16x4 MFMAs (instruction e sums k = 4 q + e over q = 0..3) on a 16-window tile.
__global__ void chain(const float* h /*[16 windows][16 k]*/, const float* w /*[16 k][32 cols]*/, float* out4 /*[8][32]*/,
                      float* out16 /*[16][32]*/, float c0) {
    const int l = threadIdx.x, b = l >> 2, i = l & 3;
    // 16x16x4: lane (j = l & 15, q = l >> 4): A[row j][k = q], B[k = q][col j]
    for (int ct = 0; ct < 2; ++ct) {
        f32x4 acc = {c0, c0, c0, c0};
        for (int e = 0; e < 4; ++e) {
            const int j = l & 15, q = l >> 4, k = 4 * q + e;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(h[j * 16 + k], w[k * 32 + ct * 16 + j], acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out16[(4 * (l >> 4) + r) * 32 + ct * 16 + (l & 15)] = acc[r];
    }
    // 4x4x1: block b = (wg = b >> 3, cq = b & 7); A lane (b, i) = h[4 wg + i][k], B lane (b, j) = w[k][4 cq + j]
    f32x4 acc = {c0, c0, c0, c0};
    for (int e = 0; e < 4; ++e)
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + e;
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(h[(4 * (b >> 3) + i) * 16 + k], w[k * 32 + 4 * (b & 7) + i], acc, 0, 0, 0);
        }
    for (int r = 0; r < 4; ++r) out4[(4 * (b >> 3) + r) * 32 + 4 * (b & 7) + i] = acc[r];
}

template <int NACC>
__global__ void rate(float* out, int n, long long* cycles) {
    f32x4 acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
    for (int g = 1; g < NACC; ++g) s += acc[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NACC>
__global__ void rate16(float* out, int n) {
    f32x4 acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
    f32x4 s = acc[0];
    for (int g = 1; g < NACC; ++g) s += acc[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CBSZ, int ABID, int BLGP>
int layout_case(const char* what) {
    std::vector<float> a(64), b(64);
    for (int l = 0; l < 64; ++l) { a[l] = 1.0f + l; b[l] = 100.0f + l; }
    float *da, *db; f32x4* dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    one<CBSZ, ABID, BLGP><<<1, 64>>>(da, db, dd);
    std::vector<f32x4> d(64);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    // hypothesis: D[block][row r][col j] in lane 4 block + j, VGPR r;  A[block][row i] from lane 4 ablock + i with
    // ablock = (block & ~((1 << CBSZ) - 1)) | ABID when CBSZ > 0;  B lane l' = BLGP 1: l & 31, BLGP 2: 32 + (l & 31)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, j = l & 3;
            const int ablk = CBSZ ? ((blk & ~((1 << CBSZ) - 1)) | ABID) : blk;
            int bl = 4 * blk + j;
            if (BLGP == 1) bl &= 31;
            if (BLGP == 2) bl = 32 + (bl & 31);
            const float want = a[4 * ablk + r] * b[bl];
            if (d[l][r] != want) {
                if (bad < 4) printf("   lane %d r %d: got %g want %g\n", l, r, d[l][r], want);
                ++bad;
            }
        }
    printf("%-40s cbsz %d abid %d blgp %d: %s (%d mismatches)\n", what, CBSZ, ABID, BLGP, bad ? "DIFFERENT" : "as assumed", bad);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    return 0;
}

int main() {
    layout_case<0, 0, 0>("plain layout");
    layout_case<3, 0, 0>("A of block 0 / 8 to 8 blocks");
    layout_case<3, 5, 0>("A of block 5 / 13 to 8 blocks");
    layout_case<2, 3, 0>("A of block 3 / 7 / 11 / 15 to 4 blocks");
    layout_case<0, 0, 1>("B of lanes 0-31 to all");
    layout_case<0, 0, 2>("B of lanes 32-63 to all");
    layout_case<3, 6, 2>("both");
    layout_case<4, 0, 0>("A of block 0 to all 16 blocks");
    layout_case<4, 11, 0>("A of block 11 to all 16 blocks");
    layout_case<4, 11, 1>("A of block 11 to all, B of lanes 0-31");
    {   // bit identity of the k-ordered chain
        int differ = 0, total = 0;
        float *dh, *dw, *o4, *o16;
        CK(hipMalloc(&dh, 1024)); CK(hipMalloc(&dw, 2048)); CK(hipMalloc(&o4, 1024)); CK(hipMalloc(&o16, 2048));
        srand(7);
        for (int trial = 0; trial < 200; ++trial) {
            std::vector<float> h(256), w(512), r4(256), r16(512);
            for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
            for (auto& x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
            CK(hipMemcpy(dh, h.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), 2048, hipMemcpyHostToDevice));
            chain<<<1, 64>>>(dh, dw, o4, o16, trial % 3 ? 0.0f : 0.37f);
            CK(hipMemcpy(r4.data(), o4, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(r16.data(), o16, 2048, hipMemcpyDeviceToHost));
            for (int wdw = 0; wdw < 8; ++wdw)
                for (int c = 0; c < 32; ++c) {
                    ++total;
                    if (r4[wdw * 32 + c] != r16[wdw * 32 + c]) ++differ;
                }
        }
        printf("k-ordered chain of sixteen 4x4x1 against four 16x16x4: %d of %d values differ\n", differ, total);
    }
    {
        float* out; long long* cyc; long long h;
        CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 8));
        const int n = 2000;
        rate<1><<<1, 64>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("1 wave, 1 accumulator : %.2f cycles per MFMA (dependent latency)\n", (double)h / (n * 8.0));
        rate<2><<<1, 64>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("1 wave, 2 accumulators: %.2f cycles per MFMA\n", (double)h / (n * 16.0));
        rate<3><<<1, 64>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("1 wave, 3 accumulators: %.2f cycles per MFMA\n", (double)h / (n * 24.0));
        rate<4><<<1, 64>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("1 wave, 4 accumulators: %.2f cycles per MFMA\n", (double)h / (n * 32.0));
        rate<3><<<1, 256>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("4 waves (1 per SIMD), 3 accumulators: %.2f cycles per MFMA per wave\n", (double)h / (n * 24.0));
        rate<3><<<1, 512>>>(out, n, cyc); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("8 waves (2 per SIMD), 3 accumulators: %.2f cycles per MFMA per wave (wave 0's clock only)\n", (double)h / (n * 24.0));
        // whole-chip throughput by wall clock: 1024 workgroups of 4 or 8 waves
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int waves = 4; waves <= 8; waves += 4) {
            float ms;
            rate<3><<<1024, 64 * waves>>>(out, n, cyc);
            CK(hipEventRecord(e0)); rate<3><<<1024, 64 * waves>>>(out, n, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("4x4x1 : 1024 workgroups x %d waves: %.3f ms, %.1f TFLOP/s\n", waves, ms, 1024.0 * waves * n * 24.0 * 512 / (ms * 1e-3) / 1e12);
            rate16<3><<<1024, 64 * waves>>>(out, n);
            CK(hipEventRecord(e0)); rate16<3><<<1024, 64 * waves>>>(out, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("16x16x4: 1024 workgroups x %d waves: %.3f ms, %.1f TFLOP/s\n", waves, ms, 1024.0 * waves * n * 6.0 * 2048 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
