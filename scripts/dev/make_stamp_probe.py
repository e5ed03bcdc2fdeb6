#!/usr/bin/env python3
"""Developer probe: a copy of helen_amd/csrc under build/probe/ whose two-tile kernels (gru_x3_il_kernel,
gru_fused_bf16_il_kernel) carry s_memtime stamps, built as build/lib_stamps.so -- the timing library
scripts/dev/region_stamps.py reads.  The product sources are not touched; the stamps cost SGPRs and ~40 cycles each, so
launch times of this library are only comparable with each other.
Per wave of workgroup 0 and per region of the last steady-state trip: [0] after the barrier, [1] after the wait for the gi
DMA (x3) / after the slot stream (bf16), [2] after the slot stream (x3) / after the VMEM wait (bf16), [3] after the LDS
drain in front of the barrier; and, region r0 only, a stamp in front of every fifth (x3) / fourth (bf16) slot-step.
    python scripts/dev/make_stamp_probe.py && HELEN_HIP_LIB=$PWD/build/lib_stamps.so python scripts/dev/region_stamps.py"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "helen_amd", "csrc")
DST = os.path.join(ROOT, "build", "probe", "helen_amd", "csrc")


def sub(text, old, new, what):
    if old not in text:
        sys.exit("make_stamp_probe: the source changed under the patch (%s)" % what)
    return text.replace(old, new, 1)


def patch_x3(s):
    s = sub(s, "template <bool DEC>\n__global__", "__device__ unsigned long long helen_dbg_x3[2][2][8][16];\n"
            "__device__ unsigned long long helen_dbg_x3f[2][2][8][16];\ntemplate <bool DEC>\n__global__", "x3 globals")
    s = sub(s, "    f32x4 Pr[2], Pz[2], Pn[2];\n", "    f32x4 Pr[2], Pz[2], Pn[2];\n    unsigned long long st[16], sf[16];\n"
            "#pragma unroll\n    for (int k = 0; k < 16; ++k) st[k] = sf[k] = 0;\n", "x3 arrays")
    s = sub(s, "        f32x4* const base = smem + x * kPerTile;\n",
            "        f32x4* const base = smem + x * kPerTile;\n        constexpr int ri = cur * 2 + x;\n"
            "        if constexpr (steady) st[ri * 4 + 0] = __builtin_amdgcn_s_memtime();\n", "x3 region start")
    s = sub(s, "        if (gates) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n",
            "        if (gates) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
            "        if constexpr (steady) { st[ri * 4 + 1] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n",
            "x3 gi wait")
    s = sub(s, "            __builtin_amdgcn_sched_barrier(0);\n            if constexpr (i >= kLead && i - kLead < NM) mfma_item(",
            "            __builtin_amdgcn_sched_barrier(0);\n"
            "            if constexpr (steady && ri == 0 && i % 5 == 0 && i / 5 < 16) sf[i / 5] = __builtin_amdgcn_s_memtime();\n"
            "            if constexpr (i >= kLead && i - kLead < NM) mfma_item(", "x3 stream")
    s = sub(s, "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        __builtin_amdgcn_s_barrier();",
            "        if constexpr (steady) st[ri * 4 + 2] = __builtin_amdgcn_s_memtime();\n"
            "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
            "        if constexpr (steady) { st[ri * 4 + 3] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n"
            "        __builtin_amdgcn_s_barrier();", "x3 region end")
    s = sub(s, "    for (; s < T; ++s) step(No{}, s);\n",
            "    for (; s < T; ++s) step(No{}, s);\n    if (blockIdx.x == 0 && lane == 0) {\n#pragma unroll\n"
            "        for (int k = 0; k < 16; ++k) {\n            helen_dbg_x3[DEC][dir][v][k] = st[k];\n"
            "            helen_dbg_x3f[DEC][dir][v][k] = sf[k];\n        }\n    }\n", "x3 write-out")
    return s


def patch_bf16(s):
    s = sub(s, "template <int MI, bool DEC>\n__global__ __launch_bounds__(512, 1) void gru_fused_bf16_il_kernel(",
            "__device__ unsigned long long helen_dbg_bf16[2][2][8][16];\n__device__ unsigned long long helen_dbg_bf16f[2][2][8][16];\n"
            "template <int MI, bool DEC>\n__global__ __launch_bounds__(512, 1) void gru_fused_bf16_il_kernel(", "bf16 globals")
    s = sub(s, "    f32x4 Pr[2], Pz[2], Pn[2], Pg[2];\n", "    f32x4 Pr[2], Pz[2], Pn[2], Pg[2];\n    unsigned long long st[16], sf[16];\n"
            "#pragma unroll\n    for (int k = 0; k < 16; ++k) st[k] = sf[k] = 0;\n", "bf16 arrays")
    s = sub(s, "        f32x4* const obase = smem + o * kPerTile;\n",
            "        f32x4* const obase = smem + o * kPerTile;\n        constexpr int ri = cur * 2 + x;\n"
            "        if constexpr (steady) { st[ri * 4 + 0] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n",
            "bf16 region start")
    s = sub(s, "            __builtin_amdgcn_sched_barrier(0);\n            if constexpr (i >= kLead && i - kLead < NM) mfma_item(",
            "            __builtin_amdgcn_sched_barrier(0);\n"
            "            if constexpr (steady && ri == 0 && i % 4 == 0 && i / 4 < 12) sf[i / 4] = __builtin_amdgcn_s_memtime();\n"
            "            if constexpr (i >= kLead && i - kLead < NM) mfma_item(", "bf16 stream")
    s = sub(s, "        __builtin_amdgcn_sched_barrier(0);\n        ring_rd[x] = ",
            "        __builtin_amdgcn_sched_barrier(0);\n        if constexpr (steady) st[ri * 4 + 1] = __builtin_amdgcn_s_memtime();\n        ring_rd[x] = ",
            "bf16 stream end")
    s = sub(s, "        // this phase's results become tile x's pending gate math\n",
            "        if constexpr (steady && ri == 0) sf[12] = __builtin_amdgcn_s_memtime();\n"
            "        // this phase's results become tile x's pending gate math\n", "bf16 h stored")
    s = sub(s, "        if (issued == 0) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");",
            "        if constexpr (steady && ri == 0) sf[13] = __builtin_amdgcn_s_memtime();\n"
            "        if (issued == 0) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");", "bf16 stores issued")
    s = sub(s, "        else asm volatile(\"s_waitcnt vmcnt(2)\" ::: \"memory\");\n        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n",
            "        else asm volatile(\"s_waitcnt vmcnt(2)\" ::: \"memory\");\n"
            "        if constexpr (steady) st[ri * 4 + 2] = __builtin_amdgcn_s_memtime();\n"
            "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
            "        if constexpr (steady) { st[ri * 4 + 3] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n",
            "bf16 region end")
    s = sub(s, "    for (; s < T; ++s) step(No{}, s);\n    // the gates of tile 1's last step",
            "    for (; s < T; ++s) step(No{}, s);\n    if (blockIdx.x == 0 && lane == 0) {\n#pragma unroll\n"
            "        for (int k = 0; k < 16; ++k) {\n            helen_dbg_bf16[DEC][dir][v][k] = st[k];\n"
            "            helen_dbg_bf16f[DEC][dir][v][k] = sf[k];\n        }\n    }\n    // the gates of tile 1's last step", "bf16 write-out")
    return s


def patch_pair(s):
    s = sub(s, "template <bool DEC>\n", "__device__ unsigned long long helen_dbg_pair[2][2][8][16];\n"
            "__device__ unsigned long long helen_dbg_pairf[2][2][8][16];\ntemplate <bool DEC>\n", "pair globals")
    s = sub(s, "    auto half_step = [&](auto X, auto CUR, auto STEADY, int s) __attribute__((always_inline)) {\n",
            "    unsigned long long st[16], sf[16];\n#pragma unroll\n    for (int k = 0; k < 16; ++k) st[k] = sf[k] = 0;\n"
            "    auto half_step = [&](auto X, auto CUR, auto STEADY, int s) __attribute__((always_inline)) {\n", "pair arrays")
    s = sub(s, "        const f32x4* hb = hx + slane;\n",
            "        const f32x4* hb = hx + slane;\n        constexpr int ri = cur * 2 + x;\n"
            "        if constexpr (steady) st[ri * 4 + 0] = __builtin_amdgcn_s_memtime();\n", "pair start")
    s = sub(s, "            __builtin_amdgcn_sched_barrier(0);\n#pragma unroll\n            for (int e = 0; e < 2; ++e)\n",
            "            __builtin_amdgcn_sched_barrier(0);\n            if (steady && ri == 0) sf[m] = __builtin_amdgcn_s_memtime();\n"
            "#pragma unroll\n            for (int e = 0; e < 2; ++e)\n", "pair groups")
    s = sub(s, "        if (DEC && has_prev2) {\n            if (v == ((s - 2) & 3))",
            "        if constexpr (steady) st[ri * 4 + 1] = __builtin_amdgcn_s_memtime();\n"
            "        if (DEC && has_prev2) {\n            if (v == ((s - 2) & 3))", "pair M end")
    s = sub(s, "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        a_pref",
            "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
            "        if constexpr (steady) { st[ri * 4 + 2] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n"
            "        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n"
            "        if constexpr (steady) { st[ri * 4 + 3] = __builtin_amdgcn_s_memtime(); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); }\n"
            "        a_pref", "pair barrier")
    s = sub(s, "    for (; s < T; ++s) step(No{}, s);                     // the last one or two steps: no step s+1 to feed\n",
            "    for (; s < T; ++s) step(No{}, s);                     // the last one or two steps: no step s+1 to feed\n"
            "    if (pair_index == 0 && lane == 0) {\n#pragma unroll\n        for (int k = 0; k < 16; ++k) {\n"
            "            helen_dbg_pair[DEC][dir][v][k] = st[k];\n            helen_dbg_pairf[DEC][dir][v][k] = sf[k];\n        }\n    }\n",
            "pair write-out")
    return s


def patch_api(s):
    fn = ""
    for name in ("x3", "x3f", "bf16", "bf16f", "pair", "pairf"):
        fn += ("__attribute__((visibility(\"default\"))) int helen_debug_%s(unsigned long long* out) { return (int)hipMemcpyFromSymbol("
               "out, HIP_SYMBOL(helen::helen_dbg_%s), sizeof(unsigned long long) * 2 * 2 * 8 * 16); }\n" % (name, name))
    return sub(s, "const char* helen_last_error(void) { return g_err; }\n", "const char* helen_last_error(void) { return g_err; }\n" + fn, "api")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "lib_stamps.so")
    shutil.rmtree(os.path.join(ROOT, "build", "probe"), ignore_errors=True)
    os.makedirs(os.path.dirname(DST))
    shutil.copytree(SRC, DST, ignore=shutil.ignore_patterns("*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(ROOT, "build", "probe", "include"))
    for name, fn in (("kernels_x3_il.h", patch_x3), ("kernels_fused_bf16_il.h", patch_bf16), ("kernels_gru_pair.h", patch_pair),
                     ("api.hip", patch_api)):
        path = os.path.join(DST, name)
        with open(path) as f:
            text = f.read()
        with open(path, "w") as f:
            f.write(fn(text))
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out, "api.hip"]
    subprocess.run(cmd, cwd=DST, check=True, stderr=subprocess.DEVNULL)
    print(out)


if __name__ == "__main__":
    main()
