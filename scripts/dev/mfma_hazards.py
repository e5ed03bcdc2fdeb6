"""Developer check for kernels whose MFMAs are inline asm (kernels_fused_bf16_w4.h): hipcc does not know those statements
are MFMAs and inserts no wait states behind them, so any copy or spill of an accumulator that IT places close behind one
would read the register before the matrix pipe has written it.  This scans a kernel's ISA for a non-MFMA instruction
that reads the destination of a v_mfma_f32_16x16x32_bf16 with fewer than NEED wait states in between (an s_nop N counts
N+1, every other instruction 1, an intervening MFMA of the same kind 16 -- the pipe is in order).

    python scripts/dev/mfma_hazards.py <file.s | libhelen_hip.so> [kernel-name-substring]

Exit status 1 when a candidate is found.  (gfx940/950 need 11 wait states between an 8-pass XDL write and a VALU read;
NEED = 20 leaves a margin.)"""
import re
import subprocess
import sys

NEED = 20
WINDOW = 24


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels_of(path):
    """{name: [instruction lines]} from an assembly listing or (through llvm-objdump) a shared library's code object."""
    if path.endswith(".s"):
        text = open(path).read()
    else:
        llvm = "/opt/rocm/lib/llvm/bin/"
        import os
        import tempfile
        tmp = tempfile.mkdtemp()
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
        subprocess.check_call([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path])
        subprocess.check_call([llvm + "clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat, "--output=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
        text = subprocess.check_output([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
    out, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:\s*(?:;.*)?$", line.strip())
        if m:
            name = m.group(1)
            out[name] = []
            continue
        t = line.strip()
        if name is None or not t or t.startswith((";", ".", "//")):
            continue
        t = re.sub(r"\s*//.*$", "", t)          # objdump's address comments
        t = re.sub(r"\s*;.*$", "", t)
        if t:
            out[name].append(t)
    return out


def scan(lines):
    found = []
    for i, l in enumerate(lines):
        if not l.startswith(("v_mfma_f32_16x16x32", "v_mfma_f32_16x16x4_f32 v")):      # (the asm ones: VGPR destination)
            continue
        dst = regs(l.split(None, 1)[1].split(", ")[0])
        waited = 0
        for k in range(i + 1, min(i + 1 + WINDOW, len(lines))):
            m = lines[k]
            parts = m.split(None, 1)
            if m.startswith("s_nop"):
                waited += int(parts[1], 0) + 1
            elif len(parts) < 2 or m.startswith("s_"):
                waited += 1
            if m.startswith("s_nop") or len(parts) < 2 or m.startswith("s_"):
                if waited >= NEED:
                    break
                continue
            ops = parts[1].split(", ")
            if m.startswith("v_mfma"):
                src = set().union(*[regs(x) for x in ops[1:3]])       # A and B; C of an MFMA is interlocked
                if not (dst & src):
                    waited += 16 if m.startswith("v_mfma_f32_16x16x32") else 8
                    if waited >= NEED:
                        break
                    continue
            elif m.startswith(("ds_write", "global_store", "scratch_store", "buffer_store")):
                src = set().union(*[regs(x) for x in ops])
            else:
                src = set().union(*[regs(x) for x in ops[1:]])
            if dst & src:
                found.append((i, l, k - i, waited, m))
                break
            waited += 1
            if waited >= NEED:
                break
    return found


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "w4_kernel"
    bad = 0
    for name, lines in kernels_of(path).items():
        if want not in name:
            continue
        n_mfma = sum(l.startswith("v_mfma_f32_16x16x32") for l in lines)
        found = scan(lines)
        print("%s: %d MFMAs, %d hazard candidates" % (name, n_mfma, len(found)))
        for i, l, d, w, m in found:
            print("   %s\n      +%d instructions, %d wait states: %s" % (l, d, w, m))
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
