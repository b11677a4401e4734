"""ISA-level instruction mix of one kernel's main loop, from the disassembly of a code object.

    python scripts/isa_mix.py <library.so> <kernel-name regex> [steps per loop iteration]

Extracts the gfx950 code object from the fat binary (clang-offload-bundler), disassembles it (llvm-objdump), finds the
kernel, takes the LARGEST backward branch as its main loop and classifies every instruction of the loop body.  For the
headline kernel k_forward_sp<float, 2,3,1,2,2, Pat, NTR=2, DENSE_P=false, CK=8> one iteration = one chunk of 8 time steps
(the chunk's 8 Riccati recompute steps included), so `steps per loop iteration` = 8."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("fma / mul / add (fp arithmetic)", r"^v_(fma|fmac|mul|add|sub|mad|pk_fma|pk_mul|pk_add)_(f32|f64|legacy_f32)"),
    ("transcendental (rsq, rcp, log, exp, sqrt)", r"^v_(rsq|rcp|log|exp|sqrt|sin|cos)_"),
    ("fp compare / min / max / med", r"^v_(cmp|cmpx|max|min|med3|max3|min3)_.*(f32|f64)"),
    ("v_cndmask (select)", r"^v_cndmask"),
    ("v_mov / v_accvgpr (register moves)", r"^v_(mov|accvgpr|swap)"),
    ("conversion (cvt, ldexp, frexp, fract, rndne)", r"^v_(cvt|ldexp|frexp|fract|rndne|trunc|floor|ceil)"),
    ("integer / address VALU", r"^v_(add|sub|lshl|lshr|ashr|and|or|xor|mul_lo|mul_hi|mad_u|mad_i|bfe|add3|lshl_add|cmp_.*[ui](32|64)|max_u32|min_u32|addc|subb|readfirstlane|readlane|writelane)"),
    ("vector memory (global / buffer / scratch)", r"^(global|buffer|scratch|flat)_"),
    ("LDS", r"^ds_"),
    ("s_waitcnt / s_nop", r"^s_(waitcnt|nop|sleep)"),
    ("scalar memory", r"^s_(load|buffer_load|store)"),
    ("scalar ALU / branch", r"^s_"),
]


def disassemble(so):
    sec = subprocess.run([f"{LLVM}/llvm-readelf", "-S", so], capture_output=True, text=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    off, size = int(m.group(1), 16), int(m.group(2), 16)
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        data = open(so, "rb").read()[off:off + size]
        open(fat, "wb").write(data)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout


def kernel_body(asm, pat):
    out, on = [], False
    for line in asm.splitlines():
        h = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if h:
            if on:
                break
            name = subprocess.run(["c++filt", h.group(1)], capture_output=True, text=True).stdout.strip()
            on = re.search(pat, name) is not None
            if on:
                out.append(("name", name))
            continue
        if on:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m:
                out.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def main():
    so, pat = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    body = kernel_body(disassemble(so), pat)
    if not body:
        raise SystemExit("kernel not found")
    print("kernel:", body[0][1][:200])
    ins = body[1:]
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(-?\d+)\s*$", args)
            if not m:
                continue
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off                     # simm16 printed unsigned
            tgt = a + 4 + 4 * off
            if tgt in addr and addr[tgt] < i and i - addr[tgt] > 100:
                loops.append((addr[tgt], i))
    if not loops:
        raise SystemExit("no backward branch found")
    # the main loop = the largest INNERMOST big loop (the compiler may wrap the guarded first / last chunk and the whole-chunk
    # loop into an outer pseudo-loop with shared code: that one contains a clearly smaller big loop and is skipped)
    inner = [L for L in loops if not any(M != L and M[0] >= L[0] and M[1] <= L[1] and (M[1] - M[0]) < 0.9 * (L[1] - L[0])
                                         for M in loops)]
    best = max(inner, key=lambda L: L[1] - L[0])
    loop = ins[best[0]:best[1] + 1]
    cnt = collections.Counter()
    ops = collections.Counter()
    for _, op, _ in loop:
        for name, rx in CLASSES:
            if re.search(rx, op):
                cnt[name] += 1
                break
        else:
            cnt["other: " + op] += 1
        ops[op] += 1
    total = sum(cnt.values())
    valu = sum(v for k, v in cnt.items() if not any(s in k for s in ("s_waitcnt", "scalar", "vector memory", "LDS")))
    print(f"kernel instructions {len(ins)}, main loop {len(loop)} instructions = {len(loop) / steps:.1f} per time step "
          f"({steps} steps per iteration); VALU {valu} = {valu / steps:.1f} per step")
    for name, _ in CLASSES + [(k, None) for k in cnt if k.startswith("other")]:
        if cnt[name]:
            print(f"  {cnt[name]:6d}  {cnt[name] / steps:7.1f} / step  {100.0 * cnt[name] / total:5.1f} %   {name}")
    print("  top opcodes:", ", ".join(f"{op} {n}" for op, n in ops.most_common(14)))


if __name__ == "__main__":
    main()
