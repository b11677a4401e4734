"""Register / scratch / LDS / code-size table of every kernel in a built library (from the code object's metadata notes).

    python scripts/kernel_resources.py <library.so> [kernel-name regex]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so, td):
    sec = subprocess.run([f"{LLVM}/llvm-readelf", "-S", so], capture_output=True, text=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    off, size = int(m.group(1), 16), int(m.group(2), 16)
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
    open(fat, "wb").write(open(so, "rb").read()[off:off + size])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    return co


def main():
    so, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    with tempfile.TemporaryDirectory() as td:
        co = code_object(so, td)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout
    size = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            rows.append(cur)
            cur = {}
        cur[k] = v
    if cur.get("name"):
        rows.append(cur)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spillV':>6} {'scratch':>7} {'lds':>6} {'code B':>8}  kernel")
    for r in rows:
        name = subprocess.run(["c++filt", r.get("name", "")], capture_output=True, text=True).stdout.strip()
        if not re.search(pat, name):
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"lqg::Mask<[^>]*>\{[^}]*\}\}?", "M", short)[:150]
        print(f"{r.get('vgpr_count', '?'):>5} {r.get('agpr_count', '?'):>5} {r.get('sgpr_count', '?'):>5} {r.get('vgpr_spill_count', '?'):>6} "
              f"{r.get('private_segment_fixed_size', '?'):>7} {r.get('group_segment_fixed_size', '?'):>6} {size.get(r.get('name', ''), 0):>8}  {short}")


if __name__ == "__main__":
    main()
