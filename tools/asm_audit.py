"""Diagnostic: full waits inside the loops of the engine's kernels.  Compiles one source of pygrank_amd/csrc to gfx950 assembly and
prints, per kernel, the vector-memory loads, the `s_waitcnt vmcnt(0)` and how many of those sit in blocks the compiler marks as part
of a loop.  A load under a run-time branch (`cond ? load : 0`, `if (p != nullptr) v = p[i]`) gets a basic block and such a wait of
its own; so does the head of a loop that a path reaches with loads still pending ("waits").
Usage: python tools/asm_audit.py pgh_pb.hip [kernel name fragment]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    csrc = os.path.join(ROOT, "pygrank_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, src)],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    i, kernels = 0, []
    while i < len(lines):
        m = re.match(r"^(_Z\S+):\s", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            kernels.append((m.group(1), i, j))
            i = j
        i += 1
    for name, a, b in kernels:
        body = lines[a:b]
        if (pat and pat not in name) or not any("s_endpgm" in ln for ln in body):
            continue
        loads = sum(1 for ln in body if re.search(r"\t(global|buffer)_load", ln))
        full = sum(1 for ln in body if re.search(r"s_waitcnt\s+vmcnt\(0\)", ln))
        in_loop, inside = 0, False
        for ln in body:
            if re.match(r"^\.LBB", ln):
                inside = "in Loop" in ln or "Loop Header" in ln
            if inside and re.search(r"s_waitcnt\s+vmcnt\(0\)", ln):
                in_loop += 1
        print(f"{name[:100]:102s} loads={loads:4d} vmcnt(0)={full:3d} in loops={in_loop:3d}")


if __name__ == "__main__":
    main()
