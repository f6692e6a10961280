"""Diagnostic: registers / LDS / scratch of the kernels in a `hipcc -S --cuda-device-only` dump whose name matches a pattern.
Usage: python tools/kernel_resources.py file.s k_pb_finish"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
    block = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    if pat not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    get = lambda key: re.search(r"\." + key + r":\s+(\d+)", block).group(1)
    print(f"vgpr {get('vgpr_count'):>4} agpr {get('agpr_count'):>3} sgpr {get('sgpr_count'):>4} lds {get('group_segment_fixed_size'):>7} "
          f"scratch {get('private_segment_fixed_size'):>5}  {name[:150]}")
