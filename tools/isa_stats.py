"""Diagnostic: instruction statistics of one kernel in a `hipcc -S --cuda-device-only` dump.
Usage: python tools/isa_stats.py file.s <mangled-name-substring> [--dump out.s]"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and pat in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
body = text[start:end]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
count = lambda rx: sum(1 for l in body if re.search(rx, l))
print(f"lines {len(body)} vmem_loads {count(r'(global|buffer)_load')} vmem_stores {count(r'(global|buffer)_store')} "
      f"waitcnt_vm {count(r's_waitcnt.*vmcnt')} lds_atomics {count(r'ds_add_u64')} ds_read {count(r'ds_read')} "
      f"barriers {count(r's_barrier')} scratch {count(r'scratch_')} v_fma_f64 {count(r'v_fma_f64')}")
