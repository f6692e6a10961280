"""Repeated builds of the bench graph: wall time and slowest phase of each (allocation / sort stalls show as one phase 10x its usual)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pygrank_amd as pg
from pygrank_amd import _lib as L
from pygrank_amd.synthetic import rmat_graph
pg.load_backend("hip")
lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
keep = None
for i in range(n):
    L.check(lib.pgh_sync())
    t0 = time.perf_counter()
    adj = rmat_graph(23, 16, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
    L.check(lib.pgh_sync())
    dt = time.perf_counter() - t0
    buf = C.create_string_buffer(4096)
    L.check(lib.pgh_last_build_profile(buf, 4096))
    phases = [(k, float(v)) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)]
    worst = max(phases, key=lambda kv: kv[1])
    t1 = time.perf_counter()
    del adj
    L.check(lib.pgh_sync())
    print(f"build {i + 1:2d}: {dt * 1e3:7.1f} ms; slowest phase {worst[1]:7.1f} ms ({worst[0]}); destroy {1e3 * (time.perf_counter() - t1):6.1f} ms", flush=True)
