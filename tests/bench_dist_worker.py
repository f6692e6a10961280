"""One rank of `bench.py --gpus 2` on the CPU (gloo) against the host test double (tests/test_distributed_cpu.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygrank_amd import _lib  # noqa: E402

_lib._install_test_double(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libpgh_host_oracle.so")))
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
