"""One rank of `bench.py --gpus 2` on the CPU (gloo) against the host test double (tests/test_distributed_cpu.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_double  # noqa: E402

host_double.install()
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
