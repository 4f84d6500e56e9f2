"""TEST INFRASTRUCTURE: bench.py's launcher and rank logic on a machine without a GPU.

    python tests/bench_rehearsal.py --gpus 2 --rows 20000 ...

swaps the device entry points for tests/fake_hip.py (NumPy), lets the ranks talk over gloo, and runs bench.main()
unchanged -- ``--gpus N`` then starts N copies of THIS script (bench.py starts sys.argv[0]).  The line it prints says
"rehearsal" instead of "synthetic": a CPU stand-in never reports as a measurement.  bench.py itself knows nothing
of this file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("AKS_HOST_ALLOC", "torch")      # this worker uses torch tensors / process groups: the interop backend
os.environ["AKS_BENCH_BACKEND"] = "gloo"

import bench  # noqa: E402
import fake_hip  # noqa: E402

fake_hip.install()
_emit = bench.emit


def emit(line):
    out = json.loads(line)
    if isinstance(out, dict) and "data" in out:
        out["data"] = "rehearsal (NumPy stand-in for the device, numbers meaningless)"
    _emit(json.dumps(out))


bench.emit = emit

if __name__ == "__main__":
    sys.exit(bench.main())
