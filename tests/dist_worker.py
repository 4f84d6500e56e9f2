"""Worker of the multi-rank tests: one process per rank (torch.distributed.run).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/dist_worker.py --backend gloo --device cpu --out DIR

``--device cpu``  : CPU tensors + tests/fake_hip.py (host logic of the row-sharded driver: partition,
                    ghost plan, all-to-all, all-reduces, stage chaining) -- runs anywhere.
``--device cuda`` : the real HIP kernels; every rank uses GPU 0 and ``gloo`` carries the collectives
                    through host memory (how two ranks are rehearsed on the one-GPU box).
``--backend nccl --device cuda`` : RCCL itself, rank r on GPU r (needs as many GPUs as ranks): the C-driven path
                    as bench.py --gpus N runs it.
Each case solves the same problem as a single-process CPU oracle run and writes its verdict.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arnoldi-py_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("AKS_HOST_ALLOC", "torch")      # this worker uses torch tensors / process groups: the interop backend
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out", required=True)
    ap.add_argument("--native-mock", action="store_true",
                    help="(with --device cuda) the C-DRIVEN multi-rank path: the library's own communicator, RCCL "
                         "replaced by tests/mock_rccl (shared memory), because the ranks share one GPU")
    args = ap.parse_args()
    if args.native_mock:           # before arnoldi_amd is imported: _hip reads AKS_LIB_PATH at import
        os.environ["AKS_LIB_PATH"] = os.path.join(ROOT, "tests", "mock_rccl", "libarnoldi_hip.so")
        os.environ["AKS_COMM_OVER_GLOO"] = "1"
        os.environ["AKS_GRAPH"] = "0"          # the stand-in synchronises streams: nothing to capture

    if args.backend == "nccl":     # RCCL proper: one GPU per rank (a box with >= world_size GPUs)
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    if args.device == "cpu":
        import fake_hip

        fake_hip.install()
    elif args.backend != "nccl":
        torch.cuda.set_device(0)

    from arnoldi_amd.dist import Comm
    from dist_cases import run_cases

    verdict = run_cases(Comm(), rank, world)

    with open(os.path.join(args.out, f"rank{rank}.json"), "w") as f:
        json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
