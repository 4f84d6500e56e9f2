"""cProfile of the FIRST partial_schur call of a process (README example); run with AKS_HOST_ALLOC=hip to see what a torch-free
process pays before its first solve (round 4: 4.97 s of 5.29 s were dlopen of the library -> librccl.so; now loaded lazily).
    AKS_HOST_ALLOC=hip python profiles/hip_first_call_profile.py"""
import os, sys, time, cProfile, pstats, io
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, os.path.join(ROOT,"arnoldi-py_amd"))
import numpy as np
import arnoldi_amd
from arnoldi_amd.matrices import mark
from arnoldi_amd.utils import arg_largest_real
A=mark(50); np.random.seed(0)
pr=cProfile.Profile(); pr.enable()
Q,T,h=arnoldi_amd.partial_schur(A,5,max_dim=20,sort_function=arg_largest_real,stopping_criterion=1e-8)
pr.disable()
out=io.StringIO(); pstats.Stats(pr,stream=out).sort_stats("tottime").print_stats(14); print(out.getvalue()[:3500])
