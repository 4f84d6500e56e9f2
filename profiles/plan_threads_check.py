"""The binned plan of a random CSR matrix for 1, 2, 3, 8, 16 planner threads: the same bytes (sha256 of all seven arrays) and the time of aks_pb_plan_create.
    python profiles/plan_threads_check.py N"""
import sys, os, time, hashlib, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'arnoldi-py_amd'))
import numpy as np
from arnoldi_amd import _hip, matrices
lib=_hip.load()
def plan(M):
    indptr=np.ascontiguousarray(M.indptr,np.int32); indices=np.ascontiguousarray(M.indices,np.int32); values=np.ascontiguousarray(M.data)
    cplx=int(np.iscomplexobj(values))
    sz=_hip.PbSizes()
    t=time.perf_counter()
    p=lib.aks_pb_plan_create(indptr.ctypes.data,indices.ctypes.data,values.ctypes.data,cplx,M.shape[0],M.shape[1],C.byref(sz))
    dt=time.perf_counter()-t
    assert p, lib.aks_last_error()
    val=np.empty(sz.nnz_pad,values.dtype); lcol=np.empty(sz.nnz_pad,np.uint16); sb=np.empty(sz.n_slabs,np.int32); se=np.empty(sz.n_slabs,np.int32)
    runs=np.empty((sz.n_runs,4),np.uint32); rb=np.empty(sz.n_rowblocks+1,np.int32); lrow=np.empty(sz.n_lrow,np.uint16)
    assert lib.aks_pb_plan_export(p,val.ctypes.data,lcol.ctypes.data,sb.ctypes.data,se.ctypes.data,runs.ctypes.data,rb.ctypes.data,lrow.ctypes.data)==0
    lib.aks_pb_plan_destroy(p)
    h=hashlib.sha256()
    for a in (val,lcol,sb,se,runs,rb,lrow): h.update(a.tobytes())
    return h.hexdigest()[:16], dt
n=int(sys.argv[1])
A=matrices.random_csr(n,5,1234)
for nt in (1,2,3,8,16):
    os.environ["AKS_PLAN_THREADS"]=str(nt)
    print(nt, *plan(A))
