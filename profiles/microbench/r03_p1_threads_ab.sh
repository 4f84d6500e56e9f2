#!/bin/bash
# phase 1 of the binned SpMV with 256 / 512 / 1024 threads per workgroup and 2..16 entries per thread per trip:
# pb_abi_bench under rocprofv3 (run_variants.sh), one build after the other, the shipped build first and last
cd $GRAFT_REPO_ROOT
make -s -C arnoldi-py_amd all >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Iinclude -o profiles/microbench/pb_abi_bench profiles/microbench/pb_abi_bench.cpp -Larnoldi-py_amd/arnoldi_amd/lib -larnoldi_hip -Wl,-rpath,'$ORIGIN/../../arnoldi-py_amd/arnoldi_amd/lib' 2>&1 | grep -v warning | tail -2
V=profiles/microbench/variants
./profiles/microbench/run_variants.sh arnoldi-py_amd/arnoldi_amd/lib $V/p1_1024_2 $V/p1_1024_8 $V/p1_512_4 $V/p1_512_8 $V/p1_256_8 $V/p1_256_16 arnoldi-py_amd/arnoldi_amd/lib > gpurun_out/r03_p1_threads_ab.txt 2>&1
grep "==\|k_pb_phase" gpurun_out/r03_p1_threads_ab.txt
