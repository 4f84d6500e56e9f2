set -x
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ticket or dgks or arnoldi or stress_grid or markov_golden" > gpurun_out/r05_j1_tests.log 2>&1 && \
python profiles/ab_bench.py 3 arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so arnoldi-py_amd/arnoldi_amd/lib/ab/libnorel.so -- --workload laplace2d --n 1000000 --nev 10 --max-dim 40 --steps 20 > gpurun_out/r05_ticket_release_ab.txt 2>&1 && \
python profiles/ab_bench.py 3 arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so arnoldi-py_amd/arnoldi_amd/lib/ab/libnorel.so -- --workload laplace3d --n 2000000 --nev 10 --max-dim 40 --steps 20 >> gpurun_out/r05_ticket_release_ab.txt 2>&1 && \
python profiles/ab_bench.py 3 arnoldi-py_amd/arnoldi_amd/lib/libarnoldi_hip.so arnoldi-py_amd/arnoldi_amd/lib/ab/libnorel.so -- --workload laplace3d --n 16000000 --nev 10 --max-dim 40 --steps 5 >> gpurun_out/r05_ticket_release_ab.txt 2>&1 && \
python -m pytest tests -x -q -m gpu > gpurun_out/r05_j1_full.log 2>&1
echo rc=$?
tail -3 gpurun_out/r05_j1_tests.log; cat gpurun_out/r05_ticket_release_ab.txt; tail -5 gpurun_out/r05_j1_full.log
