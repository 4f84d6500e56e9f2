make -C tests/mock_rccl > /dev/null 2>&1
timeout -k 5 300 python profiles/r05_oneshot_cost.py 2 4 6 > gpurun_out/r05_oneshot_cost.txt 2>&1
echo "cost rc=$?"; cat gpurun_out/r05_oneshot_cost.txt | cut -c1-600
timeout -k 5 500 python -m pytest tests/test_gpu_sharded_full.py -q -m gpu -x -s -k "one_shot" > gpurun_out/r05_j10_oneshot.log 2>&1
echo "thread-rank one-shot tests rc=$?"; grep -n "one-shot x\|passed\|failed\|Error" gpurun_out/r05_j10_oneshot.log | cut -c1-300
export AKS_LIB_PATH=$GRAFT_REPO_ROOT/tests/mock_rccl/libarnoldi_hip.so AKS_GRAPH=0
timeout -k 5 120 python tests/thread_ranks_worker.py --case repro --ranks 8 --repeats 1 --out gpurun_out/os_base_8.json > gpurun_out/os_base_8.log 2>&1
AKS_ALLREDUCE=oneshot GPU_MAX_HW_QUEUES=32 timeout -k 5 120 python tests/thread_ranks_worker.py --case repro --ranks 8 --repeats 1 --out gpurun_out/os_one_8.json > gpurun_out/os_one_8.log 2>&1
echo "8 ranks by hand rc=$?"
python - <<'PY'
import json
for f in ("os_base_8", "os_one_8"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); print(f, d["allreduce_path"], d["sha"], d["info"][-1])
    except Exception as e: print(f, "no result:", e)
PY
unset AKS_LIB_PATH AKS_GRAPH
bash tools/jobs/r05_job6.sh
