cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u4 -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u4.json > gpurun_out/sell_u4.log 2>&1
python3 profiles/r05_sell_trace_summary.py gpurun_out/sell_trace_u4 gpurun_out/sell_phases_u4.json u4 > gpurun_out/r05_sell_summary2.txt 2>&1
AKS_LIB_PATH=$GRAFT_REPO_ROOT/arnoldi-py_amd/arnoldi_amd/lib/ab/libsellu8.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u8 -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u8.json > gpurun_out/sell_u8.log 2>&1
python3 profiles/r05_sell_trace_summary.py gpurun_out/sell_trace_u8 gpurun_out/sell_phases_u8.json u8 >> gpurun_out/r05_sell_summary2.txt 2>&1
cat gpurun_out/r05_sell_summary2.txt
rm -rf gpurun_out/sell_trace_u4 gpurun_out/sell_trace_u8
echo "=== capture crash probe under rocgdb"
timeout -k 5 240 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex run -ex bt -ex "thread apply all bt 14" --args python3 profiles/r05_capture_crash_probe.py > gpurun_out/r05_capture_probe_gdb.txt 2>&1
echo "gdb rc=$?"
grep -n "\[probe\]\|SIGSEGV\|signal\|^#" gpurun_out/r05_capture_probe_gdb.txt | head -80 | cut -c1-260
echo "=== torch-free bench rehearsal test"
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_multi_rank" > gpurun_out/r05_j11_bench.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r05_j11_bench.log | cut -c1-300
