set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u4 -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u4.json > gpurun_out/sell_u4.log 2>&1
echo rc=$?
AKS_LIB_PATH=$GRAFT_REPO_ROOT/arnoldi-py_amd/arnoldi_amd/lib/ab/libsellu8.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u8 -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u8.json > gpurun_out/sell_u8.log 2>&1
echo rc=$?
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sell_trace_u4b -- python3 profiles/r05_sell_in_solve.py gpurun_out/sell_phases_u4b.json > gpurun_out/sell_u4b.log 2>&1
python3 profiles/r05_sell_trace_summary.py gpurun_out/sell_trace_u4 gpurun_out/sell_phases_u4.json u4 > gpurun_out/r05_sell_summary.txt 2>&1
python3 profiles/r05_sell_trace_summary.py gpurun_out/sell_trace_u8 gpurun_out/sell_phases_u8.json u8 >> gpurun_out/r05_sell_summary.txt 2>&1
python3 profiles/r05_sell_trace_summary.py gpurun_out/sell_trace_u4b gpurun_out/sell_phases_u4b.json u4again >> gpurun_out/r05_sell_summary.txt 2>&1
cat gpurun_out/r05_sell_summary.txt
rm -rf gpurun_out/sell_trace_u4 gpurun_out/sell_trace_u8 gpurun_out/sell_trace_u4b
