mkdir -p gpurun_out/r6
timeout 1500 bash tools/flake.sh 12 > gpurun_out/r6/call36_flake.log 2>&1
grep -c "rc=0" gpurun_out/r6/call36_flake.log; grep -v "rc=0" gpurun_out/r6/call36_flake.log | head -5
FLAKE_RUNS=4 timeout 900 bash tools/flake_seq.sh > gpurun_out/r6/call36_guard.log 2>&1
grep "guard" gpurun_out/r6/call36_guard.log | cut -c1-160
