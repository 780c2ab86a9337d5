mkdir -p gpurun_out/r6
timeout 3000 bash tools/flake.sh 20 > gpurun_out/r6/call18_flake.log 2>&1
cat gpurun_out/r6/call18_flake.log | cut -c1-120
