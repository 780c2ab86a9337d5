mkdir -p gpurun_out/r6
timeout 1500 python tools/train_sanity.py 30000 > gpurun_out/r6/call34_sanity_30000.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r6/call34_sanity_30000.log | cut -c1-250
