mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6/call12_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call12_pytest.log)"
bash tools/evidence.sh r06z > gpurun_out/r6/call12_evidence.log 2>&1
tail -5 gpurun_out/r6/call12_evidence.log | cut -c1-300
bash tools/pmc_sq.sh r06 > gpurun_out/r6/call12_sq.log 2>&1; head -40 gpurun_out/sq_r06.txt
