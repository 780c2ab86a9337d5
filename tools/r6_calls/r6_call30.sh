mkdir -p gpurun_out/r6
for i in 1 2; do
  timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r6/call30_pytest_$i.log 2>&1
  echo "run $i pytest rc=$? $(tail -1 gpurun_out/r6/call30_pytest_$i.log)"
done
