mkdir -p gpurun_out/r6
for i in 1 2; do
  echo "--- product"; timeout 300 python tools/bench_wgrad.py 2>/dev/null | tail -5
  echo "--- trimmed"; NSKY_LIB=$PWD/scratch/r6/libtrim.so timeout 300 python tools/bench_wgrad.py 2>/dev/null | tail -5
done
