# the abort behind tests/test_gpu_eval_latents.py with thread-local and relaxed capture modes
run() { NSKY_CAPTURE_MODE=$1 python tools/flake_seq.py test_gpu_eval_latents.py > /tmp/fs.log 2>&1; rc=$?; echo "[mode=$1] rc=$rc $(grep -c 'eval methods' /tmp/fs.log)"; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  run thread_local
  run relaxed
done
