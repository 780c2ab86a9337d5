# the abort behind tests/test_gpu_eval_latents.py with and without the synchronisation in front of the fit graph's destruction
run() { NSKY_FIT_SYNC=$2 python tools/flake_seq.py "$1" > /tmp/fs.log 2>&1; rc=$?; echo "[sync=$2] rc=$rc $(grep -c 'eval methods' /tmp/fs.log)"; }
for i in 1 2 3 4 5 6 7 8 9 10; do
  run test_gpu_eval_latents.py 0
  run test_gpu_eval_latents.py 1
done
