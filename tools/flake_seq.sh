# the abort behind tests/test_gpu_eval_latents.py when the evaluation-method body has already run once in the fresh process (its kernels'
# code objects are loaded -- the three identical native backtraces of the abort end in the runtime's lazy code-object load, comgr's metadata parse)
mkdir -p gpurun_out/flake
gcc -shared -fPIC -o /tmp/abort_bt.so tools/abort_bt.c || exit 1
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  LD_PRELOAD=/tmp/abort_bt.so NSKY_FLAKE_PRELOAD=1 python tools/flake_seq.py test_gpu_eval_latents.py > /tmp/fs.log 2>&1; rc=$?
  echo "[preload] rc=$rc $(grep -c 'eval methods' /tmp/fs.log)"
  [ $rc -ne 0 ] && tail -c 9000 /tmp/fs.log > gpurun_out/flake/pre_$i.log
done
