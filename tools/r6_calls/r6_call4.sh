mkdir -p gpurun_out/r6
ls -la /usr/local/graft/lib/ > gpurun_out/r6/call4_diag.log 2>&1
nm -D /usr/local/graft/lib/*execguard* 2>/dev/null | grep -i " T \| W " | head -60 >> gpurun_out/r6/call4_diag.log
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
export NSKY_FLAKE_SWEEP_STEPS=1
LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" python -X faulthandler -c '
import ctypes, sys
print("start", flush=True)
l = ctypes.CDLL(None); print("guard", hasattr(l, "heap_guard_sweep"), flush=True)
import pytest; print("pytest", flush=True)
import torch; print("torch", flush=True)
sys.path.insert(0, "."); from neusky_amd import hip; print("hip", flush=True)
x = torch.zeros(4, device="cuda"); print(float(x.sum()), flush=True)
' >> gpurun_out/r6/call4_diag.log 2>&1; echo "inline rc=$?" >> gpurun_out/r6/call4_diag.log
LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" python -X faulthandler tools/flake_seq.py test_gpu_eval_latents.py > gpurun_out/r6/call4_seq.log 2>&1; echo "seq rc=$?" >> gpurun_out/r6/call4_diag.log
tail -c 3000 gpurun_out/r6/call4_seq.log
cat gpurun_out/r6/call4_diag.log | tail -40
timeout 900 python -m pytest tests/test_gpu_trainer_surface.py tests/test_gpu_graph.py tests/test_gpu_two_ranks.py -m gpu -x -q > gpurun_out/r6/call4_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call4_pytest.log)"
