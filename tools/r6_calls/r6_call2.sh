mkdir -p gpurun_out/r6
echo "LD_PRELOAD='$LD_PRELOAD'"
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
LD_PRELOAD=/tmp/heap_guard.so python -c 'print("replace ok")'; echo "replace rc=$?"
LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" python -c 'print("prepend ok")'; echo "prepend rc=$?"
LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" python -c 'import torch; print(torch.cuda.is_available()); x=torch.zeros(4,device="cuda"); print(x.sum().item())'; echo "prepend+gpu rc=$?"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/call2_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call2_pytest.log)"
