# CPU self-test of tools/heap_guard.c: every deliberate error of heap_guard_selftest.c is reported, a clean run and python + torch are quiet
set -e
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
gcc -O0 -o /tmp/heap_guard_selftest tools/heap_guard_selftest.c -ldl
expect() { out=$(HEAP_GUARD_CONTINUE=1 LD_PRELOAD=/tmp/heap_guard.so /tmp/heap_guard_selftest $1 2>&1 || true); echo "$out" | grep -q "$2" && echo "mode $1: ok ($2)" || { echo "mode $1: MISSING '$2'"; echo "$out" | head -5; exit 1; }; }
expect 0 "clean exit"
expect 1 "write BEHIND block"
expect 2 "STALE POINTER"
expect 3 "never handed out"
expect 4 "1 damaged"
expect 5 "HEADER of block"
LD_PRELOAD=/tmp/heap_guard.so /tmp/heap_guard_selftest 0 2>&1 | grep -q heap_guard && { echo "clean run is not quiet"; exit 1; }
LD_PRELOAD=/tmp/heap_guard.so python3 -c "
import ctypes, torch
g = ctypes.CDLL(None); g.heap_guard_sweep.restype = ctypes.c_long
x = torch.nn.Linear(64, 64)(torch.randn(8, 64, requires_grad=True)).sum(); x.backward()
assert g.heap_guard_sweep(b'python + torch') == 0 and g.heap_guard_violations() == 0
print('python + torch: quiet')"
