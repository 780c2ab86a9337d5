"""python tools/flake_seq.py <comma-separated test files to run first> [-k expression] : run them through pytest IN THIS PROCESS, then the evaluation-method
test body (tests/test_gpu_eval_methods.run_eval_methods) in the same process -- which preceding tests does the intermittent abort need?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import ctypes
import pytest

# tools/heap_guard.c preloaded (LD_PRELOAD=/tmp/heap_guard.so): every live and quarantined host block is verified after each test, and -- with
# NSKY_FLAKE_SWEEP_STEPS=1 -- before every Adam launch of the eval-latent fit and on both sides of every graph replay,
# so the first damaged block is reported within one step of the write, with the library that allocated it
_libc = ctypes.CDLL(None)
GUARD = hasattr(_libc, "heap_guard_sweep")
if GUARD:
    _libc.heap_guard_sweep.restype = ctypes.c_long
    _libc.heap_guard_sweep.argtypes = [ctypes.c_char_p]


def sweep(tag):
    if GUARD and _libc.heap_guard_sweep(tag.encode()):
        print("HEAP DAMAGE first seen at:", tag, flush=True)
        import traceback
        traceback.print_stack()
        os._exit(77)


class SweepPlugin:
    def pytest_runtest_setup(self, item):
        sweep("before " + item.name)

    def pytest_runtest_teardown(self, item):
        sweep("after " + item.name)


if GUARD and os.environ.get("NSKY_FLAKE_SWEEP_STEPS"):
    import torch
    from neusky_amd import hip as _hip
    _adam, _replay = _hip.adam_step, torch.cuda.CUDAGraph.replay
    _count = [0]

    def adam_step(*a, **k):
        _count[0] += 1
        sweep(f"before adam_step call {_count[0]}")
        return _adam(*a, **k)

    def replay(self):
        sweep(f"before graph replay (adam calls so far {_count[0]})")
        r = _replay(self)
        sweep(f"after graph replay (adam calls so far {_count[0]})")
        return r

    _hip.adam_step = adam_step
    torch.cuda.CUDAGraph.replay = replay

if os.environ.get("NSKY_FLAKE_PRELOAD"):  # every kernel of the evaluation-method body loaded while the heap is fresh: is the abort tied to FIRST use?
    import test_gpu_eval_methods as t0
    t0.run_eval_methods("FiLM")
    print("preload: eval methods FiLM ok", flush=True)
files = [f for f in sys.argv[1].split(",") if f]
keep = ["-k", sys.argv[2]] if len(sys.argv) > 2 else []  # optional pytest -k expression
if files:
    # (-s: the guard reports on fd 2, which pytest's capture would swallow)
    rc = pytest.main(["-m", "gpu", "-q", "-s", "-p", "no:cacheprovider", "-p", "no:faulthandler"] + keep + [os.path.join("tests", f) for f in files], plugins=[SweepPlugin()])
    print("pytest rc", rc, flush=True)
    sweep("after the preceding tests")
if os.environ.get("NSKY_FLAKE_COLLECT"):  # destroy what the preceding tests left (dead pipelines, their graphs and pools) HERE, step by step
    import faulthandler, gc, torch
    faulthandler.enable()
    print("collect: start", flush=True)
    torch.cuda.synchronize(); print("collect: synchronised", flush=True)
    print("collect: gc", gc.collect(), flush=True)
    torch.cuda.synchronize(); print("collect: synchronised again", flush=True)
    torch.cuda.empty_cache(); print("collect: cache emptied", flush=True)
    torch.cuda.synchronize(); print("collect: done", flush=True)
import test_gpu_eval_methods as t
for c in ("FiLM", "Attention"):
    t.run_eval_methods(c)
    sweep("after eval methods " + c)
    print("eval methods", c, "ok", flush=True)
