"""python tools/flake_seq.py <comma-separated test files to run first> [-k expression] : run them through pytest IN THIS PROCESS, then the evaluation-method
test body (tests/test_gpu_eval_methods.run_eval_methods) in the same process -- which preceding tests does the intermittent abort need?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import pytest
if os.environ.get("NSKY_FLAKE_PRELOAD"):  # every kernel of the evaluation-method body loaded while the heap is fresh: is the abort tied to FIRST use?
    import test_gpu_eval_methods as t0
    t0.run_eval_methods("FiLM")
    print("preload: eval methods FiLM ok", flush=True)
files = [f for f in sys.argv[1].split(",") if f]
keep = ["-k", sys.argv[2]] if len(sys.argv) > 2 else []  # optional pytest -k expression
if files:
    rc = pytest.main(["-m", "gpu", "-q", "-p", "no:cacheprovider"] + keep + [os.path.join("tests", f) for f in files])
    print("pytest rc", rc, flush=True)
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
    print("eval methods", c, "ok", flush=True)
