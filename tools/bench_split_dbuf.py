import sys, os, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in ["0","4"]:
    env=dict(os.environ, NSKY_GEMM_VARIANT=v)
    out=subprocess.run([sys.executable, os.path.join(ROOT,"tools","bench_gemm_split.py")],env=env,capture_output=True,text=True)
    print("variant",v, "(double-buffered)" if v=="0" else "(single buffer)"); print("\n".join(l for l in out.stdout.splitlines() if l.startswith("prec") and not l.startswith("prec 0")))
