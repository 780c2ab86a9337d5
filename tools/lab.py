"""Lab switches for the scripts under tools/ -- the package itself reads no process variable.

    import lab; lab.apply()          # before anything imports neusky_amd.hip

NSKY_LIB=<path>            an experimental build of the library instead of neusky_amd/libneusky_hip.so
NSKY_ASYNC_WGRAD=0         weight gradients on the launching stream (same arithmetic; same-box A/B runs)
NSKY_CAPTURE_MODE=global   capture mode of every HIP graph (flake hunt)
NSKY_FROZEN_DX=f32         input gradients through frozen dense layers on the exact-fp32 MFMA (attention decoder A/B)
NSKY_RETIRE_SECONDS=<s>    idle time after which a retired graph is destroyed (ops.RETIRE_SECONDS; 1e9: never)
NSKY_ORDER_BY_STREAM=1     cached weight preparations ordered by whole-stream waits (rounds 3-5) instead of events
NSKY_FIT_STREAM=0          the DDF-fit rows ride in the visibility rows' launches (rounds 3-5) instead of a launch of their own
NSKY_FILM_ASYNC=0          the FiLM chains' weight gradients on the launching stream (implies NSKY_FIT_STREAM=0: timing only)
"""
import importlib.util
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def use_library(path: str) -> None:
    """load neusky_amd.hip against another shared library (must run before the first import of neusky_amd.hip)"""
    import neusky_amd
    assert "neusky_amd.hip" not in sys.modules, "neusky_amd.hip is already bound to the in-tree library"
    spec = importlib.util.find_spec("neusky_amd.hip")
    src = open(spec.origin).read()
    marker = 'LIB_PATH = os.path.join(_HERE, "libneusky_hip.so")'
    assert marker in src
    mod = importlib.util.module_from_spec(spec)
    sys.modules["neusky_amd.hip"] = mod
    exec(compile(src.replace(marker, f"LIB_PATH = {os.path.abspath(path)!r}"), spec.origin, "exec"), mod.__dict__)
    neusky_amd.hip = mod


def apply() -> None:
    if os.environ.get("NSKY_LIB"):
        use_library(os.environ["NSKY_LIB"])
    from neusky_amd import hip, ops
    if os.environ.get("NSKY_ASYNC_WGRAD", "1") == "0":
        ops.ASYNC_WGRAD = False
    if os.environ.get("NSKY_CAPTURE_MODE"):
        ops.CAPTURE_MODE = os.environ["NSKY_CAPTURE_MODE"]
    if os.environ.get("NSKY_FROZEN_DX", "bf16x3") != "bf16x3":
        ops.FROZEN_DX_PRECISION = hip.PREC_F32
    if os.environ.get("NSKY_FIT_STREAM", "1") == "0" or os.environ.get("NSKY_FILM_ASYNC", "1") == "0":
        from neusky_amd.models.neusky_model import NeuSkyFactoModel
        NeuSkyFactoModel.start_ddf_fit = lambda self, prep: None
    if os.environ.get("NSKY_FILM_ASYNC", "1") == "0":
        orig = ops.FilmSirenFn._backward_fused

        def in_line(*a, **k):
            keep, ops.ASYNC_WGRAD = ops.ASYNC_WGRAD, False
            try:
                return orig(*a, **k)
            finally:
                ops.ASYNC_WGRAD = keep
        ops.FilmSirenFn._backward_fused = staticmethod(in_line)
    if os.environ.get("NSKY_RETIRE_SECONDS"):
        ops.RETIRE_SECONDS = float(os.environ["NSKY_RETIRE_SECONDS"])
    if os.environ.get("NSKY_ORDER_BY_STREAM") == "1":
        import torch

        def by_stream(mark, hit_seq):
            cur = torch.cuda.current_stream()
            if hit_seq == ops._STEP_SEQ[0] and mark[0] != cur:
                cur.wait_stream(mark[0])
        ops._order_after = by_stream
    if os.environ.get("NSKY_PRINT_STREAMS") == "1":  # which pool streams play which role in each captured step
        import torch
        from neusky_amd.pipelines import train_graph as tgm
        orig_body = tgm.TrainGraph._body

        def body(self, step):
            m = self.pipeline.model
            ids = {"current": torch.cuda.current_stream().stream_id, "capturing": torch.cuda.is_current_stream_capturing(),
                   "illum": getattr(getattr(m, "_illum_stream", None), "stream_id", None), "fit": getattr(getattr(m, "_fit_stream", None), "stream_id", None),
                   "roles": {d: {r: st.stream_id for r, st in v.items()} for d, v in ops._ROLE_STREAMS.items()}}
            print("NSKY streams", ids, flush=True)
            return orig_body(self, step)
        tgm.TrainGraph._body = body
