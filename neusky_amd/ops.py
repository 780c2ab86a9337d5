"""torch.autograd.Function wrappers that chain the HIP kernels of libneusky_hip.so.

Every Function's forward AND backward run exclusively on the C ABI in neusky_amd.hip (fp32 MFMA dense
layers with fused epilogues, hash-grid encode, render-stage kernels); PyTorch only owns the device
buffers and the autograd graph between Functions.  Hand-derived backward passes replace what the
reference obtains from torch autograd (incl. the double backward of sdf_albedo_field.py:235-238).

Conventions: activation matrices are row-major float32 with a leading dimension that is a multiple
of 4; weights arrive zero-padded to multiples of 4 in both dimensions (`pad_weight`).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import hip


# Arithmetic of the dense contractions (operands and results are fp32 in memory either way; see include/neusky_hip.h).  Two policies:
#   splith (default): every large contraction -- the FiLM-SIREN chains, the sdf value chain, the SDF / albedo field and all their weight
#                    gradients -- runs on the fused chain kernels at ANY row count: fp32-grade products from fp16 hi + residual planes on
#                    power-of-two pre-scaled operands, three fp16 MFMAs per product (~2^-22).  Round 4 retired the row-count switches to
#                    the per-layer kernels and with them the last 2-term bf16 (2^-16) backward products: what still runs per layer (shapes
#                    the chain kernels do not take, narrow heads, proposal layers) uses the fp16 split forward and the EXACT fp32 MFMA backward.
#   f32            : every product on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), per-layer kernels only: the measurement reference
#                    (bench.py's `fp32_exact` line; tests/test_gpu_full_size.py compares the two kernel sets).
_POLICIES = ("f32", "splith")
_FWD = {"splith": hip.PREC_F16X2}
_POLICY = "splith"  # the product runs ONE policy; no process variable or config key selects another
FWD_PRECISION = _FWD.get(_POLICY, hip.PREC_F32)
BWD_PRECISION = hip.PREC_F32  # per-layer backward GEMMs: exact fp32 MFMA under either policy


def set_precision_policy(policy: str) -> None:
    """MEASUREMENT / TEST HOOK ONLY (bench.py's `fp32_exact` leg, tests/test_gpu_full_size.py, tests/test_gpu_sdf_chain.py): "f32" sends
    every product to the exact-fp32 MFMA on the per-layer kernels -- the independent kernel set the fused chains are compared with.
    Nothing in neusky_amd/ calls it."""
    global FWD_PRECISION, _POLICY
    if policy not in _POLICIES:
        raise ValueError(policy)
    _POLICY = policy
    FWD_PRECISION = _FWD.get(policy, hip.PREC_F32)


def fgemm(A, W, Cout, M, N, K, **k):
    """forward-pass dense layer (W = [out, in] weight, k-contiguous)"""
    if FWD_PRECISION == hip.PREC_F16X2 and K % _PLANES_K_STEP == 0 and K >= _PLANES_MIN_K and N > 64 and M >= 4096:
        return hip.gemm_planes(A, _planes(W, N, K, False, FWD_PRECISION), Cout, M, N, K, precision=FWD_PRECISION, **k)
    return hip.gemm(A, W, Cout, M, N, K, precision=FWD_PRECISION, **k)


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def _slab_resident(sk) -> bool:
    """`sk` (a registered slab parameter) holds its pre-zeroed slab view as .grad and takes gradients in place (from its second pass on)"""
    g = sk.grad
    return g is not None and getattr(sk, "_nsky_sunk", False) and g.shape == sk.shape and g.is_contiguous()


class _PadFn(torch.autograd.Function):
    """zero padding of a small weight / bias to multiples of 4: one copy into a zero-filled (arena) buffer forward, a VIEW of the
    incoming gradient backward (torch's F.pad costs a pad kernel each way).  When the padded tensor is a slab parameter, the backward
    DEFERS: the view is added to the parameter's slab slot at the end of the pass (finish_pass, behind the final join), so the node
    neither returns a gradient nor waits for the side stream that may still be accumulating `g` -- the FiLM chains' weight gradients
    (2 ms for the DDF network) then really run beside the rest of the backward pass instead of being joined by the next node."""

    @staticmethod
    def forward(ctx, w, shape, persistent=False):
        out = torch.zeros(shape, device=w.device) if persistent else zeros(shape, device=w.device)
        if w.dim() == 2:
            out[:w.shape[0], :w.shape[1]].copy_(w)
        else:
            out[:w.shape[0]].copy_(w)
        ctx.orig = tuple(w.shape)
        ctx.sink = w if getattr(w, "_nsky_grad_sink", False) else None
        ctx.set_materialize_grads(False)  # (a consumer that deferred its share itself sends nothing: FilmSirenFn)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        o = ctx.orig
        gs = g[:o[0], :o[1]] if len(o) == 2 else g[:o[0]]
        sk = ctx.sink
        if sk is not None:
            if _slab_resident(sk):
                _DEFERRED_PADS.append((gs, sk.grad))
                _queue_end_of_pass()
                return None, None, None
            sk._nsky_sunk = True  # (slab-resident from the next zero fill on)
        join_weight_gradients()  # (g may be a weight-gradient accumulator of the side stream)
        return gs, None, None


def _padded(w, shape, persistent):
    out = _PadFn.apply(w, shape, persistent)
    out._nsky_pad_of = w if getattr(w, "_nsky_grad_sink", False) else None  # (read by the consumer's forward: FilmSirenFn)
    return out


def pad_weight(w: torch.Tensor, persistent: bool = False) -> torch.Tensor:
    """[out, in] -> zero padded [pad4(out), pad4(in)] (autograd-tracked).  persistent: the copy outlives the step (a frozen
    network's cache): it gets an allocation of its own instead of a region of the step's zero arena."""
    o, i = w.shape
    if o % 4 == 0 and i % 4 == 0:
        return w if w.is_contiguous() else w.contiguous()
    return _padded(w, (pad4(o), pad4(i)), persistent)


def pad_bias(b: torch.Tensor, persistent: bool = False) -> torch.Tensor:
    if b.shape[0] % 4 == 0:
        return b if b.is_contiguous() else b.contiguous()
    return _padded(b, (pad4(b.shape[0]),), persistent)


WGRAD_STREAM_MIN_ROWS = 32768


def flush_wgrad(batch: dict) -> None:
    """run the weight-gradient problems grad_weight queued in `batch`: one launch per row count"""
    for M, items in batch.items():
        hip.wgrad_native_batch([it[0] for it in items], M)
    batch.clear()


def _splits(M: int, n_out: int = 256, k_in: int = 256) -> int:
    """split-K factor of a weight-gradient GEMM (reduction over M rows).  The split kernel keeps 2 workgroups per CU
    resident (512 on the chip), so the launch is sized to a whole number of such waves: tiles x splits = 512 w with
    the smallest w that keeps a split under 4096 rows (520 workgroups cost two full waves, 512 cost one); short
    reductions keep at least 128 rows per split."""
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    smin = max(1, (M + 4095) // 4096)
    waves = (tiles * smin + 511) // 512
    return max(1, min((512 * waves) // tiles, M // 128))


def grad_weight(dZ, X, M, n_out, k_in, like, bias_like=None, bias_rows=None, acc=None, a_native_nt=0, b_native_nt=0, a_scale_max=None,
                batch=None):
    """dW[n_out, k_in] = dZ[:M, :n_out]^T @ X[:M, :k_in] (split-K over M, atomically reduced).
    With bias_like, also returns db[n_out] = column sums of dZ (over the first `bias_rows` rows when the tail rows are
    tangent rows that carry no bias) from the same pass over dZ.  acc=(dW, db): accumulate into existing buffers
    (row-chunked callers).  batch: a dict the caller flushes with flush_wgrad at the end of its backward -- long row-major
    reductions are then only queued here and run as ONE launch per row count."""
    if acc is not None:
        dW, db = acc
    else:
        if bias_like is not None:  # one zero-filled slab for both accumulators (16-byte aligned views)
            nw = (like.numel() + 3) // 4 * 4
            flat = zeros(nw + bias_like.numel(), device=like.device)
            dW, db = flat[:like.numel()].view_as(like), flat[nw:].view_as(bias_like)
        else:
            dW, db = zeros_like(like), None
    if (_POLICY == "splith" and a_native_nt == 0 and b_native_nt == 0 and a_scale_max is None and M >= WGRAD_STREAM_MIN_ROWS
            and n_out >= 64 and k_in >= 64 and n_out % 4 == 0 and k_in % 4 == 0 and ld(dZ) % 4 == 0 and ld(X) % 4 == 0):
        # long reductions over row-major operands (the field's layers): the streaming kernel, dW and db in one pass
        prob = hip.wgrad_problem_rowmajor(dZ, n_out, X, k_in, M, dW, db, 0 if bias_rows is None else bias_rows)
        if batch is not None:
            batch.setdefault(M, []).append((prob, dZ, X))  # (the operands are kept alive until the flush)
        else:
            hip.wgrad_native_batch([prob], M)
        return dW if db is None else (dW, db)
    splits = _splits(M, n_out, k_in)
    kw = dict(a_kcontig=False, b_kcontig=False, k_splits=splits, precision=BWD_PRECISION, a_native_nt=a_native_nt, b_native_nt=b_native_nt)
    if a_scale_max is not None and k_in > 64:  # fp32-grade products: fp16 hi + scaled residual, the gradient pre-scaled by its maximum
        kw.update(precision=hip.PREC_F16X2, a_scale_max=a_scale_max)
    if acc is not None and splits <= 1:
        kw["beta"] = 1.0  # single-pass epilogue: add to what the previous chunks left
    if db is None:
        hip.gemm(dZ, X, dW, n_out, k_in, M, **kw)
        return dW
    if bias_rows is None or bias_rows == M:
        hip.gemm(dZ, X, dW, n_out, k_in, M, a_rowsum=db, **kw)
    elif bias_rows % 32 == 0:  # row sums over the leading (value) rows only, from the same pass over dZ
        hip.gemm(dZ, X, dW, n_out, k_in, M, a_rowsum=db, rowsum_k_limit=bias_rows, **kw)
    else:
        hip.gemm(dZ, X, dW, n_out, k_in, M, **kw)
        hip.colsum(dZ, bias_rows, n_out, db)
    return dW, db


# Gradient accumulators shared by every backward node of ONE backward pass that trains the same prepared weight (the geo /
# colour matrices feed three SDFAlbedoFn calls and the sdf probe): the node autograd runs first allocates the zero-filled
# (dW, db) pair and returns it, the later ones add into it in place (split-K atomics / beta = 1) and return None -- autograd
# runs the weight's producer only after all of them, so it sees the finished sum without 2 x 15 separate zero fills and adds.
_SHARED_GRADS: dict = {}
# A bias that is an optimizer-slab parameter (engine._Group.bind registers it: register_grad_sink) accumulates straight into its slab view
# and the nodes return None for it: no AccumulateGrad node ever touches the accumulator (which, while _SHARED_GRADS still refers to it,
# AccumulateGrad would CLONE on the main stream -- under a captured graph in front of the side stream's last adds: found as two wrong bias
# gradients under graph replay when the weight gradients first moved to the side stream).
_GRAD_SINKS: dict = {}   # data_ptr of a slab parameter -> weak reference to the parameter (a dead pipeline's entries must not catch a new
                         # pipeline's parameter that the allocator placed at the same address)
_SUNK_BIAS: set = set()  # data_ptrs of the slab views handed out as bias accumulators in this backward pass


def register_grad_sink(p) -> None:
    """`p` (a parameter whose .grad is a view of the optimizer's gradient slab, zeroed by the optimizer before every pass) takes its bias
    gradients IN PLACE from the second pass on: the backward nodes return None for it.  The contract of engine.Optimizers and of `.backward()`;
    `torch.autograd.grad(loss, [p])` on a registered parameter sees no gradient -- read `p.grad` after `.backward()` instead."""
    import weakref
    _GRAD_SINKS[p.data_ptr()] = weakref.ref(p)


def _grad_sink_of(t):
    ref = _GRAD_SINKS.get(t.data_ptr())
    if ref is None:
        return None
    sk = ref()
    if sk is None or sk.data_ptr() != t.data_ptr() or sk.shape != t.shape:
        if sk is None:
            del _GRAD_SINKS[t.data_ptr()]
        return None
    return sk


_PASS = {"queued": False}  # the end-of-pass callback of the running backward pass has been queued


def _queue_end_of_pass() -> None:
    if not _PASS["queued"]:
        _PASS["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass)


def finish_pass() -> None:
    """the end of a backward pass, on the stream `backward()` was called on: the last join of the side streams, the padded parameters'
    deferred gradients into their slab slots, then everything this pass shared or kept alive is released.  Idempotent: autograd runs it
    as a final callback, and whoever reads the slab first (the pipeline's exchange hook, queued in front of it) calls it itself."""
    join_weight_gradients()
    for gs, view in _DEFERRED_PADS:
        view.add_(gs)
    reset_pass_state()


_end_of_pass = finish_pass


def reset_pass_state() -> None:
    """forget the per-pass state.  Runs at the end of every backward pass AND at every step boundary (begin_step, the pipeline's
    get_train_loss_dict): autograd skips the final callbacks of a backward pass that raised (out of memory, a failed capture), and
    stale entries would make the next pass skip its join, return None for gradients that were never accumulated, and re-wait a side
    stream of an aborted capture."""
    _SHARED_GRADS.clear()
    _SUNK_BIAS.clear()
    _WGRAD_PENDING.clear()
    _WGRAD_KEEP.clear()
    _DEFERRED_PADS.clear()
    _PASS["queued"] = False


def first_only(first, t):
    """what a node returns to autograd for a shared accumulator: the tensor from the node that allocated it, None from the later ones
    (they added in place) and None for a bias that accumulates in its own slab view"""
    return t if (first and t is not None and t.data_ptr() not in _SUNK_BIAS) else None


def shared_grad(like, bias_like):
    """-> (dW, db, first)"""
    # per stream: nodes replayed on different streams are ordered only along autograd's edges, so they do not share a buffer
    sid = torch.cuda.current_stream().cuda_stream
    key = (like.data_ptr(), tuple(like.shape), 0 if bias_like is None else bias_like.data_ptr(), sid)
    hit = _SHARED_GRADS.get(key)
    if hit is not None:
        return hit[1], hit[2], False
    _queue_end_of_pass()  # forget the shared accumulators when the pass ends
    db = None
    sk = _grad_sink_of(bias_like) if bias_like is not None else None
    if sk is not None:
        if sk.grad is not None and sk.grad.shape == bias_like.shape and sk.grad.is_contiguous() and getattr(sk, "_nsky_sunk", False):
            db = sk.grad  # the slab view, zeroed by zero_grad_all
            _SUNK_BIAS.add(db.data_ptr())
        else:
            sk._nsky_sunk = True  # sinks from the next zero_grad_all on (which then keeps .grad as the slab view)
    nw = (like.numel() + 3) // 4 * 4
    flat = zeros(nw + (bias_like.numel() if (bias_like is not None and db is None) else 0), device=like.device)
    dW = flat[:like.numel()].view_as(like)
    if db is None and bias_like is not None:
        db = flat[nw:].view_as(bias_like)
    _SHARED_GRADS[key] = (like, dW, db)
    return dW, db, True


# ---------------------------------------------------------------------------------------------------------------------
# Weight gradients off the critical path.  The weight gradients of a chain (the streaming kernel: 2 ms for the DDF network) feed nothing
# in the backward pass -- they land in the optimizer slab (`sink` parameters) or in accumulators only the weight-norm / padding nodes at
# the very end of the pass read -- while the input gradients the same node returns are what the next nodes wait for (the hash-table
# scatter, an LDS-atomic kernel that leaves the matrix pipes and most of the HBM bandwidth idle).  So a backward node may launch its
# weight-gradient kernels on a SIDE stream: forked behind the node's own kernels, joined where their results are first read
# (join_weight_gradients: the weight-norm / padding backward nodes, and a callback at the end of the backward pass).  The operands are
# kept ALIVE until that join (references in _WGRAD_KEEP), so the allocator cannot hand their memory to a later kernel of the main stream;
# after the join any reuse is ordered behind the side stream's work.  (Not Tensor.record_stream: with it the flaky-subset runs of
# tools/flake.sh aborted one time in eight -- "free(): invalid pointer", host heap -- somewhere between the allocator's deferred events and
# the captured graphs' private pools; the tree before the side stream: 0 of 8, with plain references: see DESIGN section 4.)
# capture mode of every HIP graph of this package (the train step, the eval-latent fit, the render chunk).  thread_local: only the capturing
# thread is policed -- other host threads (the RCCL watchdog, a loader staging the next batch) may legally touch the runtime during a
# capture.  The captures themselves make no unsafe call from any thread ("global" passes too).  (tools/lab.py sets another for the flake hunt of DESIGN section 7; nothing in the package does)
CAPTURE_MODE = "thread_local"
ASYNC_WGRAD = True  # (tools/lab.py clears it for same-box A/B runs; nothing in the package does; arithmetic is identical either way)
# ---- one stream per ROLE and device for the life of the process -------------------------------------------------------------------
# torch.cuda.Stream() hands out 32 pool streams per priority round-robin: the 33rd object IS the 1st stream again.  Rounds 2-5 created
# streams per model / per capture (illumination stream, warm-up side stream, torch.cuda.graph's own capture stream), so in a process that
# builds many pipelines two "different" streams of one captured step end up the same pool stream -- a capturing stream then waits on an
# event recorded on itself, or forks into itself -- and round 6's extra stream moved the coincidences onto a case in which
# hipStreamEndCapture segfaults (deterministic, in the 80th test of the GPU suite).  The package's streams are therefore created ONCE,
# together (consecutive pool slots: distinct), and every capture runs on the package's own capture stream.
ROLES = ("capture", "illumination", "ddf_fit", "wgrad")
_ROLE_STREAMS: dict = {}  # device index -> {role: stream}


def role_stream(role: str, device=None) -> "torch.cuda.Stream":
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    streams = _ROLE_STREAMS.get(idx)
    if streams is None:
        with torch.cuda.device(idx):
            streams = _ROLE_STREAMS[idx] = {r: torch.cuda.Stream() for r in ROLES}
        assert len({s.stream_id for s in streams.values()}) == len(ROLES)
    return streams[role]


_WGRAD_PENDING: list = []  # side streams with unjoined work of THIS backward pass
_WGRAD_KEEP: list = []     # operands of the unjoined launches
_DEFERRED_PADS: list = []  # (gradient view, slab view) of padded slab parameters: added by finish_pass


# ---- retiring a captured graph -------------------------------------------------------------------------------------------------
# Round 6 root-caused the host-heap damage that rounds 4-5 chased behind the eval-latent fits: a use-after-free in the
# HIP runtime (libamdhip64 7.0.51831).  hipGraphLaunch holds a reference to the executable graph until the launch's last command
# completes; if the caller's handle is destroyed first (torch.cuda.CUDAGraph dropped right after its last replay) that reference is the
# LAST one and is released by the completion callback on ROCr's async-events thread -- HsaAmdSignalHandler -> VirtualGPU::
# updateCommandsState -> Event::processCallbacks -> GraphExec::~GraphExec -> Stream::terminate -> HostQueue::terminate frees the graph's
# internal streams' roc::VirtualGPU objects, and HsaAmdSignalHandler then touches the VirtualGPU it was called for: freed memory whenever
# the completing command ran on one of the graph's own streams (tools/heap_guard.c with HEAP_GUARD_FENCE_SIZE=920 faults at that
# instruction; tools/hip_graph_destroy_uaf.py reproduces it with torch alone).  A synchronize in front of the destruction is not enough:
# the callback runs on another thread, some time after the signal the synchronize waited for.  So no captured graph of this package is
# destroyed near its last launch: it is RETIRED -- kept alive here -- and destroyed by a later call, on the caller's thread, once it has
# been idle for seconds (the destructor then holds the last reference and runs outside the signal handler).
_RETIRED_GRAPHS: list = []  # [(graph or object holding graphs, time.monotonic() at retirement)]
RETIRE_SECONDS = 2.0


def retire_graph(obj) -> None:
    """keep `obj` (a torch.cuda.CUDAGraph, or anything whose destruction destroys captured graphs) alive past its last launch; destroy
    what was retired more than RETIRE_SECONDS ago, after a device synchronize"""
    import time
    now = time.monotonic()
    if obj is not None:
        _RETIRED_GRAPHS.append((obj, now))
    if any(now - t > RETIRE_SECONDS for _, t in _RETIRED_GRAPHS) and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        keep = [(o, t) for o, t in _RETIRED_GRAPHS if now - t <= RETIRE_SECONDS]
        del _RETIRED_GRAPHS[:]
        _RETIRED_GRAPHS.extend(keep)  # (the dropped entries' destructors run here, on this thread)


def async_weight_gradients(launch, operands) -> None:
    """run `launch()` (weight-gradient kernels only: nothing the caller returns may depend on them) on the side stream"""
    if not ASYNC_WGRAD:
        launch()
        return
    main = torch.cuda.current_stream()
    side = role_stream("wgrad", main.device)
    if side == main:  # (a caller running the step on a stream of its own that IS this pool stream: in line)
        launch()
        return
    side.wait_stream(main)
    with torch.cuda.stream(side):
        launch()
    _WGRAD_KEEP.append([t for t in operands if t is not None])
    _queue_end_of_pass()
    if side not in _WGRAD_PENDING:
        _WGRAD_PENDING.append(side)


def join_weight_gradients() -> None:
    """the current stream waits for every weight-gradient launch made so far in this backward pass.  A join never forgets anything: the
    model runs on two streams and a mid-pass join on one of them (a weight-norm / padding node) must leave the side streams pending for
    the other and for the end-of-pass callback, and the operands stay referenced until that callback (_end_of_pass) -- dropping them at
    a mid-pass join would let the joining stream's allocator hand out memory the side stream may still read for another consumer.
    A redundant wait costs an event."""
    if _WGRAD_PENDING:
        cur = torch.cuda.current_stream()
        for side in _WGRAD_PENDING:
            cur.wait_stream(side)


def join_unless_sunk(*bias_grads) -> None:
    """a bias accumulator that is not -- yet -- slab-resident (the first step, a caller without the engine's slabs) goes to autograd from
    the node that allocated it and is read by the leaf's AccumulateGrad as soon as the LAST node sharing it has run, whichever that is:
    EVERY node that added to one on the side stream joins before it returns, not only the one that hands the tensor over (round 5: the
    first eager step of one pipeline in eight lost a later node's share of `glin0.bias`, tools/flake_graph.py).  Slab-resident biases
    (every step after the first) are read by nothing before the end of the pass: no join, the launches stay off the critical path."""
    if any(t is not None and t.data_ptr() not in _SUNK_BIAS for t in bias_grads):
        join_weight_gradients()


def grad_bias(dZ, M, n_out, like):
    db = zeros_like(like)
    hip.colsum(dZ, M, n_out, db)
    return db


# Per-step cache of pre-split weight planes (hip.split_planes) for the LDS-DMA dense-layer kernel.  Keyed by the weight's
# storage; the weight tensor itself is held so its address cannot be recycled inside the step.  Dropped by begin_step()
# (the optimiser has changed the weights) -- models call it from their own begin_step.
_PLANES: dict = {}
_PLANES_K_STEP = 4   # the LDS-DMA kernel takes any K % 4 == 0 (a partial last k-tile multiplies the planes' zero padding)
_PLANES_MIN_K = 36   # below: the layer is all epilogue, the 32-deep k-tile mostly padding


# ---- zero arena ---------------------------------------------------------------------------------------------------------
# A train step needs ~10^2 zero-initialised tensors (split-k / atomic accumulators, padded outputs, gradient slabs of weights
# that are not optimizer-slab parameters).  Each torch.zeros is a fill launch; here they are carved out of ONE buffer that
# begin_step allocates and zero-fills once, sized by what the previous step asked for.  Every region is handed out once, so
# it is zero when it is handed out; the buffer lives as long as any view of it (autograd keeps them through the backward).
_ARENA = {"buf": None, "off": 0, "need": 0, "cap": 0}
_ARENA_ALIGN = 64  # floats: 256-byte aligned regions
_ARENA_MAX_REGION = 1 << 22  # floats; larger requests get their own allocation


def zeros(*shape, device) -> torch.Tensor:
    """float32 zeros of `shape`, from the step's arena when it has room (otherwise a torch.zeros of its own)"""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    n = 1
    for d in shape:
        n *= int(d)
    if n > _ARENA_MAX_REGION:  # (a whole hash table's gradient, when it is not an optimizer-slab parameter)
        return torch.zeros(shape, device=device)
    a = _ARENA
    n_al = (n + _ARENA_ALIGN - 1) // _ARENA_ALIGN * _ARENA_ALIGN
    a["need"] += n_al
    buf = a["buf"]
    if buf is not None and n > 0 and buf.device == torch.device(device) and a["off"] + n_al <= a["cap"]:
        # .data: same storage, but its own version counter and no view relation to the arena -- to autograd each region is an
        # independent tensor (in-place writes to one region must not invalidate tensors saved from another)
        out = buf[a["off"]:a["off"] + n].view(shape).data
        a["off"] += n_al
        return out
    return torch.zeros(shape, device=device)


def zeros_like(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        return torch.zeros_like(t)
    return zeros(tuple(t.shape), device=t.device)


_STEP_SEQ = [0]  # bumped by begin_step: cache entries remember the step they were prepared in


def stable_id(t: torch.Tensor):
    """identity of the PARAMETER a derived weight tensor (weight norm, padding) was made from, when its maker recorded one
    (`_nsky_src`): unlike the data pointer of the per-step temporary it is the same in every step and inside a graph capture"""
    return getattr(t, "_nsky_src", None) or t.data_ptr()


_STREAM_BUFS: dict = {}  # persistent (stream bytes, table) pairs of the packed weight streams, by (owner key, layout size)


def _stream_buffers(key, nbytes, n_table, device):
    """The packed weight stream of a network is re-packed every optimizer step INTO THE SAME BUFFER: zero-filled once (the pad slabs of
    partial groups are streamed through LDS but never written by a pack, so they stay zero as long as the layout -- part of the key --
    is the same), one fill launch per network and direction less per step, and a stable address under graph capture."""
    k = (key, int(nbytes), int(n_table), str(device))
    hit = _STREAM_BUFS.get(k)
    if hit is None:
        # eviction: never wholesale and never a buffer a captured HIP graph may hold the address of (its pack kernel would write the
        # stream into memory the allocator has handed to somebody else): only entries that were last touched OUTSIDE a capture, oldest first
        if len(_STREAM_BUFS) > 64:
            for old_k in [kk for kk, v in sorted(_STREAM_BUFS.items(), key=lambda kv: kv[1][2]) if not v[3]][:len(_STREAM_BUFS) - 64]:
                del _STREAM_BUFS[old_k]
        hit = _STREAM_BUFS[k] = [torch.zeros(int(nbytes), dtype=torch.uint8, device=device), torch.empty(int(n_table), device=device), 0, False]
    hit[2] = _STEP_SEQ[0]
    hit[3] = hit[3] or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    return hit[0], hit[1]


def _ready_mark():
    """(stream, event recorded behind what the caller has just launched on it): what a cache entry keeps of its preparation"""
    s = torch.cuda.current_stream()
    e = torch.cuda.Event()
    e.record(s)
    return s, e


def _order_after(mark, hit_seq) -> None:
    """a cached weight preparation (planes / packed stream) made on another stream of THIS step: order the current stream after it --
    after the PREPARATION (an event recorded behind it), not after everything the other stream was handed since: a wait for the whole
    stream made the small DDF-fit backward, enqueued late by autograd, wait for the main stream's entire backward (round 6).
    An entry that survives from an earlier step (a frozen network's stream) needs no edge -- steps are ordered by their caller --
    and must not get one: the stream it was made on may be the legacy stream, which a capturing stream cannot wait for."""
    cur = torch.cuda.current_stream()
    if hit_seq == _STEP_SEQ[0] and mark[0] != cur:
        cur.wait_event(mark[1])


def begin_step(device=None) -> None:
    _STEP_SEQ[0] += 1
    reset_pass_state()
    _PLANES.clear()
    _SDF_STREAMS.clear()
    _FIELD_STREAMS.clear()
    for key in [k for k, hit in _FILM_STREAMS.items() if any(t.requires_grad for t in hit[0])]:
        del _FILM_STREAMS[key]  # (streams packed from frozen weights -- keyed by storage and version -- stay)
    a = _ARENA
    a["cap"] = max(a["cap"], a["need"])
    a["need"], a["off"] = 0, 0
    a["buf"] = torch.zeros(a["cap"], device=device) if (device is not None and a["cap"] > 0 and torch.device(device).type == "cuda") else None


def _planes(W, n_rows, n_k, transpose, precision):
    key = (W.data_ptr(), W._version, ld(W), n_rows, n_k, transpose, precision)
    hit = _PLANES.get(key)
    if hit is None:
        hit = _PLANES[key] = (W, hip.split_planes(W, n_rows, n_k, transpose, precision), _ready_mark(), _STEP_SEQ[0])
    else:  # split on another stream (parallel passes of one step share the weights): order this stream after it
        _order_after(hit[2], hit[3])
    return hit[1]


def ld(t):
    return hip.ld(t)


# input gradients through FROZEN dense layers (the RENI++ attention decoder's 37 linear layers per step): three bf16 terms per operand,
# six MFMAs per product (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi: ~2^-24, bf16 keeps fp32's exponent range so a gradient needs no
# pre-scaling) instead of the exact-fp32 MFMA at 1/16 of the bf16 rate
FROZEN_DX_PRECISION = hip.PREC_BF16X3


def grad_input(dZ, W, M, k_in, n_red, out, precision=None, **epi):
    """dX[M, k_in] = dZ[M, n_red] @ W[n_red, k_in]   (W stored [out, in] = [n_red, k_in])."""
    return hip.gemm(dZ, W, out, M, k_in, n_red, a_kcontig=True, b_kcontig=False, precision=BWD_PRECISION if precision is None else precision, **epi)


# =============================================================================================
# weight norm (+ row / column re-ordering and zero padding for the consumer) in one launch each way
class WeightNormFn(torch.autograd.Function):
    """W[r,c] = g[sr] v[sr,sc] / ||v[sr]||, sr = row_map[r], sc = col_map[c] (-1 = structural zero): nn.utils.weight_norm
    as the reference's SDF / colour layers carry it (sdf_albedo_field.py:147-161), emitted directly in the layout the
    field kernels read (rows [feat | sdf | 0 0 0], columns [feat | pad | x PE | pad], ...).  Every row of v must be listed
    in row_map (so dv is written completely)."""

    @staticmethod
    def forward(ctx, v, g, row_map, col_map, inverse_col):
        v = v.contiguous()
        out = torch.empty(row_map.numel(), col_map.numel(), device=v.device)
        inv = torch.empty(v.shape[0], device=v.device)
        hip.weight_norm_fwd(v, g, row_map, col_map, out, inv)
        ctx.save_for_backward(v, g, inv, row_map, inverse_col)
        return out

    @staticmethod
    def backward(ctx, d_out):
        join_weight_gradients()  # d_out is a weight-gradient accumulator the side stream may still be adding to
        v, g, inv, row_map, inverse_col = ctx.saved_tensors
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        hip.weight_norm_bwd(d_out.contiguous(), v, g, inv, row_map, inverse_col, dv, dg)
        return dv, dg, None, None, None


def weight_norm_maps(out_features: int, in_features: int, rows, cols, device):
    """int32 device maps for WeightNormFn from python lists of source indices (-1 = zero row / column)"""
    assert sorted(r for r in rows if r >= 0) == list(range(out_features)), "every source row exactly once"
    used = [c for c in cols if c >= 0]
    assert len(set(used)) == len(used) and all(c < in_features for c in used)
    inverse = [-1] * in_features
    for oc, sc in enumerate(cols):
        if sc >= 0:
            inverse[sc] = oc
    t = lambda x: torch.tensor(x, dtype=torch.int32, device=device)  # noqa: E731
    return t(rows), t(cols), t(inverse)


# =============================================================================================
# hash-grid encode
# =============================================================================================
class HashEncodeFn(torch.autograd.Function):
    """rows = [x | PE(x) | hash(x)] (+ three tangent row blocks).  Output [P, ld] or stacked [4P, ld]."""

    @staticmethod
    def forward(ctx, x, table, geom, mode, include_x, pe_freqs, pe_max_exp, tangents, need_dx):
        P = x.shape[0]
        width = (3 if include_x else 0) + 6 * pe_freqs + 2 * geom.n_levels
        ldy = pad4(width)
        x = x.contiguous()
        out = torch.empty((4 * P if tangents else P), ldy, device=x.device)
        T = out[P:].view(3, P, ldy) if tangents else None
        hip.encode_fwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, out[:P], T)
        ctx.save_for_backward(x, table)
        ctx.cfg = (geom, mode, include_x, pe_freqs, pe_max_exp, tangents, need_dx, P, ldy)
        # Parameters re-homed into an optimizer slab (engine._Group) own a pre-zeroed gradient view: the backward scatters
        # straight into it instead of zero-filling a 49 MB table of its own and adding that to .grad afterwards
        sink = table if getattr(table, "_nsky_grad_sink", False) else None
        base = table._base
        if sink is None and base is not None and getattr(base, "_nsky_grad_sink", False) and base.numel() == table.numel() \
                and base.is_contiguous() and table.is_contiguous() and base.data_ptr() == table.data_ptr():
            sink = base  # the [n, 2] view of a flat parameter (tcnn's `params` layout): its gradient view is the same memory
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, table = ctx.saved_tensors
        geom, mode, include_x, pe_freqs, pe_max_exp, tangents, need_dx, P, ldy = ctx.cfg
        if not ctx.needs_input_grad[1]:  # a frozen table (the eval-latent fit): only the input gradient, no scatter
            if not need_dx:
                return None, None, None, None, None, None, None, None, None
            dx = torch.empty(P, 3, device=x.device)
            hip.encode_bwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, d_out.contiguous()[:P], None, None, dx)
            return dx, None, None, None, None, None, None, None, None
        d_out = d_out.contiguous()
        sink = ctx.sink
        accumulate = sink is not None and sink.grad is not None and sink.grad.is_contiguous() and sink.grad.numel() == table.numel()
        if accumulate or (sink is not None and sink.grad is None):
            # engine.Optimizers.zero_grad_all: this parameter's .grad stays its slab view.  (A parameter first used after step 0 -- a
            # loss enabled later -- arrives here with .grad dropped: it takes the slow path once and sinks from the next step on.)
            sink._nsky_sunk = True
        dtable = sink.grad.view_as(table) if accumulate else zeros_like(table)
        dx = torch.empty(P, 3, device=x.device) if need_dx else None
        dT = d_out[P:].view(3, P, ldy) if tangents else None
        hip.encode_bwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, d_out[:P], dT, dtable, dx)
        return dx, (None if accumulate else dtable), None, None, None, None, None, None, None


# =============================================================================================
# generic dense layer (used for the small proposal networks and heads)
# =============================================================================================
_ACT = {"none": hip.EPI_NONE, "relu": hip.EPI_RELU}


class DenseFn(torch.autograd.Function):
    """Y[M, n_out] = act(X[M, :k] @ Wp^T + b); X has ld = Wp.shape[1] (padded), Wp/b padded."""

    @staticmethod
    def forward(ctx, X, Wp, bp, n_out, act, need_dx):
        M, K = X.shape[0], Wp.shape[1]
        # (zero-filled only when it has pad columns the kernel does not write: a [262144, 64] proposal layer is 67 MB)
        Y = torch.empty(M, n_out, device=X.device) if n_out == Wp.shape[0] else zeros(M, Wp.shape[0], device=X.device)
        fgemm(X, Wp, Y, M, n_out, K, bias=bp, epi=_ACT[act])
        ctx.save_for_backward(X, Wp, bp, Y)
        ctx.cfg = (n_out, act, need_dx)
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, Wp, bp, Y = ctx.saved_tensors
        n_out, act, need_dx = ctx.cfg
        M, K = X.shape[0], Wp.shape[1]
        dZ = dY.contiguous()
        if act == "relu":
            dZ = torch.ops.aten.threshold_backward(dZ, Y, 0.0)  # dY where Y > 0, one pass (no mask tensor)
        dW = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:  # (a frozen layer -- the RENI++ decoder -- takes no weight gradient)
            dW, db = grad_weight(dZ, X, M, n_out, K, Wp, bp)
        dX = None
        if need_dx:
            dX = torch.empty(M, K, device=X.device)
            frozen = not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
            # (the split kernels stage whole 32-deep k-tiles: a 3- or 4-row head keeps the exact kernel, whose loads are guarded)
            split_ok = frozen and _POLICY != "f32" and Wp.shape[0] % 32 == 0 and K > 64
            grad_input(dZ, Wp, M, K, Wp.shape[0], dX, precision=FROZEN_DX_PRECISION if split_ok else None)
        return dX, dW, db, None, None, None


# =============================================================================================
# FiLM-SIREN (DDF network, RENI-shaped illumination decoder)
# =============================================================================================
_FILM_STREAMS: dict = {}


def forget_film_streams(wb) -> None:
    """drop the packed streams built from the padded weight list `wb` (FiLMSiren.invalidate_weight_cache)"""
    for key in [k for k, hit in _FILM_STREAMS.items() if hit[0] and wb and hit[0][0] is wb[0]]:
        del _FILM_STREAMS[key]


def _film_fused_ok(M, H, Hm, n_map, n_film, mw, fw, ow, x, cond) -> bool:
    return (FWD_PRECISION == hip.PREC_F16X2 and ld(x) <= 16
            and hip.film_supported(H, Hm, n_map, n_film, mw[0].shape[1], fw[0].shape[1], ow.shape[0])
            and ld(x) >= fw[0].shape[1] and ld(cond) >= mw[0].shape[1])


def _film_stream(wb, n_map, n_film, mw, mb, mwo, mbo, fw, fb, ow, ob, direction=0):
    """per-step cache of one packed weight stream of a network (direction 0 forward, 1 FiLM backward, 2 mapping backward;
    dropped by begin_step when any of its weights is trainable: the optimiser changed them) -> (descriptor, stream bytes, bias / scale table)"""
    key = (wb[0].data_ptr(), wb[0]._version, wb[-2].data_ptr(), n_map, n_film, direction)
    hit = _FILM_STREAMS.get(key)
    if hit is None:
        net = hip.film_net(mw[0].shape[1], fw[0].shape[1], ow.shape[0], mw, mb, mwo, mbo, fw, fb, ow, ob)
        nbytes, _ = hip.film_stream_layout(net, direction)
        # zero-filled once: the pad slabs of partial groups are streamed through LDS but never multiplied
        stream, table = _stream_buffers(("film", stable_id(wb[0]), n_map, n_film, direction), nbytes, hip.FILM_TABLE_FLOATS, wb[0].device)
        hip.film_pack(net, stream, table, direction)
        hit = _FILM_STREAMS[key] = (list(wb), net, stream, table, _ready_mark(), _STEP_SEQ[0])
    else:  # packed on another stream of the same step: order this stream after it
        _order_after(hit[4], hit[5])
    return hit[1], hit[2], hit[3]


class FilmSirenFn(torch.autograd.Function):
    """neusky/utils/siren.py:108-208 as a chain of fp32-MFMA layers with fused epilogues.

    args: x [M, pad4(in)], cond [M, pad4(cond)], n_map, n_film, train_weights, then padded tensors
          map_w0,map_b0,...,map_wo,map_bo, film_w0,film_b0,...,out_w,out_b.
    Returns the raw head output [M, pad4(out)] (activation applied by the caller)."""

    @staticmethod
    def _unpack(wb, n_map, n_film):
        mw = [wb[2 * i] for i in range(n_map)]
        mb = [wb[2 * i + 1] for i in range(n_map)]
        mwo, mbo = wb[2 * n_map], wb[2 * n_map + 1]
        o = 2 * n_map + 2
        fw = [wb[o + 2 * i] for i in range(n_film)]
        fb = [wb[o + 2 * i + 1] for i in range(n_film)]
        return mw, mb, mwo, mbo, fw, fb, wb[o + 2 * n_film], wb[o + 2 * n_film + 1], o

    @staticmethod
    def forward(ctx, x, cond, n_map, n_film, train_weights, need_dcond, *wb):
        M = x.shape[0]
        ctx.need_dx = x.requires_grad
        dev = x.device
        mw, mb, mwo, mbo, fw, fb, ow, ob, o = FilmSirenFn._unpack(wb, n_map, n_film)
        H, Hm = fw[0].shape[0], mw[0].shape[0]
        n_out_p = ow.shape[0]
        # nothing has to survive the call when no input needs a gradient (render / eval): two ping-pong activation
        # buffers, no pre-activation side output
        save = any(ctx.needs_input_grad)
        if _film_fused_ok(M, H, Hm, n_map, n_film, mw, fw, ow, x, cond):
            # one launch for the whole network (csrc/film_chain.hip): the [M, 2 n_film H] frequency / phase matrix is never
            # formed and no activation makes a round trip through HBM between layers; kept for the backward: mapping
            # activations, FiLM pre-activations and outputs
            net, stream, scales = _film_stream(wb, n_map, n_film, mw, mb, mwo, mbo, fw, fb, ow, ob)
            Mp = hip.film_rows(M)  # saved activations are tile-native [ceil32(M), H] matrices (include/neusky_hip.h)
            ys = [torch.empty(Mp, H, device=dev) for _ in range(n_film if save else min(2, n_film))]
            hs = [torch.empty(Mp, Hm, device=dev) for _ in range(n_map)] if save else None
            zs = [torch.empty(Mp, H, device=dev) for _ in range(n_film)] if save else None
            res = torch.empty(M, n_out_p, device=dev)
            hip.film_chain_fwd(net, stream, scales, cond, x, M, hs, zs, ys if save else [ys[i % len(ys)] for i in range(n_film)], res)
            if not save:
                return res
            ctx.save_for_backward(x, cond, x.new_empty(0), *hs, *ys, *zs, *wb)
            ctx.cfg = (n_map, n_film, train_weights, need_dcond, M, H, Hm)
            ctx.sinks = [w if getattr(w, "_nsky_grad_sink", False) else None for w in wb]
            ctx.pad_sinks = [getattr(w, "_nsky_pad_of", None) for w in wb]  # slab parameters behind a padding copy (_PadFn defers)
            return res
        if save:
            hs = [torch.empty(M, Hm, device=dev) for _ in range(n_map)]
            ys = [torch.empty(M, H, device=dev) for _ in range(n_film)]
            zs = [torch.empty(M, H, device=dev) for _ in range(n_film)]
        else:
            pp = [torch.empty(M, max(H, Hm), device=dev) for _ in range(2)]
            hs = [pp[i & 1][:, :Hm] for i in range(n_map)]
            ys = [pp[(n_map + i) & 1][:, :H] for i in range(n_film)]
            zs = [None] * n_film
        FP = torch.empty(M, 2 * n_film * H, device=dev)
        res = zeros(M, n_out_p, device=dev)
        for r0, r1 in [(0, M)]:
            m = r1 - r0
            # mapping network: (Linear, LeakyReLU(0.2)) * n  -> Linear to 2*n_film*H  (siren.py:114-119)
            h = cond[r0:r1]
            for i in range(n_map):
                fgemm(h, mw[i], hs[i][r0:r1], m, Hm, mw[i].shape[1], bias=mb[i], epi=hip.EPI_LEAKY, p0=0.2)
                h = hs[i][r0:r1]
            fp = FP[r0:r1]
            fgemm(h, mwo, fp, m, 2 * n_film * H, Hm, bias=mbo)
            # FiLM layers: sin((15 F + 30) (W y + b) + P)  (siren.py:141-144, :200)
            y = x[r0:r1]
            for i in range(n_film):
                fgemm(y, fw[i], ys[i][r0:r1], m, H, fw[i].shape[1], bias=fb[i], epi=hip.EPI_FILM, p0=15.0, p1=30.0,
                      aux0=fp[:, i * H:(i + 1) * H], aux1=fp[:, (n_film + i) * H:(n_film + i + 1) * H],
                      out1=zs[i][r0:r1] if save else None)
                y = ys[i][r0:r1]
            fgemm(y, ow, res[r0:r1], m, n_out_p, H, bias=ob)
        if not save:
            return res
        ctx.save_for_backward(x, cond, FP, *hs, *ys, *zs, *wb)
        ctx.cfg = (n_map, n_film, train_weights, need_dcond, M, H, Hm)
        # weights that ARE optimizer-slab parameters (no padding copy in between) take their gradient in place
        ctx.sinks = [w if getattr(w, "_nsky_grad_sink", False) else None for w in wb]
        return res

    @staticmethod
    def _backward_fused(ctx, d_res, x, cond, hs, ys, zs, wb, mw, mb, mwo, mbo, fw, fb, ow, ob, o):
        """backward of the fused forward: two chain kernels (csrc/film_chain.hip) produce every pre-activation gradient as a
        tile-native matrix (F / phase are re-formed in registers: no [M, 2 n_film H] matrix is read or recomputed through HBM)
        and d_cond; the parameter gradients are weight-gradient GEMMs straight over those matrices."""
        n_map, n_film, train_w, need_dcond, M, H, Hm = ctx.cfg
        train_w = train_w and any(ctx.needs_input_grad[6:])  # (frozen weights -- the eval-latent fit -- take no gradient)
        dev = x.device
        Mp = hip.film_rows(M)
        d_res = d_res.contiguous()
        net1, s1, t1 = _film_stream(wb, n_map, n_film, mw, mb, mwo, mbo, fw, fb, ow, ob, 1)
        net2, s2, t2 = _film_stream(wb, n_map, n_film, mw, mb, mwo, mbo, fw, fb, ow, ob, 2)
        dzs = [torch.empty(Mp, H, device=dev) for _ in range(n_film)]
        dfp = torch.empty(Mp, 2 * n_film * H, device=dev)
        rowmax = torch.empty(Mp, device=dev)
        gmax = zeros(n_film + 1 + n_map, device=dev)
        d_x = torch.empty(M, ld(x), device=dev) if ctx.need_dx else None  # the DDF's multi-view rays (ddf_model.py:297-322)
        hip.film_chain_bwd_film(net1, s1, t1, M, d_res, hs[-1], zs, dzs, dfp, rowmax, gmax[:n_film + 1], d_x)
        d_cond = torch.empty(M, ld(cond), device=dev) if need_dcond else None
        want_map = need_dcond or train_w
        dpres = [torch.empty(Mp, Hm, device=dev) for _ in range(n_map)] if want_map else None
        if want_map:
            hip.film_chain_bwd_map(net2, s2, t2, M, dfp, rowmax, hs, dpres, d_cond, gmax[n_film + 1:])
        grads: List[Optional[torch.Tensor]] = [None] * len(wb)
        sunk = [False] * len(wb)
        if train_w:
            for idx, t in enumerate(wb):
                sk = ctx.sinks[idx]
                if sk is not None and sk.grad is not None and sk.grad.shape == t.shape and sk.grad.is_contiguous():
                    grads[idx], sunk[idx] = sk.grad, True
                    sk._nsky_sunk = True
                elif sk is not None and sk.grad is None:
                    sk._nsky_sunk = True  # (first used after step 0: sinks from the next zero_grad_all on)
            sizes = [0 if sunk[idx] else (t.numel() + 3) // 4 * 4 for idx, t in enumerate(wb)]
            flat = zeros(max(sum(sizes), 4), device=dev)
            off = 0
            for idx, t in enumerate(wb):
                if not sunk[idx]:
                    grads[idx] = flat[off:off + t.numel()].view_as(t)
                    off += sizes[idx]
            acc = lambda iw: (grads[iw], grads[iw + 1])  # noqa: E731
            nt, ntm = H // 32, Hm // 32
            # layers whose two operands are tile-native (FiLM layers 1.., the mapping head, mapping layers 1..): ONE launch of the
            # streaming weight-gradient kernel for all of them; the rest (row-major d_res / x / cond operands) per layer
            native = []

            def wgrad(dZ, X, n_out, k_in, like, bias_like, iw, a_nt, b_nt, smax, x_scale=64.0):
                # x_scale: power of two applied to X before its fp16 split: 2^6 for sine outputs (|y| <= 1), 2^3 for the mapping
                # network's LeakyReLU activations (unbounded in principle: |h| up to 8000 stays inside fp16's range)
                if a_nt > 0 and b_nt > 0 and a_nt % 4 == 0 and b_nt % 4 == 0 and smax is not None:
                    native.append(hip.wgrad_problem(dZ, a_nt, X, b_nt, M, grads[iw], grads[iw + 1], smax, x_scale))
                elif a_nt == 0 and b_nt > 0 and n_out <= 4 and dZ.dim() == 2 and dZ.shape[1] == 4 and dZ.is_contiguous():
                    # the narrow head (1 or 3 outputs, padded to 4): a column sum of the tile-native activation under the rows' weights,
                    # exact fp32 on the vector units in one pass (the exact-fp32 GEMM spent 0.22 ms on this [M, 4]^T [M, H] product)
                    hip.native_weighted_colsum(X, b_nt, M, grads[iw], grads[iw + 1], w4=dZ, n_out=n_out)
                else:
                    grad_weight(dZ, X, M, n_out, k_in, like, bias_like, acc=acc(iw), a_native_nt=a_nt, b_native_nt=b_nt, a_scale_max=smax)

            def launch():
                wgrad(d_res, ys[-1], ow.shape[0], H, ow, ob, o + 2 * n_film, 0, nt, None)
                for i in range(n_film - 1, 0, -1):
                    wgrad(dzs[i], ys[i - 1], H, H, fw[i], fb[i], o + 2 * i, nt, nt, gmax[i:i + 1])
                wgrad(dzs[0], x, H, fw[0].shape[1], fw[0], fb[0], o, nt, 0, None)
                wgrad(dfp, hs[-1], 2 * n_film * H, Hm, mwo, mbo, 2 * n_map, 2 * n_film * nt, ntm, gmax[n_film:n_film + 1], 8.0)
                for l in range(n_map - 1, 0, -1):
                    wgrad(dpres[l], hs[l - 1], Hm, Hm, mw[l], mb[l], 2 * l, ntm, ntm, gmax[n_film + 1 + l:n_film + 2 + l], 8.0)
                k0 = mw[0].shape[1]
                wgrad(dpres[0], cond, Hm, k0, mw[0], mb[0], 0, ntm, 0, gmax[n_film + 1:n_film + 2] if k0 > 64 else None)
                if native:
                    hip.wgrad_native_batch(native, M)

            # every gradient lands in the optimizer slab -- directly, or (a padded copy of a slab parameter) as an add of this node's
            # accumulator deferred to the end of the pass -- so nothing the backward pass runs before its end depends on the launches: they
            # go to the side stream.  (The deferred share is NOT handed to autograd: the padding node may have a second producer -- the
            # DDF-fit rows are a node of their own -- and autograd would sum the two accumulators while this one is still being written.)
            pad = [(not sunk[i]) and ctx.pad_sinks[i] is not None and _slab_resident(ctx.pad_sinks[i]) for i in range(len(wb))]
            if all(sunk[i] or pad[i] for i in range(len(wb))):
                async_weight_gradients(launch, [d_res, x, cond, dfp, gmax, *ys, *hs, *dzs, *(dpres or [])])
                for i, t in enumerate(wb):
                    if pad[i]:
                        o = ctx.pad_sinks[i].shape
                        _DEFERRED_PADS.append((grads[i][:o[0], :o[1]] if len(o) == 2 else grads[i][:o[0]], ctx.pad_sinks[i].grad))
                        sunk[i] = True
                _queue_end_of_pass()
            else:
                launch()
        return (d_x, d_cond, None, None, None, None, *[None if sunk[i] else g for i, g in enumerate(grads)])

    @staticmethod
    def backward(ctx, d_res):
        n_map, n_film, train_w, need_dcond, M, H, Hm = ctx.cfg
        train_w = train_w and any(ctx.needs_input_grad[6:])  # (frozen weights -- the eval-latent fit -- take no gradient)
        sv = ctx.saved_tensors
        x, cond, FP = sv[0], sv[1], sv[2]
        hs = list(sv[3:3 + n_map])
        ys = list(sv[3 + n_map:3 + n_map + n_film])
        zs = list(sv[3 + n_map + n_film:3 + n_map + 2 * n_film])
        wb = sv[3 + n_map + 2 * n_film:]
        mw, mb, mwo, mbo, fw, fb, ow, ob, o = FilmSirenFn._unpack(wb, n_map, n_film)
        dev = x.device
        if FP.numel() == 0:
            return FilmSirenFn._backward_fused(ctx, d_res, x, cond, hs, ys, zs, wb, mw, mb, mwo, mbo, fw, fb, ow, ob, o)
        grads: List[Optional[torch.Tensor]] = [None] * len(wb)
        d_res = d_res.contiguous()
        n_out_p = ow.shape[0]
        NF = 2 * n_film * H
        sunk = [False] * len(wb)
        if train_w:  # gradient accumulators shared by every row chunk: the parameter's own .grad view where the weight is an
            # optimizer-slab parameter, otherwise views of ONE zero-filled slab
            for idx, t in enumerate(wb):
                sk = ctx.sinks[idx]
                if sk is not None and sk.grad is not None and sk.grad.shape == t.shape and sk.grad.is_contiguous():
                    grads[idx], sunk[idx] = sk.grad, True
                    sk._nsky_sunk = True
                elif sk is not None and sk.grad is None:
                    sk._nsky_sunk = True  # (first used after step 0: sinks from the next zero_grad_all on)
            sizes = [0 if sunk[idx] else (t.numel() + 3) // 4 * 4 for idx, t in enumerate(wb)]
            flat = zeros(max(sum(sizes), 4), device=dev)
            off = 0
            for idx, t in enumerate(wb):
                if not sunk[idx]:
                    grads[idx] = flat[off:off + t.numel()].view_as(t)
                    off += sizes[idx]
        acc = (lambda iw: (grads[iw], grads[iw + 1])) if train_w else (lambda iw: None)
        d_x = torch.empty(M, fw[0].shape[1], device=dev) if ctx.need_dx else None
        d_cond = torch.empty(M, mw[0].shape[1], device=dev) if need_dcond else None
        chunks = [(0, M)]
        mc = max(r1 - r0 for r0, r1 in chunks)
        dFP_buf = torch.empty(mc, NF, device=dev)
        dZa, dZb = torch.empty(mc, H, device=dev), torch.empty(mc, H, device=dev)
        dPa, dPb = torch.empty(mc, Hm, device=dev), torch.empty(mc, Hm, device=dev)
        for r0, r1 in chunks:
            m = r1 - r0
            fp, dFP = FP[r0:r1], dFP_buf[:m]
            dr = d_res[r0:r1]
            if train_w:
                grad_weight(dr, ys[-1][r0:r1], m, n_out_p, H, ow, ob, acc=acc(o + 2 * n_film))
            # walk the FiLM layers backwards; each dX GEMM applies the FiLM backward epilogue of the layer below
            i = n_film - 1
            dZ, dZo = dZa[:m], dZb[:m]
            grad_input(dr, ow, m, H, n_out_p, dZ, epi=hip.EPI_BWD_FILM, p0=15.0, p1=30.0, aux0=zs[i][r0:r1],
                       aux1=fp[:, i * H:(i + 1) * H], aux2=fp[:, (n_film + i) * H:(n_film + i + 1) * H],
                       out1=dFP[:, i * H:(i + 1) * H], out2=dFP[:, (n_film + i) * H:(n_film + i + 1) * H])
            while True:
                y_in = ys[i - 1][r0:r1] if i > 0 else x[r0:r1]
                k_in = fw[i].shape[1]
                if train_w:
                    grad_weight(dZ, y_in, m, H, k_in, fw[i], fb[i], acc=acc(o + 2 * i))
                if i == 0:
                    if d_x is not None:  # gradient w.r.t. the encoded direction rows (DDF multi-view rays, ddf_model.py:297-322)
                        grad_input(dZ, fw[0], m, k_in, H, d_x[r0:r1])
                    break
                j = i - 1
                grad_input(dZ, fw[i], m, H, H, dZo, epi=hip.EPI_BWD_FILM, p0=15.0, p1=30.0, aux0=zs[j][r0:r1],
                           aux1=fp[:, j * H:(j + 1) * H], aux2=fp[:, (n_film + j) * H:(n_film + j + 1) * H],
                           out1=dFP[:, j * H:(j + 1) * H], out2=dFP[:, (n_film + j) * H:(n_film + j + 1) * H])
                dZ, dZo = dZo, dZ
                i = j
            # mapping network
            if train_w:
                grad_weight(dFP, hs[-1][r0:r1], m, NF, Hm, mwo, mbo, acc=acc(2 * n_map))
            dpre, dpo = dPa[:m], dPb[:m]
            grad_input(dFP, mwo, m, Hm, NF, dpre, epi=hip.EPI_BWD_LEAKY, p0=0.2, aux0=hs[-1][r0:r1])
            for i in range(n_map - 1, -1, -1):
                h_in = hs[i - 1][r0:r1] if i > 0 else cond[r0:r1]
                k_in = mw[i].shape[1]
                if train_w:
                    grad_weight(dpre, h_in, m, Hm, k_in, mw[i], mb[i], acc=acc(2 * i))
                if i > 0:
                    grad_input(dpre, mw[i], m, Hm, Hm, dpo, epi=hip.EPI_BWD_LEAKY, p0=0.2, aux0=hs[i - 1][r0:r1])
                    dpre, dpo = dpo, dpre
                elif d_cond is not None:
                    grad_input(dpre, mw[0], m, k_in, Hm, d_cond[r0:r1])
        return (d_x, d_cond, None, None, None, None, *[None if sunk[i] else g for i, g in enumerate(grads)])


# =============================================================================================
# SDF + albedo field (geo net with forward-mode tangents, colour net)
# =============================================================================================
class SDFAlbedoFn(torch.autograd.Function):
    """forward_geonetwork + analytic d sdf/dx + colour net (sdf_albedo_field.py:211-269).

    ET: stacked encode rows [4N, 72] = [E; T0; T1; T2] from HashEncodeFn(tangents=True).
    Padded / permuted weights (see fields/sdf_albedo_field.py):
      W0 [256,72] b0; W1 [256,256] b1; W2 [260,256] rows = [feat(256) | sdf | 0 0 0], b2 [260];
      Wc0 [256,300] columns = [feat(256) | sdf-slot,3 pad (zero) | x(3) PE(36) | pad], bc0; Wc1, bc1; Wc2 [4,256], bc2 [4].
    Returns sdf [N], gradients [N,3], albedo [N,3]."""

    @staticmethod
    def forward(ctx, ET, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2, beta, want_albedo=True):
        N = ET.shape[0] // 4
        dev = ET.device
        Hd = W0.shape[0]
        Kin = W0.shape[1]
        A0 = torch.empty(4 * N, Hd, device=dev)
        S0 = torch.empty(N, Hd, device=dev)
        fgemm(ET[:N], W0, A0[:N], N, Hd, Kin, bias=b0, epi=hip.EPI_SOFTPLUS, p0=beta, out1=S0)
        fgemm(ET[N:], W0, A0[N:], 3 * N, Hd, Kin, epi=hip.EPI_MUL_AUX, aux0=S0, row_mod=N)
        A1 = torch.empty(4 * N, Hd, device=dev)
        S1 = torch.empty(N, Hd, device=dev)
        fgemm(A0[:N], W1, A1[:N], N, Hd, Hd, bias=b1, epi=hip.EPI_SOFTPLUS, p0=beta, out1=S1)
        fgemm(A0[N:], W1, A1[N:], 3 * N, Hd, Hd, epi=hip.EPI_MUL_AUX, aux0=S1, row_mod=N)
        GF = W2.shape[0] - 4  # geo feature dim (256)
        ldc = Wc0.shape[1]
        # (not zero-filled as a whole -- 118 MB at the step's 98 304 samples: only its pad columns, which meet zero weight columns)
        CIN = torch.empty(N, ldc, device=dev)
        fgemm(A1[:N], W2, CIN, N, GF + 1, Hd, bias=b2)  # [feat | sdf] straight into the colour-net input
        npe = 39  # x (3) + PE6 (36) columns of the encode row
        CIN[:, GF + 1:GF + 4] = 0.0
        CIN[:, GF + 4:GF + 4 + npe] = ET[:N, :npe]
        if GF + 4 + npe < ldc:
            CIN[:, GF + 4 + npe:] = 0.0
        G = zeros(3 * N, 4, device=dev)
        fgemm(A1[N:], W2[GF:GF + 1], G, 3 * N, 1, Hd)  # d sdf / d x_k = tangent . w_sdf
        Hc = Wc0.shape[0]
        if want_albedo:
            C0 = torch.empty(N, Hc, device=dev)
            fgemm(CIN, Wc0, C0, N, Hc, ldc, bias=bc0, epi=hip.EPI_RELU)
            C1 = torch.empty(N, Hc, device=dev)
            fgemm(C0, Wc1, C1, N, Hc, Hc, bias=bc1, epi=hip.EPI_RELU)
            ALB = zeros(N, 4, device=dev)
            fgemm(C1, Wc2, ALB, N, 3, Hc, bias=bc2, epi=hip.EPI_SIGMOID, p0=1.0)
        else:  # geometry-only pass (DDF-fit ground truth, hash-grid density probe): the colour net's output is never read
            C0 = C1 = ALB = CIN.new_empty(0)
        ctx.save_for_backward(ET, A0, S0, A1, S1, CIN, C0, C1, ALB, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2)
        ctx.cfg = (N, Hd, Kin, GF, ldc, Hc, beta)
        ctx.want_albedo = want_albedo
        ctx.set_materialize_grads(False)
        sdf = CIN[:, GF].clone()
        grad = G.view(3, N, 4)[:, :, 0].t().contiguous()
        if not want_albedo:
            alb = zeros(N, 3, device=sdf.device)
            ctx.mark_non_differentiable(alb)
            return sdf, grad, alb
        return sdf, grad, ALB[:, :3].clone()

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_alb):
        ET, A0, S0, A1, S1, CIN, C0, C1, ALB, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ctx.saved_tensors
        N, Hd, Kin, GF, ldc, Hc, beta = ctx.cfg
        dev = ET.device
        colour = ctx.want_albedo and g_alb is not None
        if not colour:
            # no gradient reaches the colour net (geometry-only pass, or albedo unused downstream): only the sdf slot of the
            # geo net's [feat | sdf | 0 0 0] output carries one
            return SDFAlbedoFn._backward_geo(ctx, g_sdf, g_grad, None, {})
        # ---- colour net
        dpc2 = zeros(N, 4, device=dev)
        if g_alb is not None:
            alb = ALB[:, :3]
            dpc2[:, :3] = g_alb * alb * (1.0 - alb)
        join_weight_gradients()  # this path adds on the current stream, to accumulators a fused node of the same pass may be filling on the side stream
        dWc2, dbc2, f_c2 = shared_grad(Wc2, bc2)
        wq: dict = {}  # the layers' weight gradients are queued and run as one launch per row count at the end
        grad_weight(dpc2, C1, N, 4, Hc, Wc2, bc2, acc=(dWc2, dbc2), batch=wq)
        dpc1 = torch.empty(N, Hc, device=dev)
        grad_input(dpc2, Wc2, N, Hc, 4, dpc1, epi=hip.EPI_BWD_RELU, aux0=C1)
        dWc1, dbc1, f_c1 = shared_grad(Wc1, bc1)
        grad_weight(dpc1, C0, N, Hc, Hc, Wc1, bc1, acc=(dWc1, dbc1), batch=wq)
        dpc0 = torch.empty(N, Hc, device=dev)
        grad_input(dpc1, Wc1, N, Hc, Hc, dpc0, epi=hip.EPI_BWD_RELU, aux0=C0)
        dWc0, dbc0, f_c0 = shared_grad(Wc0, bc0)
        grad_weight(dpc0, CIN, N, Hc, ldc, Wc0, bc0, acc=(dWc0, dbc0), batch=wq)
        dCIN = torch.empty(N, ldc, device=dev)
        grad_input(dpc0, Wc0, N, ldc, Hc, dCIN)
        # the sdf slot / pad columns of Wc0 are structural zeros: overwrite them with the upstream sdf gradient
        dCIN[:, GF:GF + 4] = 0.0
        if g_sdf is not None:
            dCIN[:, GF] = g_sdf
        return SDFAlbedoFn._backward_geo(ctx, g_sdf, g_grad, (dCIN, dWc0, dbc0, f_c0, dWc1, dbc1, f_c1, dWc2, dbc2, f_c2), wq)

    @staticmethod
    def _backward_geo(ctx, g_sdf, g_grad, colour, wq):
        """geo-net part of the backward; colour = (dCIN, colour-net weight gradients ...) or None when the colour net took none;
        wq: the weight-gradient problems queued so far (flushed here, at the end)"""
        ET, A0, S0, A1, S1, CIN, C0, C1, ALB, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ctx.saved_tensors
        N, Hd, Kin, GF, ldc, Hc, beta = ctx.cfg
        dev = ET.device
        if colour is not None:
            dCIN, dWc0, dbc0, f_c0, dWc1, dbc1, f_c1, dWc2, dbc2, f_c2 = colour
            dH = dCIN[:, :GF + 4]
        else:
            dCIN = None
            dH = zeros(N, GF + 4, device=dev)
            if g_sdf is not None:
                dH[:, GF] = g_sdf
        # ---- geo net, last layer (value rows)
        join_weight_gradients()  # (the geometry-only entry of this path: adds on the current stream, see backward)
        dW2, db2, f_2 = shared_grad(W2, b2)
        grad_weight(dH, A1[:N], N, GF + 4, Hd, W2, b2, acc=(dW2, db2), batch=wq)
        dA1v = torch.empty(N, Hd, device=dev)
        grad_input(dH, W2, N, Hd, GF + 4, dA1v)
        # tangent rows of the last layer: grad_k = ta1_k . w_sdf
        if g_grad is None:
            g_grad = zeros(N, 3, device=dev)
        g_grad = g_grad.contiguous()
        w2s = W2[GF].contiguous()
        # ---- layer 1 (reverse over forward); the same pass over the tangent rows accumulates d w_sdf += sum_{k,n} g_grad[n,k] ta1_k[n,:]
        # straight into the sdf row of dW2
        D1 = torch.empty(4 * N, Hd, device=dev)
        hip.softplus_tangent_bwd(dA1v, S1, A1[N:], None, g_grad, w2s, beta, N, Hd, D1[:N], D1[N:], wsum=dW2[GF])
        dW1, db1, f_1 = shared_grad(W1, b1)
        grad_weight(D1, A0, 4 * N, Hd, Hd, W1, b1, bias_rows=N, acc=(dW1, db1), batch=wq)
        dA0 = torch.empty(4 * N, Hd, device=dev)
        grad_input(D1, W1, 4 * N, Hd, Hd, dA0)
        # ---- layer 0
        D0 = torch.empty(4 * N, Hd, device=dev)
        hip.softplus_tangent_bwd(dA0[:N], S0, A0[N:], dA0[N:], None, None, beta, N, Hd, D0[:N], D0[N:])
        dW0, db0, f_0 = shared_grad(W0, b0)
        grad_weight(D0, ET, 4 * N, Hd, Kin, W0, b0, bias_rows=N, acc=(dW0, db0), batch=wq)
        dET = torch.empty(4 * N, Kin, device=dev)
        grad_input(D0, W0, 4 * N, Kin, Hd, dET)
        if dCIN is not None:  # x / PE columns of the colour-net input came straight from the encode row
            dET[:N, :39] += dCIN[:, GF + 4:GF + 4 + 39]
        flush_wgrad(wq)
        k = first_only  # later nodes of the pass added in place; slab-resident biases return nothing
        if colour is None:
            return (dET, k(f_0, dW0), k(f_0, db0), k(f_1, dW1), k(f_1, db1), k(f_2, dW2), k(f_2, db2), None, None, None, None, None, None,
                    None, None)
        return (dET, k(f_0, dW0), k(f_0, db0), k(f_1, dW1), k(f_1, db1), k(f_2, dW2), k(f_2, db2), k(f_c0, dWc0), k(f_c0, dbc0),
                k(f_c1, dWc1), k(f_c1, dbc1), k(f_c2, dWc2), k(f_c2, dbc2), None, None)


# =============================================================================================
# SDF + albedo field as chain kernels (csrc/field_chain.hip)
# =============================================================================================
_FIELD_STREAMS: dict = {}


def _field_pack(kind, weights, layers_fn):
    """per-step cache of one packed weight stream of the field (dropped by begin_step) -> (stream, scales, groups)"""
    key = (kind,) + tuple((w.data_ptr(), w._version) for w in weights)
    hit = _FIELD_STREAMS.get(key)
    if hit is None:
        dev = weights[0].device
        pk = hip.chain_pack(layers_fn(), dev, lambda nb, nt: _stream_buffers((kind,) + tuple(stable_id(w) for w in weights), nb, nt, dev))
        hit = _FIELD_STREAMS[key] = (weights, pk, _ready_mark(), _STEP_SEQ[0])
    else:
        _order_after(hit[2], hit[3])
    return hit[1]


def field_fused_ok(ET, W0, W1, W2, Wc0, Wc1) -> bool:
    return (FWD_PRECISION == hip.PREC_F16X2 and ET.shape[0] >= 4 and ld(ET) == W0.shape[1]
            and hip.field_supported(W0.shape[1], W0.shape[0], W2.shape[0] - 4, Wc1.shape[0], Wc0.shape[1]))


class FieldChainFn(torch.autograd.Function):
    """SDFAlbedoFn's contract (same arguments, same outputs) on the fused field kernels: geometry network with forward-mode tangents
    in the quad layout, colour path, and the hand-derived reverse of both; every product fp32-grade (fp16 hi + residual planes on
    power-of-two pre-scaled operands), weight gradients by the streaming tile-native kernel."""

    NPE = 39  # x (3) + PE6 (36): the leading columns of an encode row that also feed the colour net

    @staticmethod
    def forward(ctx, ET, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2, beta, want_albedo=True):
        N = ET.shape[0] // 4
        dev = ET.device
        GF = W2.shape[0] - 4
        Kin = W0.shape[1]
        save = any(ctx.needs_input_grad)
        w_sdf, b_sdf = W2[GF], b2[GF:GF + 1]
        net = hip.field_net(Kin, FieldChainFn.NPE, beta, b0, b1, w_sdf, b_sdf, b2[:GF], bc0, bc1, Wc2, bc2)
        Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
        a0q, a1q = torch.empty(Mq, 256, device=dev), torch.empty(Mq, 256, device=dev)
        Eq = torch.empty(Mq, 128, device=dev) if save else None
        a1max = torch.empty(N, device=dev)
        sdf, grad = torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
        pk = _field_pack("geo_fwd", (W0, W1), lambda: [hip.chain_layer(W0, 256, Kin), hip.chain_layer(W1, 256, 256)])
        # largest |Eq|, |a0q| (tangent rows included): the operand scales of the first two layers' weight gradients
        qmax = zeros(2, device=dev) if save else None
        hip.field_geo_fwd(net, pk, ET, N, a0q, a1q, Eq, a1max, sdf, grad, qmax)
        ctx.qmax = qmax
        if want_albedo:
            a1v = torch.empty(Mp, 256, device=dev) if save else None
            feat, c0, c1 = torch.empty(Mp, 256, device=dev), torch.empty(Mp, 256, device=dev), torch.empty(Mp, 256, device=dev)
            xpe = torch.empty(Mp, 128, device=dev)
            alb = torch.empty(N, 4, device=dev)
            pk = _field_pack("col_fwd", (W2, Wc0, Wc1), lambda: [hip.chain_layer(W2, 256, 256), hip.chain_layer(Wc0, 256, Wc0.shape[1]),
                                                                  hip.chain_layer(Wc1, 256, 256)])
            hip.field_colour_fwd(net, pk, ET, N, a1q, a1max, a1v, feat, xpe, c0, c1, alb)
        else:
            a1v = feat = xpe = c0 = c1 = alb = None
        ctx.want_albedo = want_albedo
        ctx.cfg = (N, GF, Kin, beta)
        ctx.set_materialize_grads(False)
        if save:
            e = ET.new_empty(0)
            ctx.save_for_backward(ET, a0q, a1q, Eq, *((a1v, feat, xpe, c0, c1, alb) if want_albedo else (e, e, e, e, e, e)),
                                  W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2)
        if not want_albedo:
            out_alb = zeros(N, 3, device=dev)
            ctx.mark_non_differentiable(out_alb)
            return sdf, grad, out_alb
        return sdf, grad, alb[:, :3].contiguous()

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_alb):
        ET, a0q, a1q, Eq, a1v, feat, xpe, c0, c1, alb, W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ctx.saved_tensors
        N, GF, Kin, beta = ctx.cfg
        dev = ET.device
        colour = ctx.want_albedo and g_alb is not None
        net = hip.field_net(Kin, FieldChainFn.NPE, beta, b0, b1, W2[GF], b2[GF:GF + 1], b2[:GF], bc0, bc1, Wc2, bc2)
        Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
        gmax = zeros(8, device=dev)
        g_sdf = None if g_sdf is None else g_sdf.contiguous()
        g_grad = None if g_grad is None else g_grad.contiguous()
        da1v = dxpe = None
        if colour:
            dpc2 = torch.empty(N, 4, device=dev)
            dpc1, dpc0, dfeat, da1v = (torch.empty(Mp, 256, device=dev) for _ in range(4))
            dxpe = torch.empty(N, 40, device=dev)
            pk = _field_pack("col_bwd", (Wc1, Wc0, W2), lambda: [hip.chain_layer(Wc1, 256, 256, True), hip.chain_layer(Wc0, Wc0.shape[1], 256, True),
                                                                  hip.chain_layer(W2, 256, 256, True)])
            hip.field_colour_bwd(net, pk, N, g_alb.contiguous(), alb, c0, c1, dpc2, dpc1, dpc0, dfeat, dxpe, da1v, gmax[:3])
        d1q, d0q = torch.empty(Mq, 256, device=dev), torch.empty(Mq, 256, device=dev)
        dET = torch.empty(4 * N, ld(ET), device=dev) if ctx.needs_input_grad[0] else None
        pk = _field_pack("geo_bwd", (W1, W0), lambda: [hip.chain_layer(W1, 256, 256, True), hip.chain_layer(W0, Kin, 256, True)])
        hip.field_geo_bwd(net, pk, N, g_sdf, g_grad, da1v, dxpe, a0q, a1q, d1q, d0q, dET, gmax[4:6])
        none6 = (None,) * 6
        if not any(ctx.needs_input_grad[1:13]):  # frozen field (the eval-latent fit): only the encode rows' gradient
            return (dET, *none6, *none6, None, None)
        # ---- parameter gradients: every dense layer's by the streaming tile-native kernel (value rows only for the biases of the
        # quad-native gradients), the two narrow output layers' by weighted column sums
        dW2, db2, f_2 = shared_grad(W2, b2)
        dW1, db1, f_1 = shared_grad(W1, b1)
        dW0, db0, f_0 = shared_grad(W0, b0)
        R4 = 4 * N
        qmax = ctx.qmax
        # (the accumulators are read by the weight-norm nodes at the end of the pass, which join the side stream: async_weight_gradients)

        def launch_geo():
            hip.wgrad_native_batch([hip.wgrad_problem(d1q, 8, a0q, 8, R4, dW1, db1, gmax[4:5], 8.0, bias_row_mod=4, b_scale_max=qmax[1:2])], R4)
            hip.wgrad_native_batch([hip.wgrad_problem(d0q, 8, Eq, 4, R4, dW0, db0, gmax[5:6], 64.0, width_b=Kin, bias_row_mod=4, b_scale_max=qmax[0:1])], R4)
            hip.native_weighted_colsum(a1q, 8, R4, dW2[GF], db2[GF:GF + 1], g_sdf=g_sdf, g_grad=g_grad)

        k = first_only  # later nodes of the pass added in place; slab-resident biases return nothing
        if not colour:
            async_weight_gradients(launch_geo, [d1q, a0q, d0q, Eq, a1q, g_sdf, g_grad, gmax, qmax])
            join_unless_sunk(db0, db1, db2)
            return (dET, k(f_0, dW0), k(f_0, db0), k(f_1, dW1), k(f_1, db1), k(f_2, dW2), k(f_2, db2), *none6, None, None)
        dWc2, dbc2, f_c2 = shared_grad(Wc2, bc2)
        dWc1, dbc1, f_c1 = shared_grad(Wc1, bc1)
        dWc0, dbc0, f_c0 = shared_grad(Wc0, bc0)

        def launch_all():
            launch_geo()
            hip.wgrad_native_batch([hip.wgrad_problem(dfeat, 8, a1v, 8, N, dW2[:GF], db2[:GF], gmax[2:3], 8.0),
                                    hip.wgrad_problem(dpc1, 8, c0, 8, N, dWc1, dbc1, gmax[0:1], 8.0),
                                    hip.wgrad_problem(dpc0, 8, feat, 8, N, dWc0[:, :GF], dbc0, gmax[1:2], 8.0)], N)
            hip.wgrad_native_batch([hip.wgrad_problem(dpc0, 8, xpe, 4, N, dWc0[:, GF:], None, gmax[1:2], 64.0, width_b=Wc0.shape[1] - GF)], N)
            hip.native_weighted_colsum(c1, 8, N, dWc2, dbc2, w4=dpc2, n_out=3)

        async_weight_gradients(launch_all, [d1q, a0q, d0q, Eq, a1q, g_sdf, g_grad, gmax, qmax, dfeat, a1v, dpc1, c0, dpc0, feat, xpe, c1, dpc2])
        join_unless_sunk(db0, db1, db2, dbc0, dbc1, dbc2)
        return (dET, k(f_0, dW0), k(f_0, db0), k(f_1, dW1), k(f_1, db1), k(f_2, dW2), k(f_2, db2), k(f_c0, dWc0), k(f_c0, dbc0),
                k(f_c1, dWc1), k(f_c1, dbc1), k(f_c2, dWc2), k(f_c2, dbc2), None, None)


def field_apply(ET, *args):
    """the field on stacked encode rows: the fused chain kernels where they apply, otherwise the per-layer path"""
    W0, _, W1, _, W2, _, Wc0, _, Wc1 = args[:9]
    if field_fused_ok(ET, W0, W1, W2, Wc0, Wc1):
        return FieldChainFn.apply(ET, *args)
    return SDFAlbedoFn.apply(ET, *args)


_SDF_STREAMS: dict = {}


def _sdf_stream(W0, b0, W1, b1, W2, b2, GF, beta, direction):
    """per-step cache of the packed weight stream of the sdf value chain (direction 0 forward, 1 backward; dropped by begin_step)"""
    key = (W0.data_ptr(), W0._version, W1.data_ptr(), W2.data_ptr(), GF, direction)
    hit = _SDF_STREAMS.get(key)
    if hit is None:
        keep = (W0, b0, W1, b1, W2, b2, W2[GF], b2[GF:GF + 1])
        net = hip.sdf_net(W0, b0, W1, b1, keep[6], keep[7], beta)
        nbytes, _ = hip.sdf_stream_layout(net, direction)
        stream, table = _stream_buffers(("sdf", stable_id(W0), stable_id(W1), GF, direction), nbytes, hip.FILM_TABLE_FLOATS, W0.device)
        hip.sdf_pack(net, stream, table, direction)
        hit = _SDF_STREAMS[key] = (keep, net, stream, table, _ready_mark(), _STEP_SEQ[0])
    else:
        _order_after(hit[4], hit[5])
    return hit[1], hit[2], hit[3]


class SDFValueFn(torch.autograd.Function):
    """get_sdf_at_pos (sdf_albedo_field.py:169-174): value-only geo net on encode rows E [M,72] -> sdf [M].
    Long batches (the DDF termination points) run the fused value chain: one kernel each way plus the weight gradients."""

    @staticmethod
    def forward(ctx, E, W0, b0, W1, b1, W2, b2, beta, train_weights):
        M = E.shape[0]
        dev = E.device
        Hd, Kin = W0.shape
        GF = W2.shape[0] - 4
        ctx.fused = (FWD_PRECISION == hip.PREC_F16X2 and hip.sdf_supported(Kin, Hd)
                     and ld(E) >= Kin and ld(E) % 4 == 0)
        ctx.cfg = (M, Hd, Kin, GF, beta, train_weights)
        if ctx.fused:
            net, stream, table = _sdf_stream(W0, b0, W1, b1, W2, b2, GF, beta, 0)
            Mp = hip.film_rows(M)
            A0 = torch.empty(Mp, Hd, device=dev); A1 = torch.empty(Mp, Hd, device=dev)
            sdf = torch.empty(M, device=dev)
            hip.sdf_chain_fwd(net, stream, table, E, M, A0, A1, sdf)
            if any(ctx.needs_input_grad):
                # the backward's stream is packed here, in a quiet stretch of the step: launched from the backward, its 11 small
                # workgroups (34 KB of LDS each) queue behind the illumination decoder's chain kernels on the second stream, which hold
                # every CU's LDS (0.28 ms on the timeline for 10 us of work)
                _sdf_stream(W0, b0, W1, b1, W2, b2, GF, beta, 1)
            ctx.save_for_backward(E, A0, A1, W0, b0, W1, b1, W2, b2)
            return sdf
        A0 = torch.empty(M, Hd, device=dev); S0 = torch.empty(M, Hd, device=dev)
        fgemm(E, W0, A0, M, Hd, Kin, bias=b0, epi=hip.EPI_SOFTPLUS, p0=beta, out1=S0)
        A1 = torch.empty(M, Hd, device=dev); S1 = torch.empty(M, Hd, device=dev)
        fgemm(A0, W1, A1, M, Hd, Hd, bias=b1, epi=hip.EPI_SOFTPLUS, p0=beta, out1=S1)
        out = zeros(M, 4, device=dev)
        fgemm(A1, W2[GF:GF + 1], out, M, 1, Hd, bias=b2[GF:GF + 1])
        ctx.save_for_backward(E, A0, S0, A1, S1, W0, b0, W1, b1, W2, b2)
        return out[:, 0]

    @staticmethod
    def _backward_fused(ctx, g_sdf):
        E, A0, A1, W0, b0, W1, b1, W2, b2 = ctx.saved_tensors
        M, Hd, Kin, GF, beta, train_w = ctx.cfg
        train_w = train_w and any(ctx.needs_input_grad[1:7])  # (frozen weights -- the eval-latent fit -- take no gradient)
        dev = E.device
        net, stream, table = _sdf_stream(W0, b0, W1, b1, W2, b2, GF, beta, 1)
        Mp = hip.film_rows(M)
        dZ1 = torch.empty(Mp, Hd, device=dev); dZ0 = torch.empty(Mp, Hd, device=dev)
        dE = torch.empty(M, ld(E), device=dev) if ctx.needs_input_grad[0] else None
        gmax = zeros(4, device=dev)
        dW0 = db0 = dW1 = db1 = dW2 = db2 = None
        f0 = f1 = f2 = False
        if train_w:
            dW2, db2, f2 = shared_grad(W2, b2)
            dW1, db1, f1 = shared_grad(W1, b1)
            dW0, db0, f0 = shared_grad(W0, b0)
        hip.sdf_chain_bwd(net, stream, table, M, g_sdf.contiguous(), A0, A1, dZ1, dZ0, dE,
                          dW2[GF] if train_w else None, db2[GF:GF + 1] if train_w else None, gmax)
        if train_w:
            nt = Hd // 32

            def launch():
                # softplus outputs are unbounded in principle: 2^3 keeps |a| up to 8000 inside fp16's range (as for the mapping network)
                hip.wgrad_native_batch([hip.wgrad_problem(dZ1, nt, A0, nt, M, dW1, db1, gmax[0:1], 8.0)], M)
                grad_weight(dZ0, E, M, Hd, Kin, W0, b0, acc=(dW0, db0), a_native_nt=nt, b_native_nt=0, a_scale_max=gmax[1:2])

            async_weight_gradients(launch, [dZ1, A0, dZ0, E, gmax])
            join_unless_sunk(db0, db1)  # (db2's share came from the chain kernel above, on this stream)
            if not f2: dW2 = None
            if not f1: dW1 = None
            if not f0: dW0 = None
            db0, db1, db2 = first_only(f0, db0), first_only(f1, db1), first_only(f2, db2)
        return dE, dW0, db0, dW1, db1, dW2, db2, None, None

    @staticmethod
    def backward(ctx, g_sdf):
        if ctx.fused:
            return SDFValueFn._backward_fused(ctx, g_sdf)
        E, A0, S0, A1, S1, W0, b0, W1, b1, W2, b2 = ctx.saved_tensors
        M, Hd, Kin, GF, beta, train_w = ctx.cfg
        train_w = train_w and any(ctx.needs_input_grad[1:7])
        dev = E.device
        g = zeros(M, 4, device=dev)
        g[:, 0] = g_sdf
        w2s = zeros(4, Hd, device=dev)
        w2s[0] = W2[GF]
        dZ1 = torch.empty(M, Hd, device=dev)
        grad_input(g, w2s, M, Hd, 4, dZ1, epi=hip.EPI_MUL_AUX, aux0=S1)  # da1 * softplus'(z1)
        dZ0 = torch.empty(M, Hd, device=dev)
        grad_input(dZ1, W1, M, Hd, Hd, dZ0, epi=hip.EPI_MUL_AUX, aux0=S0)
        dE = torch.empty(M, Kin, device=dev)
        grad_input(dZ0, W0, M, Kin, Hd, dE)
        dW0 = db0 = dW1 = db1 = dW2 = db2 = None
        if train_w:
            join_weight_gradients()  # (as in SDFAlbedoFn.backward: adds on the current stream)
            dW2, db2, f2 = shared_grad(W2, b2)
            hip.weighted_colsum(A1, M, Hd, g, 4, dW2[GF])
            db2[GF] += g_sdf.sum()
            dW1, db1, f1 = shared_grad(W1, b1)
            wq: dict = {}
            grad_weight(dZ1, A0, M, Hd, Hd, W1, b1, acc=(dW1, db1), batch=wq)
            dW0, db0, f0 = shared_grad(W0, b0)
            grad_weight(dZ0, E, M, Hd, Kin, W0, b0, acc=(dW0, db0), batch=wq)
            flush_wgrad(wq)
            if not f2: dW2 = None
            if not f1: dW1 = None
            if not f0: dW0 = None
            db0, db1, db2 = first_only(f0, db0), first_only(f1, db1), first_only(f2, db2)
        return dE, dW0, db0, dW1, db1, dW2, db2, None, None


# =============================================================================================
# render stages
# =============================================================================================
class HemiCompositeFn(torch.autograd.Function):
    """renderers.py:60-130 on compact inputs -> rgb [R,3]."""

    @staticmethod
    def forward(ctx, albedo, normals, weights, dirs, cam_colours, cam_of_ray, vis, bg):
        R = albedo.shape[0]
        a, n, w = albedo.contiguous(), normals.contiguous(), weights.contiguous()
        dirs, cam_colours, bg = dirs.contiguous(), cam_colours.contiguous(), bg.contiguous()
        vis = vis.contiguous() if vis is not None else None
        rgb = torch.empty(R, 3, device=a.device)
        lin = torch.empty(R, 3, device=a.device)
        hip.hemi_composite_fwd(a, n, w, dirs, cam_colours, cam_of_ray, vis, bg, rgb, lin)
        ctx.save_for_backward(a, n, w, dirs, cam_colours, cam_of_ray, bg, lin, *([vis] if vis is not None else []))
        ctx.has_vis = vis is not None
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        sv = ctx.saved_tensors
        a, n, w, dirs, cam_colours, cam_of_ray, bg, lin = sv[:8]
        vis = sv[8] if ctx.has_vis else None
        R, S, _ = a.shape
        dev = a.device
        da, dn = torch.empty_like(a), torch.empty_like(n)
        dw = torch.empty_like(w)
        dcol = zeros_like(cam_colours)
        dvis = torch.empty_like(vis) if vis is not None else None
        dbg = torch.empty_like(bg)
        hip.hemi_composite_bwd(a, n, w, dirs, cam_colours, cam_of_ray, vis, bg, lin, d_rgb.contiguous(), da, dn, dw, dcol, dvis, dbg)
        return da, dn, dw, None, dcol, None, dvis, dbg


class SplitRowsFn(torch.autograd.Function):
    """t [N, ...] -> (t[:n], t[n:]) as views; the backward is ONE concatenation.  (Two slices of a tensor that requires grad cost
    autograd a zero fill and a copy each plus the add that joins them: 5 launches per tensor.)"""

    @staticmethod
    def forward(ctx, t, n):
        ctx.n, ctx.rest = n, (t.shape[0] - n,) + tuple(t.shape[1:])
        ctx.set_materialize_grads(False)
        return t.narrow(0, 0, n), t.narrow(0, n, t.shape[0] - n)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        if gb is None:
            gb = zeros(*ctx.rest, device=ga.device)
        if ga is None:
            ga = zeros(ctx.n, *ctx.rest[1:], device=gb.device)
        return torch.cat((ga, gb), 0), None


class NeusWeightsFn(torch.autograd.Function):
    """NeuS alpha + transmittance + weights; returns weights [R,S], T_bg [R], accumulation [R], depth [R] (unclipped)."""

    @staticmethod
    def forward(ctx, sdf, grad, ray_dirs, starts, ends, variance, anneal):
        R, S = sdf.shape
        dev = sdf.device
        sdf, grad, ray_dirs = sdf.contiguous(), grad.contiguous(), ray_dirs.contiguous()
        starts, ends = starts.contiguous(), ends.contiguous()
        w = torch.empty(R, S, device=dev)
        tb, acc, dep = torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
        hip.neus_weights_fwd(sdf, grad, ray_dirs, starts, ends, variance, anneal, None, w, tb, acc, dep)
        ctx.save_for_backward(sdf, grad, ray_dirs, starts, ends, variance, w)
        ctx.anneal = anneal
        ctx.set_materialize_grads(False)  # absent output gradients arrive as None, not as zero-filled tensors
        ctx.mark_non_differentiable(acc, dep)
        return w, tb, acc, dep

    @staticmethod
    def backward(ctx, dw, dtb, _dacc, _ddep):
        sdf, grad, ray_dirs, starts, ends, variance, w = ctx.saved_tensors
        dsdf, dgrad = torch.empty_like(sdf), torch.empty_like(grad)
        dvar = zeros_like(variance)
        dw = zeros_like(w) if dw is None else dw.contiguous()
        hip.neus_weights_bwd(sdf, grad, ray_dirs, starts, ends, variance, ctx.anneal, dw,
                             None if dtb is None else dtb.contiguous(), dsdf, dgrad, dvar)
        return dsdf, dgrad, None, None, None, dvar, None


class RayReduceFn(torch.autograd.Function):
    """per-ray renderer reductions in one pass each way (hip.ray_reduce_*): weights [R,S], starts / ends [R,S], normals / albedo
    [R,S,3] or None -> p2p [R,1] (expected depth, clipped as nerfstudio's DepthRenderer does, then to max_clamp if > 0),
    accumulation [R,1], normal [R,3], albedo on white [R,3] (empty tensors for absent inputs)"""

    @staticmethod
    def forward(ctx, weights, starts, ends, normals, albedo, max_clamp):
        R, S = weights.shape[0], weights.shape[1]
        dev = weights.device
        w = weights.detach().reshape(R, S).contiguous()
        st, en = starts.detach().reshape(R, S).contiguous(), ends.detach().reshape(R, S).contiguous()
        nr = None if normals is None else normals.detach().reshape(R, S, 3).contiguous()
        al = None if albedo is None else albedo.detach().reshape(R, S, 3).contiguous()
        sums = torch.empty(R, 8, device=dev)
        bounds = _bounds_init(dev).clone()
        p2p, acc = torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev)
        normal = torch.empty(R, 3, device=dev) if nr is not None else None
        alb = torch.empty(R, 3, device=dev) if al is not None else None
        hip.ray_reduce_fwd(w, st, en, nr, al, max_clamp, sums, bounds, p2p, acc, normal, alb)
        ctx.save_for_backward(w, st, en, nr, al, sums, bounds)
        ctx.cfg = (R, S, max_clamp, tuple(weights.shape), None if normals is None else tuple(normals.shape),
                   None if albedo is None else tuple(albedo.shape))
        ctx.set_materialize_grads(False)
        empty = torch.empty(0, device=dev)
        return p2p, acc, (normal if normal is not None else empty), (alb if alb is not None else empty)

    @staticmethod
    def backward(ctx, d_p2p, d_acc, d_normal, d_alb):
        w, st, en, nr, al, sums, bounds = ctx.saved_tensors
        R, S, max_clamp, wshape, nshape, ashape = ctx.cfg
        dev = w.device
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_w = torch.empty(R, S, device=dev)
        want_n = nr is not None and ctx.needs_input_grad[3] and d_normal is not None
        want_a = al is not None and ctx.needs_input_grad[4] and d_alb is not None
        d_nr = torch.empty(R, S, 3, device=dev) if want_n else None
        d_al = torch.empty(R, S, 3, device=dev) if want_a else None
        hip.ray_reduce_bwd(w, st, en, nr, al, sums, bounds, max_clamp, c(d_p2p), c(d_acc), c(d_normal) if nr is not None else None,
                           c(d_alb) if al is not None else None, d_w, d_nr, d_al)
        return (d_w.view(wshape), None, None, None if d_nr is None else d_nr.view(nshape), None if d_al is None else d_al.view(ashape), None)


_BOUNDS_INIT: dict = {}


def _bounds_init(device) -> torch.Tensor:
    t = _BOUNDS_INIT.get(str(device))
    if t is None:
        t = _BOUNDS_INIT[str(device)] = torch.tensor([float("inf"), float("-inf")], device=device)
    return t


class NormalizeFn(torch.autograd.Function):
    """F.normalize(g, p=2, dim=-1) for [..., 3] (sdf_albedo_field.py:256), one kernel each way"""

    @staticmethod
    def forward(ctx, g):
        gc = g.detach().contiguous()
        n = torch.empty_like(gc)
        hip.normalize3_fwd(gc, n)
        ctx.save_for_backward(gc)
        return n

    @staticmethod
    def backward(ctx, d_n):
        (gc,) = ctx.saved_tensors
        d_g = torch.empty_like(gc)
        hip.normalize3_bwd(gc, d_n.contiguous(), d_g)
        return d_g


class RENIGridInputsFn(torch.autograd.Function):
    """latents [U,L,3] x directions [D,3] (+ R ray rows: ray_dirs [R,3] with the latent set ray_latent [R] of each) ->
    (cond [U D + R, pad4(3 L)], xrow [U D + R, 12]): the RENI++ decoder's rotation-invariant inputs of every pair and of the batch's
    own rays (hip.reni_grid_inputs_*); differentiable w.r.t. the latent codes"""

    @staticmethod
    def forward(ctx, latents, directions, ray_dirs=None, ray_latent=None):
        U, L, _ = latents.shape
        D = directions.shape[0]
        R = 0 if ray_dirs is None else ray_dirs.shape[0]
        Z, d = latents.detach().contiguous(), directions.detach().contiguous()
        rd = None if ray_dirs is None else ray_dirs.detach().contiguous()
        rl = None if ray_latent is None else ray_latent.to(torch.int64).contiguous()
        cond = torch.empty(U * D + R, pad4(3 * L), device=Z.device)
        xrow = torch.empty(U * D + R, 12, device=Z.device)
        hip.reni_grid_inputs_fwd(Z, d, rd, rl, cond, xrow)
        ctx.save_for_backward(Z, d, rd, rl)
        ctx.mark_non_differentiable(xrow)
        ctx.set_materialize_grads(False)
        return cond, xrow

    @staticmethod
    def backward(ctx, d_cond, _dx):
        Z, d, rd, rl = ctx.saved_tensors
        if d_cond is None:
            return None, None, None, None
        d_Z = torch.empty_like(Z)
        hip.reni_grid_inputs_bwd(Z, d, rd, rl, d_cond if d_cond.stride(1) == 1 else d_cond.contiguous(), d_Z)
        return d_Z, None, None, None


class RENIOutputFn(torch.autograd.Function):
    """raw head output [U D + R, 4] of the decoder -> (grid [U, D, 3], rays [R, 3]) = exp(raw) x the per-image scale
    (hip.reni_output_*; neusky_model.py:488-549); differentiable w.r.t. raw and scale"""

    @staticmethod
    def forward(ctx, raw, scale, ray_latent, U, D):
        raw, sc = raw.contiguous(), scale.detach().contiguous()
        R = raw.shape[0] - U * D
        rl = ray_latent.to(torch.int64).contiguous() if R else None
        grid = torch.empty(U, D, 3, device=raw.device)
        rays = torch.empty(R, 3, device=raw.device) if R else None
        hip.reni_output_fwd(raw, sc, rl, U, D, grid, rays)
        ctx.save_for_backward(raw, sc, rl)
        ctx.cfg = (U, D, R)
        ctx.set_materialize_grads(False)
        return grid, rays

    @staticmethod
    def backward(ctx, d_grid, d_rays):
        raw, sc, rl = ctx.saved_tensors
        U, D, R = ctx.cfg
        d_raw = torch.empty_like(raw)
        d_scale = zeros_like(sc) if ctx.needs_input_grad[1] else None
        hip.reni_output_bwd(raw, sc, rl, U, D, R, None if d_grid is None else d_grid.contiguous(), None if d_rays is None else d_rays.contiguous(),
                            d_raw, d_scale)
        return d_raw, d_scale, None, None, None


class PointAlphasFn(torch.autograd.Function):
    """NeuS alphas [P,3] of P isolated samples for three interval lengths (hip.point_alphas_*; the hash-grid density probe,
    neusky_model.py:715-732); differentiable w.r.t. sdf, gradients and the variance parameter"""

    @staticmethod
    def forward(ctx, sdf, grad, dirs, gaps, variance, anneal):
        s, g, d = sdf.detach().reshape(-1).contiguous(), grad.detach().reshape(-1, 3).contiguous(), dirs.detach().reshape(-1, 3).contiguous()
        alphas = torch.empty(s.numel(), 3, device=s.device)
        hip.point_alphas_fwd(s, g, d, gaps, variance, anneal, alphas)
        ctx.save_for_backward(s, g, d, variance)
        ctx.cfg = (tuple(float(v) for v in gaps), float(anneal), tuple(sdf.shape), tuple(grad.shape))
        return alphas

    @staticmethod
    def backward(ctx, d_alphas):
        s, g, d, variance = ctx.saved_tensors
        gaps, anneal, sdf_shape, grad_shape = ctx.cfg
        d_sdf, d_grad = torch.empty_like(s), torch.empty_like(g)
        d_var = zeros_like(variance)
        hip.point_alphas_bwd(s, g, d, gaps, variance, anneal, d_alphas.contiguous(), d_sdf, d_grad, d_var)
        return d_sdf.view(sdf_shape), d_grad.view(grad_shape), None, None, d_var, None


class SigmoidColumnFn(torch.autograd.Function):
    """t [M] = scale * sigmoid(raw[:, 0]) on a chain's padded [M, 4] head output (the DDF's termination distance,
    directional_distance_field.py:297-299), one kernel each way (hip.sigmoid_column_*)"""

    @staticmethod
    def forward(ctx, raw, scale):
        raw = raw.contiguous()
        t = torch.empty(raw.shape[0], device=raw.device)
        hip.sigmoid_column_fwd(raw, scale, t)
        ctx.save_for_backward(raw)
        ctx.scale = float(scale)
        return t

    @staticmethod
    def backward(ctx, d_t):
        raw, = ctx.saved_tensors
        d_raw = torch.empty_like(raw)
        hip.sigmoid_column_bwd(raw, ctx.scale, d_t.contiguous(), d_raw)
        return d_raw, None


class TermPointsFn(torch.autograd.Function):
    """The DDF's predicted termination points of the visibility rows and (optionally) of the fit rays, in ONE [M + N, 3] buffer:
    sphere_pts[m] - sel_dirs[m % Dv] t_hat[m]  |  fit_pos[n] + fit_dirs[n] t_main[n]   (neusky_model.py:1716-1724, ddf_model.py:243);
    differentiable w.r.t. the two distance vectors (hip.ray_points_*)"""

    @staticmethod
    def forward(ctx, sphere_pts, sel_dirs, t_hat, fit_pos, fit_dirs, t_main):
        M = sphere_pts.shape[0]
        N = 0 if fit_pos is None else fit_pos.shape[0]
        sp, sd = sphere_pts.detach().contiguous(), sel_dirs.detach().contiguous()
        out = torch.empty(M + N, 3, device=sp.device)
        hip.ray_points_fwd(sp, sd, -1.0, t_hat.detach().contiguous(), out[:M])
        fd = None
        if N:
            fd = fit_dirs.detach().contiguous()
            hip.ray_points_fwd(fit_pos.detach().contiguous(), fd, 1.0, t_main.detach().contiguous(), out[M:])
        ctx.save_for_backward(sd, fd)
        ctx.cfg = (M, N)
        return out

    @staticmethod
    def backward(ctx, d_out):
        sd, fd = ctx.saved_tensors
        M, N = ctx.cfg
        d_out = d_out.contiguous()
        d_hat = torch.empty(M, device=d_out.device)
        hip.ray_points_bwd(sd, -1.0, d_out[:M], d_hat)
        d_main = None
        if N:
            d_main = torch.empty(N, device=d_out.device)
            hip.ray_points_bwd(fd, 1.0, d_out[M:], d_main)
        return None, None, d_hat, None, None, d_main


class DDFQueryRowsFn(torch.autograd.Function):
    """Every row the DDF network is evaluated on in a train step, in ONE pair of buffers: the R x Dv visibility rows
    (hip.visibility_rays; not differentiable) followed by the DDF-fit rows (fit rays | multi-view | sky; hip.ddf_fit_rows_fwd,
    ddf_model.py:193-360).  Differentiable input: the fit rays' ground-truth termination distance, through the multi-view
    rows' directions.  fit = None: visibility rows only.
    -> pts_all [M+E,3], xrow_all [M+E,16], surf_dist [M], term_dist [M], mv_points [N,3], sky_gt [Ns], distance_weight [N]"""

    @staticmethod
    def forward(ctx, term_dist_fit, origins, ray_dirs, depth, sel_dirs, radius, fit):
        dev = origins.device
        R, Dv = origins.shape[0], sel_dirs.shape[0]
        M = R * Dv
        N = fit["positions"].shape[0] if fit is not None else 0
        n_mv = N if (fit is not None and fit["want_mv"]) else 0
        sky_o = fit["sky_o"] if fit is not None else None
        Ns = sky_o.shape[0] if sky_o is not None else 0
        E = N + n_mv + Ns
        pts_all = torch.empty(M + E, 3, device=dev)
        xrow_all = torch.empty(M + E, 16, device=dev)
        surf_dist = torch.empty(M, device=dev)
        term_dist = torch.empty(M, device=dev)
        hip.visibility_rays(origins, ray_dirs, depth, sel_dirs, radius, pts_all[:M], xrow_all[:M], surf_dist, term_dist)
        mv_points = torch.empty(n_mv, 3, device=dev)
        sky_gt = torch.empty(Ns, device=dev)
        dist_w = torch.empty(N, device=dev) if (fit is not None and fit["want_weight"]) else None
        if E > 0:
            t = term_dist_fit.detach().reshape(-1).contiguous()
            hip.ddf_fit_rows_fwd(fit["positions"], fit["directions"], t, fit["mv_points_in"], fit["seed"], fit["counter"], sky_o, fit["sky_d"],
                                 radius, bool(n_mv), fit["weight_exp"], fit["weight_include_z"], pts_all[M:], xrow_all[M:],
                                 mv_points if n_mv else None, sky_gt if Ns else None, dist_w)
            ctx.save_for_backward(fit["positions"], fit["directions"], t, mv_points)
        ctx.cfg = (M, N, n_mv, tuple(term_dist_fit.shape) if term_dist_fit is not None else None)
        ctx.set_materialize_grads(False)
        outs = (pts_all, xrow_all, surf_dist, term_dist, mv_points, sky_gt, dist_w if dist_w is not None else torch.empty(0, device=dev))
        ctx.mark_non_differentiable(outs[0], *outs[2:])
        return outs

    @staticmethod
    def backward(ctx, _dp, d_xrow, *_):
        M, N, n_mv, tshape = ctx.cfg
        if n_mv == 0 or d_xrow is None or not ctx.needs_input_grad[0]:
            return (None,) * 7
        positions, directions, t, mv_points = ctx.saved_tensors
        d_t = torch.empty(N, device=positions.device)
        hip.ddf_fit_rows_bwd(positions, directions, t, mv_points, d_xrow[M + N:M + 2 * N], d_t)
        return (d_t.view(tshape),) + (None,) * 6


class DDFFitRowsFn(torch.autograd.Function):
    """The DDF-fit rows ALONE (fit rays | multi-view | sky; hip.ddf_fit_rows_fwd, ddf_model.py:193-360) in buffers of their own, for
    NeuSkyFactoModel.start_ddf_fit: 263 456 rows are 8 full rounds of the four-wave chain kernels plus 11 workgroups, i.e. a ninth
    round on 11 of 256 CUs (8 % of the DDF forward and of its FiLM backward), so the 1 312 fit rows run as a small launch of their
    own, early, on a third stream, beside kernels that leave most of the chip idle, and the visibility rows are exactly 8 rounds.
    Differentiable input: the fit rays' ground-truth termination distance, through the multi-view rows' directions.
    -> pts [E,3], xrow [E,16], mv_points [N,3], sky_gt [Ns], distance_weight [N]"""

    @staticmethod
    def forward(ctx, term_dist_fit, radius, fit):
        dev = fit["positions"].device
        N = fit["positions"].shape[0]
        n_mv = N if fit["want_mv"] else 0
        sky_o = fit["sky_o"]
        Ns = sky_o.shape[0] if sky_o is not None else 0
        E = N + n_mv + Ns
        pts = torch.empty(E, 3, device=dev)
        xrow = torch.empty(E, 16, device=dev)
        mv_points = torch.empty(n_mv, 3, device=dev)
        sky_gt = torch.empty(Ns, device=dev)
        dist_w = torch.empty(N, device=dev) if fit["want_weight"] else None
        t = term_dist_fit.detach().reshape(-1).contiguous()
        hip.ddf_fit_rows_fwd(fit["positions"], fit["directions"], t, fit["mv_points_in"], fit["seed"], fit["counter"], sky_o, fit["sky_d"],
                             radius, bool(n_mv), fit["weight_exp"], fit["weight_include_z"], pts, xrow,
                             mv_points if n_mv else None, sky_gt if Ns else None, dist_w)
        ctx.save_for_backward(fit["positions"], fit["directions"], t, mv_points)
        ctx.cfg = (N, n_mv, tuple(term_dist_fit.shape))
        ctx.set_materialize_grads(False)
        outs = (pts, xrow, mv_points, sky_gt, dist_w if dist_w is not None else torch.empty(0, device=dev))
        ctx.mark_non_differentiable(outs[0], *outs[2:])
        return outs

    @staticmethod
    def backward(ctx, _dp, d_xrow, *_):
        N, n_mv, tshape = ctx.cfg
        if n_mv == 0 or d_xrow is None or not ctx.needs_input_grad[0]:
            return None, None, None
        positions, directions, t, mv_points = ctx.saved_tensors
        d_t = torch.empty(N, device=positions.device)
        hip.ddf_fit_rows_bwd(positions, directions, t, mv_points, d_xrow[N:2 * N], d_t)
        return d_t.view(tshape), None, None


class VisibilityFinishFn(torch.autograd.Function):
    """vis [R,D] from DDF distances (neusky_model.py:1724-1753)."""

    @staticmethod
    def forward(ctx, t_hat, surf_dist, threshold, scale, sel_index, R, Dv, D, lower_value):
        vis = torch.full((R, D), float(lower_value), device=t_hat.device)
        t_hat = t_hat.contiguous()
        hip.visibility_finish_fwd(t_hat, surf_dist, threshold, scale, sel_index, R, Dv, D, vis)
        ctx.save_for_backward(t_hat, surf_dist, threshold, sel_index)
        ctx.cfg = (scale, R, Dv, D)
        return vis

    @staticmethod
    def backward(ctx, d_vis):
        t_hat, surf_dist, threshold, sel_index = ctx.saved_tensors
        scale, R, Dv, D = ctx.cfg
        d_t = torch.empty_like(t_hat)
        d_thr = zeros_like(threshold)
        hip.visibility_finish_bwd(t_hat, surf_dist, threshold, scale, sel_index, R, Dv, D, d_vis.contiguous(), d_t, d_thr)
        return d_t, None, d_thr, None, None, None, None, None, None


class InterlevelFn(torch.autograd.Function):
    """per-ray interlevel loss sums of one proposal level (see include/neusky_hip.h); gradient w.r.t. the proposal weights"""

    @staticmethod
    def forward(ctx, c, w, sb, wp):
        c, w, sb, wp = c.contiguous(), w.contiguous(), sb.contiguous(), wp.contiguous()
        per_ray = torch.empty(w.shape[0], device=w.device)
        hip.interlevel_fwd(c, w, sb, wp, per_ray)
        ctx.save_for_backward(c, w, sb, wp)
        return per_ray

    @staticmethod
    def backward(ctx, g):
        c, w, sb, wp = ctx.saved_tensors
        d_wp = torch.empty_like(wp)
        hip.interlevel_bwd(c, w, sb, wp, g.contiguous(), d_wp)
        return None, None, None, d_wp


class ProposalMLPFn(torch.autograd.Function):
    """the proposal networks' density MLP (Linear + ReLU -> Linear(., 1)) on hash-encoded rows, both layers in registers
    (hip.proposal_mlp_*): feat [P, ld], lin0 weight / bias, lin1 weight / bias (unpadded torch parameters) -> raw [P, 1]"""

    @staticmethod
    def forward(ctx, feat, w0, b0, w1, b1):
        feat = feat.contiguous()
        w0c, w1c = w0.detach().contiguous(), w1.detach().contiguous()
        raw = torch.empty(feat.shape[0], 1, device=feat.device)
        hip.proposal_mlp_fwd(feat, w0c, b0.detach(), w1c, b1.detach(), raw)
        ctx.save_for_backward(feat, w0c, b0.detach(), w1c, b1.detach())
        ctx.sinks = [t if getattr(t, "_nsky_grad_sink", False) else None for t in (w0, b0, w1, b1)]
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        feat, w0, b0, w1, b1 = ctx.saved_tensors
        sinks = ctx.sinks
        if all(sk is not None and sk.grad is not None and sk.grad.is_contiguous() for sk in sinks):
            # optimizer-slab parameters: the kernel accumulates straight into their .grad views of the (zero-filled) gradient slab --
            # no AccumulateGrad copy per parameter, nothing for collect_grads to gather
            for sk in sinks:
                sk._nsky_sunk = True
            d_feat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
            hip.proposal_mlp_bwd(feat, w0, b0, w1, b1, d_raw.contiguous(), d_feat, sinks[0].grad, sinks[1].grad, sinks[2].grad, sinks[3].grad)
            return d_feat, None, None, None, None
        for sk in sinks:
            if sk is not None and sk.grad is None:
                sk._nsky_sunk = True  # (sinks from the next zero_grad_all on)
        flat = zeros(w0.numel() + b0.numel() + w1.numel() + 4, device=feat.device)
        n0, nb = w0.numel(), b0.numel()
        dw0, db0 = flat[:n0].view_as(w0), flat[n0:n0 + nb]
        dw1, db1 = flat[n0 + nb:n0 + nb + w1.numel()].view_as(w1), flat[n0 + nb + w1.numel():n0 + nb + w1.numel() + 1]
        d_feat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        hip.proposal_mlp_bwd(feat, w0, b0, w1, b1, d_raw.contiguous(), d_feat, dw0, db0, dw1, db1)
        return d_feat, dw0, db0, dw1, db1


class DensityWeightsFn(torch.autograd.Function):
    """proposal-network weights from the raw density head: raw [R*n, ld] (column 0), ebins [R,n+1] -> weights [R,n]
    (trunc_exp density + RaySamples.get_weights in one launch each way; see include/neusky_hip.h)"""

    @staticmethod
    def forward(ctx, raw, ebins):
        R, n = ebins.shape[0], ebins.shape[1] - 1
        ebins = ebins.contiguous()
        w = torch.empty(R, n, device=raw.device)
        hip.density_weights_fwd(raw, ebins, w)
        ctx.save_for_backward(raw, ebins)
        return w

    @staticmethod
    def backward(ctx, dw):
        raw, ebins = ctx.saved_tensors
        d_raw = torch.empty_like(raw)
        hip.density_weights_bwd(raw, ebins, dw.contiguous(), d_raw)
        return d_raw, None


class TruncExpFn(torch.autograd.Function):
    """nerfstudio trunc_exp (density activation of the proposal networks)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


# =============================================================================================
# fused closed-form loss terms (one launch each way per model)
# =============================================================================================
def _c(t):
    return None if t is None else t.detach().contiguous().float()


class TotalLossFn(torch.autograd.Function):
    """total = sum_s scale_s sum_i coef_s[i] x_s[i] over up to 8 small tensors, one launch each way (hip.weighted_total_*): the step's
    objective straight from the UNSCALED fused loss terms, the per-ray interlevel sums, ... -- what nerfstudio's scale_dict +
    functools.reduce(torch.add, loss_dict.values()) + interlevel_loss' mean form with a dozen scalar launches each way.
    args: metas [(has_coef, scale)], then the tensors x_0, (coef_0), x_1, ..."""

    @staticmethod
    def forward(ctx, metas, *tensors):
        parts, k = [], 0
        for has_coef, scale in metas:
            x = tensors[k].contiguous()
            c = tensors[k + 1].contiguous() if has_coef else None
            k += 2 if has_coef else 1
            parts.append((x, c, scale))
        total = torch.empty(1, device=parts[0][0].device)
        hip.weighted_total_fwd(parts, total)
        ctx.parts, ctx.metas = parts, metas
        return total[0]

    @staticmethod
    def backward(ctx, g):
        g = g.reshape(1).contiguous()
        grads, out, k = [], [None], 0
        for (x, c, sc), (has_coef, _) in zip(ctx.parts, ctx.metas):
            d = torch.empty_like(x) if ctx.needs_input_grad[1 + k] else None
            grads.append(d)
            out.append(d)
            if has_coef:
                out.append(None)
            k += 2 if has_coef else 1
        hip.weighted_total_bwd(ctx.parts, g, grads)
        return tuple(out)


class MainLossesFn(torch.autograd.Function):
    """terms[8] = unscaled {rgb_l1, eikonal, fg_mask, hashgrid_density, ground_plane, sky_pixel, visibility_sigmoid,
    sdf_level_set} of NeuSkyFactoModel.get_loss_dict's train branch (neusky_model.py:933-1035); an absent input (None) leaves
    its term at 0.  See include/neusky_hip.h."""

    @staticmethod
    def forward(ctx, rgb, image, mask, eik, weights, normal, hdr_bg, grid, sdf_term, vis_thr, sky_alpha, vis_target):
        R = image.shape[0]
        ins = [_c(rgb), _c(image), _c(mask), _c(eik), _c(weights), _c(normal), _c(hdr_bg), _c(grid), _c(sdf_term), _c(vis_thr)]
        S = weights.shape[1] if weights is not None else (eik.shape[1] if eik is not None else 0)
        d = hip.MainLossesDesc(R=R, S=S, P=(grid.numel() // 3 if grid is not None else 0),
                               M=(sdf_term.numel() if sdf_term is not None else 0), sky_alpha=float(sky_alpha), vis_target=float(vis_target))
        for name, t in zip(("rgb", "image", "mask", "eik", "weights", "normal", "hdr_bg", "grid", "sdf_term", "vis_thr"), ins):
            setattr(d, name, hip.ptr(t))
        terms = torch.empty(hip.N_MAIN_TERMS, device=image.device)
        wsum = torch.empty(R, device=image.device) if weights is not None else None
        hip.main_losses_fwd(d, terms, wsum)
        ctx.desc, ctx.keep, ctx.wsum = d, ins, wsum
        ctx.shapes = [None if t is None else t.shape for t in (rgb, eik, weights, normal, hdr_bg, grid, sdf_term, vis_thr)]
        return terms

    @staticmethod
    def backward(ctx, g):
        need = ctx.needs_input_grad
        idx = (0, 3, 4, 5, 6, 7, 8, 9)  # rgb, eik, weights, normal, hdr_bg, grid, sdf_term, vis_thr among the forward arguments
        outs = []
        for k, i in enumerate(idx):
            shp = ctx.shapes[k]
            outs.append(torch.empty(shp, device=g.device) if (need[i] and shp is not None) else None)
        hip.main_losses_bwd(ctx.desc, ctx.wsum, g.contiguous(), *outs)
        res = [None] * 12
        for k, i in enumerate(idx):
            res[i] = outs[k]
        return tuple(res)


class DDFLossesFn(torch.autograd.Function):
    """terms[5] = unscaled {depth_l1, sdf_l2, sdf_l1, multi_view, sky_ray} of DDFModel.get_loss_dict (ddf_model.py:407-493)"""

    @staticmethod
    def forward(ctx, expected, term, mask, dist_weight, sdf, mv_expected, mv_term, sky_expected, sky_term, flags):
        ins = [_c(expected), _c(term), _c(mask), _c(dist_weight), _c(sdf), _c(mv_expected), _c(mv_term), _c(sky_expected), _c(sky_term)]
        d = hip.DDFLossesDesc(Mr=expected.numel(), Mm=(mv_expected.numel() if mv_expected is not None else 0),
                              Ms=(sky_expected.numel() if sky_expected is not None else 0), **flags)
        for name, t in zip(("expected", "term", "mask", "dist_weight", "sdf", "mv_expected", "mv_term", "sky_expected", "sky_term"), ins):
            setattr(d, name, hip.ptr(t))
        terms = torch.empty(hip.N_DDF_TERMS, device=expected.device)
        hip.ddf_losses_fwd(d, terms)
        ctx.desc, ctx.keep = d, ins
        ctx.shapes = [None if t is None else t.shape for t in (expected, sdf, mv_expected, sky_expected, term, mv_term)]
        return terms

    @staticmethod
    def backward(ctx, g):
        need = ctx.needs_input_grad
        idx = (0, 4, 5, 7, 1, 6)  # expected, sdf, mv_expected, sky_expected, term, mv_term among the forward arguments
        outs = [torch.empty(ctx.shapes[k], device=g.device) if (need[i] and ctx.shapes[k] is not None) else None for k, i in enumerate(idx)]
        hip.ddf_losses_bwd(ctx.desc, g.contiguous(), *outs)
        res = [None] * 10
        for k, i in enumerate(idx):
            res[i] = outs[k]
        return tuple(res)


# =============================================================================================
# attention core of the RENI++ transformer decoder (csrc/attention.hip)
# =============================================================================================
class FrozenFeedForwardFn(torch.autograd.Function):
    """y = W2 relu(W1 x + b1) + b2 with constant weights (the frozen RENI++ decoder's feed-forward block) as ONE autograd node: the ReLU
    mask is applied in the epilogue of the second layer's input-gradient product (EPI_BWD_RELU on the saved hidden rows), so no
    threshold pass over the [M, hidden] gradient exists.  x [M, K] contiguous, K and the widths multiples of 4."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        x = x.contiguous()
        M = x.shape[0]
        h = torch.empty(M, W1.shape[0], device=x.device)
        fgemm(x, W1, h, M, W1.shape[0], W1.shape[1], bias=b1, epi=hip.EPI_RELU)
        y = torch.empty(M, W2.shape[0], device=x.device)
        fgemm(h, W2, y, M, W2.shape[0], W2.shape[1], bias=b2)
        ctx.save_for_backward(h, W1, W2)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, W1, W2 = ctx.saved_tensors
        M = h.shape[0]
        dz = torch.empty_like(h)
        prec = FROZEN_DX_PRECISION if _POLICY != "f32" else None
        grad_input(dy.contiguous(), W2, M, W2.shape[1], W2.shape[0], dz, precision=prec, epi=hip.EPI_BWD_RELU, aux0=h)
        dx = torch.empty(M, W1.shape[1], device=h.device)
        grad_input(dz, W1, M, W1.shape[1], W1.shape[0], dx, precision=prec)
        return dx, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (x + r, LayerNorm(x + r)) of a [M, W] row stream in one pass (csrc/attention.hip: add_layer_norm_kernel); r None: s is x.
    gamma / beta are taken as constants (the frozen RENI++ decoder): callers with trainable norms use torch."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps):
        x = x.contiguous()
        M, W = x.shape
        y, stats = torch.empty_like(x), torch.empty(M, 2, device=x.device)
        s = x if r is None else torch.empty_like(x)
        hip.add_layer_norm_fwd(x, None if r is None else r.contiguous(), gamma, beta, eps, None if r is None else s, y, stats)
        ctx.save_for_backward(s, stats, gamma)
        ctx.has_r = r is not None
        ctx.set_materialize_grads(False)  # (an unused s -- the last norm of the stack -- hands None to backward, not a zero matrix)
        return s, y

    @staticmethod
    def backward(ctx, ds_in, dy):
        s, stats, gamma = ctx.saved_tensors
        if dy is None:  # only the sum was used
            return ds_in, (ds_in if ctx.has_r else None), None, None, None
        ds = torch.empty_like(s)
        hip.add_layer_norm_bwd(s, stats, gamma, dy.contiguous(), None if ds_in is None else ds_in.contiguous(), ds)
        return ds, (ds if ctx.has_r else None), None, None, None


def add_layer_norm(x, r, ln):
    """(x + r, ln(x + r)) for a torch.nn.LayerNorm `ln` over the last dimension; r may be None"""
    W = x.shape[-1]
    if W not in (64, 128, 256, 512) or ln.weight.requires_grad or ln.bias.requires_grad:
        raise NotImplementedError("add_layer_norm (csrc/attention.hip): frozen norms over 64 / 128 / 256 / 512 features (the RENI++ decoder's)")
    s, y = AddLayerNormFn.apply(x.reshape(-1, W), None if r is None else r.reshape(-1, W), ln.weight, ln.bias, ln.eps)
    return s.view(x.shape), y.view(x.shape)


class AttnCoreFn(torch.autograd.Function):
    """O[u, d] = per-head softmax_n(q~ . K~_n) V~ combined with (d_x, d_y, 1) -- model_components/illumination.py:AttentionDecoder.
    Q [U D (+ R), H] (H = 16 heads_n; rows u D + d, then the R ray rows), dirs [U, D, 3] (no gradient), Kt / Vt [U, heads, L, 48]
    -> O like Q.  ray_dirs [R, 3], ray_perm [R], ray_seg [U + 1] (int32: the rays sorted by camera, hip.attn_core_rays_fwd): a ray's
    row attends to the keys / values of its camera.  Saves Q, O and two row statistics per head (instead of the [U, heads, D, L] score /
    probability matrices of the batched-product form)."""

    @staticmethod
    def forward(ctx, Q, dirs, Kt, Vt, scale, ray_dirs=None, ray_perm=None, ray_seg=None):
        Q, dirs, Kt, Vt = Q.contiguous(), dirs.contiguous(), Kt.contiguous(), Vt.contiguous()
        U, D = dirs.shape[:2]
        H, nh, N = Q.shape[-1], Kt.shape[1], U * D
        R = 0 if ray_dirs is None else ray_dirs.shape[0]
        ctx.shape = Q.shape
        Q = Q.reshape(N + R, H)
        O = torch.empty_like(Q)
        rmax = torch.empty(nh * (N + R), device=Q.device)
        rsum = torch.empty(nh * (N + R), device=Q.device)
        hip.attn_core_fwd(Q[:N].view(U, D, H), dirs, Kt, Vt, scale, O[:N].view(U, D, H), rmax[:nh * N].view(U, nh, D), rsum[:nh * N].view(U, nh, D))
        if R:
            ray_dirs = ray_dirs.contiguous()
            hip.attn_core_rays_fwd(Q[N:], ray_dirs, ray_perm, ray_seg, Kt, Vt, scale, O[N:], rmax[nh * N:].view(R, nh), rsum[nh * N:].view(R, nh))
        ctx.save_for_backward(Q, dirs, Kt, Vt, O, rmax, rsum, ray_dirs, ray_perm, ray_seg)
        ctx.scale = scale
        return O.view(ctx.shape)

    @staticmethod
    def backward(ctx, dO):
        Q, dirs, Kt, Vt, O, rmax, rsum, ray_dirs, ray_perm, ray_seg = ctx.saved_tensors
        U, D = dirs.shape[:2]
        H, nh, N = Q.shape[-1], Kt.shape[1], U * D
        R = Q.shape[0] - N
        dO = dO.contiguous().reshape(N + R, H)
        dQ, dKt, dVt = torch.empty_like(Q), torch.empty_like(Kt), torch.empty_like(Vt)
        hip.attn_core_bwd(Q[:N].view(U, D, H), dirs, Kt, Vt, O[:N].view(U, D, H), rmax[:nh * N].view(U, nh, D), rsum[:nh * N].view(U, nh, D),
                          dO[:N].view(U, D, H), ctx.scale, dQ[:N].view(U, D, H), dKt, dVt)
        if R:  # adds the rays' part to dKt / dVt (same stream, behind the grid rows' kernels)
            hip.attn_core_rays_bwd(Q[N:], ray_dirs, ray_perm, ray_seg, Kt, Vt, O[N:], rmax[nh * N:].view(R, nh), rsum[nh * N:].view(R, nh), dO[N:],
                                   ctx.scale, dQ[N:], dKt, dVt)
        return dQ.view(ctx.shape), None, dKt, dVt, None, None, None, None
