"""engine.Optimizers on the GPU: gradients autograd keeps in tensors of its own are moved into the slab by ONE launch
(hip.gather_segments / nsky_gather_segments) instead of one add kernel per parameter; parameters whose backward kernels write
into the slab themselves keep their slab view."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gather_segments_copies_every_run():
    from neusky_amd import hip
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 4, 5, 257, 65536, 100003, 7]
    src_slab = torch.randn(sum(sizes) + 16, generator=g).to(DEV)
    pairs, off = [], 0
    dst_slab = torch.full((sum(sizes) + 64,), -1.0, device=DEV)
    doff = 1  # destinations off the 16-byte grid too
    for n in sizes:
        pairs.append((src_slab[off:off + n], dst_slab[doff:doff + n]))
        off += n
        doff += n + 1
    pairs += [(torch.randn(70, 5, device=DEV), torch.empty(70, 5, device=DEV)) for _ in range(70)]  # more than one launch's worth
    hip.gather_segments(pairs)
    torch.cuda.synchronize()
    for s, d in pairs:
        assert torch.equal(s, d)
    doff = 1
    for n in sizes:  # the gaps between destinations are untouched
        assert float(dst_slab[doff - 1]) == -1.0 and float(dst_slab[doff + n]) == -1.0
        doff += n + 1


def test_optimizers_collect_grads_equals_accumulate_grad():
    from neusky_amd.engine import AdamOptimizerConfig, Optimizers
    torch.manual_seed(1)
    lin1, lin2 = torch.nn.Linear(37, 19).to(DEV), torch.nn.Linear(19, 3).to(DEV)
    unused = torch.nn.Parameter(torch.randn(5, device=DEV))
    cfg = {"a": {"optimizer": AdamOptimizerConfig(lr=1e-2), "scheduler": None}, "b": {"optimizer": AdamOptimizerConfig(lr=1e-3), "scheduler": None}}
    opt = Optimizers(cfg, {"a": list(lin1.parameters()) + [unused], "b": list(lin2.parameters())})
    params = list(lin1.parameters()) + list(lin2.parameters())
    x = torch.randn(64, 37, device=DEV)

    def loss_of(ps, xin):
        w1, b1, w2, b2 = ps
        return (torch.tanh(xin @ w1.T + b1) @ w2.T + b2).square().mean()

    for it in range(3):
        opt.zero_grad_all()
        if it > 0:  # from the second step on autograd keeps the gradients itself
            assert all(p.grad is None for p in params)
        loss_of(params, x).backward()
        opt.collect_grads()
        clones = [p.detach().clone().requires_grad_(True) for p in params]
        want = torch.autograd.grad(loss_of(clones, x), clones)
        for p, w in zip(params, want):
            assert p.grad.data_ptr() >= opt.flat_g.data_ptr() and p.grad.data_ptr() < opt.flat_g.data_ptr() + 4 * opt.flat_g.numel()
            assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7)
        assert bool((unused.grad == 0).all())
        before = [p.detach().clone() for p in params]
        opt.optimizer_scheduler_step_all(it)
        assert all(not torch.equal(b, p.detach()) for b, p in zip(before, params)), "Adam saw the collected gradients"
    # a second backward without zero_grad_all accumulates (torch semantics)
    opt.zero_grad_all()
    loss_of(params, x).backward()
    opt.collect_grads()
    first = [p.grad.clone() for p in params]
    loss_of(params, x).backward()
    opt.collect_grads()
    for p, f in zip(params, first):
        assert torch.allclose(p.grad, 2 * f, rtol=1e-5, atol=1e-7)


def test_copy_segments_mixed_element_types():
    """hip.copy_segments: byte runs of any dtype (the graphed step's input hand-over), aligned or not, one launch"""
    from neusky_amd import hip
    g = torch.Generator().manual_seed(3)
    base = torch.randn(5000, generator=g).to(DEV)
    srcs = [torch.randn(1024, 3, generator=g).to(DEV), torch.randint(0, 99, (1024, 1), generator=g).to(DEV),
            (torch.rand(1024, 4, generator=g) > 0.5).to(DEV), torch.randn(1024, 1, generator=g).to(DEV),
            base[1:4001], (torch.rand(333, generator=g) > 0.5).to(DEV), torch.randint(0, 255, (777,), generator=g, dtype=torch.uint8).to(DEV)]
    dsts = [torch.zeros_like(s) for s in srcs]
    dsts[4] = torch.zeros(4003, device=DEV)[3:]  # 4-byte aligned only
    hip.copy_segments(list(zip(srcs, dsts)))
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
