"""Which python lines launch the small torch kernels of one eager train step (run on the GPU box).
Counts aten ops per (file:line in neusky_amd/) with a TorchDispatchMode; backward ops are attributed to 'backward:<op>'."""
import sys, os, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from neusky_amd.utils.randomise import randomise

pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(6)]
for i in range(2):
    train_iteration(pipe, opt, 1000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
torch.cuda.synchronize()
SKIP = {"aten.view.default", "aten._unsafe_view.default", "aten.reshape.default", "aten.expand.default", "aten.slice.Tensor",
        "aten.select.int", "aten.unsqueeze.default", "aten.squeeze.dim", "aten.t.default", "aten.transpose.int", "aten.detach.default",
        "aten.alias.default", "aten.as_strided.default", "aten.permute.default", "aten.empty.memory_format", "aten.empty_like.default",
        "aten.unbind.int", "aten.split.Tensor", "aten.empty_strided.default", "aten.squeeze.default", "aten.view_as.default",
        "aten.is_same_size.default", "aten.lift_fresh.default", "aten._local_scalar_dense.default", "aten.narrow.default",
        "aten.unfold.default", "aten.new_empty.default", "aten.split_with_sizes.default", "aten.chunk.default"}
sites = collections.Counter()
views = collections.Counter()
ops_at = collections.defaultdict(collections.Counter)
large = []


class Counter(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name in ("aten.slice.Tensor", "aten.select.int", "aten.index.Tensor") and isinstance(args[0], torch.Tensor) and args[0].requires_grad:
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/neusky_amd/" in fr.filename and "hip.py" not in fr.filename:
                    views[f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name} {name}"] += 1
                    break
        if name not in SKIP:
            site = None
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/neusky_amd/" in fr.filename and "hip.py" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            if site is None:
                node = torch._C._current_autograd_node()
                site = "backward:" + (node.name() if node is not None else "other")
            sites[site] += 1
            ops_at[site][name] += 1
            big = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor) and a.numel() >= (1 << 18)]
            if big and not site.startswith("ops.py"):  # plain torch ops on large tensors: candidates for a kernel of their own
                large.append((site, name, big))
        return func(*args, **(kwargs or {}))


with Counter():
    train_iteration(pipe, opt, 2000, ray_bundle=batches[4][0], batch=batches[4][1])
torch.cuda.synchronize()
print("total non-view aten ops:", sum(sites.values()))
byfile = collections.Counter()
for s, n in sites.items():
    byfile[s.split(":")[0]] += n
print("by file:", dict(byfile.most_common()))
for s, n in sites.most_common(400):
    top = ", ".join(f"{k.replace('aten.', '')}x{v}" for k, v in ops_at[s].most_common(8))
    print(f"{n:5d}  {s:60s} {top}")
print("---- slices / selects / index of tensors that require grad (each costs a zeros + copy pair in backward)")
for s_, n in views.most_common(40):
    print(f"{n:5d}  {s_}")
print("---- torch ops on tensors of >= 2^18 elements outside ops.py")
for site, name, shapes in large:
    print(f"  {site:58s} {name:34s} {shapes}")
