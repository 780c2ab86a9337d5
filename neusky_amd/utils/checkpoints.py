"""Loading reference (nerfstudio-trainer format) checkpoints into the HIP pipeline - SURVEY.md section 8(f) item 3.

Wire format as read by the reference itself: `torch.load(".../nerfstudio_models/step-%09d.ckpt")["pipeline"]` is a flat
state dict whose keys start with `_model.` (neusky/pipelines/neusky_pipeline.py:174-194), with the DDF under
`_model.visibility_field.` (:186) and the SDF field under `_model.field.` (neusky/models/neusky_model.py:286-297 does the
same for the RENI decoder).  What maps 1:1 onto this implementation, because the layouts are fixed by in-tree code or
by torch itself:

  _model.field.encoding.params                      tcnn hash table, level-major [entry][2] flat (fp16 or fp32)
  _model.field.{glin,clin}{l}.{weight_g,weight_v,bias}   torch weight_norm (old style) or
  _model.field.{glin,clin}{l}.parametrizations.weight.original{0,1}   (new style: g, v)
  _model.field.deviation_network.variance
  _model.visibility_field.field.position_encoding.params
  _model.visibility_field.field.ddf.{mapping_network.network.{2 l}, net.{i}.layer, final_layer}.{weight,bias}
                                                    the DDF FiLM-SIREN under the module names of neusky/utils/siren.py:108-208 (the only
                                                    in-tree statement of that network; this package's FiLMSiren keeps the same names)
  _model.proposal_networks.{i}.mlp_base.0.tcnn_encoding.params | .encoding.tcnn_encoding.params | .encoding.params
                                                    proposal hash tables in tcnn's level-major layout
  _model.proposal_networks.{i}.mlp_base.1.layers.{l}.{weight,bias} | .network.layers.{l}.{weight,bias}
                                                    proposal density MLPs in their fp32 torch form (nerfstudio MLP, implementation="torch")
                                                    [UNVERIFIED-UPSTREAM key names: nerfstudio's source is absent from the reference tree]
  _model.{train,eval}_illumination_latents, _model.{train,eval}_scale, _model.eval_rotation, _model.visibility_threshold
  any other key that is the name of a parameter / buffer of THIS package's pipeline (its own checkpoints; the illumination decoder)

Not mapped (their tensor layouts live in packages whose source is absent from the reference tree - SURVEY F2): proposal networks
saved as ONE packed fp16 blob (`tcnn.NetworkWithInputEncoding`), nerfstudio's torch hash table (fixed 2^k entries per level, no dense
levels: not tcnn's indexing) and the real RENI++ decoder.  They are returned in `unmapped`.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def _put(dst: torch.nn.Parameter, src: torch.Tensor, name: str) -> None:
    if dst.numel() != src.numel():
        raise ValueError(f"{name}: checkpoint has {tuple(src.shape)} ({src.numel()} values), model expects {tuple(dst.shape)}")
    with torch.no_grad():
        dst.copy_(src.to(dtype=dst.dtype, device=dst.device).reshape(dst.shape))


def load_reference_pipeline_state(pipeline, state: Dict[str, torch.Tensor], strict_shapes: bool = True) -> Tuple[List[str], List[str]]:
    """state = ckpt["pipeline"].  Returns (loaded keys, unmapped keys)."""
    model = pipeline.model
    targets: Dict[str, torch.nn.Parameter] = {}
    f = model.field
    targets["_model.field.encoding.params"] = f.encoding.params
    targets["_model.field.deviation_network.variance"] = f.deviation_network.variance
    for kind, n in (("glin", 3), ("clin", 3)):
        for l in range(n):
            lin = getattr(f, f"{kind}{l}")
            base = f"_model.field.{kind}{l}."
            targets[base + "weight_g"] = lin.weight_g
            targets[base + "weight_v"] = lin.weight_v
            targets[base + "bias"] = lin.bias
            targets[base + "parametrizations.weight.original0"] = lin.weight_g
            targets[base + "parametrizations.weight.original1"] = lin.weight_v
    for name in ("train_illumination_latents", "train_scale", "eval_illumination_latents", "eval_scale", "eval_rotation"):
        if hasattr(model, name):
            targets["_model." + name] = getattr(model, name)
    if model.visibility_field is not None:
        targets["_model.visibility_threshold"] = model.visibility_threshold
        vf = model.visibility_field.field
        targets["_model.visibility_field.field.position_encoding.params"] = vf.position_encoding.params
        for n, p_ in vf.ddf.named_parameters():  # mapping_network.network.{2l}.*, net.{i}.layer.*, final_layer.* (siren.py:108-208)
            targets["_model.visibility_field.field.ddf." + n] = p_
    for i, net in enumerate(model.proposal_networks):
        base = f"_model.proposal_networks.{i}."
        for k in ("mlp_base.0.tcnn_encoding.params", "encoding.tcnn_encoding.params", "encoding.params"):
            targets[base + k] = net.encoding.params
        for l, lin in enumerate((net.lin0, net.lin1)):
            for stem in (f"mlp_base.1.layers.{l}.", f"network.layers.{l}.", f"lin{l}."):
                targets[base + stem + "weight"] = lin.weight
                targets[base + stem + "bias"] = lin.bias
    own = dict(pipeline.named_parameters())
    own.update(dict(pipeline.named_buffers()))
    for k, v in own.items():  # this package's own names (its checkpoints; the illumination decoder)
        targets.setdefault(k, v)
    loaded, unmapped = [], []
    for k, v in state.items():
        if k in targets:
            try:
                _put(targets[k], v, k)
                loaded.append(k)
            except ValueError:
                if strict_shapes:
                    raise
                unmapped.append(k)
        elif k.startswith("_model."):
            unmapped.append(k)
    if loaded and hasattr(model, "begin_step") and next(model.parameters()).is_cuda:
        model.begin_step()  # prepared (weight-normed / padded / packed) copies of the old weights are stale
    return loaded, unmapped


# ---------------------------------------------------------------------------------------------------------------------
# saving / resuming THIS package's runs, in the layout nerfstudio's trainer uses (what the reference's load code expects
# to find: `<dir>/nerfstudio_models/step-%09d.ckpt` with the pipeline state dict under "pipeline", neusky_pipeline.py:174-194)
def checkpoint_path(base_dir, step: int):
    import os
    return os.path.join(str(base_dir), "nerfstudio_models", f"step-{int(step):09d}.ckpt")


def save_checkpoint(base_dir, step: int, pipeline, optimizers=None) -> str:
    """{"step", "pipeline": state_dict, "optimizers": Adam moments + step counts per group, "schedulers": {}, "rng": ...} (the
    schedulers are pure functions of the global step and carry no state).  Returns the file written."""
    import os
    path = checkpoint_path(base_dir, step)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    from .utils import device_rng_state
    dm = getattr(pipeline, "datamanager", None)
    ckpt = {"step": int(step), "pipeline": {k: v.detach().cpu().clone() for k, v in pipeline.state_dict().items()},
            # what an exact resume needs besides parameters and Adam state: the call counters of the in-kernel generators, the
            # datamanager's host generator and the step of the last eval-latent fit (neusky_pipeline.py:202-210)
            "rng": {"device": device_rng_state(), "datamanager": dm.state_dict() if hasattr(dm, "state_dict") else None,
                    "step_of_last_latent_optimisation": int(getattr(pipeline, "step_of_last_latent_optimisation", 0))},
            "optimizers": {} if optimizers is None else {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                                                        for k, st in optimizers.state_dict().items()},
            "schedulers": {}}
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, pipeline, optimizers=None) -> int:
    """resume: parameters are copied IN PLACE (they are views into the optimizer slabs and stay so), Adam moments and the
    per-group bias-correction counters are restored; returns the step to continue from."""
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    pipeline.load_state_dict(ckpt["pipeline"], strict=True)
    if optimizers is not None:
        if not ckpt.get("optimizers"):
            raise ValueError(f"{path}: no optimizer state to resume from")
        optimizers.load_state_dict(ckpt["optimizers"])
    rng = ckpt.get("rng")
    if rng:
        from .utils import load_device_rng_state
        load_device_rng_state(rng.get("device", {}))
        dm = getattr(pipeline, "datamanager", None)
        if hasattr(dm, "load_state_dict") and rng.get("datamanager") is not None:
            dm.load_state_dict(rng["datamanager"])  # every generator the datamanager draws rays with (train / eval / host)
        if hasattr(pipeline, "step_of_last_latent_optimisation"):
            pipeline.step_of_last_latent_optimisation = int(rng.get("step_of_last_latent_optimisation", 0))
    if hasattr(pipeline.model, "begin_step"):
        pipeline.model.begin_step()  # prepared (weight-normed / padded / split) copies of the old weights are stale
    return int(ckpt["step"])
