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
  _model.{train,eval}_illumination_latents, _model.{train,eval}_scale, _model.visibility_threshold

Not mapped (their tensor layouts live in packages whose source is absent from the reference tree - SURVEY F2): the
proposal networks (`tcnn.NetworkWithInputEncoding` packs MLP + grid into one fp16 blob), the DDF FiLM-SIREN
(`reni.field_components.film_siren.FiLMSiren` parameter names) and the RENI++ decoder.  They are returned in `unmapped`.
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
    for name in ("train_illumination_latents", "train_scale", "eval_illumination_latents", "eval_scale"):
        targets["_model." + name] = getattr(model, name)
    if model.visibility_field is not None:
        targets["_model.visibility_threshold"] = model.visibility_threshold
        targets["_model.visibility_field.field.position_encoding.params"] = model.visibility_field.field.position_encoding.params
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
    return loaded, unmapped


# ---------------------------------------------------------------------------------------------------------------------
# saving / resuming THIS package's runs, in the layout nerfstudio's trainer uses (what the reference's load code expects
# to find: `<dir>/nerfstudio_models/step-%09d.ckpt` with the pipeline state dict under "pipeline", neusky_pipeline.py:174-194)
def checkpoint_path(base_dir, step: int):
    import os
    return os.path.join(str(base_dir), "nerfstudio_models", f"step-{int(step):09d}.ckpt")


def save_checkpoint(base_dir, step: int, pipeline, optimizers=None) -> str:
    """{"step", "pipeline": state_dict, "optimizers": Adam moments + step counts per group, "schedulers": {}} (the
    schedulers are pure functions of the global step and carry no state).  Returns the file written."""
    import os
    path = checkpoint_path(base_dir, step)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    ckpt = {"step": int(step), "pipeline": {k: v.detach().cpu().clone() for k, v in pipeline.state_dict().items()},
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
    if hasattr(pipeline.model, "begin_step"):
        pipeline.model.begin_step()  # prepared (weight-normed / padded / split) copies of the old weights are stale
    return int(ckpt["step"])
