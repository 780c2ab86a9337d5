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
