"""Record the plugin surface of the reference: names, parameter names and defaults of the methods a
nerfstudio trainer / the reference's own code call on the Pipeline, Model, DDF model and Field classes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_plugin.py      # writes tests/golden/plugin_signatures.json

Only names travel (a JSON of strings); the reference is imported through the stub importer and never copied."""
from __future__ import annotations

import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stub_importer  # noqa: E402

_ref_stub_importer.install()

import neusky.fields.directional_distance_field as rddf  # noqa: E402
import neusky.fields.sdf_albedo_field as rsdf  # noqa: E402
import neusky.models.ddf_model as rdm  # noqa: E402
import neusky.models.neusky_model as rnm  # noqa: E402
import neusky.pipelines.neusky_pipeline as rpipe  # noqa: E402

SURFACE = {
    "NeuSkyPipeline": (rpipe.NeuSkyPipeline, ["__init__", "get_train_loss_dict", "get_eval_loss_dict", "get_eval_image_metrics_and_images",
                                              "get_average_eval_image_metrics", "get_param_groups", "generate_ddf_samples", "global_scale",
                                              "_optimise_evaluation_latents", "_setup_visibility_field"]),
    "NeuSkyFactoModel": (rnm.NeuSkyFactoModel, ["__init__", "forward", "get_outputs", "get_loss_dict", "get_metrics_dict",
                                                "get_outputs_for_camera_ray_bundle", "get_image_metrics_and_images", "fit_latent_codes_for_eval",
                                                "sample_illumination", "sample_and_forward_field", "compute_visibility", "generate_ddf_ground_truth",
                                                "get_param_groups", "get_illumination_field", "populate_modules"]),
    "DDFModel": (rdm.DDFModel, ["__init__", "forward", "get_outputs", "get_loss_dict", "get_metrics_dict", "get_param_groups",
                                "get_localised_transforms"]),
    "SDFAlbedoField": (rsdf.SDFAlbedoField, ["__init__", "forward", "get_outputs", "get_colors", "get_sdf_at_pos"]),
    "DirectionalDistanceField": (rddf.DirectionalDistanceField, ["__init__", "forward", "get_outputs"]),
}
out = {}
for cls_name, (cls, methods) in SURFACE.items():
    out[cls_name] = {}
    for m in methods:
        fn = inspect.unwrap(getattr(cls, m))
        sig = inspect.signature(fn)
        params = [p for p in sig.parameters.values() if p.name != "self"]
        out[cls_name][m] = {"params": [p.name for p in params if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)],
                            "required": [p.name for p in params if p.default is p.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)],
                            "var_keyword": any(p.kind == p.VAR_KEYWORD for p in params)}
out["param_group_keys"] = ["fields", "proposal_networks", "illumination_field", "visibility_sigmoid", "ddf_field"]  # neusky_model.py:379-398, ddf_model.py:151-156
path = os.path.join(HERE, "plugin_signatures.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
