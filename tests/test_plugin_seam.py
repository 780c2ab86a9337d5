"""The nerfstudio plugin surface: every method the reference defines on its Pipeline / Model / DDF
model / Field classes exists here under the same name with the same leading parameter names (fixture recorded FROM the
reference by tests/golden/make_golden_plugin.py), the optimizer group keys match, and - with a `nerfstudio` package
importable - the method specification and the class hierarchy are built from nerfstudio's own types."""
import inspect
import json
import os
import subprocess
import sys
import textwrap

import torch

from util_step import small_pipeline_config

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _classes():
    from neusky_amd.fields.directional_distance_field import DirectionalDistanceField
    from neusky_amd.fields.sdf_albedo_field import SDFAlbedoField
    from neusky_amd.models.ddf_model import DDFModel
    from neusky_amd.models.neusky_model import NeuSkyFactoModel
    from neusky_amd.pipelines.neusky_pipeline import NeuSkyPipeline
    return dict(NeuSkyPipeline=NeuSkyPipeline, NeuSkyFactoModel=NeuSkyFactoModel, DDFModel=DDFModel, SDFAlbedoField=SDFAlbedoField,
                DirectionalDistanceField=DirectionalDistanceField)


# constructor kwargs the reference passes by keyword and this package swallows in **kwargs (neusky_pipeline.py:162-172)
KWARGS_OK = {("NeuSkyFactoModel", "__init__"): {"grad_scaler", "train_metadata", "eval_metadata"},
             ("DDFModel", "__init__"): {"scene_box", "num_train_data", "num_val_data", "num_test_data", "train_metadata", "eval_metadata",
                                        "grad_scaler", "test_mode", "ddf_radius"}}


def test_reference_method_surface():
    ref = json.load(open(os.path.join(HERE, "golden", "plugin_signatures.json")))
    ours = _classes()
    problems = []
    for cls_name, methods in ref.items():
        if cls_name == "param_group_keys":
            continue
        for m, r in methods.items():
            fn = getattr(ours[cls_name], m, None)
            if fn is None:
                problems.append(f"{cls_name}.{m} missing")
                continue
            sig = inspect.signature(fn)
            mine = [p.name for p in sig.parameters.values() if p.name != "self" and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
            var_kw = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
            allowed = KWARGS_OK.get((cls_name, m), set()) if var_kw else set()
            want = [p for p in r["params"] if p not in allowed]
            lead = [p for p in mine if p in want or p not in r["params"]][:len(want)]
            if [p for p in mine if p in r["params"]] != [p for p in r["params"] if p in mine] or any(p not in mine for p in want):
                problems.append(f"{cls_name}.{m}: reference parameters {r['params']} vs {mine}")
            # every parameter this package ADDS is optional, so reference-style calls keep working
            extra_required = [p.name for p in sig.parameters.values() if p.name not in r["params"] and p.name != "self"
                              and p.default is p.empty and p.kind == p.POSITIONAL_OR_KEYWORD]
            if extra_required and not (cls_name, m) in KWARGS_OK:
                problems.append(f"{cls_name}.{m}: extra required parameters {extra_required}")
            del lead
    assert not problems, problems


def test_param_group_keys_and_eval_methods_present():
    ref = json.load(open(os.path.join(HERE, "golden", "plugin_signatures.json")))
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=8, images=3).setup(device="cpu")
    assert sorted(pipe.get_param_groups().keys()) == sorted(ref["param_group_keys"])
    from neusky_amd.plugin import HAVE_NERFSTUDIO, build_method_specification
    spec = build_method_specification(pipe.config)
    assert sorted(spec.config.optimizers.keys()) == sorted(ref["param_group_keys"]) and spec.config.method_name == "neusky"
    assert not HAVE_NERFSTUDIO  # the build image has none: the stand-in branch is what ran
    for name in ("num_train_data", "num_val_data", "num_test_data"):
        assert name in dict(pipe.named_buffers())
    # least-squares global scale (neusky_pipeline.py:212-225)
    pred, gt = torch.tensor([[1.0, 2.0], [3.0, 4.0]]), torch.tensor([[2.0, 4.0], [6.0, 8.0]])
    assert torch.allclose(pipe.global_scale(pred, gt), gt)


def test_with_nerfstudio_importable_the_seam_uses_its_types(tmp_path):
    """a minimal stand-in `nerfstudio` package (only the names neusky_config.py:10-19 imports) on PYTHONPATH: the entry point
    object must then BE nerfstudio's MethodSpecification(TrainerConfig) with nerfstudio optimizer / scheduler configs, and the
    Pipeline / Model / Field classes must subclass nerfstudio's bases and still construct and expose the method surface"""
    pkg = tmp_path / "nerfstudio"
    files = {
        "__init__.py": "",
        "configs/__init__.py": "", "configs/base_config.py": ("from dataclasses import dataclass\nfrom typing import Any, Type\n@dataclass\nclass ViewerConfig:\n    num_rays_per_chunk: int = 32768\n"
                                   "@dataclass\nclass InstantiateConfig:\n    _target: Type\n    def setup(self, **kwargs) -> Any:\n        return self._target(self, **kwargs)\n"),
        "engine/__init__.py": "",
        "engine/optimizers.py": "from dataclasses import dataclass\n@dataclass\nclass AdamOptimizerConfig:\n    lr: float = 1e-3\n    eps: float = 1e-8\n",
        "engine/schedulers.py": ("from dataclasses import dataclass\nfrom typing import Optional\n@dataclass\nclass CosineDecaySchedulerConfig:\n    warm_up_end: int = 0\n"
                                 "    learning_rate_alpha: float = 0.0\n    max_steps: int = 1\n@dataclass\nclass ExponentialDecaySchedulerConfig:\n"
                                 "    lr_final: Optional[float] = None\n    max_steps: int = 1\n    warmup_steps: int = 0\n"),
        "engine/trainer.py": ("from dataclasses import dataclass, field\nfrom typing import Any, Dict\n@dataclass\nclass TrainerConfig:\n    method_name: str = ''\n"
                              "    experiment_name: str = ''\n    steps_per_eval_image: int = 0\n    steps_per_eval_batch: int = 0\n    steps_per_save: int = 0\n"
                              "    steps_per_eval_all_images: int = 0\n    max_num_iterations: int = 0\n    mixed_precision: bool = False\n    pipeline: Any = None\n"
                              "    optimizers: Dict[str, Any] = field(default_factory=dict)\n    viewer: Any = None\n    vis: str = ''\n"),
        "fields/__init__.py": "", "fields/base_field.py": "from torch import nn\nclass Field(nn.Module):\n    def __init__(self):\n        raise RuntimeError('base ctor must not run')\n",
        "models/__init__.py": "", "models/base_model.py": "from torch import nn\nclass Model(nn.Module):\n    def __init__(self, config, scene_box, num_train_data, **kw):\n        raise RuntimeError('base ctor must not run')\n",
        "pipelines/__init__.py": "", "pipelines/base_pipeline.py": "from torch import nn\nclass Pipeline(nn.Module):\n    pass\n",
        "plugins/__init__.py": "", "plugins/types.py": "from dataclasses import dataclass\nfrom typing import Any\n@dataclass\nclass MethodSpecification:\n    config: Any\n    description: str\n",
    }
    for rel, body in files.items():
        f = pkg / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(body)
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import nerfstudio.plugins.types as T, nerfstudio.engine.trainer as TR, nerfstudio.engine.optimizers as OP
        import nerfstudio.pipelines.base_pipeline as BP, nerfstudio.models.base_model as BM, nerfstudio.fields.base_field as BF
        import neusky_amd.plugin as plugin
        from neusky_amd.configs.neusky_config import NeuSky
        assert plugin.HAVE_NERFSTUDIO
        assert isinstance(NeuSky, T.MethodSpecification) and isinstance(NeuSky.config, TR.TrainerConfig)
        assert NeuSky.config.method_name == "neusky" and NeuSky.config.viewer.num_rays_per_chunk == 1 << 15
        assert all(isinstance(v["optimizer"], OP.AdamOptimizerConfig) and v["optimizer"].eps == 1e-15 for v in NeuSky.config.optimizers.values())
        # the Adam groups are nerfstudio AdamOptimizerConfigs whose target is the fused SlabAdam (torch.optim.Optimizer interface)
        import torch
        from neusky_amd.optimizers import SlabAdam
        oc = NeuSky.config.optimizers["fields"]["optimizer"]
        assert oc._target is SlabAdam and issubclass(SlabAdam, torch.optim.Optimizer)
        opt = oc._target([torch.nn.Parameter(torch.zeros(3))], lr=oc.lr, eps=oc.eps)   # what OptimizerConfig.setup(params) does
        assert opt.param_groups[0]["eps"] == 1e-15 and set(opt.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
        assert all(isinstance(v["optimizer"], OP.AdamOptimizerConfig) and not hasattr(v["optimizer"], "_target")
                   for v in plugin.neusky_trainer_optimizers(fused=False).values())
        from neusky_amd.pipelines.neusky_pipeline import NeuSkyPipeline
        from neusky_amd.models.neusky_model import NeuSkyFactoModel
        from neusky_amd.models.ddf_model import DDFModel
        from neusky_amd.fields.sdf_albedo_field import SDFAlbedoField
        assert issubclass(NeuSkyPipeline, BP.Pipeline) and issubclass(NeuSkyFactoModel, BM.Model) and issubclass(DDFModel, BM.Model)
        assert issubclass(SDFAlbedoField, BF.Field)
        import nerfstudio.configs.base_config as BC
        from neusky_amd.pipelines.neusky_pipeline import NeuSkyPipelineConfig
        from neusky_amd.models.neusky_model import NeuSkyFactoModelConfig
        from neusky_amd.fields.sdf_albedo_field import SDFAlbedoFieldConfig
        assert all(issubclass(c, BC.InstantiateConfig) for c in (NeuSkyPipelineConfig, NeuSkyFactoModelConfig, SDFAlbedoFieldConfig))
        assert isinstance(NeuSky.config.pipeline, BC.InstantiateConfig) and isinstance(NeuSky.config.pipeline.model, BC.InstantiateConfig)
        from util_step import small_pipeline_config
        pipe = small_pipeline_config(R=8, images=3).setup(device="cpu")   # what nerfstudio's trainer does with config.pipeline
        assert isinstance(pipe, BP.Pipeline) and isinstance(pipe.model, BM.Model) and isinstance(pipe.model.field, BF.Field)
        assert sorted(pipe.get_param_groups()) == sorted(NeuSky.config.optimizers)
        print("seam ok")
    """) % (str(tmp_path), HERE)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and "seam ok" in out.stdout, out.stderr[-2000:]
