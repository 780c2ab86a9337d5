"""The nerfstudio side of the drop-in seam.

When `nerfstudio` is importable, the `neusky` method specification is built from nerfstudio's OWN types
(`MethodSpecification(TrainerConfig(...))` with nerfstudio optimizer / scheduler / viewer configs, exactly the call forms of
neusky/configs/neusky_config.py:33-242) and this package's Pipeline / Model / Field classes subclass nerfstudio's
`Pipeline` / `Model` / `Field` and every config dataclass derives from nerfstudio's `InstantiateConfig`, so `ns-train neusky` resolves the entry point, type-checks and drives the HIP pipeline with
nerfstudio's own Trainer.  When it is not (the build image has no nerfstudio), attribute-compatible stand-ins are used and
`neusky_amd.engine` plays the trainer.  The constructors of this package never call the nerfstudio bases' `__init__`
(`nn.Module.__init__` only): the bases contribute the type identity and their default helper methods, nothing else.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict

from torch import nn

try:  # pragma: no cover - exercised by tests/test_plugin_seam.py through a stand-in package
    from nerfstudio.configs.base_config import InstantiateConfig as ConfigBase
    from nerfstudio.configs.base_config import ViewerConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig as NSAdamOptimizerConfig
    from nerfstudio.engine.schedulers import CosineDecaySchedulerConfig as NSCosineDecaySchedulerConfig
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig as NSExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.fields.base_field import Field as FieldBase
    from nerfstudio.models.base_model import Model as ModelBase
    from nerfstudio.pipelines.base_pipeline import Pipeline as PipelineBase
    from nerfstudio.plugins.types import MethodSpecification
    HAVE_NERFSTUDIO = True
except ImportError:
    HAVE_NERFSTUDIO = False
    FieldBase = ModelBase = PipelineBase = nn.Module
    ViewerConfig = None

    @dataclass
    class ConfigBase:  # nerfstudio.configs.base_config.InstantiateConfig: `_target` + setup(**kwargs) -> _target(self, **kwargs)
        def setup(self, **kwargs) -> Any:
            return self._target(self, **kwargs)

    @dataclass
    class TrainerConfig:  # the members neusky_config.py:34-42,238-240 sets
        method_name: str = "neusky"
        experiment_name: str = "lk2"
        steps_per_eval_image: int = 5000
        steps_per_eval_batch: int = 100002
        steps_per_save: int = 5000
        steps_per_eval_all_images: int = 100000
        max_num_iterations: int = 100001
        mixed_precision: bool = False
        pipeline: Any = None
        optimizers: Dict[str, Any] = field(default_factory=dict)
        viewer: Any = None
        vis: str = "viewer"

    @dataclass
    class MethodSpecification:
        config: TrainerConfig
        description: str


if HAVE_NERFSTUDIO:
    @dataclass
    class SlabAdamOptimizerConfig(NSAdamOptimizerConfig):  # pragma: no cover - needs nerfstudio; tests/test_plugin_seam.py runs it on the stand-in
        """nerfstudio's AdamOptimizerConfig with `_target = neusky_amd.optimizers.SlabAdam`: the same update as ONE launch per group
        (OptimizerConfig.setup passes lr / eps / weight_decay to the target; max_norm is handled by the trainer)"""
        _target: Any = field(default_factory=lambda: __import__("neusky_amd.optimizers", fromlist=["SlabAdam"]).SlabAdam)


def neusky_trainer_optimizers(fused: bool = True) -> Dict[str, Dict[str, Any]]:
    """neusky/configs/neusky_config.py:216-237 in the types of the trainer that will consume them.  fused (default): the Adam groups are
    `SlabAdamOptimizerConfig` -- torch.optim.Adam's update on the fused kernel; False: nerfstudio's own AdamOptimizerConfig."""
    if not HAVE_NERFSTUDIO:
        from .engine import neusky_optimizers
        return neusky_optimizers()
    Adam = SlabAdamOptimizerConfig if fused else NSAdamOptimizerConfig
    cos = lambda: NSCosineDecaySchedulerConfig(warm_up_end=500, learning_rate_alpha=0.05, max_steps=100001)  # noqa: E731
    return {
        "proposal_networks": {"optimizer": Adam(lr=1e-2, eps=1e-15), "scheduler": cos()},
        "fields": {"optimizer": Adam(lr=1e-3, eps=1e-15), "scheduler": cos()},
        "illumination_field": {"optimizer": Adam(lr=1e-2, eps=1e-15),
                               "scheduler": NSExponentialDecaySchedulerConfig(lr_final=1e-5, max_steps=100001)},
        "visibility_sigmoid": {"optimizer": Adam(lr=1e-3, eps=1e-15),
                               "scheduler": NSExponentialDecaySchedulerConfig(warmup_steps=4000, lr_final=1e-4, max_steps=100001)},
        "ddf_field": {"optimizer": Adam(lr=1e-4, eps=1e-15), "scheduler": cos()},
    }


def build_method_specification(pipeline_config) -> "MethodSpecification":
    kw: Dict[str, Any] = dict(
        method_name="neusky", experiment_name="lk2", steps_per_eval_image=5000, steps_per_eval_batch=100002, steps_per_save=5000,
        steps_per_eval_all_images=100000, max_num_iterations=100001, mixed_precision=False, pipeline=pipeline_config,
        optimizers=neusky_trainer_optimizers(), vis="viewer")
    if HAVE_NERFSTUDIO:
        kw["viewer"] = ViewerConfig(num_rays_per_chunk=1 << 15)
    return MethodSpecification(config=TrainerConfig(**kw), description="Base config for NeuSky (MI355X HIP hot path).")
