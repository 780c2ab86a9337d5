"""The `neusky` method - mirrors the MethodSpecification at neusky/configs/neusky_config.py:33-242.

With nerfstudio installed, `NeuSky` below is what the `nerfstudio.method_configs` entry point
(pyproject.toml of this repo, same group and name as the reference's pyproject.toml:19-22) resolves, so
`ns-train neusky` lands on the HIP pipeline.  nerfstudio is not installed in the build image, so the
TrainerConfig wrapper is reproduced as a plain dataclass with the same member names.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict

from ..engine import neusky_optimizers
from ..pipelines.neusky_pipeline import NeuSkyPipelineConfig


@dataclass
class TrainerConfig:
    method_name: str = "neusky"
    experiment_name: str = "lk2"
    steps_per_eval_image: int = 5000
    steps_per_save: int = 5000
    max_num_iterations: int = 100001
    mixed_precision: bool = False
    pipeline: NeuSkyPipelineConfig = field(default_factory=NeuSkyPipelineConfig)
    optimizers: Dict[str, Any] = field(default_factory=neusky_optimizers)
    vis: str = "viewer"


@dataclass
class MethodSpecification:
    config: TrainerConfig
    description: str


NeuSky = MethodSpecification(config=TrainerConfig(), description="Base config for NeuSky (MI355X HIP hot path).")
