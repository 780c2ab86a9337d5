"""The `neusky` method - mirrors the MethodSpecification at neusky/configs/neusky_config.py:33-242.

`NeuSky` is what the `nerfstudio.method_configs` entry point (pyproject.toml of this repo, same group and name as the
reference's pyproject.toml:19-22) resolves.  With nerfstudio installed it is nerfstudio's own
`MethodSpecification(TrainerConfig(...))` (neusky_amd/plugin.py); without it, attribute-compatible stand-ins.
"""
from __future__ import annotations

from ..pipelines.neusky_pipeline import NeuSkyPipelineConfig
from ..plugin import MethodSpecification, TrainerConfig, build_method_specification  # noqa: F401  (re-exported types)

NeuSky = build_method_specification(NeuSkyPipelineConfig())
