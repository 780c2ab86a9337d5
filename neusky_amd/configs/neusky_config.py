"""The `neusky` method - mirrors the MethodSpecification at neusky/configs/neusky_config.py:33-242.

`NeuSky` is what the `nerfstudio.method_configs` entry point (pyproject.toml of this repo, same group and name as the
reference's pyproject.toml:19-22) resolves.  With nerfstudio installed it is nerfstudio's own
`MethodSpecification(TrainerConfig(...))` (neusky_amd/plugin.py); without it, attribute-compatible stand-ins.
"""
from __future__ import annotations

from ..data.dataparsers import NeRFOSRCityScapesDataParserConfig
from ..data.image_datamanager import NeuSkyDataManagerConfig
from ..pipelines.neusky_pipeline import NeuSkyPipelineConfig
from ..plugin import MethodSpecification, TrainerConfig, build_method_specification  # noqa: F401  (re-exported types)

# the datamanager of neusky_config.py:46-64: NeRF-OSR scene parsed from disk (`ns-train neusky --data <NeRF-OSR/Data>`), every image and
# mask resident on the device, 1024 train / eval rays per batch.  (bench.py and the tests build their pipelines on the synthetic
# datamanager instead: no dataset in the image.)
NeuSky = build_method_specification(NeuSkyPipelineConfig(datamanager=NeuSkyDataManagerConfig(
    dataparser=NeRFOSRCityScapesDataParserConfig(scene="site1", auto_scale_poses=True, crop_to_equal_size=True, pad_to_equal_size=False,
                                                 scene_scale=1.0, mask_vegetation=True, mask_out_of_view_frustum_objects=True,
                                                 session_holdout_indices=[0, 0, 0, 0, 0]),
    train_num_images_to_sample_from=-1, train_num_times_to_repeat_images=-1, images_on_gpu=True, masks_on_gpu=True,
    train_num_rays_per_batch=1024, eval_num_rays_per_batch=1024)))


def synthetic_pipeline_config():
    """the method's pipeline config on the synthetic NeRF-OSR-lk2-shaped datamanager (300 train cameras, 1280 x 823 frames, random pixels):
    what bench.py, the tools and the tests build their pipelines from -- there is no dataset in the image"""
    import copy
    from ..data.synthetic_datamanager import SyntheticDataManagerConfig
    cfg = copy.deepcopy(NeuSky.config.pipeline)
    cfg.datamanager = SyntheticDataManagerConfig()
    return cfg
