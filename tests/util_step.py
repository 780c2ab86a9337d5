"""Helpers shared by the end-to-end parity tests and smoke(): build a small pipeline, mirror its
parameters into the oracle's dict, draw one explicit set of random inputs."""
from __future__ import annotations

import torch

from oracle import neusky_oracle as O


def small_pipeline_config(R=16, num_prop=(24, 12), S=8, D=24, latent_dim=8, grid_res=4, vmf=(2, 8), sky=8, images=5):
    from neusky_amd.data.synthetic_datamanager import SyntheticDataManagerConfig
    from neusky_amd.model_components.ddf_sampler import VMFDDFSamplerConfig
    from neusky_amd.model_components.illumination import IcosahedronSamplerConfig, RENIFieldConfig
    from neusky_amd.models.neusky_model import NeuSkyFactoModelConfig
    from neusky_amd.pipelines.neusky_pipeline import NeuSkyPipelineConfig
    model = NeuSkyFactoModelConfig(num_proposal_samples_per_ray=tuple(num_prop), num_neus_samples_per_ray=S,
                                   illumination_field=RENIFieldConfig(latent_dim=latent_dim),
                                   illumination_sampler=IcosahedronSamplerConfig(num_directions=D))
    model.loss_inclusions["hashgrid_density_loss"]["grid_resolution"] = grid_res
    return NeuSkyPipelineConfig(
        datamanager=SyntheticDataManagerConfig(num_train_images=images, num_eval_images=2, train_num_rays_per_batch=R),
        model=model, visibility_train_sampler=VMFDDFSamplerConfig(num_samples_on_sphere=vmf[0], num_rays_per_sample=vmf[1]),
        num_sky_rays=sky)


from neusky_amd.utils.randomise import randomise  # noqa: E402,F401  (lives in the package: bench.py uses it without the oracle)


def oracle_params(pipeline, dtype=torch.float64):
    m = pipeline.model
    c = lambda t: t.detach().cpu().to(dtype).clone().requires_grad_(True)
    p = {}
    f = m.field
    p["field.table"] = c(f.encoding.table)
    for l in range(3):
        lin = getattr(f, f"glin{l}")
        p[f"field.glin{l}.v"], p[f"field.glin{l}.g"], p[f"field.glin{l}.b"] = c(lin.weight_v), c(lin.weight_g), c(lin.bias)
        lin = getattr(f, f"clin{l}")
        p[f"field.clin{l}.v"], p[f"field.clin{l}.g"], p[f"field.clin{l}.b"] = c(lin.weight_v), c(lin.weight_g), c(lin.bias)
    p["field.variance"] = c(f.deviation_network.variance)

    def film(prefix, net):
        lins = net.mapping_network.linears()
        for i, lin in enumerate(lins[:-1]):
            p[f"{prefix}map_w{i}"], p[f"{prefix}map_b{i}"] = c(lin.weight), c(lin.bias)
        p[f"{prefix}map_wo"], p[f"{prefix}map_bo"] = c(lins[-1].weight), c(lins[-1].bias)
        for i, l in enumerate(net.net):
            p[f"{prefix}film_w{i}"], p[f"{prefix}film_b{i}"] = c(l.layer.weight), c(l.layer.bias)
        p[f"{prefix}out_w"], p[f"{prefix}out_b"] = c(net.final_layer.weight), c(net.final_layer.bias)

    d = m.visibility_field.field
    p["ddf.table"] = c(d.position_encoding.table)
    film("ddf.", d.ddf)
    film("reni.", m.illumination_field.network)
    for i, net in enumerate(m.proposal_networks):
        p[f"prop{i}.table"] = c(net.encoding.table)
        p[f"prop{i}.w0"], p[f"prop{i}.b0"] = c(net.lin0.weight), c(net.lin0.bias)
        p[f"prop{i}.w1"], p[f"prop{i}.b1"] = c(net.lin1.weight), c(net.lin1.bias)
    p["train_latents"], p["train_scale"] = c(m.train_illumination_latents), c(m.train_scale)
    p["visibility_threshold"] = c(m.visibility_threshold)
    return p


def oracle_step_cfg(pipeline):
    mc = pipeline.model.config
    return O.StepCfg(
        num_prop=tuple(mc.num_proposal_samples_per_ray), num_final=mc.num_neus_samples_per_ray,
        grid_res=mc.loss_inclusions["hashgrid_density_loss"]["grid_resolution"],
        prop_grids=tuple(O.HashGridCfg(n_levels=a["num_levels"], log2_hashmap_size=a["log2_hashmap_size"], max_res=a["max_res"])
                         for a in mc.proposal_net_args_list))


def make_randoms(pipeline, R, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    from neusky_amd.model_components.illumination import random_rotation
    mc = pipeline.model.config
    res = mc.loss_inclusions["hashgrid_density_loss"]["grid_resolution"]
    n_lvl = mc.num_proposal_iterations + 1
    sc = pipeline.config.visibility_train_sampler
    ddf_rb = pipeline.visibility_train_sampler.generate_ddf_samples(sc.num_samples_on_sphere, sc.num_rays_per_sample, generator=g)
    Mv = ddf_rb.origins.shape[0]
    mv = torch.randn(Mv, 3, generator=g)
    mv = mv / mv.norm(dim=-1, keepdim=True)
    sky = pipeline.datamanager.get_sky_ray_bundle(pipeline.config.num_sky_rays)
    r = {
        "jitters": [torch.rand(R, 1, generator=g) for _ in range(n_lvl)],
        "ddf_jitters": [torch.rand(Mv, 1, generator=g) for _ in range(n_lvl)],
        "light_rotation": random_rotation(g),
        "grid_perturb": torch.rand(res**3, 3, generator=g), "grid_dirs": torch.randn(res**3, 3, generator=g),
        "ddf_rays": (ddf_rb.origins.cpu(), ddf_rb.directions.cpu()), "mv_points": mv,
        "sky_ray_bundle": sky,
    }
    return r


def randoms_to(r, device):
    out = dict(r)
    out["jitters"] = [j.to(device) for j in r["jitters"]]
    out["ddf_jitters"] = [j.to(device) for j in r["ddf_jitters"]]
    out["ddf_rays"] = tuple(t.to(device) for t in r["ddf_rays"])
    out["mv_points"] = r["mv_points"].to(device)
    return out


def oracle_randoms(r, light_dirs, dtype=torch.float64):
    return {
        "jitters": [j.to(dtype) for j in r["jitters"]], "ddf_jitters": [j.to(dtype) for j in r["ddf_jitters"]],
        "grid_perturb": r["grid_perturb"], "grid_dirs": r["grid_dirs"],
        "ddf_rays": tuple(t.cpu().to(dtype) for t in r["ddf_rays"]), "mv_points": r["mv_points"],
        "sky_o": r["sky_ray_bundle"].origins.cpu().to(dtype), "sky_d": r["sky_ray_bundle"].directions.cpu().to(dtype),
    }
