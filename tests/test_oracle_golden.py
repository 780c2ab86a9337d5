"""Pins the CPU oracle against the golden vectors captured from the reference's own in-tree
functions (tests/golden/make_golden.py).  CPU only; no GPU, no /root/reference access."""
import os

import numpy as np
import torch

import inputs as gi
from oracle import neusky_oracle as O

T = torch.from_numpy
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_g1_srgb():
    g = load("g1_srgb")
    close(O.linear_to_srgb(T(g["x"])), g["y"], rtol=0, atol=0)  # same torch ops -> bit-exact


def test_g2_ray_sphere():
    g = load("g2_sphere")
    close(O.ray_sphere_intersection_free(T(g["p"]), T(g["d_unit"]), 1.0), g["y_free"], rtol=0, atol=0)
    close(O.ray_sphere_intersection_clamped(T(g["p2"]), T(g["d_raw"]), 1.0), g["y_meth"], rtol=0, atol=0)
    close(O.ray_sphere_intersection_clamped(T(g["p2"]), T(g["d_raw"]), 2.0), g["y_meth_r2"], rtol=0, atol=0)


def _lamb(inp, training):
    return O.lambertian_render(T(inp["albedo"]), T(inp["normals"]), T(inp["dirs"]), T(inp["cam_colours"]),
                               T(inp["cam_of_ray"]), T(inp["vis"]), T(inp["bg"]), T(inp["weights"]), training)


def test_g3_lambertian_small():
    g = load("g3_lambertian_small")
    close(_lamb(g, True), g["rgb_train"], rtol=2e-6, atol=1e-6)
    close(_lamb(g, False), g["rgb_eval"], rtol=2e-6, atol=1e-6)
    # the stored inputs are exactly what inputs.py regenerates
    again = gi.lambertian_inputs(seed=3, R=8, S=4, D=16, U=3)
    for k, v in again.items():
        np.testing.assert_array_equal(v, g[k])


def test_g3_lambertian_big():
    g = load("g3_lambertian_big")
    R, S, D, U = [int(v) for v in g["shape"]]
    inp = gi.lambertian_inputs(seed=int(g["seed"]), R=R, S=S, D=D, U=U)
    chk = float(sum(np.asarray(v, np.float64).sum() for v in inp.values()))
    assert chk == float(g["input_checksum"])  # RNG stream reproduced bit-for-bit
    close(_lamb(inp, True), g["rgb_train"], rtol=1e-5, atol=2e-6)


def _standin_ddf(origins, directions, radius=1.0):
    oz, dz = origins[:, 2], directions[:, 2]
    t = torch.where(dz < -1e-6, (0.1 - oz) / dz.clamp(max=-1e-6), torch.full_like(oz, 2 * radius))
    t = t.clamp(0.0, 2 * radius)
    sdf = (origins + directions * t[:, None])[:, 2:3] - 0.1
    return {"expected_termination_dist": t, "sdf_at_termination": sdf}


def test_g4_visibility():
    for tag, (only_upper, lower_vis) in {"upper_lower1": (True, True), "upper_lower0": (True, False),
                                         "all_shadow": (False, True)}.items():
        g = load(f"g4_visibility_{tag}")
        inp = gi.visibility_inputs(seed=4, R=16, S=3, D=42, n_outside=2)
        out = O.compute_visibility(T(inp["origins"][:, 0]), T(inp["directions"][:, 0]), T(inp["depth"]),
                                   T(inp["dirs"]), torch.tensor(0.35), torch.tensor(25.0), 1.0, _standin_ddf,
                                   only_upper, lower_vis)
        R, S, D = 16, 3, 42
        # the reference repeats the [R,D] visibility over S (neusky_model.py:1755-1759).  In the configured
        # upper-hemisphere branch the result is ray-major [R,S,D]; in the (unconfigured) all-directions
        # branch `visibility` is still flat [R*D] at :1755, so unsqueeze/repeat yields SAMPLE-major [S,R,D] -
        # a reference quirk outside the `neusky` config, pinned here only to document it.
        ref_vis = g["visibility"].reshape(R, S, D) if only_upper else g["visibility"].reshape(S, R, D).transpose(1, 0, 2)
        for s in range(S):
            close(out["visibility"], ref_vis[:, s], rtol=1e-5, atol=1e-6)
        close(out["expected_termination_dist"], g["expected_termination_dist"], rtol=1e-5, atol=1e-6)
        close(out["termination_dist"], g["termination_dist"], rtol=1e-5, atol=1e-6)
        close(out["sdf_at_termination"], g["sdf_at_termination"], rtol=1e-5, atol=1e-6)
        if "difference" in g:
            close(out["difference"], g["difference"], rtol=1e-5, atol=1e-6)


def test_g5_local_frame():
    g = load("g5_local_frame")
    inp = gi.sphere_rays(seed=5, M=48, radius=1.0)
    close(O.local_frame(T(inp["positions"])), g["rot"], rtol=1e-6, atol=1e-7)
    close(O.ddf_local_direction(T(inp["positions"]), T(inp["directions"])), g["d_loc"], rtol=1e-6, atol=1e-7)


def test_g6_ddf_model_plumbing():
    g = load("g6_ddf")
    inp = gi.sphere_rays(seed=6, M=40, radius=1.0)
    p, d = T(inp["positions"]), T(inp["directions"])
    Wp, Wd, Wc = T(g["Wp"]), T(g["Wd"]), T(g["Wc"])

    def ddf_fn(pos, dirs):  # same stand-ins as the generator, through OUR frame/encoding/head
        dl = O.ddf_local_direction(pos, dirs)
        x = torch.cat([dl, O.nerf_encoding(dl, 2, 0.0, 2.0, False)], -1)
        cond = torch.cat([pos, torch.sin(pos @ Wp)], -1)
        return torch.sigmoid((x @ Wd + torch.tanh(cond @ Wc))[..., 0]) * 2.0

    out = O.ddf_model_outputs(p, d, T(g["term"]), T(g["mv_points"]), T(g["sky_o"]), T(g["sky_d"]), 1.0, ddf_fn,
                              lambda x: x.norm(dim=-1, keepdim=True) - 0.5)
    for k in ["expected_termination_dist", "distance_weight", "sdf_at_termination",
              "multi_view_expected_termination_dist", "sky_ray_termination_dist",
              "sky_ray_expected_termination_dist"]:
        close(out[k], g["out_" + k], rtol=2e-5, atol=2e-6)
    ld = O.ddf_losses(out["expected_termination_dist"], T(g["term"]), T(g["mask"]), out["distance_weight"],
                      out["sdf_at_termination"], out["multi_view_expected_termination_dist"],
                      out["multi_view_termintation_dist"], out["sky_ray_expected_termination_dist"],
                      out["sky_ray_termination_dist"])
    for k, v in ld.items():
        close(v, g["loss_" + k], rtol=2e-5, atol=1e-6)


def test_g7_losses():
    g = load("g7_losses")
    out = {"rgb": T(g["rgb"]), "eik_grad": T(g["eik"]), "weights": T(g["w"]), "normal": T(g["normal"]),
           "hdr_background_colours": T(g["hdr_bg"]), "grid_density": T(g["grid_density"]),
           "sdf_at_termination": T(g["sdf_term"])}
    ld = O.neusky_losses(out, T(g["image"]), T(g["mask"]), torch.tensor(2.0))
    keys = [k[5:] for k in g if k.startswith("loss_")]  # (the train branch; "evalloss_" etc. below)
    assert sorted(keys) == sorted(ld.keys())
    for k in keys:
        close(ld[k], g["loss_" + k], rtol=1e-5, atol=1e-7)
    m3 = T(g["mask"][:, 3].astype(np.float32))[:, None].expand(-1, 3)
    close(O.sky_pixel_loss(O.linear_to_srgb(T(g["hdr_bg"])), T(g["image"]), m3, 0.1), g["sky_direct"], rtol=1e-6)
    # the evaluation / eval-latent-fitting branch (neusky_model.py:1036-1059), the `neusky` inclusions, every optional term, and
    # the nerf_osr_envmap method (no sky-pixel term)
    for tag, kw in (("evalloss_", {}), ("evalall_", dict(rgb_l2=True, cosine_colour=True)),
                    ("evalosr_", dict(rgb_l2=True, cosine_colour=True, sky_pixel=False))):
        le = O.neusky_eval_losses(T(g["rgb"]), T(g["hdr_bg"]), T(g["image"]), T(g["mask"]), **kw)
        ekeys = [k[len(tag):] for k in g if k.startswith(tag)]
        assert sorted(ekeys) == sorted(le.keys()), (tag, ekeys, sorted(le))
        for k in ekeys:
            close(le[k], g[tag + k], rtol=1e-5, atol=1e-7)
    # coefficient quirk: 'eikonal_loss' is NOT scaled by the 'eikonal loss' coefficient
    sc = O.scale_dict(ld, O.NEUSKY_LOSS_COEFFICIENTS)
    assert float(sc["eikonal_loss"]) == float(ld["eikonal_loss"])
    assert abs(float(sc["ground_plane_loss"]) - 0.1 * float(ld["ground_plane_loss"])) < 1e-7


def film_weights(cfg):
    from make_golden_weights import film_siren_weights
    in_dim, map_in, hidden, layers, mh, ml, out_dim, seed = [int(v) for v in cfg]
    return film_siren_weights(seed, in_dim, map_in, hidden, layers, mh, ml, out_dim)


def test_g8_film_siren():
    for tag in ["small", "full"]:
        g = load(f"g8_film_siren_{tag}")
        w = {"ddf." + k: T(v) for k, v in film_weights(g["cfg"]).items()}
        y = O.film_siren(T(g["x"]), T(g["cond"]), w)
        close(y, g["y"], rtol=2e-4, atol=2e-5)
        y64 = O.film_siren(T(g["x"]).double(), T(g["cond"]).double(), {k: v.double() for k, v in w.items()})
        close(y64, g["y"], rtol=3e-4, atol=3e-5)


def test_g9_sample_illumination():
    g = load("g9_sample_illumination")
    R, S, D, NT, L = [int(v) for v in g["shape"]]
    A = T(g["A"])

    def decode(lat, dirs, scale):
        return torch.exp(torch.tanh(torch.einsum("blc,bc->bl", lat, dirs) @ A) * scale[:, None])

    cols, inv, bg = O.sample_illumination(T(g["cam"]), T(g["ray_dirs"]), T(gi.fibonacci_sphere(D)), T(g["latents"]),
                                          T(g["scales"]), decode)
    per_ray = cols[inv]  # [R,D,3]
    ref = g["hdr_illumination_colours"].reshape(R, S, D, 3)
    for s in range(S):
        close(per_ray, ref[:, s], rtol=1e-6, atol=1e-7)
    close(bg, g["hdr_background_colours"], rtol=1e-6, atol=1e-7)
    ref_dirs = g["illumination_directions"].reshape(R, S, D, 3)
    np.testing.assert_array_equal(ref_dirs[3, 1], gi.fibonacci_sphere(D))


def test_g11_field_plumbing():
    g = load("g11_field_outputs")
    R, S = g["starts"].shape[:2]
    o, d, starts = T(g["o"]), T(g["d"]), T(g["starts"])
    x = (o[:, None] + d[:, None] * starts).reshape(-1, 3).requires_grad_(True)
    h = torch.nn.functional.softplus(x @ T(g["geo_w1"]).T + T(g["geo_b1"]), beta=100) @ T(g["geo_w2"]).T + T(g["geo_b2"])
    sdf, feat = h[:, :1], h[:, 1:]
    grads = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True)[0]
    p = {}
    for i in range(3):
        p[f"field.clin{i}.v"], p[f"field.clin{i}.g"], p[f"field.clin{i}.b"] = T(g[f"cw{i}"]), T(g[f"cg{i}"]), T(g[f"cb{i}"])
    close(O.colour_network(x, feat, p).view(R, S, 3), g["albedo"], rtol=1e-5, atol=1e-6)
    close(sdf.view(R, S, 1), g["sdf"], rtol=1e-5, atol=1e-6)
    close(grads.view(R, S, 3), g["gradients"], rtol=1e-5, atol=1e-6)
    close(torch.nn.functional.normalize(grads.view(R, S, 3), dim=-1), g["normals"], rtol=1e-5, atol=1e-6)
