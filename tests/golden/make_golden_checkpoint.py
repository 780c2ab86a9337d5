"""Reference-format checkpoint fixture: tests/golden/ref_state_dict.npz.

Run ONCE in the build container, where /root/reference is mounted:   python tests/golden/make_golden_checkpoint.py
It imports the reference through the stub importer, builds the reference's OWN modules

  * `neusky.utils.siren.DDFFiLMSiren` (siren.py:147-208: mapping network, FiLM layers, head) with the reference's initialisers,
  * the colour network exactly as `SDFAlbedoField.__init__` builds it (sdf_albedo_field.py:147-161: nn.Linear + torch weight_norm under
    the attribute names `clin{l}`), evaluated by the reference's unbound `SDFAlbedoField.get_colors` (:185-209),
  * a geometry stack under nerfstudio's attribute names `glin{l}` (torch weight_norm; nerfstudio's initialiser is absent from the
    reference tree, so only the KEY NAMES and the weight-norm semantics of these layers are pinned here, not their values),

seeds them, rounds every weight to fp16 (so the fixture stays small: the rounded values are loaded back into the reference modules
BEFORE the expected outputs are computed), dumps `state_dict()` under the prefixes a NeuSky checkpoint uses
(neusky_pipeline.py:174-194: `_model.field.*`, `_model.visibility_field.field.ddf.*`) and the outputs of the reference modules on
seeded inputs.  tests/test_checkpoints.py loads the file with utils.checkpoints.load_reference_pipeline_state and reproduces the
outputs through the HIP kernels.
"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_stub_importer import install  # noqa: E402

install("/root/reference")
import neusky.fields.sdf_albedo_field as rsdf  # noqa: E402
import neusky.utils.siren as rsiren  # noqa: E402

# shapes of the fixture (a reduced NeuSky: the test builds its pipeline with the same numbers)
DDF = dict(input_dim=15, mapping_network_input_dim=35, siren_hidden_features=128, siren_hidden_layers=2, mapping_network_features=128,
           mapping_network_layers=2, out_features=1)
FIELD = dict(hidden_dim=64, geo_feat_dim=64, hidden_dim_color=64, grid_dim=32, pe_dim=36)
M_DDF, N_COL = 4096, 512


def fixture_inputs():
    """seeded inputs, generated identically by the test (numpy Generator streams are stable across platforms)"""
    g = np.random.default_rng(20240704)
    x = g.uniform(-1.0, 1.0, (M_DDF, DDF["input_dim"])).astype(np.float32)
    cond = (0.5 * g.standard_normal((M_DDF, DDF["mapping_network_input_dim"]))).astype(np.float32)
    pts = g.uniform(-0.8, 0.8, (N_COL, 3)).astype(np.float32)
    feat = (0.3 * g.standard_normal((N_COL, FIELD["geo_feat_dim"]))).astype(np.float32)
    return x, cond, pts, feat


def nerf_encoding_standin(x, num_freq, max_exp):
    """nerfstudio NeRFEncoding(in_dim=3, num_frequencies, min 0, max max_exp, include_input=False): external to the reference tree; the
    same stand-in as make_golden.py (sin of all scaled inputs, then the same with a pi / 2 phase)"""
    freqs = 2.0 ** torch.linspace(0.0, max_exp, num_freq)
    s = (2 * torch.pi * x)[..., None] * freqs
    s = s.reshape(*s.shape[:-2], -1)
    return torch.sin(torch.cat([s, s + torch.pi / 2.0], -1))


def round_to_half_(module):
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(p.half().float())


def main():
    torch.manual_seed(4242)
    torch.set_num_threads(4)
    x, cond, pts, feat = fixture_inputs()
    out = {}
    # ---- the DDF network (reference module, reference initialisers)
    ddf = rsiren.DDFFiLMSiren(**DDF)
    with torch.no_grad():  # trained-looking biases (the initialisers leave nn.Linear's defaults; any values do)
        for p in ddf.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
    round_to_half_(ddf)
    for k, v in ddf.state_dict().items():
        out["_model.visibility_field.field.ddf." + k] = v.numpy().astype(np.float16)
    with torch.no_grad():
        res = ddf(torch.cat([torch.from_numpy(cond), torch.from_numpy(x)], -1))  # forward(): [positions | directions] (:189-197)
    out["expect.ddf_raw"] = res.numpy()
    # ---- the colour network: the constructor's own loop (sdf_albedo_field.py:147-161), then the reference's get_colors
    f = FIELD
    host = torch.nn.Module()
    host.config = NS(geo_feat_dim=f["geo_feat_dim"], predict_shininess=False)
    dims = [3 + f["pe_dim"] + f["geo_feat_dim"]] + [f["hidden_dim_color"]] * 2 + [3]
    host.num_layers_color = len(dims)
    for l in range(host.num_layers_color - 1):
        lin = torch.nn.utils.weight_norm(torch.nn.Linear(dims[l], dims[l + 1]))
        setattr(host, "clin" + str(l), lin)
    # nerfstudio's geometry layers: names + weight-norm parameterisation only (values: plain nn.Linear initialisation)
    gdims = [3 + f["pe_dim"] + f["grid_dim"]] + [f["hidden_dim"]] * 2 + [1 + f["geo_feat_dim"]]
    for l in range(len(gdims) - 1):
        setattr(host, "glin" + str(l), torch.nn.utils.weight_norm(torch.nn.Linear(gdims[l], gdims[l + 1])))
    with torch.no_grad():
        for n, p in host.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(0.5 + torch.rand_like(p))  # g != |v|: the loader must honour both factors
    round_to_half_(host)
    host.position_encoding = lambda p_: nerf_encoding_standin(p_, 6, 5.0)
    host.relu, host.sigmoid = torch.nn.ReLU(), torch.nn.Sigmoid()
    with torch.no_grad():
        rgb = rsdf.SDFAlbedoField.get_colors(host, torch.from_numpy(pts), torch.from_numpy(feat))
    out["expect.albedo"] = rgb.numpy()
    sd = {k: v for k, v in host.state_dict().items()}
    assert set(k.split(".")[-1] for k in sd) == {"weight_g", "weight_v", "bias"}, sorted(sd)  # (old-style weight_norm key names)
    for k, v in sd.items():
        out["_model.field." + k] = v.numpy().astype(np.float16)
    out["_model.field.deviation_network.variance"] = np.array([0.3], dtype=np.float16)
    # effective (weight-normed) geometry weights of the reference parameterisation, for the loader's semantics check
    with torch.no_grad():  # torch's weight_norm: w = g v / |v| per output row (dim 0)
        out["expect.glin1_weight"] = torch._weight_norm(host.glin1.weight_v, host.glin1.weight_g, 0).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_state_dict.npz"), **out)
    n = sum(v.size for k, v in out.items() if k.startswith("_model."))
    print(f"wrote ref_state_dict.npz: {len([k for k in out if k.startswith('_model.')])} state tensors, {n} values")
    assert not any("__pycache__" in r for r, _, _ in os.walk("/root/reference")), "bytecode leaked into reference"


if __name__ == "__main__":
    main()
