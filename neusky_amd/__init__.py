"""neusky_amd - MI355X-native NeuSky hot path (hand-written HIP behind a C ABI, PyTorch-ROCm host code).

The package mirrors the nerfstudio Field/Model/Pipeline surface of JADGardner/neusky for the per-ray
train/render step only.  All arithmetic of the hot path runs in
`libneusky_hip.so` (built from neusky_amd/csrc by `__graft_entry__.build()`); there is no CPU or
eager-PyTorch fallback - importing `neusky_amd.hip` without the library raises.
"""
__version__ = "0.1.0"
