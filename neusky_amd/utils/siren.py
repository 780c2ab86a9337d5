"""FiLM-SIREN parameter holder.  Arithmetic spec = the only in-tree statement of it in the reference,
neusky/utils/siren.py:108-208 (`DDFFiLMSiren`, `CustomMappingNetwork`, `FiLMLayer`); the configured
network is `reni.field_components.film_siren.FiLMSiren` whose source is absent (SURVEY.md A9 - parity
unpinned beyond siren.py).  Same sub-module / parameter names as siren.py so its state_dict loads.
The forward runs in `ops.FilmSirenFn` (fp32 MFMA layers with fused LeakyReLU / FiLM-sine epilogues)."""
from __future__ import annotations

import math
from typing import List

import torch
from torch import nn

from .. import ops


class _FiLMLayer(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int):
        super().__init__()
        self.layer = nn.Linear(input_dim, hidden_dim)


class _Mapping(nn.Module):
    def __init__(self, in_features, layers, hidden, out_dim):
        super().__init__()
        mods: List[nn.Module] = []
        for _ in range(layers):
            mods += [nn.Linear(in_features, hidden), nn.LeakyReLU(0.2, inplace=True)]
            in_features = hidden
        mods.append(nn.Linear(hidden, out_dim))
        self.network = nn.Sequential(*mods)
        for m in self.network:
            if isinstance(m, nn.Linear):  # siren.py:85-88,123
                nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        with torch.no_grad():
            self.network[-1].weight *= 0.25  # siren.py:124-125

    def linears(self) -> List[nn.Linear]:
        return [m for m in self.network if isinstance(m, nn.Linear)]


class FiLMSiren(nn.Module):
    def __init__(self, in_dim: int, hidden_layers: int, hidden_features: int, mapping_network_in_dim: int,
                 mapping_network_layers: int, mapping_network_features: int, out_dim: int, outermost_linear: bool = True,
                 out_activation=None):
        super().__init__()
        if not outermost_linear or out_activation is not None:
            raise NotImplementedError("only the linear head of the `neusky` config is implemented")
        self.in_dim, self.cond_dim, self.out_dim = in_dim, mapping_network_in_dim, out_dim
        self.n_film, self.n_map, self.hidden = hidden_layers, mapping_network_layers, hidden_features
        self.net = nn.ModuleList([_FiLMLayer(in_dim, hidden_features)] +
                                 [_FiLMLayer(hidden_features, hidden_features) for _ in range(hidden_layers - 1)])
        self.final_layer = nn.Linear(hidden_features, out_dim)
        self.mapping_network = _Mapping(mapping_network_in_dim, mapping_network_layers, mapping_network_features,
                                        hidden_layers * hidden_features * 2)
        with torch.no_grad():  # siren.py:91-105,185-187
            for i, l in enumerate(self.net):
                n_in = l.layer.weight.shape[1]
                bound = 1.0 / n_in if i == 0 else math.sqrt(6.0 / n_in) / 25.0
                l.layer.weight.uniform_(-bound, bound)
            self.final_layer.weight.uniform_(-math.sqrt(6.0 / hidden_features) / 25.0, math.sqrt(6.0 / hidden_features) / 25.0)

    def invalidate_weight_cache(self) -> None:
        """once per optimisation step -- unless every weight is frozen (the RENI++ decoder): its padded copies and, through them,
        its packed weight streams (ops._film_stream) then stay valid from step to step"""
        c = getattr(self, "_wcache", None)
        frozen = not any(p.requires_grad for p in self.parameters())
        # kept only when the copies were BUILT while the network was frozen and it still is: a cache filled during a train step
        # (trainable weights, updated since through raw pointers by hip.adam_step) must not survive a later freeze -- the
        # eval-latent fit freezes every other parameter for its duration (NeuSkyFactoModel.fit_latent_codes_for_eval)
        if c and c.get("frozen", False) and frozen:
            return
        if c and "wb" in c:
            ops.forget_film_streams(c["wb"])  # packed streams keyed by the dropped copies (frozen ones are not purged by begin_step)
        self._wcache = {}

    def padded_weights(self):
        c = getattr(self, "_wcache", None)
        ver = tuple(p._version for p in self.parameters())  # (a frozen net keeps its cache over steps: a checkpoint load must drop it)
        if c is not None and "wb" in c and c["grad"] == torch.is_grad_enabled() and c["ver"] == ver:
            return c["wb"]
        wb = self._padded_weights_uncached()
        if c is not None:
            c["wb"], c["grad"], c["ver"] = wb, torch.is_grad_enabled(), ver
            c["frozen"] = not any(p.requires_grad for p in self.parameters())
        return wb

    def _padded_weights_uncached(self):
        # copies of a frozen network live across steps: they get allocations of their own, not regions of one step's zero arena
        keep = not any(p.requires_grad for p in self.parameters())
        wb = []
        for lin in self.mapping_network.linears():
            wb += [ops.pad_weight(lin.weight, keep), ops.pad_bias(lin.bias, keep)]
        for l in self.net:
            wb += [ops.pad_weight(l.layer.weight, keep), ops.pad_bias(l.layer.bias, keep)]
        wb += [ops.pad_weight(self.final_layer.weight, keep), ops.pad_bias(self.final_layer.bias, keep)]
        return wb

    def forward(self, x: torch.Tensor, conditioning_input: torch.Tensor, train_weights: bool = True, padded_output: bool = False) -> torch.Tensor:
        """x [M, pad4(in_dim)], conditioning_input [M, pad4(cond_dim)] (zero padded columns) -> [M, out_dim]
        (padded_output: the kernels' own [M, pad4(out_dim)] matrix, for a caller whose next kernel reads it as it is)"""
        need_dcond = conditioning_input.requires_grad
        out = ops.FilmSirenFn.apply(x, conditioning_input, self.n_map, self.n_film, train_weights, need_dcond,
                                    *self.padded_weights())
        return out if padded_output else out[:, :self.out_dim]
