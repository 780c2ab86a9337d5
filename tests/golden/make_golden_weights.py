"""Seeded FiLM-SIREN weights shared by make_golden.py and the tests (pure numpy)."""
import inputs as gi


def film_siren_weights(seed, in_dim, map_in, hidden, layers, map_hidden, map_layers, out_dim):
    """Seeded weights in the layout of neusky/utils/siren.py DDFFiLMSiren (shared with the tests)."""
    g = gi.rng(seed)
    w = {}
    fan = map_in
    for i in range(map_layers):
        w[f"map_w{i}"], w[f"map_b{i}"] = gi.seeded_linear(g, map_hidden, fan, (6.0 / fan) ** 0.5 * 0.6)
        fan = map_hidden
    w["map_wo"], w["map_bo"] = gi.seeded_linear(g, layers * hidden * 2, fan, 0.25 * (6.0 / fan) ** 0.5 * 0.6)
    fan = in_dim
    for i in range(layers):
        sc = 1.0 / fan if i == 0 else (6.0 / fan) ** 0.5 / 25.0
        w[f"film_w{i}"], w[f"film_b{i}"] = gi.seeded_linear(g, hidden, fan, sc)
        fan = hidden
    w["out_w"], w["out_b"] = gi.seeded_linear(g, out_dim, fan, (6.0 / fan) ** 0.5 / 25.0)
    return w
