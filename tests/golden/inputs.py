"""Seeded synthetic inputs shared by the golden-vector generator (make_golden.py) and the parity
tests.  Pure numpy (PCG64 streams are platform-stable), so the GPU box regenerates bit-identical
inputs without access to /root/reference.  Every builder returns float32 / int64 numpy arrays.
"""
from __future__ import annotations

import numpy as np


def rng(seed: int) -> np.random.Generator:
    return np.random.default_rng(seed)


def unit(v: np.ndarray) -> np.ndarray:
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def fibonacci_sphere(n: int) -> np.ndarray:
    """n near-uniform unit directions (documented stand-in for reni's icosphere sampler)."""
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0**0.5) * i
    d = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)
    return d.astype(np.float32)


def lambertian_inputs(seed: int, R: int, S: int, D: int, U: int):
    """Compact (un-broadcast) inputs of the hemisphere integral + composite."""
    g = rng(seed)
    albedo = g.uniform(0.0, 1.0, (R, S, 3)).astype(np.float32)
    normals = unit(g.normal(size=(R, S, 3))).astype(np.float32)
    # a few degenerate normals / zero weights to exercise count==0 and empty rays
    normals[0, 0] = 0.0
    dirs = fibonacci_sphere(D)
    cam_colours = np.exp(g.normal(0.0, 1.0, (U, D, 3))).astype(np.float32)  # HDR, positive
    cam_of_ray = g.integers(0, U, (R,)).astype(np.int64)
    vis = g.uniform(0.0, 1.0, (R, D)).astype(np.float32)
    bg = np.exp(g.normal(0.0, 0.5, (R, 3))).astype(np.float32)
    alpha = g.uniform(0.0, 0.3, (R, S)).astype(np.float32)
    T = np.cumprod(np.concatenate([np.ones((R, 1), np.float32), 1.0 - alpha + 1e-7], 1), 1)
    w = (alpha * T[:, :-1]).astype(np.float32)
    if R > 1:
        w[1] = 0.0  # ray that hits nothing: rgb == sRGB(bg)
    return dict(albedo=albedo, normals=normals, dirs=dirs, cam_colours=cam_colours,
                cam_of_ray=cam_of_ray, vis=vis, bg=bg, weights=w)


def visibility_inputs(seed: int, R: int, S: int, D: int, n_outside: int = 2):
    """Inputs of compute_visibility: camera rays inside the unit sphere, rendered
    depth per ray, D light directions.  The first `n_outside` rays get a depth that lands outside
    the DDF sphere to exercise the fix-up branch (neusky_model.py:1674-1683)."""
    g = rng(seed)
    o = g.uniform(-0.4, 0.4, (R, 3)).astype(np.float32)
    d = unit(g.normal(size=(R, 3))).astype(np.float32)
    depth = g.uniform(0.1, 0.5, (R, 1)).astype(np.float32)
    depth[:n_outside] = 3.0
    origins = np.repeat(o[:, None, :], S, 1)
    directions = np.repeat(d[:, None, :], S, 1)
    dirs = fibonacci_sphere(D)
    return dict(origins=origins, directions=directions, depth=depth, dirs=dirs)


def sphere_rays(seed: int, M: int, radius: float = 1.0, near_pole: int = 2):
    """Points on the DDF sphere + inward directions (SURVEY §8 A7 local frame)."""
    g = rng(seed)
    p = unit(g.normal(size=(M, 3)))
    if near_pole:
        p[:near_pole] = unit(np.array([[1e-3, 2e-3, 1.0], [-2e-3, 1e-3, -1.0]])[:near_pole])
    p = (p * radius).astype(np.float32)
    d = unit(-p / radius + 0.5 * g.normal(size=(M, 3))).astype(np.float32)
    return dict(positions=p, directions=d)


def seeded_linear(g: np.random.Generator, out_f: int, in_f: int, scale: float):
    w = g.uniform(-scale, scale, (out_f, in_f)).astype(np.float32)
    b = g.uniform(-scale, scale, (out_f,)).astype(np.float32)
    return w, b
