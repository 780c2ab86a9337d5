"""Seeded on-disk fixtures for the dataparser tests: a miniature NeRF-OSR scene folder (lk2 layout, Cityscapes palette
masks, sessions, eval masks, out-of-view object masks, env maps) and a miniature synthetic scene (transforms.json layout).
Pure data generation (numpy + PIL); used by tests/golden/make_golden_dataparser.py (to run the reference's parsers on
it) and by tests/test_dataparsers.py (to run ours on the same bytes)."""
from __future__ import annotations

import json
import os

import numpy as np
from PIL import Image

PALETTE = [[128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153], [250, 170, 30],
           [220, 220, 0], [107, 142, 35], [152, 251, 152], [70, 130, 180], [220, 20, 60], [255, 0, 0], [0, 0, 142], [0, 0, 70],
           [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32]]
SESSIONS = ["01-08_07_30", "01-08_10_00", "05-08_15_00"]


def _look_at_opencv(eye, target, roll):
    """camera-to-world, x right / y down / z forward, with a small roll about the optical axis"""
    z = target - eye
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    c, s = np.cos(roll), np.sin(roll)
    x, y = c * x + s * y, -s * x + c * y
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye
    return m


def _write_mat(path, m):
    with open(path, "w") as f:
        f.write(" ".join(f"{v:.9g}" for v in np.asarray(m, dtype=np.float64).reshape(-1)))


def _segmentation(rng, w, h):
    """blocks of palette colours (plus an unlabelled black block) covering every mask class at least somewhere"""
    seg = np.zeros((h, w, 3), dtype=np.uint8)
    bs = 5
    for by in range(0, h, bs):
        for bx in range(0, w, bs):
            k = int(rng.integers(0, len(PALETTE) + 1))
            seg[by:by + bs, bx:bx + bs] = PALETTE[k] if k < len(PALETTE) else [0, 0, 0]
    return seg


def build_nerfosr(root: str, seed: int = 7) -> str:
    """<root>/lk2/final/... ; returns the `data` folder to hand to the parser config (scene='site1')"""
    rng = np.random.default_rng(seed)
    scene_dir = os.path.join(root, "lk2", "final")
    plan = {"train": 2, "validation": 1, "test": 2}  # images per session
    sizes = [(40, 30), (44, 30), (40, 34), (46, 36)]
    counter = 0
    for split, per_session in plan.items():
        for sub in ("rgb", "pose", "intrinsics", "cityscapes_mask", "mask", "out_of_view_frustum_objects_mask"):
            os.makedirs(os.path.join(scene_dir, split, sub), exist_ok=True)
        for s_i, session in enumerate(SESSIONS):
            for j in range(per_session):
                name = f"{session}_IMG_{1000 + counter}"
                w, h = sizes[counter % len(sizes)]
                ang = 2 * np.pi * (counter / 15.0) + rng.normal(0, 0.05)
                eye = np.array([6 * np.cos(ang), 6 * np.sin(ang), 1.5 + rng.normal(0, 0.2)]) + np.array([20.0, -10.0, 3.0])
                pose = _look_at_opencv(eye, np.array([20.0, -10.0, 3.5]) + rng.normal(0, 0.1, 3), rng.normal(0, 0.02))
                K = np.eye(4)
                K[0, 0] = K[1, 1] = 35.0 + counter
                K[0, 2], K[1, 2] = w / 2.0, h / 2.0
                _write_mat(os.path.join(scene_dir, split, "pose", name + ".txt"), pose)
                _write_mat(os.path.join(scene_dir, split, "intrinsics", name + ".txt"), K)
                Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(scene_dir, split, "rgb", name + ".png"))
                Image.fromarray(_segmentation(rng, w, h)).save(os.path.join(scene_dir, split, "cityscapes_mask", name + ".png"))
                if split == "test" and j == 1:  # eval masks only for the non-hold-out frame of each session
                    m = (rng.random((h, w)) > 0.3).astype(np.uint8) * 255
                    Image.fromarray(m if s_i else np.stack([m] * 3, -1)).save(os.path.join(scene_dir, split, "mask", name + ".png"))
                if split == "train" and j == 0:
                    o = np.zeros((h, w, 3), dtype=np.uint8)
                    o[: h // 3, w // 2:] = 255
                    Image.fromarray(o).save(os.path.join(scene_dir, split, "out_of_view_frustum_objects_mask", name + ".png"))
                counter += 1
    for session in SESSIONS:
        d = os.path.join(scene_dir, "ENV_MAP_CC", session)
        os.makedirs(d, exist_ok=True)
        Image.fromarray(rng.integers(0, 256, (8, 16, 3), dtype=np.uint8)).save(os.path.join(d, "envmap.png"))
    return root


def build_synthetic(root: str, seed: int = 11) -> str:
    """<root>/renders/scene/{transforms.json,points3d.ply,train,validation} + <root>/hdris ; returns the scene folder"""
    rng = np.random.default_rng(seed)
    scene = os.path.join(root, "renders", "scene")
    os.makedirs(os.path.join(root, "hdris"), exist_ok=True)
    open(os.path.join(root, "hdris", "sunny.exr"), "wb").close()
    frames = []
    w, h = 24, 18
    k = 0
    for split, n in (("train", 5), ("validation", 2)):
        for sub in ("rgb", "cityscapes_mask", "albedo", "normal"):
            os.makedirs(os.path.join(scene, split, sub), exist_ok=True)
        for i in range(n):
            name = f"{i:04d}"
            ang = 2 * np.pi * k / 7.0
            eye = np.array([4 * np.cos(ang), 4 * np.sin(ang), 1.0 + 0.3 * rng.normal()])
            cv = _look_at_opencv(eye, rng.normal(0, 0.05, 3), 0.0)
            gl = cv.copy()
            gl[:3, 1:3] *= -1
            fr = {"file_path": f"{split}/rgb/{name}.png", "transform_matrix": gl.tolist()}
            if k % 2 == 0:
                fr.update(fl_x=30.0 + k, fl_y=31.0 + k, cx=w / 2 + 0.5, cy=h / 2 - 0.5)
            if k % 3 == 0:
                fr.update(envmap_name="sunny", envmap_rotation=0.25 * k)
            frames.append(fr)
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(scene, split, "rgb", name + ".png"))
            Image.fromarray(_segmentation(rng, w, h)).save(os.path.join(scene, split, "cityscapes_mask", name + ".png"))
            if split == "validation":
                open(os.path.join(scene, split, "albedo", name + ".exr"), "wb").close()
                if i == 0:
                    open(os.path.join(scene, split, "normal", name + ".exr"), "wb").close()
            k += 1
    # one frame of the json has no image on disk and one image has no frame: both must be skipped
    frames.append({"file_path": "train/rgb/9999.png", "transform_matrix": np.eye(4).tolist()})
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(scene, "train", "rgb", "extra.png"))
    with open(os.path.join(scene, "transforms.json"), "w") as f:
        json.dump({"fl_x": 28.0, "fl_y": 28.5, "cx": w / 2, "cy": h / 2, "w": w, "h": h, "frames": frames}, f)
    pts = rng.normal(0, 1.0, (200, 3)).astype("<f4")
    pts[:10] *= 30.0  # outliers
    rec = np.zeros(200, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    with open(os.path.join(scene, "points3d.ply"), "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 200\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        f.write(rec.tobytes())
    return scene
