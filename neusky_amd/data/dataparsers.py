"""On-disk parsers for NeRF-OSR (+ Cityscapes masks) and for the synthetic multi-illumination layout
(SURVEY.md section 8(f) item 4).  Pure host I/O: folders -> cameras + file lists -> image / 4-channel mask stacks
that `DeviceImageDataManager.from_dataset` uploads once.

Mirrors, by behaviour:
  * `NeRFOSRCityScapes._generate_dataparser_outputs`  neusky/data/dataparsers/nerfosr_cityscapes_dataparser.py:220-468
    (scene aliases :226-231, `final` / `final_clean` roots :233-238, poses of ALL splits normalised together :241-281,
     crop / pad principal points :248-261, z-shift to the camera plane :274, auto scale :276-281, equirect env-map
     cameras :316-333, session maps :337-363, Cityscapes palette masks :370-395, out-of-view object masks :398-415,
     test eval masks + hold-out check :418-441, metadata keys :445-463);
  * `get_camera_params` :135-172 (OpenCV -> OpenGL: columns 1,2 of the pose negated, :164);
  * `CustomNeuskyDataparser` neusky/data/dataparsers/custom_neusky_dataparser.py:167-596 (transforms.json with
     per-frame intrinsics falling back to the global ones :178-207, GT EXR layer discovery :216-262, HDRI lookup
     :264-296, SfM-point centring :298-387, split slicing with fall-back to train :498-512);
  * `NeuSkyDataset` neusky/data/datasets/neusky_dataset.py:113-344 (centre crop / pad / rescale of images :151-185, the
     mask stack [static, fg, ground, sky] :221-319, palette matching :321-338, env maps :340-344).

nerfstudio's `camera_utils.auto_orient_and_center_poses` / `focus_of_attention` / `rotation_matrix` are an external
dependency absent from /root/reference; they are restated here from the published nerfstudio algorithm
(orientation 'up' / 'vertical' / 'pca' / 'none', centring 'poses' / 'focus' / 'none') -- parity for that piece is
UNPINNED (no reference source to run); everything else in this file is pinned by tests/golden/dataparser_*.npz, which
were produced by running the reference's parser and dataset on a seeded on-disk fixture with this restatement
injected for the missing nerfstudio function (tests/golden/make_golden_dataparser.py).
"""
from __future__ import annotations

import glob
import json
import math
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

# Cityscapes train-id classes and their palette (public Cityscapes label definition; the reference keeps the same
# table at nerfosr_cityscapes_dataparser.py:48-91)
CITYSCAPE_CLASSES = {
    "classes": ["road", "sidewalk", "building", "wall", "fence", "pole", "traffic light", "traffic sign", "vegetation",
                "terrain", "sky", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"],
    "colours": [[128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153],
                [250, 170, 30], [220, 220, 0], [107, 142, 35], [152, 251, 152], [70, 130, 180], [220, 20, 60],
                [255, 0, 0], [0, 0, 142], [0, 0, 70], [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32]],
}
TRANSIENT_CLASSES = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
FOREGROUND_CLASSES = ["road", "sidewalk", "building", "wall", "fence", "pole", "traffic light", "traffic sign", "terrain"]
GT_LAYER_CHANNELS = {"albedo": 3, "normal": 3, "depth": 1, "roughness": 1, "metallic": 1, "ior": 1, "transmission": 1}
IMAGE_EXTS = ("*.png", "*.jpg", "*.JPG", "*.PNG")
SCENE_ALIASES = {"site1": "lk2", "site2": "st", "site3": "lwp"}
SCENES_WITHOUT_SESSIONS = ("trevi", "europa", "rathaus", "schloss")


# --------------------------------------------------------------------------------------------- containers
@dataclass
class Cameras:
    """the slice of nerfstudio's `Cameras` the path touches: per-camera pinhole (or equirect) parameters"""
    camera_to_worlds: torch.Tensor  # [N,3,4], nerfstudio/OpenGL convention
    fx: torch.Tensor
    fy: torch.Tensor
    cx: torch.Tensor
    cy: torch.Tensor
    camera_type: str = "perspective"

    def __len__(self) -> int:
        return int(self.camera_to_worlds.shape[0])


@dataclass
class Semantics:
    filenames: List[str]
    classes: List[str]
    colors: torch.Tensor  # [C,3] uint8


@dataclass
class DataparserOutputs:
    image_filenames: List[str]
    cameras: Cameras
    scene_box: Dict[str, torch.Tensor]
    mask_filenames: Optional[List[str]] = None
    metadata: Dict[str, Any] = field(default_factory=dict)
    dataparser_scale: float = 1.0


def find_files(directory: str, exts=IMAGE_EXTS, recursive: bool = False) -> List[str]:
    """sorted paths under `directory` matching any of the glob patterns (:94-115); [] when the folder is missing"""
    if not os.path.isdir(directory):
        return []
    found: List[str] = []
    for pattern in exts:
        found += glob.glob(os.path.join(directory, "**", pattern) if recursive else os.path.join(directory, pattern),
                           recursive=recursive)
    return sorted(found)


def parse_4x4_txt(path: str) -> np.ndarray:
    """a NeRF-OSR pose / intrinsics text file: 16 whitespace-separated numbers, row-major (:118-132)"""
    with open(path, encoding="UTF-8") as f:
        vals = [float(t) for t in f.read().split()]
    if len(vals) != 16:
        raise ValueError(f"{path}: expected 16 numbers, found {len(vals)}")
    return np.asarray(vals, dtype=np.float64).reshape(4, 4).astype(np.float32)


def get_camera_params(scene_dir: str, split: str) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """intrinsics [N,4,4], camera-to-world [N,4,4] in the OpenGL convention, N  (:135-172)"""
    k_files = find_files(f"{scene_dir}/{split}/intrinsics", exts=("*.txt",))
    p_files = find_files(f"{scene_dir}/{split}/pose", exts=("*.txt",))
    n = len(p_files)
    if n == 0:
        return torch.zeros(0, 4, 4), torch.zeros(0, 4, 4), 0
    K = np.stack([parse_4x4_txt(k_files[i]) for i in range(n)])
    P = np.stack([parse_4x4_txt(p_files[i]) for i in range(n)])
    P[:, 0:3, 1:3] *= -1.0  # x right / y down / z forward  ->  x right / y up / z backward
    return torch.from_numpy(K), torch.from_numpy(P), n


# --------------------------------------------------------------------- nerfstudio pose normalisation (restated)
def rotation_matrix(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """the rotation taking direction a onto direction b (Rodrigues form; antiparallel inputs are nudged)"""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    c = torch.dot(a, b)
    if c < -1 + 1e-8:
        eps = (torch.rand(3) - 0.5) * 0.01
        return rotation_matrix(a + eps, b)
    s = torch.linalg.norm(v)
    skew = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return torch.eye(3) + skew + skew @ skew * ((1 - c) / (s ** 2 + 1e-8))


def focus_of_attention(poses: torch.Tensor, initial_focus: torch.Tensor) -> torch.Tensor:
    """the point closest (least squares) to the optical axes of the cameras that look towards it; cameras facing
    away from the running estimate are dropped and the solve repeated until the active set is stable"""
    dirs = -poses[:, :3, 2:3]
    orig = poses[:, :3, 3:4]
    focus = initial_focus
    active = torch.sum(dirs.squeeze(-1) * (focus - orig.squeeze(-1)), dim=-1) > 0
    done = False
    while int(active.sum()) > 1 and not done:
        dirs, orig = dirs[active], orig[active]
        m = torch.eye(3) - dirs * dirs.transpose(-2, -1)
        mtm = m.transpose(-2, -1) @ m
        focus = torch.linalg.inv(mtm.mean(0)) @ (mtm @ orig).mean(0)[:, 0]
        active = torch.sum(dirs.squeeze(-1) * (focus - orig.squeeze(-1)), dim=-1) > 0
        done = bool(active.all())
    return focus


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "up", center_method: str = "poses"):
    """poses [N,3|4,4] -> (oriented poses [N,3,4], transform [3,4])"""
    origins = poses[..., :3, 3]
    mean_origin = origins.mean(0)
    diff = origins - mean_origin
    if center_method == "poses":
        translation = mean_origin
    elif center_method == "focus":
        translation = focus_of_attention(poses, mean_origin)
    elif center_method == "none":
        translation = torch.zeros_like(mean_origin)
    else:
        raise ValueError(f"Unknown value for center_method: {center_method}")
    if method == "pca":
        _, eigvec = torch.linalg.eigh(diff.T @ diff)
        eigvec = torch.flip(eigvec, dims=(-1,))
        if torch.linalg.det(eigvec) < 0:
            eigvec[:, 2] = -eigvec[:, 2]
        transform = torch.cat([eigvec, eigvec @ -translation[..., None]], dim=-1)
        oriented = transform @ poses
        if oriented.mean(0)[2, 1] < 0:
            oriented[:, 1:3] = -1 * oriented[:, 1:3]
    elif method in ("up", "vertical"):
        up = poses[:, :3, 1].mean(0)
        up = up / torch.linalg.norm(up)
        if method == "vertical":
            # the cameras' x axes span the horizontal plane when the photographer keeps the horizon level
            _, S, Vh = torch.linalg.svd(poses[:, :3, 0], full_matrices=False)
            if S[1] > 0.17 * math.sqrt(poses.shape[0]):
                cand = Vh[2, :]
                up = cand if torch.dot(cand, up) > 0 else -cand
            else:
                up = up - Vh[0, :] * torch.dot(up, Vh[0, :])
                up = up / torch.linalg.norm(up)
        rot = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rot, rot @ -translation[..., None]], dim=-1)
        oriented = transform @ poses
    elif method == "none":
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
        oriented = transform @ poses
    else:
        raise ValueError(f"Unknown value for method: {method}")
    return oriented, transform


def _normalise_poses(c2w: torch.Tensor, orientation_method: str, center_method: str, auto_scale: bool, scale_factor: float):
    """orient + centre, drop the cameras onto z = 0, scale into the unit box (:265-281)"""
    if c2w.shape[-2] == 3:
        c2w = torch.cat([c2w, torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(c2w.shape[0], 1, 4)], dim=1)
    c2w, transform = auto_orient_and_center_poses(c2w, method=orientation_method, center_method=center_method)
    c2w[:, 2, 3] -= c2w[:, 2, 3].mean(0)
    s = 1.0
    if auto_scale:
        s = s / float(torch.max(torch.abs(c2w[:, :3, 3])))
    c2w[:, :3, 3] *= s * scale_factor
    return c2w, transform


def _scene_box(scale: float) -> Dict[str, torch.Tensor]:
    return {"aabb": torch.tensor([[-scale] * 3, [scale] * 3], dtype=torch.float32)}


# ----------------------------------------------------------------------------------------------- NeRF-OSR
@dataclass
class NeRFOSRCityScapesDataParserConfig:
    """nerfosr_cityscapes_dataparser.py:175-198 on top of nerfstudio's NeRFOSRDataParserConfig fields"""
    data: Path = Path("data/NeRF-OSR/Data")
    scene: str = "site1"
    scene_scale: float = 1.0
    scale_factor: float = 1.0
    orientation_method: str = "vertical"
    center_method: str = "focus"
    auto_scale_poses: bool = True
    mask_source: str = "cityscapes"  # none | original | cityscapes
    crop_to_equal_size: bool = False
    pad_to_equal_size: bool = False
    run_segmentation_inference: bool = False
    mask_vegetation: bool = False
    session_holdout_indices: List[int] = field(default_factory=lambda: [0, 0, 0, 0, 0])
    session_env_map_scaling: float = 1.0
    session_env_map_scaling_threshold: float = 0.0
    mask_out_of_view_frustum_objects: bool = False
    include_sidewalk_in_ground_mask: bool = True

    def setup(self) -> "NeRFOSRCityScapes":
        return NeRFOSRCityScapes(self)


class NeRFOSRCityScapes:
    """Source convention: camera x right, y down, z into the scene (OpenCV / COLMAP); poses are camera-to-world;
    `mask/` images are 0 for dynamic content and 255 for static content."""

    def __init__(self, config: NeRFOSRCityScapesDataParserConfig):
        if config.crop_to_equal_size and config.pad_to_equal_size:
            raise AssertionError("Cannot crop and pad at the same time")
        self.config = config
        self.width_height: List[int] = []

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        return self._generate_dataparser_outputs(split)

    def _generate_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        cfg = self.config
        split = "validation" if split == "val" else split
        scene = SCENE_ALIASES.get(cfg.scene, cfg.scene)
        root = "final_clean" if scene == "trevi" else "final"
        scene_dir = f"{cfg.data}/{scene}/{root}"
        split_dir = f"{scene_dir}/{split}"

        K_tr, P_tr, n_train = get_camera_params(scene_dir, "train")
        K_va, P_va, n_val = get_camera_params(scene_dir, "validation")
        K_te, P_te, _ = get_camera_params(scene_dir, "test")
        K = torch.cat([K_tr, K_va, K_te], 0)
        if K.shape[0] == 0:
            raise ValueError(f"no camera files under {scene_dir}")
        self.width_height = []
        if cfg.crop_to_equal_size or cfg.pad_to_equal_size:
            pick = torch.min if cfg.crop_to_equal_size else torch.max
            pcx, pcy = pick(K[:, 0, 2]), pick(K[:, 1, 2])
            self.width_height = [int(pcx.item() * 2), int(pcy.item() * 2)]
            K[:, 0, 2] = pcx
            K[:, 1, 2] = pcy
        c2w, _ = _normalise_poses(torch.cat([P_tr, P_va, P_te], 0), cfg.orientation_method, cfg.center_method,
                                  cfg.auto_scale_poses, cfg.scale_factor)
        lo, hi = {"train": (0, n_train), "validation": (n_train, n_train + n_val), "test": (n_train + n_val, K.shape[0])}[split]
        c2w, K = c2w[lo:hi], K[lo:hi]
        cameras = Cameras(camera_to_worlds=c2w[:, :3, :4], fx=K[:, 0, 0], fy=K[:, 1, 1], cx=K[:, 0, 2], cy=K[:, 1, 2])

        image_filenames = find_files(f"{split_dir}/rgb")
        envmap_filenames = find_files(f"{scene_dir}/ENV_MAP_CC", recursive=True)
        envmap_cameras = None
        if envmap_filenames:
            with Image.open(envmap_filenames[0]) as im:
                ew, eh = im.size
            n_env = len(envmap_filenames)
            envmap_cameras = Cameras(
                camera_to_worlds=torch.tensor([[1.0, 0, 0, 0], [0, 0, 1.0, 0], [0, 1.0, 0, 0]]).repeat(n_env, 1, 1),
                fx=torch.full((n_env,), float(eh)), fy=torch.full((n_env,), float(eh)),
                cx=torch.full((n_env,), float(ew // 2)), cy=torch.full((n_env,), float(eh // 2)), camera_type="equirectangular")

        session_to_indices = indices_to_session = None
        if scene not in SCENES_WITHOUT_SESSIONS:
            # a session is a folder of ENV_MAP_CC; an image belongs to every session whose name occurs in its path,
            # sessions are numbered in the order their first image appears (:337-356)
            sessions = [os.path.basename(p) for p in glob.glob(f"{scene_dir}/ENV_MAP_CC/*")]
            by_name: Dict[str, List[int]] = {}
            for idx, fn in enumerate(image_filenames):
                for s in sessions:
                    if s in fn:
                        by_name.setdefault(s, []).append(idx)
            session_to_indices = {i: v for i, v in enumerate(by_name.values())}
            indices_to_session = {}
            for s_idx, members in session_to_indices.items():
                for idx in members:
                    indices_to_session[idx] = s_idx
            if split in ("validation", "test") and len(cfg.session_holdout_indices) != len(session_to_indices):
                raise AssertionError("number of relative eval indicies must match number of unique sessions")

        mask_filenames = None
        semantics = None
        if cfg.mask_source == "original":
            mask_filenames = find_files(f"{split_dir}/mask")
        elif cfg.mask_source == "cityscapes":
            seg_dir = f"{split_dir}/cityscapes_mask"
            if not os.path.exists(seg_dir):
                if not cfg.run_segmentation_inference:
                    raise ValueError(f"Cityscapes segmentation folder {seg_dir} does not exist and run inference is False")
                raise NotImplementedError("Segmentation inference not implemented yet")
            semantics = Semantics(filenames=find_files(seg_dir), classes=CITYSCAPE_CLASSES["classes"],
                                  colors=torch.tensor(CITYSCAPE_CLASSES["colours"], dtype=torch.uint8))

        object_masks: List[Optional[str]] = [None] * len(image_filenames)
        obj_dir = f"{split_dir}/out_of_view_frustum_objects_mask"
        if os.path.exists(obj_dir) and cfg.mask_out_of_view_frustum_objects:
            by_stem = {Path(p).stem: p for p in find_files(obj_dir)}
            object_masks = [by_stem.get(Path(fn).stem) for fn in image_filenames]

        test_eval_mask_dict: Dict[int, str] = {}
        if split == "test" and scene not in SCENES_WITHOUT_SESSIONS:
            stem = lambda p: p.split("/")[-1].split(".")[0]  # noqa: E731  (first dot, as the reference :420-421)
            index_of = {stem(fn): i for i, fn in enumerate(image_filenames)}
            for mp in find_files(f"{split_dir}/mask"):
                if stem(mp) in index_of:
                    test_eval_mask_dict[index_of[stem(mp)]] = mp
            for key, rel in zip(session_to_indices.keys(), cfg.session_holdout_indices):
                held = session_to_indices[key][rel]
                if held in test_eval_mask_dict:
                    raise ValueError(f"Image {held} is both a holdout image and an eval image, update config holdout image indices")

        metadata = {
            "semantics": semantics, "session_to_indices": session_to_indices, "indices_to_session": indices_to_session,
            "session_holdout_indices": cfg.session_holdout_indices, "envmap_filenames": envmap_filenames,
            "envmap_cameras": envmap_cameras, "depth_filenames": None, "normal_filenames": None, "include_mono_prior": False,
            "c2w_colmap": None, "crop_to_equal_size": cfg.crop_to_equal_size, "pad_to_equal_size": cfg.pad_to_equal_size,
            "width_height": self.width_height, "mask_vegetation": cfg.mask_vegetation, "test_eval_mask_dict": test_eval_mask_dict,
            "out_of_view_frustum_objects_masks": object_masks,
            "include_sidewalk_in_ground_mask": cfg.include_sidewalk_in_ground_mask,
        }
        return DataparserOutputs(image_filenames=image_filenames, cameras=cameras, scene_box=_scene_box(cfg.scene_scale),
                                 mask_filenames=mask_filenames, metadata=metadata, dataparser_scale=cfg.scale_factor)


# ------------------------------------------------------------------------------- synthetic (transforms.json)
@dataclass
class CustomNeuskyDataparserConfig:
    """custom_neusky_dataparser.py:127-160"""
    data: Path = Path("path/to/data")
    transforms_filename: str = "transforms.json"
    scene_scale: float = 1.0
    scale_factor: float = 1.0
    orientation_method: str = "vertical"
    center_method: str = "focus"
    auto_scale_poses: bool = True
    mask_vegetation: bool = False
    include_sidewalk_in_ground_mask: bool = True
    center_method_sfm: bool = False
    sfm_outlier_percentile: float = 95.0
    sfm_scale_percentile: float = 50.0
    sfm_target_radius: float = 0.5
    points3d_filename: str = "points3d.ply"

    def setup(self) -> "CustomNeuskyDataparser":
        return CustomNeuskyDataparser(self)


def load_ply_points(path: Path) -> Optional[np.ndarray]:
    """vertex positions [N,3] of a PLY whose vertex record is x,y,z float32 (+ r,g,b uint8 when binary), as the
    reference's numpy reader assumes (:312-352); None when the file cannot be read that way"""
    try:
        with open(path, "rb") as f:
            n_vertices, binary = 0, False
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("element vertex"):
                    n_vertices = int(line.split()[-1])
                if "binary_little_endian" in line:
                    binary = True
                if line == "end_header":
                    break
            if n_vertices == 0:
                return None
            if binary:
                rec = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
                data = np.frombuffer(f.read(n_vertices * rec.itemsize), dtype=rec)
                return np.stack([data["x"], data["y"], data["z"]], axis=-1)
            return np.loadtxt(f, max_rows=n_vertices)[:, :3].astype(np.float32)
    except Exception:
        return None


class CustomNeuskyDataparser:
    """transforms.json (instant-ngp / BlenderNeRF: OpenGL poses) + `<split>/{rgb,cityscapes_mask,<gt layer>}` folders"""

    def __init__(self, config: CustomNeuskyDataparserConfig):
        self.config = config

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        return self._generate_dataparser_outputs(split)

    def _load_transforms(self) -> Dict[str, dict]:
        with open(Path(self.config.data) / self.config.transforms_filename, "r") as f:
            meta = json.load(f)
        glob_k = {k: float(meta[k]) for k in ("fl_x", "fl_y", "cx", "cy")}
        frames = {}
        for fr in meta["frames"]:
            rec = {k: float(fr.get(k, v)) for k, v in glob_k.items()}
            rec["c2w"] = np.asarray(fr["transform_matrix"], dtype=np.float32)
            rec["envmap_name"] = fr.get("envmap_name")
            rec["envmap_rotation"] = fr.get("envmap_rotation")
            frames[fr["file_path"]] = rec
        return frames

    def _split_files(self, split: str, subdir: str) -> List[str]:
        name = "validation" if split == "val" else split
        return find_files(str(Path(self.config.data) / name / subdir))

    def _discover_gt_layers(self, split: str, image_filenames: List[str]) -> Dict[str, List[str]]:
        """`gt_<layer>_filenames` for every layer folder that holds an EXR for EVERY image of the split (:216-262)"""
        name = "validation" if split == "val" else split
        out = {}
        stems = [Path(p).stem for p in image_filenames]
        for layer in GT_LAYER_CHANNELS:
            d = Path(self.config.data) / name / layer
            if not d.is_dir():
                continue
            by_stem = {Path(p).stem: p for p in find_files(str(d), exts=("*.exr", "*.EXR"))}
            if by_stem and all(s in by_stem for s in stems):
                out[f"gt_{layer}_filenames"] = [by_stem[s] for s in stems]
        return out

    def _resolve_gt_envmaps(self, infos: List[dict]) -> List[Optional[dict]]:
        """`<name>.exr` under hdris/ then hdris_16k/, two levels above the scene folder (:264-296)"""
        base = Path(self.config.data).parent.parent
        out: List[Optional[dict]] = []
        for info in infos:
            hit = None
            if info.get("name") is not None:
                for d in (base / "hdris", base / "hdris_16k"):
                    if (d / f"{info['name']}.exr").exists():
                        hit = {"path": str(d / f"{info['name']}.exr"), "rotation": info.get("rotation")}
                        break
            out.append(hit)
        return out

    def _sfm_centre_and_scale(self, pts: np.ndarray) -> Tuple[np.ndarray, float]:
        """robust centre / scale of an SfM cloud (:354-387): keep the closest P% of the points to the median, centre =
        their mean, scale maps the q-th percentile distance to the target radius"""
        cfg = self.config
        d0 = np.linalg.norm(pts - np.median(pts, axis=0), axis=1)
        inl = pts[d0 <= np.percentile(d0, cfg.sfm_outlier_percentile)]
        centre = inl.mean(axis=0)
        target = np.percentile(np.linalg.norm(inl - centre, axis=1), cfg.sfm_scale_percentile)
        return centre, cfg.sfm_target_radius / max(target, 1e-6)

    def _generate_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        cfg = self.config
        frames = self._load_transforms()
        per_split: Dict[str, Dict[str, list]] = {}
        all_c2w, all_k = [], []
        for s in ("train", "val", "test"):
            masks = {Path(p).stem: p for p in self._split_files(s, "cityscapes_mask")}
            rec = {"images": [], "masks": [], "envmaps": []}
            for img in self._split_files(s, "rgb"):
                key = str(Path(img).relative_to(cfg.data))
                if key not in frames:
                    continue
                fd = frames[key]
                all_c2w.append(fd["c2w"])
                all_k.append([fd["fl_x"], fd["fl_y"], fd["cx"], fd["cy"]])
                rec["images"].append(img)
                rec["masks"].append(masks.get(Path(img).stem))
                rec["envmaps"].append({"name": fd["envmap_name"], "rotation": fd["envmap_rotation"]})
            per_split[s] = rec
        counts = {s: len(per_split[s]["images"]) for s in per_split}
        if sum(counts.values()) == 0:
            raise ValueError(f"No images found matching transforms in {cfg.data}. Run scripts/prepare_synthetic_data.py first.")
        c2w = torch.from_numpy(np.stack(all_c2w))
        if cfg.center_method_sfm:
            c2w, transform = auto_orient_and_center_poses(c2w, method=cfg.orientation_method, center_method="none")
            pts = load_ply_points(Path(cfg.data) / cfg.points3d_filename)
            if pts is not None:
                pts = (transform[:3, :3].numpy() @ pts.T).T + transform[:3, 3].numpy()
                centre, scale = self._sfm_centre_and_scale(pts)
                c2w[:, :3, 3] -= torch.from_numpy(centre).float()
                c2w[:, :3, 3] *= scale * cfg.scale_factor
            else:
                c2w[:, 2, 3] -= c2w[:, 2, 3].mean(0)
                if cfg.auto_scale_poses:
                    c2w[:, :3, 3] *= (1.0 / float(torch.max(torch.abs(c2w[:, :3, 3])))) * cfg.scale_factor
        else:
            c2w, transform = _normalise_poses(c2w, cfg.orientation_method, cfg.center_method, cfg.auto_scale_poses, cfg.scale_factor)
        orientation_rotation = transform[:3, :3].clone()

        query = "val" if split in ("val", "validation") else split
        if counts.get(query, 0) == 0:
            query = "train"  # an empty split falls back to the training frames (:507-510)
        order = ["train", "val", "test"]
        lo = sum(counts[s] for s in order[:order.index(query)])
        hi = lo + counts[query]
        k = torch.tensor(all_k[lo:hi], dtype=torch.float32).reshape(-1, 4)
        cameras = Cameras(camera_to_worlds=c2w[lo:hi, :3, :4], fx=k[:, 0], fy=k[:, 1], cx=k[:, 2], cy=k[:, 3])
        image_filenames = per_split[query]["images"]
        seg = [m for m in per_split[query]["masks"] if m is not None]
        semantics = None
        if seg and len(seg) == len(image_filenames):
            semantics = Semantics(filenames=seg, classes=CITYSCAPE_CLASSES["classes"],
                                  colors=torch.tensor(CITYSCAPE_CLASSES["colours"], dtype=torch.uint8))
        metadata = {
            "semantics": semantics, "session_to_indices": None, "indices_to_session": None, "session_holdout_indices": [],
            "envmap_filenames": [], "envmap_cameras": None, "depth_filenames": None, "normal_filenames": None,
            "include_mono_prior": False, "c2w_colmap": None, "crop_to_equal_size": False, "pad_to_equal_size": False,
            "width_height": [], "mask_vegetation": cfg.mask_vegetation, "test_eval_mask_dict": {},
            "out_of_view_frustum_objects_masks": [None] * len(image_filenames),
            "include_sidewalk_in_ground_mask": cfg.include_sidewalk_in_ground_mask,
            "orientation_rotation": orientation_rotation, "gt_envmap_info": self._resolve_gt_envmaps(per_split[query]["envmaps"]),
        }
        metadata.update(self._discover_gt_layers(query, image_filenames))
        return DataparserOutputs(image_filenames=image_filenames, cameras=cameras, scene_box=_scene_box(cfg.scene_scale),
                                 metadata=metadata, dataparser_scale=cfg.scale_factor)


# ------------------------------------------------------------------------------------------------ dataset
def _load_exr(path: str, num_channels: int) -> Optional[np.ndarray]:
    """EXR layers need `pyexr` (as in the reference, neusky_dataset.py:52-64); without it the layer is skipped"""
    try:
        import pyexr  # type: ignore
        img = pyexr.open(path).get()
    except Exception:
        return None
    if img.ndim == 2:
        img = img[:, :, None]
    return img[:, :, :num_channels].astype(np.float32)


class NeuSkyDataset:
    """image + 4-channel mask reader over a DataparserOutputs (neusky_dataset.py:113-344)"""

    def __init__(self, dataparser_outputs: DataparserOutputs, scale_factor: float = 1.0, split: str = "train"):
        self._dataparser_outputs = dataparser_outputs
        self.scale_factor = scale_factor
        self.split = split
        self.scene_box = dataparser_outputs.scene_box
        self.cameras = dataparser_outputs.cameras
        md = self.metadata = dict(dataparser_outputs.metadata)
        self.semantics: Optional[Semantics] = md["semantics"]
        self.crop_to_equal_size, self.pad_to_equal_size = md["crop_to_equal_size"], md["pad_to_equal_size"]
        if self.crop_to_equal_size or self.pad_to_equal_size:
            self.target_width, self.target_height = md["width_height"]
        md["c2w"] = dataparser_outputs.cameras.camera_to_worlds
        if md["session_to_indices"] is not None:
            md["num_sessions"] = len(md["session_to_indices"])
        self.test_eval_mask_dict = md["test_eval_mask_dict"]
        self.object_masks = md["out_of_view_frustum_objects_masks"]
        self.gt_layer_filenames = {n: md[f"gt_{n}_filenames"] for n in GT_LAYER_CHANNELS if md.get(f"gt_{n}_filenames") is not None}

    def __len__(self) -> int:
        return len(self._dataparser_outputs.image_filenames)

    # ---- geometry shared by images and masks
    def _crop_box(self, width: int, height: int) -> Tuple[int, int, int, int]:
        tw, th = self.target_width, self.target_height
        return (max((width - tw) // 2, 0), max((height - th) // 2, 0), min((width + tw) // 2, width), min((height + th) // 2, height))

    def get_numpy_image(self, image_idx: int) -> np.ndarray:
        """uint8 [H,W,3|4]: centre crop OR centre pad (black) to the common size, then bilinear rescale (:151-185)"""
        im = Image.open(self._dataparser_outputs.image_filenames[image_idx])
        if self.crop_to_equal_size:
            im = im.crop(self._crop_box(*im.size))
        if self.pad_to_equal_size:
            w, h = im.size
            canvas = Image.new("RGB", (self.target_width, self.target_height), (0, 0, 0))
            canvas.paste(im, ((self.target_width - w) // 2, (self.target_height - h) // 2))
            im = canvas
        if self.scale_factor != 1.0:
            w, h = im.size
            im = im.resize((int(w * self.scale_factor), int(h * self.scale_factor)), resample=Image.BILINEAR)
        arr = np.array(im, dtype="uint8")
        if arr.ndim == 2:
            arr = arr[:, :, None].repeat(3, axis=2)
        if arr.ndim != 3 or arr.shape[2] not in (3, 4):
            raise AssertionError(f"Image shape of {arr.shape} is in correct.")
        return arr

    def get_image(self, image_idx: int) -> torch.Tensor:
        """float32 [H,W,3] in [0,1]; an alpha channel is composited over white (nerfstudio InputDataset.get_image)"""
        img = torch.from_numpy(self.get_numpy_image(image_idx).astype("float32") / 255.0)
        if img.shape[-1] == 4:
            img = img[:, :, :3] * img[:, :, 3:4] + (1.0 - img[:, :, 3:4])
        return img

    def get_mask_from_semantics(self, idx: int, mask_classes: List[str]) -> torch.Tensor:
        """bool [H,W]: pixels whose palette colour is one of `mask_classes` (:321-338)"""
        seg = np.array(Image.open(self.semantics.filenames[idx]), dtype="int32")[:, :, :3]
        palette = self.semantics.colors.numpy().astype("int32")
        hit = np.zeros(seg.shape[:2], dtype=bool)
        for name in mask_classes:
            hit |= np.all(seg == palette[self.semantics.classes.index(name)], axis=2)
        return torch.from_numpy(hit)

    def get_mask(self, idx: int) -> torch.Tensor:
        """float [H,W,4] = [static (1 = not transient), foreground, ground, sky] (:221-319)"""
        static = None
        if self.split == "test" and idx in self.test_eval_mask_dict:
            m = torch.from_numpy(np.array(Image.open(self.test_eval_mask_dict[idx]), dtype="uint8")).float() / 255.0
            static = (m[:, :, None] if m.ndim == 2 else m[:, :, 0:1]).float()
        transient, fg = list(TRANSIENT_CLASSES), list(FOREGROUND_CLASSES)
        (transient if self.metadata["mask_vegetation"] else fg).append("vegetation")
        if static is None:
            static = (~self.get_mask_from_semantics(idx, transient))[..., None].float()
        fg_mask = self.get_mask_from_semantics(idx, fg)[..., None].float()
        ground_classes = ["road"] + (["sidewalk"] if self.metadata["include_sidewalk_in_ground_mask"] else [])
        ground = self.get_mask_from_semantics(idx, ground_classes)[..., None].float()
        sky = self.get_mask_from_semantics(idx, ["sky"])[..., None].float()
        if self.object_masks[idx] is not None:
            # (sic) the reference divides the uint8 mask by 255 and casts to bool, so any non-zero value counts (:283-287)
            obj = torch.from_numpy(np.array(Image.open(self.object_masks[idx]), dtype="uint8"))[:, :, 0]
            keep = (~(obj / 255.0).bool())[..., None].float()
            static, fg_mask = static * keep, fg_mask * keep
        mask = torch.cat([static, fg_mask, ground, sky], dim=-1)
        if self.crop_to_equal_size:
            h, w = mask.shape[:2]
            left, top, right, bottom = self._crop_box(w, h)
            mask = mask[top:bottom, left:right, :]
        if self.pad_to_equal_size:
            h, w = mask.shape[:2]
            pl, pt = (self.target_width - w) // 2, (self.target_height - h) // 2
            mask = torch.nn.functional.pad(mask.permute(2, 0, 1), (pl, self.target_width - w - pl, pt, self.target_height - h - pt),
                                           mode="constant", value=0).permute(1, 2, 0)
        if self.scale_factor != 1.0:
            h, w = mask.shape[:2]
            size = (int(h * self.scale_factor), int(w * self.scale_factor))
            mask = torch.nn.functional.interpolate(mask.permute(2, 0, 1)[None], size=size, mode="nearest")[0].permute(1, 2, 0)
        return mask

    def get_envmap(self, idx: int) -> torch.Tensor:
        """[3,H,W] float in [0,1] (:340-344)"""
        return torch.from_numpy(np.array(Image.open(self.metadata["envmap_filenames"][idx]), dtype="float32") / 255.0).permute(2, 0, 1)

    def get_metadata(self, data: Dict) -> Dict:
        idx = data["image_idx"]
        out: Dict[str, Any] = {"mask": self.get_mask(idx)}
        for layer, files in self.gt_layer_filenames.items():
            arr = _load_exr(files[idx], GT_LAYER_CHANNELS[layer]) if files[idx] is not None else None
            if arr is None:
                continue
            t = torch.from_numpy(arr)
            if self.scale_factor != 1.0:
                h, w = arr.shape[:2]
                size = (int(h * self.scale_factor), int(w * self.scale_factor))
                t = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None], size=size, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
            out[f"gt_{layer}"] = t
        return out

    def __getitem__(self, image_idx: int) -> Dict:
        data = {"image_idx": image_idx, "image": self.get_image(image_idx)}
        data.update(self.get_metadata(data))
        return data


def load_stacks(dataset: NeuSkyDataset) -> Tuple[torch.Tensor, torch.Tensor]:
    """every image and mask of a dataset as [N,H,W,3] float / [N,H,W,4] bool stacks (all frames must share one size:
    `crop_to_equal_size` or `pad_to_equal_size`, as the reference's `images_on_gpu` path needs, neusky_config.py:50-61)"""
    items = [dataset[i] for i in range(len(dataset))]
    sizes = {tuple(it["image"].shape[:2]) for it in items}
    if len(sizes) != 1:
        raise ValueError(f"frames have different sizes {sorted(sizes)}: enable crop_to_equal_size or pad_to_equal_size")
    return torch.stack([it["image"] for it in items]), torch.stack([it["mask"] for it in items]) > 0.5
