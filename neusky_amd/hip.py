"""ctypes binding of libneusky_hip.so (the C ABI declared in include/neusky_hip.h).

Fails loudly when the shared library is missing: the product path has no fallback.  Tensors are
passed as raw device pointers (`tensor.data_ptr()`), the stream as torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libneusky_hip.so")


class NeuSkyHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
        "neusky_amd has no CPU/PyTorch fallback."
    )
_lib = C.CDLL(LIB_PATH)

# ---- epilogues (keep in sync with include/neusky_hip.h)
EPI_NONE, EPI_RELU, EPI_LEAKY, EPI_SIGMOID, EPI_SOFTPLUS, EPI_FILM, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6
EPI_BWD_RELU, EPI_BWD_LEAKY, EPI_BWD_FILM, EPI_EXP = 7, 8, 9, 10
PREC_F32, PREC_BF16X2, PREC_BF16X3, PREC_F16X2 = 0, 2, 3, 4


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("a_kcontig", C.c_int32), ("b_kcontig", C.c_int32),
        ("bias", C.c_void_p),
        ("epi", C.c_int32), ("p0", C.c_float), ("p1", C.c_float),
        ("aux0", C.c_void_p), ("ldaux0", C.c_int32),
        ("aux1", C.c_void_p), ("ldaux1", C.c_int32),
        ("aux2", C.c_void_p), ("ldaux2", C.c_int32),
        ("out1", C.c_void_p), ("ldout1", C.c_int32),
        ("out2", C.c_void_p), ("ldout2", C.c_int32),
        ("row_mod", C.c_int32), ("k_splits", C.c_int32), ("beta", C.c_float), ("a_rowsum", C.c_void_p), ("precision", C.c_int32),
        ("rowsum_k_limit", C.c_int32),
        ("a_native_nt", C.c_int32), ("b_native_nt", C.c_int32), ("a_scale_max", C.c_void_p),
    ]


_lib.nsky_last_error.restype = C.c_char_p
_lib.nsky_abi_version.restype = C.c_int
ABI_VERSION = 16  # the ctypes structures below mirror this version of include/neusky_hip.h
if _lib.nsky_abi_version() != ABI_VERSION:
    raise NeuSkyHipError(f"libneusky_hip.so has ABI version {_lib.nsky_abi_version()}, this package binds version {ABI_VERSION}: rebuild (build.sh)")


def _sig(name, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = C.c_int
    fn.argtypes = list(argtypes)
    return fn


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NeuSkyHipError(f"{what} failed ({rc}): {_lib.nsky_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device address of a tensor handed to the C ABI.  A host tensor is an error: there is no CPU path behind these entry points."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NeuSkyHipError(f"host tensor {tuple(t.shape)} handed to a HIP entry point: the product path runs on the device only "
                             "(the CPU restatement is oracle/, test infrastructure)")
    return t.data_ptr()


def ld(t):
    """leading dimension (elements) of a 2-D row-major view whose rows are contiguous"""
    if t is None:
        return 0
    st = t.stride()
    if len(st) != 2 or st[1] != 1:
        raise ValueError(f"expected a 2-D view with contiguous rows, got shape {tuple(t.shape)} stride {st}")
    return st[0]


_gemm = _sig("nsky_gemm_f32", C.POINTER(GemmDesc), C.c_void_p)
_colsum = _sig("nsky_colsum_f32", C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)


def gemm(A, B, Cout, M, N, K, *, a_kcontig=True, b_kcontig=True, bias=None, epi=EPI_NONE, p0=0.0, p1=0.0,
         aux0=None, aux1=None, aux2=None, out1=None, out2=None, row_mod=0, k_splits=0, beta=0.0, a_rowsum=None, precision=0,
         rowsum_k_limit=0, a_native_nt=0, b_native_nt=0, a_scale_max=None):
    """C[M,N] = epi(sum_k A(m,k) B(n,k) + bias).  A/B/C are 2-D row-major views (rows contiguous); a_native_nt / b_native_nt:
    that operand is a tile-native matrix (include/neusky_hip.h) with this many 32-feature tiles per row."""
    d = GemmDesc(
        A=ptr(A), B=ptr(B), C=ptr(Cout), M=M, N=N, K=K, lda=ld(A), ldb=ld(B), ldc=ld(Cout),
        a_kcontig=int(a_kcontig), b_kcontig=int(b_kcontig), bias=ptr(bias), epi=epi, p0=p0, p1=p1,
        aux0=ptr(aux0), ldaux0=ld(aux0), aux1=ptr(aux1), ldaux1=ld(aux1), aux2=ptr(aux2), ldaux2=ld(aux2),
        out1=ptr(out1), ldout1=ld(out1), out2=ptr(out2), ldout2=ld(out2), row_mod=row_mod, k_splits=k_splits, beta=beta,
        a_rowsum=ptr(a_rowsum), precision=precision, rowsum_k_limit=rowsum_k_limit,
        a_native_nt=a_native_nt, b_native_nt=b_native_nt, a_scale_max=ptr(a_scale_max),
    )
    check(_gemm(C.byref(d), stream_ptr()), "nsky_gemm_f32")
    return Cout


_split_planes = _sig("nsky_split_planes", C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                     C.c_int32, C.c_int32, C.c_void_p)
_gemm_planes = _sig("nsky_gemm_f32_planes", C.POINTER(GemmDesc), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


def split_planes(W, n_rows, n_k, transpose, precision):
    """two 16-bit planes [2, rows_pad, ldp] (int16 storage) of a weight matrix for gemm_planes; see include/neusky_hip.h"""
    rows_pad, ldp = (n_rows + 255) // 256 * 256, (n_k + 31) // 32 * 32
    planes = torch.empty(2, rows_pad, ldp, dtype=torch.int16, device=W.device)
    check(_split_planes(ptr(W), n_rows, n_k, ld(W), int(transpose), precision, planes[0].data_ptr(), planes[1].data_ptr(),
                        rows_pad, ldp, stream_ptr()), "nsky_split_planes")
    return planes


def gemm_planes(A, planes, Cout, M, N, K, *, precision, bias=None, epi=EPI_NONE, p0=0.0, p1=0.0, aux0=None, aux1=None, aux2=None,
                out1=None, out2=None, row_mod=0, beta=0.0):
    """C[M,N] = epi(sum_k A(m,k) B(n,k) + bias) with B given as the planes of split_planes (LDS-DMA kernel)"""
    d = GemmDesc(
        A=ptr(A), B=None, C=ptr(Cout), M=M, N=N, K=K, lda=ld(A), ldb=0, ldc=ld(Cout), a_kcontig=1, b_kcontig=1, bias=ptr(bias),
        epi=epi, p0=p0, p1=p1, aux0=ptr(aux0), ldaux0=ld(aux0), aux1=ptr(aux1), ldaux1=ld(aux1), aux2=ptr(aux2), ldaux2=ld(aux2),
        out1=ptr(out1), ldout1=ld(out1), out2=ptr(out2), ldout2=ld(out2), row_mod=row_mod, k_splits=0, beta=beta,
        a_rowsum=None, precision=precision,
    )
    check(_gemm_planes(C.byref(d), planes[0].data_ptr(), planes[1].data_ptr(), planes.shape[2], stream_ptr()), "nsky_gemm_f32_planes")
    return Cout


_wcolsum = _sig("nsky_weighted_colsum_f32", C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)


def weighted_colsum(X, M, N, w, w_stride, out):
    """out[c] += sum_r w[r * w_stride] X[r, c]"""
    check(_wcolsum(ptr(X), M, N, ld(X), ptr(w), w_stride, ptr(out), stream_ptr()), "nsky_weighted_colsum_f32")
    return out


def colsum(X, M, N, out):
    check(_colsum(ptr(X), M, N, ld(X), ptr(out), stream_ptr()), "nsky_colsum_f32")
    return out


def abi_version() -> int:
    return _lib.nsky_abi_version()


# ------------------------------------------------------------------------------------------ hash grid
class HashGridDesc(C.Structure):
    _fields_ = [
        ("table", C.c_void_p), ("n_levels", C.c_int32), ("smoothstep", C.c_int32),
        ("scale", C.c_float * 16), ("resolution", C.c_int32 * 16), ("offset", C.c_uint32 * 17),
    ]


MODE_RAW, MODE_CONTRACT_LINF, MODE_CONTRACT_L2 = 0, 1, 2

_encode_fwd = _sig("nsky_encode_fwd", C.POINTER(HashGridDesc), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                   C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)
_encode_bwd = _sig("nsky_encode_bwd", C.POINTER(HashGridDesc), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                   C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_encode_bwd_ws = _lib.nsky_encode_bwd_workspace_bytes
_encode_bwd_ws.restype = C.c_int64
_encode_bwd_ws.argtypes = [C.POINTER(HashGridDesc), C.c_int32, C.c_int32]
_hash_indices = _sig("nsky_hash_indices", C.POINTER(HashGridDesc), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)


def grid_desc(geom, table) -> HashGridDesc:
    """geom: neusky_amd.encoding.HashGridGeometry; table: [n_params, 2] float32 device tensor."""
    assert table.is_cuda and table.dtype == torch.float32 and table.is_contiguous() and table.shape == (geom.n_params, 2)
    d = HashGridDesc(table=table.data_ptr(), n_levels=geom.n_levels, smoothstep=int(geom.smoothstep))
    for i in range(geom.n_levels):
        d.scale[i] = geom.scales[i]
        d.resolution[i] = geom.resolutions[i]
    for i in range(geom.n_levels + 1):
        d.offset[i] = geom.offsets[i]
    return d


def encode_fwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, Y, T=None):
    P = x.shape[0]
    assert x.shape == (P, 3) and x.is_contiguous() and Y.shape[0] == P
    if T is not None:
        assert T.shape == (3, P, Y.shape[1]) and T.is_contiguous() and Y.is_contiguous()
    check(_encode_fwd(C.byref(grid_desc(geom, table)), ptr(x), P, mode, int(include_x), pe_freqs, pe_max_exp, ptr(Y), ld(Y),
                      ptr(T), stream_ptr()), "nsky_encode_fwd")
    return Y


def encode_bwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, dY, dT, dtable, dx=None, workspace="auto"):
    """workspace: "auto" allocates the scratch nsky_encode_bwd_workspace_bytes asks for (large batches: the fine levels' gradient
    is then accumulated by chunk-owning workgroups in LDS); None: the direct global-atomic scatter whatever the batch"""
    P = x.shape[0]
    if dT is not None:
        assert dT.shape == (3, P, dY.shape[1]) and dT.is_contiguous() and dY.is_contiguous()
    assert dtable is None or (dtable.shape == table.shape and dtable.is_contiguous())
    desc = grid_desc(geom, table)
    if isinstance(workspace, str):
        nbytes = _encode_bwd_ws(C.byref(desc), P, int(dT is not None)) if dtable is not None else 0
        workspace = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device) if nbytes > 0 else None
    check(_encode_bwd(C.byref(desc), ptr(x), P, mode, int(include_x), pe_freqs, pe_max_exp, ptr(dY), ld(dY),
                      ptr(dT), ptr(dtable), ptr(dx), ptr(workspace), stream_ptr()), "nsky_encode_bwd")


def hash_indices(geom, table, x, mode):
    P = x.shape[0]
    out = torch.empty(P, geom.n_levels, 8, dtype=torch.int32, device=x.device)
    check(_hash_indices(C.byref(grid_desc(geom, table)), ptr(x), P, mode, ptr(out), stream_ptr()), "nsky_hash_indices")
    return out


# ------------------------------------------------------------------------------------------ sample generators
_ddf_vmf_samples = _sig("nsky_ddf_vmf_samples", C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p)


def ddf_vmf_samples(n_positions, n_directions, kappa, radius, upper_hemisphere, seed, counter, origins, directions):
    """counter: int64 device tensor [1] (advanced by one); origins / directions: [n_positions * n_directions, 3] float32"""
    n = n_positions * n_directions
    assert counter.dtype == torch.int64 and counter.numel() == 1 and counter.is_cuda
    assert origins.shape == (n, 3) and directions.shape == (n, 3) and origins.is_contiguous() and directions.is_contiguous()
    check(_ddf_vmf_samples(n_positions, n_directions, float(kappa), float(radius), int(upper_hemisphere), int(seed) & (2**64 - 1),
                           ptr(counter), ptr(origins), ptr(directions), stream_ptr()), "nsky_ddf_vmf_samples")


_illum_dirs = _sig("nsky_illumination_directions", C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p)


def illumination_directions(base, rotation, seed, counter, dirs, sel, rot_out=None):
    """base [D,3] -> dirs [D,3] = base R^T, sel [D/2] int32 (ascending indices of the upper half); R = rotation [3,3] or drawn from
    (seed, counter) when rotation is None (counter: int64 device tensor [1], advanced by one)"""
    D = base.shape[0]
    assert base.is_contiguous() and dirs.is_contiguous() and dirs.shape == (D, 3) and sel.dtype == torch.int32 and sel.numel() == D // 2
    assert rotation is None or (rotation.is_contiguous() and rotation.shape == (3, 3) and rotation.dtype == torch.float32)
    assert rotation is not None or (counter.dtype == torch.int64 and counter.numel() == 1)
    check(_illum_dirs(ptr(base), D, ptr(rotation), int(seed) & (2**64 - 1), ptr(counter), ptr(dirs), ptr(sel), ptr(rot_out), stream_ptr()),
          "nsky_illumination_directions")


_fit_rows_fwd = _sig("nsky_ddf_fit_rows_fwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_void_p)
_fit_rows_bwd = _sig("nsky_ddf_fit_rows_bwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                     C.c_void_p)


def ddf_fit_rows_fwd(positions, directions, term_dist, mv_points_in, seed, counter, sky_o, sky_d, radius, want_mv, weight_exp,
                     weight_include_z, q_pos, xrow, mv_points_out, sky_gt, distance_weight):
    """rows of the DDF-fit evaluations (fit rays | multi-view | sky) into q_pos [E,3] / xrow [E,ld]; see include/neusky_hip.h"""
    N, Ns = positions.shape[0], (0 if sky_o is None else sky_o.shape[0])
    E = N + (N if want_mv else 0) + Ns
    assert q_pos.shape[0] == E and xrow.shape[0] == E and q_pos.stride() == (3, 1)
    for t in (positions, directions, term_dist, mv_points_in, sky_o, sky_d, mv_points_out, sky_gt, distance_weight):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    check(_fit_rows_fwd(ptr(positions), ptr(directions), ptr(term_dist), N, ptr(mv_points_in), int(seed) & (2**64 - 1), ptr(counter),
                        ptr(sky_o), ptr(sky_d), Ns, float(radius), int(want_mv), float(weight_exp), int(weight_include_z), ptr(q_pos),
                        ptr(xrow), ld(xrow), ptr(mv_points_out), ptr(sky_gt), ptr(distance_weight), stream_ptr()), "nsky_ddf_fit_rows_fwd")


def ddf_fit_rows_bwd(positions, directions, term_dist, mv_points, d_xrow_mv, d_term_dist):
    N = positions.shape[0]
    assert d_xrow_mv.shape[0] == N and d_term_dist.numel() == N
    check(_fit_rows_bwd(ptr(positions), ptr(directions), ptr(term_dist), ptr(mv_points), N, ptr(d_xrow_mv), ld(d_xrow_mv), ptr(d_term_dist),
                        stream_ptr()), "nsky_ddf_fit_rows_bwd")


_reni_in_fwd = _sig("nsky_reni_grid_inputs_fwd", C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                    C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)
_reni_in_bwd = _sig("nsky_reni_grid_inputs_bwd", C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                    C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)


def reni_grid_inputs_fwd(latents, directions, ray_dirs, ray_latent, cond, xrow):
    U, L, _ = latents.shape
    D = directions.shape[0]
    R = 0 if ray_dirs is None else ray_dirs.shape[0]
    assert latents.is_contiguous() and directions.is_contiguous() and cond.shape[0] == U * D + R and xrow.shape[0] == U * D + R
    assert ray_dirs is None or (ray_dirs.is_contiguous() and ray_latent.dtype == torch.int64 and ray_latent.is_contiguous())
    check(_reni_in_fwd(ptr(latents), ptr(directions), U, L, D, ptr(ray_dirs), ptr(ray_latent), R, ptr(cond), ld(cond), ptr(xrow), ld(xrow),
                       stream_ptr()), "nsky_reni_grid_inputs_fwd")


def reni_grid_inputs_bwd(latents, directions, ray_dirs, ray_latent, d_cond, d_latents):
    U, L, _ = latents.shape
    D = directions.shape[0]
    R = 0 if ray_dirs is None else ray_dirs.shape[0]
    assert d_cond.shape[0] == U * D + R and d_latents.is_contiguous() and d_latents.shape == latents.shape
    check(_reni_in_bwd(ptr(latents), ptr(directions), U, L, D, ptr(ray_dirs), ptr(ray_latent), R, ptr(d_cond), ld(d_cond), ptr(d_latents),
                       stream_ptr()), "nsky_reni_grid_inputs_bwd")


_grid_probe = _sig("nsky_grid_probe_points", C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def grid_probe_points(lattice, gap, seed, counter, positions, directions):
    """lattice [P,3] (device), gap: three python floats; positions / directions [P,3] out; counter: int64 device tensor [1] (advanced)"""
    P = lattice.shape[0]
    assert lattice.is_contiguous() and positions.is_contiguous() and directions.is_contiguous() and positions.shape == (P, 3)
    assert counter.dtype == torch.int64 and counter.numel() == 1 and counter.is_cuda
    g3 = (C.c_float * 3)(*[float(v) for v in gap])
    check(_grid_probe(ptr(lattice), g3, P, int(seed) & (2**64 - 1), ptr(counter), ptr(positions), ptr(directions), stream_ptr()),
          "nsky_grid_probe_points")


_reni_out_fwd = _sig("nsky_reni_output_fwd", C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                     C.c_void_p)
_reni_out_bwd = _sig("nsky_reni_output_bwd", C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_void_p)


def reni_output_fwd(raw, scale, ray_latent, U, D, grid, rays):
    R = 0 if rays is None else rays.shape[0]
    assert raw.shape[0] == U * D + R and raw.stride(1) == 1 and scale.is_contiguous() and scale.numel() >= U
    assert grid.is_contiguous() and (rays is None or (rays.is_contiguous() and ray_latent.dtype == torch.int64 and ray_latent.is_contiguous()))
    check(_reni_out_fwd(ptr(raw), ld(raw), ptr(scale), ptr(ray_latent), U, D, R, ptr(grid), ptr(rays), stream_ptr()), "nsky_reni_output_fwd")


def reni_output_bwd(raw, scale, ray_latent, U, D, R, d_grid, d_rays, d_raw, d_scale):
    assert d_raw.shape == raw.shape and d_raw.is_contiguous() and raw.is_contiguous()
    assert (d_grid is None or d_grid.is_contiguous()) and (d_rays is None or d_rays.is_contiguous())
    check(_reni_out_bwd(ptr(raw), ld(raw), ptr(scale), ptr(ray_latent), U, D, R, ptr(d_grid), ptr(d_rays), ptr(d_raw), ptr(d_scale), stream_ptr()),
          "nsky_reni_output_bwd")


# ------------------------------------------------------------------------------------------ render stages
_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_hemi_fwd = _sig("nsky_hemi_composite_fwd", _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P)
_hemi_bwd = _sig("nsky_hemi_composite_bwd", _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P)
_neus_fwd = _sig("nsky_neus_weights_fwd", _P, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P)
_neus_bwd = _sig("nsky_neus_weights_bwd", _P, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P)
_vis_rays = _sig("nsky_visibility_rays", _P, _P, _P, _P, _I, _I, _F, _P, _P, _I, _P, _P, _P)
_vis_fin_fwd = _sig("nsky_visibility_finish_fwd", _P, _P, _P, _F, _P, _I, _I, _I, _P, _P)
_vis_fin_bwd = _sig("nsky_visibility_finish_bwd", _P, _P, _P, _F, _P, _I, _I, _I, _P, _P, _P, _P)


def _c(t):
    assert t is None or t.is_contiguous(), "contiguous tensor required"
    return ptr(t)


def hemi_composite_fwd(albedo, normals, weights, dirs, cam_colours, cam_of_ray, vis, bg, rgb, lin):
    R, S, _ = albedo.shape
    D = dirs.shape[0]
    assert cam_of_ray.dtype == torch.int32
    check(_hemi_fwd(_c(albedo), _c(normals), _c(weights), _c(dirs), _c(cam_colours), _c(cam_of_ray), _c(vis), _c(bg), R, S, D,
                    _c(rgb), _c(lin), stream_ptr()), "nsky_hemi_composite_fwd")


def hemi_composite_bwd(albedo, normals, weights, dirs, cam_colours, cam_of_ray, vis, bg, lin, d_rgb, d_albedo, d_normals,
                       d_weights, d_cam_colours, d_vis, d_bg):
    R, S, _ = albedo.shape
    D = dirs.shape[0]
    check(_hemi_bwd(_c(albedo), _c(normals), _c(weights), _c(dirs), _c(cam_colours), _c(cam_of_ray), _c(vis), _c(bg), _c(lin),
                    _c(d_rgb), R, S, D, _c(d_albedo), _c(d_normals), _c(d_weights), _c(d_cam_colours), _c(d_vis), _c(d_bg),
                    stream_ptr()), "nsky_hemi_composite_bwd")


def neus_weights_fwd(sdf, grad, ray_dirs, starts, ends, variance, anneal, alpha, weights, trans_bg, acc, depth):
    R, S = sdf.shape
    check(_neus_fwd(_c(sdf), _c(grad), _c(ray_dirs), _c(starts), _c(ends), _c(variance), anneal, R, S, _c(alpha), _c(weights),
                    _c(trans_bg), _c(acc), _c(depth), stream_ptr()), "nsky_neus_weights_fwd")


def neus_weights_bwd(sdf, grad, ray_dirs, starts, ends, variance, anneal, d_weights, d_trans_bg, d_sdf, d_grad, d_variance):
    R, S = sdf.shape
    check(_neus_bwd(_c(sdf), _c(grad), _c(ray_dirs), _c(starts), _c(ends), _c(variance), anneal, R, S, _c(d_weights),
                    _c(d_trans_bg), _c(d_sdf), _c(d_grad), _c(d_variance), stream_ptr()), "nsky_neus_weights_bwd")


_ray_reduce_fwd = _sig("nsky_ray_reduce_fwd", _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P)
_ray_reduce_bwd = _sig("nsky_ray_reduce_bwd", _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P)
_normalize3_fwd = _sig("nsky_normalize3_fwd", _P, C.c_int64, _P, _P)
_normalize3_bwd = _sig("nsky_normalize3_bwd", _P, _P, C.c_int64, _P, _P)


def ray_reduce_fwd(weights, starts, ends, normals, albedo, max_clamp, sums, bounds, p2p, accumulation, normal, albedo_acc):
    R, S = weights.shape
    check(_ray_reduce_fwd(_c(weights), _c(starts), _c(ends), _c(normals), _c(albedo), R, S, float(max_clamp), _c(sums), _c(bounds), _c(p2p),
                          _c(accumulation), _c(normal), _c(albedo_acc), stream_ptr()), "nsky_ray_reduce_fwd")


def ray_reduce_bwd(weights, starts, ends, normals, albedo, sums, bounds, max_clamp, d_p2p, d_accumulation, d_normal, d_albedo_acc,
                   d_weights, d_normals, d_albedo):
    R, S = weights.shape
    check(_ray_reduce_bwd(_c(weights), _c(starts), _c(ends), _c(normals), _c(albedo), _c(sums), _c(bounds), R, S, float(max_clamp),
                          _c(d_p2p), _c(d_accumulation), _c(d_normal), _c(d_albedo_acc), _c(d_weights), _c(d_normals), _c(d_albedo),
                          stream_ptr()), "nsky_ray_reduce_bwd")


def normalize3_fwd(g, n):
    check(_normalize3_fwd(_c(g), g.numel() // 3, _c(n), stream_ptr()), "nsky_normalize3_fwd")


def normalize3_bwd(g, d_n, d_g):
    check(_normalize3_bwd(_c(g), _c(d_n), g.numel() // 3, _c(d_g), stream_ptr()), "nsky_normalize3_bwd")


def visibility_rays(origins, ray_dirs, depth, sel_dirs, radius, sphere_pts, xrow, surf_dist, term_dist=None):
    R, Dv = origins.shape[0], sel_dirs.shape[0]
    check(_vis_rays(_c(origins), _c(ray_dirs), _c(depth), _c(sel_dirs), R, Dv, radius, _c(sphere_pts), _c(xrow), ld(xrow),
                    _c(surf_dist), _c(term_dist), stream_ptr()), "nsky_visibility_rays")


def visibility_finish_fwd(t_hat, surf_dist, threshold, scale, sel_index, R, Dv, D, vis):
    assert sel_index.dtype == torch.int32
    check(_vis_fin_fwd(_c(t_hat), _c(surf_dist), _c(threshold), scale, _c(sel_index), R, Dv, D, _c(vis), stream_ptr()),
          "nsky_visibility_finish_fwd")


def visibility_finish_bwd(t_hat, surf_dist, threshold, scale, sel_index, R, Dv, D, d_vis, d_t_hat, d_threshold):
    check(_vis_fin_bwd(_c(t_hat), _c(surf_dist), _c(threshold), scale, _c(sel_index), R, Dv, D, _c(d_vis), _c(d_t_hat),
                       _c(d_threshold), stream_ptr()), "nsky_visibility_finish_bwd")


# ------------------------------------------------------------------------------------------ elementwise helpers
_sp_tan_bwd = _sig("nsky_softplus_tangent_bwd", _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P)
_pdf_sample = _sig("nsky_pdf_sample", _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P)
_adam = _sig("nsky_adam_step", _P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, _I, _F, _P)
_wn_fwd = _sig("nsky_weight_norm_fwd", _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P)
_wn_bwd = _sig("nsky_weight_norm_bwd", _P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P)


def softplus_tangent_bwd(da, s, ta, dta, ggrad, wvec, beta, N, Cc, dz, du, wsum=None):
    """wsum [Cc] (outer-product form): += sum_{n,k} ggrad[n,k] ta_k[n,:], the gradient of wvec, from the same pass"""
    check(_sp_tan_bwd(ptr(da), ptr(s), ptr(ta), ptr(dta), ptr(ggrad), ptr(wvec), beta, N, Cc, ld(s), ptr(dz), ptr(du), ptr(wsum),
                      stream_ptr()), "nsky_softplus_tangent_bwd")


def pdf_sample(weights, bins, u_base, jitter, num_bins, histogram_padding=0.01, eps=1e-5, want_inds=False):
    R, n0 = weights.shape
    assert bins.shape == (R, n0 + 1) and u_base.shape == (num_bins,)
    new_bins = torch.empty(R, num_bins, device=weights.device)
    inds = torch.empty(R, num_bins, dtype=torch.int32, device=weights.device) if want_inds else None
    check(_pdf_sample(_c(weights), _c(bins), _c(u_base), _c(jitter), R, n0, num_bins, histogram_padding, eps, _c(new_bins),
                      _c(inds), stream_ptr()), "nsky_pdf_sample")
    return new_bins, inds


_il_fwd = _sig("nsky_interlevel_fwd", _P, _P, _P, _P, _I, _I, _I, _P, _P)
_il_bwd = _sig("nsky_interlevel_bwd", _P, _P, _P, _P, _P, _I, _I, _I, _P, _P)


def interlevel_fwd(c, w, sb, wp, per_ray):
    R, S = w.shape
    check(_il_fwd(_c(c), _c(w), _c(sb), _c(wp), R, S, wp.shape[1], _c(per_ray), stream_ptr()), "nsky_interlevel_fwd")
    return per_ray


def interlevel_bwd(c, w, sb, wp, d_per_ray, d_wp):
    R, S = w.shape
    check(_il_bwd(_c(c), _c(w), _c(sb), _c(wp), _c(d_per_ray), R, S, wp.shape[1], _c(d_wp), stream_ptr()), "nsky_interlevel_bwd")
    return d_wp


_collider = _sig("nsky_sphere_collider", _P, _P, _I, _F, _F, _P, _P, _P)
_ubins = _sig("nsky_uniform_bins", _P, _P, _P, _I, _I, _P, _P, _P)
_b2s = _sig("nsky_bins_to_samples", _P, _P, _P, _P, _P, _I, _I, _P, _P, _P)


def sphere_collider(origins, directions, radius, near_plane):
    """origins, directions [R,3] -> nears, fars [R,1] (nerfstudio SphereCollider)"""
    R = origins.shape[0]
    nears = torch.empty(R, 1, device=origins.device)
    fars = torch.empty(R, 1, device=origins.device)
    check(_collider(_c(origins), _c(directions), R, radius, near_plane, ptr(nears), ptr(fars), stream_ptr()), "nsky_sphere_collider")
    return nears, fars


def uniform_bins(nears, fars, n, jitter):
    """-> sbins, ebins [R,n+1] (UniformSampler, single jitter per ray or None)"""
    R = nears.shape[0]
    sbins = torch.empty(R, n + 1, device=nears.device)
    ebins = torch.empty(R, n + 1, device=nears.device)
    check(_ubins(_c(nears), _c(fars), _c(jitter), R, n, ptr(sbins), ptr(ebins), stream_ptr()), "nsky_uniform_bins")
    return sbins, ebins


def bins_to_samples(sbins, nears, fars, origins=None, directions=None, want_ebins=True, want_positions=False):
    """sbins [R,n+1] -> (ebins [R,n+1] | None, positions [R,n,3] | None)"""
    R, n = sbins.shape[0], sbins.shape[1] - 1
    ebins = torch.empty(R, n + 1, device=sbins.device) if want_ebins else None
    pos = torch.empty(R, n, 3, device=sbins.device) if want_positions else None
    check(_b2s(_c(sbins), _c(nears), _c(fars), _c(origins), _c(directions), R, n, ptr(ebins), ptr(pos), stream_ptr()),
          "nsky_bins_to_samples")
    return ebins, pos


_dw_fwd = _sig("nsky_density_weights_fwd", _P, _I, _P, _I, _I, _P, _P)
_dw_bwd = _sig("nsky_density_weights_bwd", _P, _I, _P, _P, _I, _I, _P, _P)


def density_weights_fwd(raw, ebins, weights):
    """raw [R*n, ld] (column 0 = density-head output), ebins [R,n+1] -> weights [R,n]"""
    R, n = weights.shape
    check(_dw_fwd(ptr(raw), ld(raw), _c(ebins), R, n, _c(weights), stream_ptr()), "nsky_density_weights_fwd")
    return weights


def density_weights_bwd(raw, ebins, d_weights, d_raw):
    R, n = d_weights.shape
    check(_dw_bwd(ptr(raw), ld(raw), _c(ebins), _c(d_weights), R, n, _c(d_raw), stream_ptr()), "nsky_density_weights_bwd")
    return d_raw


def weight_norm_fwd(v, g, row_map, col_map, out, inv_norm):
    """out[r,c] = g[sr] v[sr,sc] / ||v[sr]|| under the row / column maps (int32, -1 = zero); see include/neusky_hip.h"""
    check(_wn_fwd(ptr(v), ptr(g), row_map.numel(), col_map.numel(), v.shape[1], ld(v), ptr(row_map), ptr(col_map), ptr(out),
                  ld(out), ptr(inv_norm), stream_ptr()), "nsky_weight_norm_fwd")
    return out


def weight_norm_bwd(d_out, v, g, inv_norm, row_map, inverse_col, dv, dg):
    check(_wn_bwd(ptr(d_out), ld(d_out), ptr(v), ptr(g), ptr(inv_norm), row_map.numel(), v.shape[1], ld(v), ptr(row_map),
                  ptr(inverse_col), ptr(dv), ptr(dg), stream_ptr()), "nsky_weight_norm_bwd")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    n = p.numel()
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    check(_adam(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, step, grad_scale, stream_ptr()), "nsky_adam_step")


# ------------------------------------------------------------------------------------------ attention core (RENI++ transformer decoder)
_attn_fwd = _sig("nsky_attn_core_fwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_attn_bwd = _sig("nsky_attn_core_bwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                 C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_add_ln_fwd = _sig("nsky_add_layer_norm_fwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p)
_add_ln_bwd = _sig("nsky_add_layer_norm_bwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p)


def add_layer_norm_fwd(x, r, gamma, beta, eps, s, y, stats):
    """s = x + r (r None: s = x, not stored), y = LayerNorm(s), stats [M, 2] = (mean, rstd); x, r, s, y [M, W] contiguous"""
    M, W = x.shape
    assert all(t is None or t.is_contiguous() for t in (x, r, gamma, beta, s, y, stats))
    check(_add_ln_fwd(ptr(x), ptr(r), ptr(gamma), ptr(beta), M, W, float(eps), ptr(s), ptr(y), ptr(stats), stream_ptr()), "nsky_add_layer_norm_fwd")


def add_layer_norm_bwd(s, stats, gamma, dy, ds_in, ds):
    M, W = s.shape
    assert all(t is None or t.is_contiguous() for t in (s, stats, gamma, dy, ds_in, ds))
    check(_add_ln_bwd(ptr(s), ptr(stats), ptr(gamma), ptr(dy), ptr(ds_in), M, W, ptr(ds), stream_ptr()), "nsky_add_layer_norm_bwd")


_attn_rays_fwd = _sig("nsky_attn_core_rays_fwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                      C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_attn_rays_bwd = _sig("nsky_attn_core_rays_bwd", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def attn_core_rays_fwd(Q, dirs, perm, seg, Kt, Vt, scale, O, row_max, row_sum):
    """the rays' rows, sorted by camera (perm [R] int32, seg [U + 1] int32): Q / O [R, 16 nh], dirs [R, 3], row_max / row_sum [R, nh]"""
    R, H = Q.shape
    U, nh, L = Kt.shape[0], Kt.shape[1], Kt.shape[2]
    assert H == 16 * nh and perm.dtype == torch.int32 and seg.dtype == torch.int32 and seg.numel() == U + 1
    assert all(t.is_contiguous() for t in (Q, dirs, perm, seg, Kt, Vt, O, row_max, row_sum))
    check(_attn_rays_fwd(ptr(Q), ptr(dirs), ptr(perm), ptr(seg), ptr(Kt), ptr(Vt), U, R, L, nh, float(scale), ptr(O), ptr(row_max), ptr(row_sum),
                         stream_ptr()), "nsky_attn_core_rays_fwd")
    return O


def attn_core_rays_bwd(Q, dirs, perm, seg, Kt, Vt, O, row_max, row_sum, dO, scale, dQ, dKt, dVt):
    """adds the rays' part to dKt / dVt (call after attn_core_bwd on the same stream)"""
    R, H = Q.shape
    U, nh, L = Kt.shape[0], Kt.shape[1], Kt.shape[2]
    assert all(t.is_contiguous() for t in (Q, dirs, perm, seg, Kt, Vt, O, row_max, row_sum, dO, dQ, dKt, dVt))
    check(_attn_rays_bwd(ptr(Q), ptr(dirs), ptr(perm), ptr(seg), ptr(Kt), ptr(Vt), ptr(O), ptr(row_max), ptr(row_sum), ptr(dO), U, R, L, nh,
                         float(scale), ptr(dQ), ptr(dKt), ptr(dVt), stream_ptr()), "nsky_attn_core_rays_bwd")


def attn_core_fwd(Q, dirs, Kt, Vt, scale, O, row_max, row_sum):
    """Q [U,D,16 nh], dirs [U,D,3], Kt / Vt [U,nh,L,48] -> O [U,D,16 nh], row_max / row_sum [U,nh,D] (include/neusky_hip.h)"""
    U, D, H = Q.shape
    nh, L = Kt.shape[1], Kt.shape[2]
    assert H == 16 * nh and Kt.shape[3] == 48 and all(t.is_contiguous() for t in (Q, dirs, Kt, Vt, O, row_max, row_sum))
    check(_attn_fwd(ptr(Q), ptr(dirs), ptr(Kt), ptr(Vt), U, D, L, nh, float(scale), ptr(O), ptr(row_max), ptr(row_sum), stream_ptr()),
          "nsky_attn_core_fwd")
    return O


def attn_core_bwd(Q, dirs, Kt, Vt, O, row_max, row_sum, dO, scale, dQ, dKt, dVt):
    U, D, H = Q.shape
    nh, L = Kt.shape[1], Kt.shape[2]
    assert all(t.is_contiguous() for t in (Q, dirs, Kt, Vt, O, row_max, row_sum, dO, dQ, dKt, dVt))
    # D = dO . O per row and head (from the row kernel to the token kernel), then four floats per (camera, head): the operand maxima
    drow = torch.empty(U * nh * (D + 4), device=Q.device)
    check(_attn_bwd(ptr(Q), ptr(dirs), ptr(Kt), ptr(Vt), ptr(O), ptr(row_max), ptr(row_sum), ptr(dO), U, D, L, nh, float(scale),
                    ptr(dQ), ptr(dKt), ptr(dVt), ptr(drow), stream_ptr()), "nsky_attn_core_bwd")


# ------------------------------------------------------------------------------------------ fused loss terms
TOTAL_MAX_SEGMENTS = 8


class TotalSegment(C.Structure):
    _fields_ = [("x", C.c_void_p), ("coef", C.c_void_p), ("grad", C.c_void_p), ("n", C.c_int32), ("scale", C.c_float)]


_total_fwd = _sig("nsky_weighted_total_fwd", C.POINTER(TotalSegment), C.c_int32, C.c_void_p, C.c_void_p)
_total_bwd = _sig("nsky_weighted_total_bwd", C.POINTER(TotalSegment), C.c_int32, C.c_void_p, C.c_void_p)


def weighted_total_fwd(parts, total):
    """parts: [(x, coef | None, scale)] -> total[0] = sum_s scale_s sum_i coef_s[i] x_s[i] (one launch; include/neusky_hip.h)"""
    assert 0 < len(parts) <= TOTAL_MAX_SEGMENTS
    arr = (TotalSegment * len(parts))(*[TotalSegment(ptr(x), ptr(c), None, x.numel(), float(sc)) for x, c, sc in parts])
    check(_total_fwd(arr, len(parts), ptr(total), stream_ptr()), "nsky_weighted_total_fwd")
    return total


def weighted_total_bwd(parts, g, grads):
    """grads[s] (or None) = g[0] scale_s coef_s"""
    arr = (TotalSegment * len(parts))(*[TotalSegment(ptr(x), ptr(c), ptr(d), x.numel(), float(sc)) for (x, c, sc), d in zip(parts, grads)])
    check(_total_bwd(arr, len(parts), ptr(g), stream_ptr()), "nsky_weighted_total_bwd")


class MainLossesDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("P", C.c_int32), ("M", C.c_int32),
                ("rgb", C.c_void_p), ("image", C.c_void_p), ("mask", C.c_void_p), ("eik", C.c_void_p), ("weights", C.c_void_p),
                ("normal", C.c_void_p), ("hdr_bg", C.c_void_p), ("grid", C.c_void_p), ("sdf_term", C.c_void_p), ("vis_thr", C.c_void_p),
                ("sky_alpha", C.c_float), ("vis_target", C.c_float)]


class DDFLossesDesc(C.Structure):
    _fields_ = [("Mr", C.c_int32), ("Mm", C.c_int32), ("Ms", C.c_int32),
                ("expected", C.c_void_p), ("term", C.c_void_p), ("mask", C.c_void_p), ("dist_weight", C.c_void_p), ("sdf", C.c_void_p),
                ("mv_expected", C.c_void_p), ("mv_term", C.c_void_p), ("sky_expected", C.c_void_p), ("sky_term", C.c_void_p),
                ("want_depth", C.c_int32), ("want_sdf_l2", C.c_int32), ("want_sdf_l1", C.c_int32), ("mask_to_circumference", C.c_int32),
                ("inverse_depth_weight", C.c_int32), ("radius", C.c_float)]


_main_losses_fwd = _sig("nsky_main_losses_fwd", C.POINTER(MainLossesDesc), C.c_void_p, C.c_void_p, C.c_void_p)
_main_losses_bwd = _sig("nsky_main_losses_bwd", C.POINTER(MainLossesDesc), *([C.c_void_p] * 11))
_ddf_losses_fwd = _sig("nsky_ddf_losses_fwd", C.POINTER(DDFLossesDesc), C.c_void_p, C.c_void_p)
_ddf_losses_bwd = _sig("nsky_ddf_losses_bwd", C.POINTER(DDFLossesDesc), *([C.c_void_p] * 8))
N_MAIN_TERMS, N_DDF_TERMS = 8, 5


def main_losses_fwd(desc: MainLossesDesc, terms, wsum):
    check(_main_losses_fwd(C.byref(desc), ptr(terms), ptr(wsum), stream_ptr()), "nsky_main_losses_fwd")


def main_losses_bwd(desc: MainLossesDesc, wsum, d_terms, d_rgb, d_eik, d_weights, d_normal, d_hdr_bg, d_grid, d_sdf_term, d_vis_thr):
    check(_main_losses_bwd(C.byref(desc), ptr(wsum), ptr(d_terms), ptr(d_rgb), ptr(d_eik), ptr(d_weights), ptr(d_normal), ptr(d_hdr_bg),
                           ptr(d_grid), ptr(d_sdf_term), ptr(d_vis_thr), stream_ptr()), "nsky_main_losses_bwd")


def ddf_losses_fwd(desc: DDFLossesDesc, terms):
    check(_ddf_losses_fwd(C.byref(desc), ptr(terms), stream_ptr()), "nsky_ddf_losses_fwd")


def ddf_losses_bwd(desc: DDFLossesDesc, d_terms, d_expected, d_sdf, d_mv_expected, d_sky_expected, d_term, d_mv_term):
    check(_ddf_losses_bwd(C.byref(desc), ptr(d_terms), ptr(d_expected), ptr(d_sdf), ptr(d_mv_expected), ptr(d_sky_expected), ptr(d_term),
                          ptr(d_mv_term), stream_ptr()), "nsky_ddf_losses_bwd")


# ------------------------------------------------------------------------------------------ fused FiLM-SIREN chain
FILM_MAX_LAYERS = 12
FILM_TABLE_FLOATS = 6656  # biases + reciprocal tile scales written by nsky_film_pack


class FilmNet(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("n_map", C.c_int32), ("n_film", C.c_int32),
                ("cond_dim", C.c_int32), ("x_dim", C.c_int32), ("out_dim", C.c_int32),
                ("map_w", C.c_void_p * FILM_MAX_LAYERS), ("map_b", C.c_void_p * FILM_MAX_LAYERS), ("map_ld", C.c_int32 * FILM_MAX_LAYERS),
                ("mo_w", C.c_void_p), ("mo_b", C.c_void_p), ("mo_ld", C.c_int32),
                ("film_w", C.c_void_p * FILM_MAX_LAYERS), ("film_b", C.c_void_p * FILM_MAX_LAYERS), ("film_ld", C.c_int32 * FILM_MAX_LAYERS),
                ("out_w", C.c_void_p), ("out_b", C.c_void_p), ("out_ld", C.c_int32)]


_film_layout = _sig("nsky_film_stream_layout", C.POINTER(FilmNet), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32))
_film_pack = _sig("nsky_film_pack", C.POINTER(FilmNet), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_film_fwd = _sig("nsky_film_chain_fwd", C.POINTER(FilmNet), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


_film_bwd_film = _sig("nsky_film_chain_bwd_film", C.POINTER(FilmNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
_film_bwd_map = _sig("nsky_film_chain_bwd_map", C.POINTER(FilmNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)


_wgrad_native = _sig("nsky_wgrad_native", C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                     C.c_void_p, C.c_float, C.c_void_p)


class WgradProblem(C.Structure):
    _fields_ = [("dZ", C.c_void_p), ("nnt_a", C.c_int32), ("X", C.c_void_p), ("nnt_b", C.c_int32), ("dW", C.c_void_p), ("ldw", C.c_int32),
                ("db", C.c_void_p), ("a_scale_max", C.c_void_p), ("b_scale", C.c_float),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("width_a", C.c_int32), ("width_b", C.c_int32), ("bias_rows", C.c_int32),
                ("bias_row_mod", C.c_int32), ("b_scale_max", C.c_void_p)]


WGRAD_MAX_PROBLEMS = 16
_wgrad_native_batch = _sig("nsky_wgrad_native_batch", C.POINTER(WgradProblem), C.c_int32, C.c_int32, C.c_void_p)


def wgrad_problem(dZ, nnt_a, X, nnt_b, rows, dW, db=None, a_scale_max=None, b_scale=64.0, width_a=0, width_b=0, bias_row_mod=0,
                  b_scale_max=None) -> WgradProblem:
    """width_a / width_b: features of dZ / X that exist when fewer than the tiles walked (dW is then [width_a, width_b]);
    bias_row_mod = 4: db sums the value rows (row % 4 == 0) of a quad-native dZ only; b_scale_max: device scalar max |X| (replaces the
    constant b_scale: operands that carry input tangents have no a-priori bound)"""
    wa, wb = width_a or 32 * nnt_a, width_b or 32 * nnt_b
    assert dW.stride(1) == 1 and dW.shape[0] >= wa and dW.shape[1] >= wb
    assert dZ.numel() >= film_rows(rows) * 32 * nnt_a and X.numel() >= film_rows(rows) * 32 * nnt_b
    return WgradProblem(ptr(dZ), nnt_a, ptr(X), nnt_b, ptr(dW), ld(dW), ptr(db), ptr(a_scale_max), float(b_scale), 0, 0, int(width_a), int(width_b), 0,
                        int(bias_row_mod), ptr(b_scale_max))


def wgrad_problem_rowmajor(dZ, n_out, X, k_in, rows, dW, db=None, bias_rows=0) -> WgradProblem:
    """dW[n_out, k_in] += dZ[:rows, :n_out]^T X[:rows, :k_in] over ROW-MAJOR operands (2-term bf16 products); see include/neusky_hip.h"""
    assert dW.stride(1) == 1 and dW.shape[0] >= n_out and dW.shape[1] >= k_in and n_out % 4 == 0 and k_in % 4 == 0
    assert dZ.shape[0] >= rows and X.shape[0] >= rows and ld(dZ) >= n_out and ld(X) >= k_in
    return WgradProblem(ptr(dZ), 0, ptr(X), 0, ptr(dW), ld(dW), ptr(db), None, 1.0, ld(dZ), ld(X), n_out, k_in, int(bias_rows), 0, None)


def wgrad_native_batch(problems, rows):
    """every problem: dW += dZ^T X, db += column sums of dZ over tile-native matrices sharing the batch `rows`; ONE launch
    (csrc/wgrad_native.hip).  problems: list of wgrad_problem(...), at most WGRAD_MAX_PROBLEMS."""
    for i in range(0, len(problems), WGRAD_MAX_PROBLEMS):
        chunk = problems[i:i + WGRAD_MAX_PROBLEMS]
        arr = (WgradProblem * len(chunk))(*chunk)
        check(_wgrad_native_batch(arr, len(chunk), rows, stream_ptr()), "nsky_wgrad_native_batch")


def wgrad_native(dZ, nnt_a, X, nnt_b, rows, dW, db=None, a_scale_max=None, b_scale=64.0):
    """dW[32 nnt_a, 32 nnt_b] += dZ^T X and db += column sums of dZ over two tile-native matrices (csrc/wgrad_native.hip);
    dW / db are accumulated into.  nnt_a, nnt_b: multiples of 4."""
    wgrad_problem(dZ, nnt_a, X, nnt_b, rows, dW, db, a_scale_max, b_scale)  # shape checks
    check(_wgrad_native(ptr(dZ), nnt_a, ptr(X), nnt_b, rows, ptr(dW), ld(dW), ptr(db), ptr(a_scale_max), float(b_scale), stream_ptr()),
          "nsky_wgrad_native")
    return dW


def film_supported(hidden, map_hidden, n_map, n_film, cond_dim, x_dim, out_dim) -> bool:
    """shapes the fused chain kernels are built for (anything else runs the per-layer dense kernels)"""
    return (hidden in (128, 256) and map_hidden == hidden and 1 <= n_map <= FILM_MAX_LAYERS and 1 <= n_film <= FILM_MAX_LAYERS
            and 1 <= cond_dim <= 320 and 1 <= x_dim <= 16 and 1 <= out_dim <= 4
            and n_map * hidden + 3 * n_film * hidden + 32 <= 6144)


def film_net(cond_dim, x_dim, out_dim, map_w, map_b, mo_w, mo_b, film_w, film_b, out_w, out_b) -> FilmNet:
    """describe a FiLM-SIREN by its fp32 parameter tensors (torch nn.Linear layout, rows contiguous); the caller keeps
    the tensors alive while the descriptor is in use"""
    n = FilmNet(hidden=film_w[0].shape[0], n_map=len(map_w), n_film=len(film_w), cond_dim=cond_dim, x_dim=x_dim, out_dim=out_dim)
    for i, (w, b) in enumerate(zip(map_w, map_b)):
        n.map_w[i], n.map_b[i], n.map_ld[i] = ptr(w), ptr(b), ld(w)
    for i, (w, b) in enumerate(zip(film_w, film_b)):
        n.film_w[i], n.film_b[i], n.film_ld[i] = ptr(w), ptr(b), ld(w)
    n.mo_w, n.mo_b, n.mo_ld = ptr(mo_w), ptr(mo_b), ld(mo_w)
    n.out_w, n.out_b, n.out_ld = ptr(out_w), ptr(out_b), ld(out_w)
    return n


def film_stream_layout(net: FilmNet, direction: int = 0):
    nbytes, ntiles = C.c_int64(0), C.c_int32(0)
    check(_film_layout(C.byref(net), direction, C.byref(nbytes), C.byref(ntiles)), "nsky_film_stream_layout")
    return nbytes.value, ntiles.value


def film_pack(net: FilmNet, stream_buf, scales, direction: int = 0):
    check(_film_pack(C.byref(net), direction, ptr(stream_buf), ptr(scales), stream_ptr()), "nsky_film_pack")


_ray_points_fwd = _sig("nsky_ray_points_fwd", C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
_ray_points_bwd = _sig("nsky_ray_points_bwd", C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)


def ray_points_fwd(origins, dirs, sign, t, out):
    """out[i] = origins[i] + sign t[i] dirs[i % len(dirs)]; out may be a row slice of a larger [*, 3] buffer"""
    n = origins.shape[0]
    assert origins.is_contiguous() and dirs.is_contiguous() and t.is_contiguous() and out.is_contiguous() and out.shape == (n, 3) and t.numel() == n
    check(_ray_points_fwd(ptr(origins), ptr(dirs), dirs.shape[0], float(sign), ptr(t), n, ptr(out), stream_ptr()), "nsky_ray_points_fwd")


def ray_points_bwd(dirs, sign, d_out, d_t):
    n = d_out.shape[0]
    assert dirs.is_contiguous() and d_out.is_contiguous() and d_t.is_contiguous() and d_t.numel() == n
    check(_ray_points_bwd(ptr(dirs), dirs.shape[0], float(sign), ptr(d_out), n, ptr(d_t), stream_ptr()), "nsky_ray_points_bwd")


_train_metrics = _sig("nsky_train_metrics", C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p)


def train_metrics(pred, gt, mask, peak_sq, variance=None):
    """-> out [3] float32: [PSNR-like 10 log10(peak_sq / mse of the (masked) difference), s_val, 1 / s_val] (the last two when
    variance is given); pred / gt / mask contiguous with the same number of elements"""
    n = pred.numel()
    assert pred.is_contiguous() and gt.is_contiguous() and gt.numel() == n and (mask is None or (mask.is_contiguous() and mask.numel() == n))
    assert pred.dtype == gt.dtype == torch.float32 and (mask is None or mask.dtype == torch.float32)
    out = torch.empty(3, device=pred.device)
    check(_train_metrics(ptr(pred), ptr(gt), ptr(mask), n, float(peak_sq), ptr(variance), ptr(out), stream_ptr()), "nsky_train_metrics")
    return out


_point_alphas_fwd = _sig("nsky_point_alphas_fwd", C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_float, C.c_int32,
                         C.c_void_p, C.c_void_p)
_point_alphas_bwd = _sig("nsky_point_alphas_bwd", C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_float, C.c_int32,
                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def point_alphas_fwd(sdf, grad, dirs, gaps, variance, anneal, alphas):
    P = sdf.numel()
    assert sdf.is_contiguous() and grad.is_contiguous() and dirs.is_contiguous() and alphas.is_contiguous() and alphas.shape == (P, 3)
    g3 = (C.c_float * 3)(*[float(v) for v in gaps])
    check(_point_alphas_fwd(ptr(sdf), ptr(grad), ptr(dirs), g3, ptr(variance), float(anneal), P, ptr(alphas), stream_ptr()), "nsky_point_alphas_fwd")


def point_alphas_bwd(sdf, grad, dirs, gaps, variance, anneal, d_alphas, d_sdf, d_grad, d_variance):
    P = sdf.numel()
    assert d_alphas.is_contiguous() and d_sdf.is_contiguous() and d_grad.is_contiguous()
    g3 = (C.c_float * 3)(*[float(v) for v in gaps])
    check(_point_alphas_bwd(ptr(sdf), ptr(grad), ptr(dirs), g3, ptr(variance), float(anneal), P, ptr(d_alphas), ptr(d_sdf), ptr(d_grad),
                            ptr(d_variance), stream_ptr()), "nsky_point_alphas_bwd")


_sig_col_fwd = _sig("nsky_sigmoid_column_fwd", C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p)
_sig_col_bwd = _sig("nsky_sigmoid_column_bwd", C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p)


def sigmoid_column_fwd(raw, scale, t):
    assert raw.dim() == 2 and raw.stride(1) == 1 and t.is_contiguous() and t.numel() == raw.shape[0]
    check(_sig_col_fwd(ptr(raw), ld(raw), raw.shape[0], float(scale), ptr(t), stream_ptr()), "nsky_sigmoid_column_fwd")


def sigmoid_column_bwd(raw, scale, d_t, d_raw):
    assert d_raw.shape == raw.shape and d_raw.is_contiguous() and raw.is_contiguous() and d_t.is_contiguous()
    check(_sig_col_bwd(ptr(raw), ld(raw), raw.shape[0], float(scale), ptr(d_t), ptr(d_raw), stream_ptr()), "nsky_sigmoid_column_bwd")


_prop_fwd = _sig("nsky_proposal_mlp_fwd", C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                 C.c_void_p, C.c_void_p, C.c_void_p)
_prop_bwd = _sig("nsky_proposal_mlp_bwd", C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def proposal_mlp_supported(in_dim: int, hidden: int) -> bool:
    return hidden == 16 and 1 <= in_dim <= 12


def proposal_mlp_fwd(feat, w0, b0, w1, b1, raw):
    """feat [P, ld >= in_dim] -> raw [P]; w0 [16, in_dim], b0 [16], w1 [1, 16] or [16], b1 [1] (torch nn.Linear tensors)"""
    P, in_dim = feat.shape[0], w0.shape[1]
    assert feat.stride(1) == 1 and w0.is_contiguous() and w1.is_contiguous() and raw.is_contiguous() and raw.numel() == P
    check(_prop_fwd(ptr(feat), ld(feat), P, in_dim, w0.shape[0], ptr(w0), ld(w0), ptr(b0), ptr(w1), ptr(b1), ptr(raw), stream_ptr()),
          "nsky_proposal_mlp_fwd")


def proposal_mlp_bwd(feat, w0, b0, w1, b1, d_raw, d_feat, dw0, db0, dw1, db1):
    P, in_dim = feat.shape[0], w0.shape[1]
    assert d_raw.is_contiguous() and d_raw.numel() == P and (d_feat is None or (d_feat.shape == feat.shape and d_feat.is_contiguous() and feat.is_contiguous()))
    assert dw0.shape == w0.shape and dw0.is_contiguous()
    check(_prop_bwd(ptr(feat), ld(feat), P, in_dim, w0.shape[0], ptr(w0), ld(w0), ptr(b0), ptr(w1), ptr(b1), ptr(d_raw), ptr(d_feat), ptr(dw0),
                    ptr(db0), ptr(dw1), ptr(db1), stream_ptr()), "nsky_proposal_mlp_bwd")


class Segment(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64)]


_gather_segments = _sig("nsky_gather_segments", C.POINTER(Segment), C.c_int32, C.c_void_p)


def gather_segments(pairs):
    """pairs: [(src, dst)] contiguous float32 tensors of equal numel: dst <- src for all of them in ONE launch"""
    if not pairs:
        return
    arr = (Segment * len(pairs))()
    for i, (src, dst) in enumerate(pairs):
        assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() and src.dtype == dst.dtype == torch.float32
        arr[i] = Segment(src.data_ptr(), dst.data_ptr(), src.numel())
    check(_gather_segments(arr, len(pairs), stream_ptr()), "nsky_gather_segments")


_copy_segments = _sig("nsky_copy_segments", C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p)


def copy_segments(pairs):
    """pairs: [(src, dst)] contiguous device tensors of equal dtype and numel: dst <- src for all of them in ONE launch"""
    if not pairs:
        return
    n = len(pairs)
    src, dst, nb = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
    for i, (s_, d_) in enumerate(pairs):
        assert s_.is_cuda and d_.is_cuda and s_.is_contiguous() and d_.is_contiguous() and s_.dtype == d_.dtype and s_.numel() == d_.numel()
        src[i], dst[i], nb[i] = s_.data_ptr(), d_.data_ptr(), s_.numel() * s_.element_size()
    check(_copy_segments(src, dst, nb, n, stream_ptr()), "nsky_copy_segments")


class SdfNet(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32),
                ("w0", C.c_void_p), ("ld0", C.c_int32), ("b0", C.c_void_p),
                ("w1", C.c_void_p), ("ld1", C.c_int32), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("beta", C.c_float)]


_sdf_layout = _sig("nsky_sdf_stream_layout", C.POINTER(SdfNet), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32))
_sdf_pack = _sig("nsky_sdf_pack", C.POINTER(SdfNet), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sdf_fwd = _sig("nsky_sdf_chain_fwd", C.POINTER(SdfNet), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                C.c_void_p, C.c_void_p)
_sdf_bwd = _sig("nsky_sdf_chain_bwd", C.POINTER(SdfNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def sdf_supported(in_dim: int, hidden: int) -> bool:
    return hidden == 256 and 4 <= in_dim <= 80 and in_dim % 4 == 0


def sdf_net(W0, b0, W1, b1, w2_row, b2_elem, beta) -> SdfNet:
    """the geometry network evaluated for its sdf only: W0 [hidden, in_dim], W1 [hidden, hidden], w2_row [hidden] (a contiguous
    row view), b2_elem a 1-element view; the caller keeps the tensors alive while the descriptor is in use"""
    return SdfNet(in_dim=W0.shape[1], hidden=W0.shape[0], w0=ptr(W0), ld0=ld(W0), b0=ptr(b0), w1=ptr(W1), ld1=ld(W1), b1=ptr(b1),
                  w2=ptr(w2_row), b2=ptr(b2_elem), beta=float(beta))


def sdf_stream_layout(net: SdfNet, direction: int = 0):
    nbytes, ntiles = C.c_int64(0), C.c_int32(0)
    check(_sdf_layout(C.byref(net), direction, C.byref(nbytes), C.byref(ntiles)), "nsky_sdf_stream_layout")
    return nbytes.value, ntiles.value


def sdf_pack(net: SdfNet, stream_buf, table, direction: int = 0):
    check(_sdf_pack(C.byref(net), direction, ptr(stream_buf), ptr(table), stream_ptr()), "nsky_sdf_pack")


def sdf_chain_fwd(net: SdfNet, stream_buf, table, E, M, a0_save, a1_save, sdf):
    check(_sdf_fwd(C.byref(net), ptr(stream_buf), ptr(table), ptr(E), ld(E), M, ptr(a0_save), ptr(a1_save), ptr(sdf), stream_ptr()),
          "nsky_sdf_chain_fwd")
    return sdf


def sdf_chain_bwd(net: SdfNet, stream_buf, table, M, g_sdf, a0_save, a1_save, dz1, dz0, dE, dw2, db2, gmax):
    check(_sdf_bwd(C.byref(net), ptr(stream_buf), ptr(table), M, ptr(g_sdf), ptr(a0_save), ptr(a1_save), ptr(dz1), ptr(dz0),
                   ptr(dE), 0 if dE is None else ld(dE), ptr(dw2), ptr(db2), ptr(gmax), stream_ptr()), "nsky_sdf_chain_bwd")


# ------------------------------------------------------------------------------------------ SDF / albedo field chain
CHAIN_MAX_LAYERS = 8


class ChainLayer(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ld", C.c_int32), ("rows", C.c_int32), ("K", C.c_int32), ("transposed", C.c_int32)]


class FieldNet(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("npe", C.c_int32), ("beta", C.c_float),
                ("b0", C.c_void_p), ("b1", C.c_void_p), ("w_sdf", C.c_void_p), ("b_sdf", C.c_void_p), ("b2f", C.c_void_p),
                ("bc0", C.c_void_p), ("bc1", C.c_void_p), ("wc2", C.c_void_p), ("ldc2", C.c_int32), ("bc2", C.c_void_p)]


_chain_layout = _sig("nsky_chain_stream_layout", C.POINTER(ChainLayer), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32))
_chain_pack = _sig("nsky_chain_pack", C.POINTER(ChainLayer), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_field_geo_fwd = _sig("nsky_field_geo_fwd", C.POINTER(FieldNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_field_col_fwd = _sig("nsky_field_colour_fwd", C.POINTER(FieldNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_field_col_bwd = _sig("nsky_field_colour_bwd", C.POINTER(FieldNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_field_geo_bwd = _sig("nsky_field_geo_bwd", C.POINTER(FieldNet), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)
_native_wcolsum = _sig("nsky_native_weighted_colsum", C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_int32, C.c_void_p, C.c_void_p)


def chain_layer(W, rows, K, transposed=False) -> ChainLayer:
    """one layer of a packed weight stream: W[r][k] = W[r * ld + k] (transposed: W[k * ld + r]), r < rows, k < K"""
    return ChainLayer(ptr(W), ld(W), int(rows), int(K), int(bool(transposed)))


def chain_pack(layers, device, buffers=None):
    """-> (stream bytes tensor, per-tile reciprocal scales, number of groups); see include/neusky_hip.h.
    buffers(nbytes, n_scales) -> (zero-initialised uint8 stream buffer, scales buffer): a caller's persistent pair (the pack writes
    the real slabs only, so the pad slabs of a buffer that has held the same layout before are still zero)"""
    arr = (ChainLayer * len(layers))(*layers)
    nbytes, ntiles, ngroups = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    check(_chain_layout(arr, len(layers), C.byref(nbytes), C.byref(ntiles), C.byref(ngroups)), "nsky_chain_stream_layout")
    if buffers is not None:
        stream, scales = buffers(nbytes.value, max(ntiles.value, 4))
    else:
        stream = torch.zeros(nbytes.value, dtype=torch.uint8, device=device)  # (pad slabs of partial groups are streamed, never multiplied)
        scales = torch.empty(max(ntiles.value, 4), device=device)
    check(_chain_pack(arr, len(layers), ptr(stream), ptr(scales), stream_ptr()), "nsky_chain_pack")
    return stream, scales, ngroups.value


def field_supported(in_dim: int, hidden: int, geo_feat: int, hidden_colour: int, colour_in: int) -> bool:
    """shapes the fused field kernels are built for (the `neusky` method's, neusky_config.py:66-77)"""
    return hidden == 256 and geo_feat == 256 and hidden_colour == 256 and 68 <= in_dim <= 80 and in_dim % 4 == 0 and colour_in == 300


def field_net(in_dim, npe, beta, b0, b1, w_sdf, b_sdf, b2f=None, bc0=None, bc1=None, wc2=None, bc2=None) -> FieldNet:
    return FieldNet(in_dim=in_dim, npe=npe, beta=float(beta), b0=ptr(b0), b1=ptr(b1), w_sdf=ptr(w_sdf), b_sdf=ptr(b_sdf), b2f=ptr(b2f),
                    bc0=ptr(bc0), bc1=ptr(bc1), wc2=ptr(wc2), ldc2=0 if wc2 is None else ld(wc2), bc2=ptr(bc2))


def field_geo_fwd(net: FieldNet, pack, ET, N, a0q, a1q, Eq, a1max, sdf, grad, qmax=None):
    """qmax [2] (zero-filled by the caller): max |Eq|, max |a0q| -- the b_scale_max of the first two layers' weight gradients"""
    stream, scales, groups = pack
    check(_field_geo_fwd(C.byref(net), ptr(stream), ptr(scales), groups, ptr(ET), ld(ET), N, ptr(a0q), ptr(a1q), ptr(Eq), ptr(a1max), ptr(sdf),
                         ptr(grad), ptr(qmax), stream_ptr()), "nsky_field_geo_fwd")


def field_colour_fwd(net: FieldNet, pack, ET, N, a1q, a1max, a1v, feat, xpe, c0, c1, alb):
    stream, scales, groups = pack
    check(_field_col_fwd(C.byref(net), ptr(stream), ptr(scales), groups, ptr(ET), ld(ET), N, ptr(a1q), ptr(a1max), ptr(a1v), ptr(feat), ptr(xpe),
                         ptr(c0), ptr(c1), ptr(alb), stream_ptr()), "nsky_field_colour_fwd")


def field_colour_bwd(net: FieldNet, pack, N, g_alb, alb, c0, c1, dpc2, dpc1, dpc0, dfeat, dxpe, da1v, gmax):
    stream, scales, groups = pack
    check(_field_col_bwd(C.byref(net), ptr(stream), ptr(scales), groups, N, ptr(g_alb), ptr(alb), ptr(c0), ptr(c1), ptr(dpc2), ptr(dpc1), ptr(dpc0),
                         ptr(dfeat), ptr(dxpe), ptr(da1v), ptr(gmax), stream_ptr()), "nsky_field_colour_bwd")


def field_geo_bwd(net: FieldNet, pack, N, g_sdf, g_grad, da1v, dxpe, a0q, a1q, d1q, d0q, dET, gmax):
    stream, scales, groups = pack
    check(_field_geo_bwd(C.byref(net), ptr(stream), ptr(scales), groups, N, ptr(g_sdf), ptr(g_grad), ptr(da1v), ptr(dxpe), ptr(a0q), ptr(a1q),
                         ptr(d1q), ptr(d0q), ptr(dET), 0 if dET is None else ld(dET), ptr(gmax), stream_ptr()), "nsky_field_geo_bwd")


def native_weighted_colsum(X, nt, rows, out, bias=None, w4=None, n_out=1, g_sdf=None, g_grad=None):
    """out[o] += sum_rows w[row][o] X[row] over a tile-native X [rows, 32 nt]; w4 [rows, 4], or the quad form (g_sdf / g_grad)"""
    check(_native_wcolsum(ptr(X), nt, rows, ptr(w4), n_out, ptr(g_sdf), ptr(g_grad), ptr(out), ld(out) if out.dim() == 2 else out.shape[0],
                          ptr(bias), stream_ptr()), "nsky_native_weighted_colsum")


def quad_native_to_rows(buf, N: int, width: int):
    """quad-native [ceil32(4 N), width] -> [4, N, width] (row-set major; tests)"""
    return film_native_to_rows(buf, 4 * N, width).reshape(N, 4, width).permute(1, 0, 2)


def _ptr_array(ts, n):
    arr = (C.c_void_p * FILM_MAX_LAYERS)()
    for i in range(n):
        arr[i] = None if ts is None or ts[i] is None else ts[i].data_ptr()
    return arr


def film_chain_fwd(net: FilmNet, stream_buf, scales, cond, x, M, h_save, z_save, y_save, res):
    """see include/neusky_hip.h; h_save / z_save: lists (entries may be None) or None; y_save: list of [M, hidden] tensors"""
    check(_film_fwd(C.byref(net), ptr(stream_buf), ptr(scales), ptr(cond), ld(cond), ptr(x), ld(x), M,
                    _ptr_array(h_save, net.n_map), _ptr_array(z_save, net.n_film), _ptr_array(y_save, net.n_film), ptr(res), ld(res),
                    stream_ptr()), "nsky_film_chain_fwd")
    return res


def film_rows(M: int) -> int:
    """rows of a tile-native activation matrix holding M batch rows"""
    return (M + 31) // 32 * 32


def film_native_to_rows(buf, M: int, width: int):
    """tile-native [ceil32(M), width] -> row-major [M, width] (tests / fallbacks; torch ops)"""
    R = film_rows(M) // 32
    return buf.reshape(R, width // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(R * 32, width)[:M]


def film_rows_to_native(x, width: int):
    """row-major [M, width] -> tile-native [ceil32(M), width] (zero padded rows)"""
    M = x.shape[0]
    R = film_rows(M) // 32
    full = x.new_zeros(R * 32, width)
    full[:M] = x[:, :width]
    return full.reshape(R, 32, width // 32, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().reshape(R * 32, width)


def film_chain_bwd_film(net: FilmNet, stream_buf, table, M, d_res, h_last, z_save, dz_save, dfp, dfp_rowmax, gmax, d_x=None):
    """gmax: zero-filled float tensor [n_film + 1]; d_x: optional [M, ldx] output; see include/neusky_hip.h"""
    check(_film_bwd_film(C.byref(net), ptr(stream_buf), ptr(table), M, ptr(d_res), ld(d_res), ptr(h_last), _ptr_array(z_save, net.n_film),
                         _ptr_array(dz_save, net.n_film), ptr(dfp), ptr(dfp_rowmax), ptr(gmax), ptr(d_x), ld(d_x) if d_x is not None else 0,
                         stream_ptr()), "nsky_film_chain_bwd_film")


def film_chain_bwd_map(net: FilmNet, stream_buf, table, M, dfp, dfp_rowmax, h_save, dpre_save, d_cond, gmax):
    """gmax: zero-filled float tensor [n_map]"""
    check(_film_bwd_map(C.byref(net), ptr(stream_buf), ptr(table), M, ptr(dfp), ptr(dfp_rowmax), _ptr_array(h_save, net.n_map),
                        _ptr_array(dpre_save, net.n_map), ptr(d_cond), ld(d_cond) if d_cond is not None else 0, ptr(gmax), stream_ptr()),
          "nsky_film_chain_bwd_map")
