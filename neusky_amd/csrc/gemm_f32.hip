// fp32 dense layer on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   C[M,N] = epilogue( sum_k A(m,k) B(n,k) + bias[n] )
//
// Workgroup = 256 threads = 4 waves (one per SIMD).  Block tile BM x BN x 32, each wave owns a
// (BM/WAVES_M) x (BN/WAVES_N) sub-tile built from 32x32 MFMA accumulators.  Both operand tiles live
// in LDS K-MAJOR ([k][m], row stride BM+2 floats) whatever their layout in HBM, so the MFMA operand
// fetch is one conflict-free ds_read_b32 per 32x2 fragment (lanes 0-31 read row k, lanes 32-63 row
// k+1).  k-contiguous operands are transposed on the way into LDS (8 lanes read one 128-B row
// segment, the (BM+2) stride keeps the scattered ds_write_b32 at 2-way = free).  Global->LDS is
// register staged and software pipelined: tile t+1 is fetched to VGPRs before the 64 MFMAs of tile
// t and written to the other LDS buffer after them, one barrier per tile.
#include <stdlib.h>

#include "common.h"
#include "../../include/neusky_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK_DEFAULT = 32;

struct EpiCtx {
  const float* bias;
  int epi;
  float p0, p1;
  const float* aux0; int ldaux0;
  const float* aux1; int ldaux1;
  const float* aux2; int ldaux2;
  float* out1; int ldout1;
  float* out2; int ldout2;
  int row_mod;
  float beta;
  int rs_limit;  // a_rowsum: only k < rs_limit contribute (whole 32-deep k-tiles)
  int a_nnt, b_nnt;            // > 0: the (m-contiguous) operand is a tile-native matrix with that many 32-feature tiles per row
  const float* a_scale_max;    // device scalar: largest |A| (A is multiplied by a power of two that brings it to ~2^14; undone on C)
};

// float offset of element (row k, feature t), t % 4 == 0, of a tile-native matrix with nnt 32-feature tiles per row
// (include/neusky_hip.h: 32 x 32 blocks in v_mfma_f32_32x32x16 accumulator order)
__device__ __forceinline__ long native_offset(int k, int t, int nnt) {
  return ((long)(k >> 5) * nnt + (t >> 5)) * 1024 + ((t & 31) >> 3) * 256 + ((k & 31) + 32 * ((t >> 2) & 1)) * 4;
}
__device__ __forceinline__ float pow2_scale_for(float m, float& inv) {  // m s < 2^15
  if (!(m > 0.0f) || !(m < 3.0e38f)) { inv = 1.0f; return 1.0f; }
  int e;
  (void)frexpf(m, &e);
  e = max(-100, min(100, e));
  inv = ldexpf(1.0f, e - 15);
  return ldexpf(1.0f, 15 - e);
}

// sin/cos with Cody-Waite reduction to [-pi/4, pi/4] and minimax polynomials (|err| < 2e-7 for |x| < 1e4):
// ~20 VALU ops instead of the ocml slow path; used by the FiLM epilogues where |x| = |freq * z + phase| ~ 1e2.
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
  const float k = rintf(x * 0.6366197723675814f);  // x * 2/pi
  float r = fmaf(-k, 1.5707962513e+00f, x);
  r = fmaf(-k, 7.5497894159e-08f, r);
  r = fmaf(-k, 5.3903029534e-15f, r);
  const float r2 = r * r;
  float sp = fmaf(r2, 2.7183114939e-06f, -1.9839334836e-04f);
  sp = fmaf(sp, r2, 8.3333293855e-03f);
  sp = fmaf(sp, r2, -1.6666666567e-01f);
  sp = fmaf(sp * r2, r, r);
  float cp = fmaf(r2, 2.4433157117e-05f, -1.3887316255e-03f);
  cp = fmaf(cp, r2, 4.1666645683e-02f);
  cp = fmaf(cp, r2, -0.5f);
  cp = fmaf(cp, r2, 1.0f);
  const int q = (int)k;
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// softplus_beta(v) and sigmoid(beta v) from ONE exponential: t = exp(-|beta v|) in (0, 1];
// softplus = (max(beta v, 0) + log1p(t)) / beta, sigmoid = 1/(1+t) or t/(1+t).  log1p(t) = log(u) * t / (u - 1) with
// u = fl(1 + t) cancels the rounding of 1 + t (few-ulp result for every t); hardware exp2/log2/rcp based.
// torch.nn.functional.softplus semantics: beta v > 20 returns v itself (sdf_albedo_field.py geo network, beta = 100).
__device__ __forceinline__ void softplus_sigmoid(float v, float beta, float inv_beta, float& sp, float& sg) {
  const float bv = beta * v;
  const float t = __expf(-fabsf(bv));
  const float u = 1.0f + t;
  const float rc = __builtin_amdgcn_rcpf(u);
  const float um1 = u - 1.0f;
  const float l = um1 == 0.0f ? t : __logf(u) * (t * __builtin_amdgcn_rcpf(um1));
  sp = bv > 20.0f ? v : (fmaxf(bv, 0.0f) + l) * inv_beta;
  sg = bv >= 0.0f ? rc : t * rc;
}

__device__ __forceinline__ void epilogue_store(const EpiCtx& e, float* C, int ldc, int row, int col, float acc) {
  float v = acc + (e.bias ? e.bias[col] : 0.0f);
  float r;
  switch (e.epi) {
    default:
    case NSKY_EPI_NONE: r = v; break;
    case NSKY_EPI_RELU: r = fmaxf(v, 0.0f); break;
    case NSKY_EPI_LEAKY: r = v > 0.0f ? v : e.p0 * v; break;
    case NSKY_EPI_SIGMOID: r = e.p0 * sigmoidf_(v); break;
    case NSKY_EPI_SOFTPLUS: {
      float sg;
      softplus_sigmoid(v, e.p0, 1.0f / e.p0, r, sg);
      if (e.out1) e.out1[(long)row * e.ldout1 + col] = sg;
    } break;
    case NSKY_EPI_FILM: {
      float f = e.p0 * e.aux0[(long)row * e.ldaux0 + col] + e.p1;
      float ph = e.aux1[(long)row * e.ldaux1 + col];
      float sn, cs;
      sincos_cw(fmaf(f, v, ph), sn, cs);
      r = sn;
      if (e.out1) e.out1[(long)row * e.ldout1 + col] = v;
    } break;
    case NSKY_EPI_MUL_AUX: r = v * e.aux0[(long)(row % e.row_mod) * e.ldaux0 + col]; break;
    case NSKY_EPI_BWD_RELU: r = e.aux0[(long)row * e.ldaux0 + col] > 0.0f ? v : 0.0f; break;
    case NSKY_EPI_BWD_LEAKY: r = e.aux0[(long)row * e.ldaux0 + col] > 0.0f ? v : e.p0 * v; break;
    case NSKY_EPI_BWD_FILM: {
      float z = e.aux0[(long)row * e.ldaux0 + col];
      float f = e.p0 * e.aux1[(long)row * e.ldaux1 + col] + e.p1;
      float ph = e.aux2[(long)row * e.ldaux2 + col];
      float sn, cs;
      sincos_cw(fmaf(f, z, ph), sn, cs);
      float gc = v * cs;
      r = gc * f;
      e.out1[(long)row * e.ldout1 + col] = gc * z * e.p0;
      e.out2[(long)row * e.ldout2 + col] = gc;
    } break;
    case NSKY_EPI_EXP: r = expf(fminf(v, e.p0)); break;
  }
  float* dst = C + (long)row * ldc + col;
  if (e.beta != 0.0f) r += e.beta * (*dst);
  *dst = r;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// float4 epilogue: 4 consecutive columns of one row (all operands 16-byte aligned, ld % 4 == 0)
__device__ __forceinline__ void epilogue_store4(const EpiCtx& e, float* C, int ldc, int row, int col, float4 a) {
  float v[4] = {a.x, a.y, a.z, a.w};
  if (e.bias) {
    const float4 b = ld4(e.bias + col);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  float r[4];
  switch (e.epi) {
    default:
    case NSKY_EPI_NONE:
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = v[i];
      break;
    case NSKY_EPI_RELU:
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i], 0.0f);
      break;
    case NSKY_EPI_LEAKY:
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = v[i] > 0.0f ? v[i] : e.p0 * v[i];
      break;
    case NSKY_EPI_SIGMOID:
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = e.p0 * sigmoidf_(v[i]);
      break;
    case NSKY_EPI_SOFTPLUS: {
      float sg[4];
      const float inv_beta = 1.0f / e.p0;
#pragma unroll
      for (int i = 0; i < 4; ++i) softplus_sigmoid(v[i], e.p0, inv_beta, r[i], sg[i]);
      if (e.out1) st4(e.out1 + (long)row * e.ldout1 + col, make_float4(sg[0], sg[1], sg[2], sg[3]));
    } break;
    case NSKY_EPI_FILM: {
      const float4 F = ld4(e.aux0 + (long)row * e.ldaux0 + col);
      const float4 P = ld4(e.aux1 + (long)row * e.ldaux1 + col);
      const float f[4] = {F.x, F.y, F.z, F.w}, ph[4] = {P.x, P.y, P.z, P.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sn, cs;
        sincos_cw(fmaf(fmaf(e.p0, f[i], e.p1), v[i], ph[i]), sn, cs);
        r[i] = sn;
      }
      if (e.out1) st4(e.out1 + (long)row * e.ldout1 + col, make_float4(v[0], v[1], v[2], v[3]));
    } break;
    case NSKY_EPI_MUL_AUX: {
      const float4 A = ld4(e.aux0 + (long)(row % e.row_mod) * e.ldaux0 + col);
      r[0] = v[0] * A.x; r[1] = v[1] * A.y; r[2] = v[2] * A.z; r[3] = v[3] * A.w;
    } break;
    case NSKY_EPI_BWD_RELU: {
      const float4 A = ld4(e.aux0 + (long)row * e.ldaux0 + col);
      r[0] = A.x > 0.f ? v[0] : 0.f; r[1] = A.y > 0.f ? v[1] : 0.f; r[2] = A.z > 0.f ? v[2] : 0.f; r[3] = A.w > 0.f ? v[3] : 0.f;
    } break;
    case NSKY_EPI_BWD_LEAKY: {
      const float4 A = ld4(e.aux0 + (long)row * e.ldaux0 + col);
      r[0] = A.x > 0.f ? v[0] : e.p0 * v[0]; r[1] = A.y > 0.f ? v[1] : e.p0 * v[1];
      r[2] = A.z > 0.f ? v[2] : e.p0 * v[2]; r[3] = A.w > 0.f ? v[3] : e.p0 * v[3];
    } break;
    case NSKY_EPI_BWD_FILM: {
      const float4 Z = ld4(e.aux0 + (long)row * e.ldaux0 + col);
      const float4 F = ld4(e.aux1 + (long)row * e.ldaux1 + col);
      const float4 P = ld4(e.aux2 + (long)row * e.ldaux2 + col);
      const float z[4] = {Z.x, Z.y, Z.z, Z.w}, fr[4] = {F.x, F.y, F.z, F.w}, ph[4] = {P.x, P.y, P.z, P.w};
      float o1[4], o2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float f = fmaf(e.p0, fr[i], e.p1);
        float sn, cs;
        sincos_cw(fmaf(f, z[i], ph[i]), sn, cs);
        const float gc = v[i] * cs;
        r[i] = gc * f; o1[i] = gc * z[i] * e.p0; o2[i] = gc;
      }
      st4(e.out1 + (long)row * e.ldout1 + col, make_float4(o1[0], o1[1], o1[2], o1[3]));
      st4(e.out2 + (long)row * e.ldout2 + col, make_float4(o2[0], o2[1], o2[2], o2[3]));
    } break;
    case NSKY_EPI_EXP:
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = expf(fminf(v[i], e.p0));
      break;
  }
  float* dst = C + (long)row * ldc + col;
  if (e.beta != 0.0f) {
    const float4 o = ld4(dst);
    r[0] += e.beta * o.x; r[1] += e.beta * o.y; r[2] += e.beta * o.z; r[3] += e.beta * o.w;
  }
  st4(dst, make_float4(r[0], r[1], r[2], r[3]));
}

template <int BT, bool KCONTIG, int BK>
struct TileLoader {
  // BT x BK tile; F4 float4 per thread
  static constexpr int F4 = BT * BK / 4 / 256;
  static constexpr int KQ = BK / 4;  // float4 per row (k-contiguous layout)
  float4 v[F4];

  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int t0, int T, int k0, int kend, int tid, int nnt = 0) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      int f = it * 256 + tid;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        int row = f / KQ, kq = f % KQ;
        int t = t0 + row, k = k0 + kq * 4;
        if (t < T && k < kend) x = *reinterpret_cast<const float4*>(P + (long)t * ld + k);
      } else {
        int krow = f / (BT / 4), t4 = f % (BT / 4);
        int k = k0 + krow, t = t0 + t4 * 4;
        if (k < kend && t < T) x = *reinterpret_cast<const float4*>(P + (nnt > 0 ? native_offset(k, t, nnt) : (long)k * ld + t));
      }
      v[it] = x;
    }
  }

  __device__ __forceinline__ void store(float* S, int tid) const {
    constexpr int LD = BT + 2;
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      int f = it * 256 + tid;
      if (KCONTIG) {
        int row = f / KQ, kq = f % KQ;
        float* s = S + (kq * 4) * LD + row;
        s[0] = v[it].x; s[LD] = v[it].y; s[2 * LD] = v[it].z; s[3 * LD] = v[it].w;
      } else {
        int krow = f / (BT / 4), t4 = f % (BT / 4);
        float2* s = reinterpret_cast<float2*>(S + krow * LD + t4 * 4);
        s[0] = make_float2(v[it].x, v[it].y);
        s[1] = make_float2(v[it].z, v[it].w);
      }
    }
  }
};

// Output-tile raster: a 1-D grid over (row tiles x column tiles).  Consecutive workgroup ids go round-robin to the 8
// XCDs (each with a private L2), so id % 8 labels the workgroups that share an L2; the bijective remap hands every
// label a contiguous run of logical tiles, and the logical order is column-tile fastest: the column tiles of one row
// tile (which all re-read the same A rows) run back to back on ONE L2 instead of streaming A from HBM once per
// column tile, while the (small) B operand stays L2-resident for every row tile.
__device__ __forceinline__ int xcd_contiguous_id() {
  const int nwg = gridDim.x;
  int id = blockIdx.x;
  if (nwg >= 16) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  return id;
}
__device__ __forceinline__ void tile_of_block(int tiles_n, int& m_tile, int& n_tile) {
  const int id = xcd_contiguous_id();
  n_tile = id % tiles_n;
  m_tile = id / tiles_n;
}
// Split-K launches put (output tile, k-split) on ONE grid axis, output tile fastest: the tiles of one k-split -- which read the
// same rows of both operands (a weight gradient's 2 x 2 tiles each read dZ and X of their split) -- get consecutive logical
// ids, i.e. the same XCD and neighbouring dispatch slots, so the re-reads hit that XCD's L2 instead of going out to HBM once
// per tile (PMC: the weight-gradient kernel fetched 2.1x its operand bytes with the split on grid.z).
__device__ __forceinline__ void tile_split_of_block(int tiles_m, int tiles_n, int& m_tile, int& n_tile, int& split) {
  const int id = xcd_contiguous_id();
  const int tiles = tiles_m * tiles_n;
  split = id / tiles;
  const int t = id - split * tiles;
  n_tile = t % tiles_n;
  m_tile = t / tiles_n;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool AK, bool BKC, int BK, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                       int ldc, int k_split_len, int vec4, float* __restrict__ a_rowsum, EpiCtx e) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA_S = BM + 2, LDB_S = BN + 2;
  constexpr int SMEM_MAIN = 2 * BK * (LDA_S + LDB_S), SMEM_EPI = WM * BN;
  __shared__ float smem[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA_S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int m_tile, n_tile, split;
  tile_split_of_block((M + BM - 1) / BM, (N + BN - 1) / BN, m_tile, n_tile, split);
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int kbeg = split * k_split_len;
  const int kend = min(K, kbeg + k_split_len);
  const int ntiles = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  TileLoader<BM, AK, BK> la;
  TileLoader<BN, BKC, BK> lb;
  if (ntiles > 0) {
    la.load(A, lda, m0, M, kbeg, kend, tid, e.a_nnt);
    lb.load(B, ldb, n0, N, kbeg, kend, tid, e.b_nnt);
    la.store(As, tid);
    lb.store(Bs, tid);
  }
  __syncthreads();

  int cur = 0;
  float rs_acc = 0.0f;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1 < ntiles);
    if (more) {
      la.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend, tid, e.a_nnt);
      lb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend, tid, e.b_nnt);
    }
    if (a_rowsum != nullptr && n_tile == 0 && tid < BM && kbeg + t * BK < e.rs_limit) {
      // bias gradient for free: sum_k A(m,k) of this K tile (A = dZ^T in the weight-gradient GEMM)
      const float* col = As + cur * BK * LDA_S + tid;
      float sacc = 0.0f;
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) sacc += col[kk * LDA_S];
      rs_acc += sacc;
    }
    const float* as = As + cur * BK * LDA_S + (lane >> 5) * LDA_S + wm * WM + (lane & 31);
    const float* bs = Bs + cur * BK * LDB_S + (lane >> 5) * LDB_S + wn * WN + (lane & 31);
    // fragment reads run one k-step ahead of the MFMAs that consume them (LDS latency hidden behind the
    // 4 x 64-cycle MFMAs of the current step)
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = as[(kk + 1) * 2 * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = bs[(kk + 1) * 2 * LDB_S + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc otherwise re-serialises them)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      la.store(As + (cur ^ 1) * BK * LDA_S, tid);
      lb.store(Bs + (cur ^ 1) * BK * LDB_S, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  if (a_rowsum != nullptr && n_tile == 0 && tid < BM && m0 + tid < M) atomicAdd(a_rowsum + m0 + tid, rs_acc);

  // Epilogue: park the accumulators in LDS (the operand buffers are dead after the last barrier), one
  // wave-row (WM rows) at a time, and sweep them row-major so C, the aux operands and the side outputs move
  // as whole rows.
  float* Cs = smem;
  const bool atomic = k_split_len < K;
#pragma unroll
  for (int pass = 0; pass < WAVES_M; ++pass) {
    if (pass > 0) __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int cl = wn * WN + j * 32 + (lane & 31);
            Cs[rl * BN + cl] = acc[i][j][r];
          }
    }
    __syncthreads();
    const int mrow0 = m0 + pass * WM;
    if (vec4 && !atomic) {
      constexpr int C4 = BN / 4;
      for (int c = tid; c < WM * C4; c += 256) {
        const int rl = c / C4, cl = (c % C4) * 4;
        const int row = mrow0 + rl, col = n0 + cl;
        if (row < M && col < N) epilogue_store4(e, C, ldc, row, col, *reinterpret_cast<const float4*>(Cs + rl * BN + cl));
      }
    } else {
      for (int idx = tid; idx < WM * BN; idx += 256) {
        const int rl = idx / BN, cl = idx % BN;
        const int row = mrow0 + rl, col = n0 + cl;
        if (row < M && col < N) {
          const float v = Cs[idx];
          if (atomic)
            atomicAdd(C + (long)row * ldc + col, v);
          else
            epilogue_store(e, C, ldc, row, col, v);
        }
      }
    }
  }
}

// =================================================================================================
// Split-bf16 variant: every fp32 operand element is split into NS bf16 terms (hi, [mid,] lo) while it is
// staged into LDS, and the product is rebuilt from 3 (NS=2: hh + hl + lh, ~2^-16 relative) or 6 (NS=3:
// + mm + hl' + l'h, ~fp32) v_mfma_f32_32x32x16_bf16 per output tile and 16-deep k-step, accumulated in fp32.
// One bf16 MFMA does 16x the MACs of v_mfma_f32_32x32x2_f32 in half the cycles, so the rebuilt product runs at
// 32/3 (NS=2) or 32/6 (NS=3) times the fp32 matrix rate.  Used for the backward GEMMs (gradients tolerate 1e-5).
// LDS image per operand and term: [128 rows][32 k] bf16, 80-byte rows (64 B + 16 B pad): the 16-byte operand
// fragments (8 consecutive k of one row) of any 16-lane group fall on 16 distinct 16-byte bank slots.
// =================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SROW = 40;  // bf16 elements per LDS row (32 + 8 pad)

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_hi_as_f32(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_lo_as_f32(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// fp16 split with a SCALED residual (precision NSKY_PREC_F16X2): x = h + 2^-11 l with h = fp16(x), l = fp16(2^11 (x - h)).
// x - h is exact in fp32 and at most half an fp16 ulp of x, so the scaled residual sits in fp16's normal range whenever
// x does and carries 11 more bits: h + 2^-11 l reproduces x to ~2^-22 relative (2^-36 absolute below fp16's normal
// range).  The product is rebuilt from THREE v_mfma_f32_32x32x16_f16: hh into one fp32 accumulator, h l' + l' h into a
// second one that is folded in with 2^-11 in the epilogue; only l l' (2^-22) is dropped.  fp32-grade products at the
// cost of the 2-term bf16 form -- valid for operands inside fp16's range (|x| <= 65504; larger magnitudes saturate),
// i.e. the forward layers' bounded activations and weights, not gradients.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_RES_SCALE = 2048.0f;

__device__ __forceinline__ uint32_t pack2h(float a, float b) {
  f32x2 v = {a, b};
  f16x2 r = __builtin_convertvector(v, f16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void unpack2h(uint32_t p, float& a, float& b) {
  f16x2 h = __builtin_bit_cast(f16x2, p);
  a = (float)h[0];
  b = (float)h[1];
}
__device__ __forceinline__ void split4h(const float x[4], uint2 out[2]) {
  float c[4], h[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_fmed3f(x[i], -65504.0f, 65504.0f);
  const uint32_t h01 = pack2h(c[0], c[1]), h23 = pack2h(c[2], c[3]);
  unpack2h(h01, h[0], h[1]);
  unpack2h(h23, h[2], h[3]);
  out[0] = make_uint2(h01, h23);
  out[1] = make_uint2(pack2h((c[0] - h[0]) * F16_RES_SCALE, (c[1] - h[1]) * F16_RES_SCALE),
                      pack2h((c[2] - h[2]) * F16_RES_SCALE, (c[3] - h[3]) * F16_RES_SCALE));
}

// split 4 floats into NS terms of 4 bf16 each (8 bytes per term)
template <int NS>
__device__ __forceinline__ void split4(const float x[4], uint2 out[NS]) {
  float r[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const uint32_t p01 = pack2(r[0], r[1]), p23 = pack2(r[2], r[3]);
    out[t] = make_uint2(p01, p23);
    if (t + 1 < NS) {
      r[0] -= bf16_hi_as_f32(p01); r[1] -= bf16_lo_as_f32(p01);
      r[2] -= bf16_hi_as_f32(p23); r[3] -= bf16_lo_as_f32(p23);
    }
  }
}

// Epilogue shared by the split kernels: park the 2x2 32x32 accumulators of each wave in LDS (the operand images are dead),
// one wave-row (64 tile rows) at a time, and sweep them row-major so C, the aux operands and the side outputs move as
// whole rows.  H: fold the scaled cross-term accumulator in with 2^-11.
template <bool H, int BN = 128, int TMX, int TNX>
__device__ __forceinline__ void tile_epilogue(float* Cs, f32x16 (&acc)[2][2], f32x16 (&accx)[TMX][TNX], int wm, int wn, int lane,
                                              int tid, int m0, int n0, int M, int N, float* __restrict__ C, int ldc, int vec4,
                                              bool atomic, const EpiCtx& e, float out_scale = 1.0f) {
  constexpr int WM = 64, WN = 64, NT = 2 * BN;  // BN / 64 wave columns x 2 wave rows x 64 lanes
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass > 0) __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int cl = wn * WN + j * 32 + (lane & 31);
            Cs[rl * BN + cl] = (H ? fmaf(accx[H ? i : 0][H ? j : 0][r], 1.0f / F16_RES_SCALE, acc[i][j][r]) : acc[i][j][r]) * out_scale;
          }
    }
    __syncthreads();
    const int mrow0 = m0 + pass * WM;
    if (vec4 && !atomic) {
      constexpr int C4 = BN / 4;
      for (int c = tid; c < WM * C4; c += NT) {
        const int rl = c / C4, cl = (c % C4) * 4;
        const int row = mrow0 + rl, col = n0 + cl;
        if (row < M && col < N) epilogue_store4(e, C, ldc, row, col, *reinterpret_cast<const float4*>(Cs + rl * BN + cl));
      }
    } else {
      for (int idx = tid; idx < WM * BN; idx += NT) {
        const int rl = idx / BN, cl = idx % BN;
        const int row = mrow0 + rl, col = n0 + cl;
        if (row < M && col < N) {
          const float v = Cs[idx];
          if (atomic)
            atomicAdd(C + (long)row * ldc + col, v);
          else
            epilogue_store(e, C, ldc, row, col, v);
        }
      }
    }
  }
}

template <int NS, bool KCONTIG, bool H = false, int ROWS = 128>
struct SplitLoader {  // ROWS x 32 fp32 tile -> NS bf16 images [ROWS][SROW]; ROWS = threads / 2 (non-k-contiguous form: any 8 x ROWS/4 threads)
  float4 v[4];
  float rs[4] = {0.f, 0.f, 0.f, 0.f};  // running row sums (every thread stages the same 4 tile rows on every k-tile)
  int nnt = 0;        // > 0: tile-native operand (non-k-contiguous form only)
  float scale = 1.0f; // power of two applied to every element as it is staged
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int t0, int T, int k0, int kend, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        const int f = it * 256 + tid;
        const int row = f >> 3, kq = f & 7;
        const int t = t0 + row, k = k0 + kq * 4;
        if (t < T && k < kend) x = *reinterpret_cast<const float4*>(P + (long)t * ld + k);
      } else {  // 4(k) x 4(t) block per thread: kb = tid % 8, mb = tid / 8; load row k0 + 4 kb + it
        const int kb = tid & 7, mb = tid >> 3;
        const int k = k0 + 4 * kb + it, t = t0 + 4 * mb;
        if (k < kend && t < T) x = *reinterpret_cast<const float4*>(P + (nnt > 0 ? native_offset(k, t, nnt) : (long)k * ld + t));
      }
      v[it] = make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale);
    }
  }
  // images: S + term * (128 * SROW) bf16 ; rowsum: also accumulate sum_k of the staged tile rows into rs[]
  __device__ __forceinline__ void store(__bf16* S, int tid, bool rowsum) {
    if (KCONTIG) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int f = it * 256 + tid;
        const int row = f >> 3, kq = f & 7;
        const float x[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        uint2 o[NS];
        if (H) split4h(x, o); else split4<NS>(x, o);
#pragma unroll
        for (int t = 0; t < NS; ++t) *reinterpret_cast<uint2*>(S + t * ROWS * SROW + row * SROW + kq * 4) = o[t];
        if (rowsum) rs[it] += (x[0] + x[1]) + (x[2] + x[3]);
      }
    } else {
      const int kb = tid & 7, mb = tid >> 3;
      const float m[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                             {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 o[NS];
        if (H) split4h(m[i], o); else split4<NS>(m[i], o);
#pragma unroll
        for (int t = 0; t < NS; ++t) *reinterpret_cast<uint2*>(S + t * ROWS * SROW + (4 * mb + i) * SROW + 4 * kb) = o[t];
        if (rowsum) rs[i] += (m[i][0] + m[i][1]) + (m[i][2] + m[i][3]);
      }
    }
  }
  // fold the running row sums into LDS float[128] (8 threads share a row)
  __device__ __forceinline__ void flush_rowsum(float* rsum, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(rsum + (KCONTIG ? i * 32 + (tid >> 3) : 4 * (tid >> 3) + i), rs[i]);
  }
};

template <int NS, bool AK, bool BKC, bool H = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16s_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int lda, int ldb,
    int ldc, int k_split_len, int vec4, float* __restrict__ a_rowsum, EpiCtx e) {
  constexpr int BM = 128, BN = 128, BKT = 32, WM = 64, WN = 64, TM = 2, TN = 2;
  constexpr int IMG = 128 * SROW;  // bf16 elements per image
  constexpr int STAGE = 2 * NS * IMG;  // bf16 elements of the stage (A images then B images)
  constexpr int SMEM_MAIN = STAGE * 2 + 128 * 4, SMEM_EPI = WM * BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
  __bf16* S0 = reinterpret_cast<__bf16*>(smem_raw);
  float* rsum = reinterpret_cast<float*>(smem_raw + STAGE * 2);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m_tile, n_tile, split;
  tile_split_of_block((M + BM - 1) / BM, (N + BN - 1) / BN, m_tile, n_tile, split);
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int kbeg = split * k_split_len;
  const int kend = min(K, kbeg + k_split_len);
  const int ntiles = (kend - kbeg + BKT - 1) / BKT;
  const bool want_rs = (a_rowsum != nullptr) && n_tile == 0;
  if (want_rs && tid < 128) rsum[tid] = 0.0f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  static_assert(!H || NS == 2, "the fp16 scaled-residual split has two terms");
  f32x16 accx[H ? TM : 1][H ? TN : 1];  // H: h l' + l' h, folded in with 2^-11 in the epilogue
  if (H) {
#pragma unroll
    for (int i = 0; i < (H ? TM : 1); ++i)
#pragma unroll
      for (int j = 0; j < (H ? TN : 1); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.0f;
  }
  SplitLoader<NS, AK, H> la;
  SplitLoader<NS, BKC, H> lb;
  la.nnt = e.a_nnt;
  lb.nnt = e.b_nnt;
  float a_inv = 1.0f;
  if (e.a_scale_max) la.scale = pow2_scale_for(*e.a_scale_max, a_inv);
  const int frow = lane & 31, fh = lane >> 5;

  auto compute = [&](const __bf16* As, const __bf16* Bs) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[NS][TM], bfr[NS][TN];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[s][i] = *reinterpret_cast<const bf16x8*>(As + s * IMG + (wm * WM + i * 32 + frow) * SROW + ks * 16 + fh * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[s][j] = *reinterpret_cast<const bf16x8*>(Bs + s * IMG + (wn * WN + j * 32 + frow) * SROW + ks * 16 + fh * 8);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (H) {
            const int ii = H ? i : 0, jj = H ? j : 0;
            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][i]), __builtin_bit_cast(f16x8, bfr[1][j]), accx[ii][jj], 0, 0, 0);
            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1][i]), __builtin_bit_cast(f16x8, bfr[0][j]), accx[ii][jj], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][i]), __builtin_bit_cast(f16x8, bfr[0][j]), acc[i][j], 0, 0, 0);
            continue;
          }
          if (NS == 3) {  // smallest cross terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  if (ntiles > 0) {
    la.load(A, lda, m0, M, kbeg, kend, tid);
    lb.load(B, ldb, n0, N, kbeg, kend, tid);
  }
  // one LDS stage, two workgroups per CU: tile t+1 is fetched to registers while tile t is multiplied (a two-stage variant,
  // a prefetch distance of two and a 256-wide tile were measured slower or neutral in round 1 and are gone)
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // previous tile's fragment reads are done
    la.store(S0, tid, want_rs && kbeg + t * BKT < e.rs_limit);
    lb.store(S0 + NS * IMG, tid, false);
    __syncthreads();
    if (t + 1 < ntiles) {
      la.load(A, lda, m0, M, kbeg + (t + 1) * BKT, kend, tid);
      lb.load(B, ldb, n0, N, kbeg + (t + 1) * BKT, kend, tid);
    }
    compute(S0, S0 + NS * IMG);
  }
  if (want_rs) la.flush_rowsum(rsum, tid);
  __syncthreads();
  if (want_rs && tid < 128 && m0 + tid < M) atomicAdd(a_rowsum + m0 + tid, rsum[tid] * a_inv);
  __syncthreads();

  tile_epilogue<H>(reinterpret_cast<float*>(smem_raw), acc, accx, wm, wn, lane, tid, m0, n0, M, N, C, ldc, vec4, k_split_len < K, e, a_inv);
}

// =================================================================================================
// LDS-DMA variant of the split kernels for a k-contiguous fp32 A (activations, gradients) against PRE-SPLIT weights.
//
// The lab (tools/gemm_lab) shows the register-staged kernel above spends its k-loop on three phases that add up instead
// of overlapping: the global loads (~1/3), the fp32 -> 16-bit split of both operands (~1/5) and the MFMAs (~1/2).  Here
//   * B (a weight matrix, re-read by every one of the M/128 row tiles) is split ONCE per step into two 16-bit planes
//     (nsky_split_planes) and streamed by global_load_lds_dwordx4 straight into its LDS image: no registers, no VALU;
//   * A is streamed as raw fp32 by the same LDS-DMA into a 3-deep ring and split IN PLACE: a 16-byte chunk of 4 floats
//     becomes 8 bytes of the hi plane + 8 bytes of the lo plane inside the same 32-byte pair of chunks, by the lane that
//     issued its DMA (so the wave's own vmcnt wait orders it, no barrier), one tile ahead of the MFMAs;
//   * one barrier per k-tile; tile t+2 of A and t+1 of B are in flight while tile t is multiplied.
// Stage image (A and B alike): [128 rows][128 B] = 8 chunks of 16 B per row, chunk c stored at position c ^ ((row >> 1) & 7):
// the DMA writes lane-linear (1 KB per wave instruction = 8 rows), the swizzle is applied to the SOURCE address, and the
// ds_read_b128 operand fetch (32 rows x one chunk) is bank-conflict free.  A chunks after the split: 2q = hi(k 8q..8q+7),
// 2q+1 = lo; B chunks: 0..3 = hi plane k 0..31, 4..7 = lo plane.
// LDS: 3 x 16 KB (A ring) + 2 x BN x 128 B (B ring) = 80 KB (BN = 128, two workgroups per CU) or 112 KB (BN = 256, one).
// Requires K % 4 == 0 and planes zero padded to whole 256-row x 32-k tiles; a partial last k-tile re-reads column 0 of A for
// the missing chunks (multiplied by the planes' zero padding), M tails re-read row M-1.
// =================================================================================================
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  // LDS-DMA hidden from hipcc's waitcnt bookkeeping (it would drain vmcnt(0) before every ds_read otherwise); M0 is saved
  // and restored inside the statement.  Completion is counted by hand: vmcnt_wait<N>().
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void vmcnt_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool H>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Bhi,
                                                             const uint16_t* __restrict__ Blo, float* __restrict__ C, int M, int N,
                                                             int K, int lda, int ldp, int ldc, int vec4, EpiCtx e) {
  // 128 x 128 tile, 4 waves (2 x 2), two workgroups per CU, every wave streams its share of A and of B (A ring of 3 stages,
  // B ring of 2).  (A 256-wide, wave-specialised variant measured 0.5 % slower on the step in round 1 and is gone.)
  constexpr int BN = 128, NW = BN / 32, WCOLS = BN / 64;
  constexpr int STAGE = 16384, A_STAGES = 3, B_STAGES = 2, B_OFF = A_STAGES * STAGE, B_STAGE = BN * 128;
  constexpr int NA = 16 / NW, NB = 4;  // LDS-DMA instructions per wave and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[B_OFF + B_STAGES * B_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int aw = wave, bw = wave;
  int m_tile, n_tile;
  tile_of_block((N + BN - 1) / BN, m_tile, n_tile);
  const int m0 = m_tile * 128, n0 = n_tile * BN;
  const int T = (K + 31) / 32;  // a partial last k-tile reads its missing A chunks from column 0 (any finite data: the planes
                                // are zero there), so K only has to be a multiple of 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  // DMA geometry of this lane: A instruction j of streaming wave aw fills rows 8 (NA aw + j) + lane / 8, B instruction j of
  // streaming wave bw rows 8 (NB bw + j) + lane / 8, chunk position lane % 8
  const int prow = lane >> 3, ppos = lane & 7;
  const float* a_src[NA];
  int a_col[NA];
  const uint16_t* b_src[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int r = 8 * (NA * aw + j) + prow;
    const int c = ppos ^ ((r >> 1) & 7);
    a_col[j] = 4 * c;
    a_src[j] = A + (long)min(m0 + r, M - 1) * lda + 4 * c;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int r = 8 * (NB * bw + j) + prow;
    const int c = ppos ^ ((r >> 1) & 7);
    b_src[j] = (c < 4 ? Bhi : Blo) + (long)(n0 + r) * ldp + 8 * (c & 3);
  }
  const uint32_t a_off = aw * (NA * 1024);
  auto issue_a = [&](int t) {
    const uint32_t dst = lds0 + (t % A_STAGES) * STAGE + a_off;
#pragma unroll
    for (int j = 0; j < NA; ++j) glds16(t * 32 + a_col[j] < K ? a_src[j] + t * 32 : a_src[j] - a_col[j], dst + j * 1024);
  };
  auto issue_b = [&](int t) {
    const uint32_t dst = lds0 + B_OFF + (t % B_STAGES) * B_STAGE + bw * (NB * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) glds16(b_src[j] + t * 32, dst + j * 1024);
  };
  // in-place split of the chunks this lane's DMAs delivered (tile t): raw_load reads them (before the MFMAs of the current
  // tile are issued), split_store converts and writes the two 8-byte halves (after them, so the VALU work can be scheduled
  // into the MFMA shadow)
  float4 raw[NA];
  auto raw_load = [&](int t) {
    const unsigned char* st = smem + (t % A_STAGES) * STAGE + a_off;
#pragma unroll
    for (int j = 0; j < NA; ++j) raw[j] = *reinterpret_cast<const float4*>(st + j * 1024 + lane * 16);
  };
  auto split_store = [&](int t) {
    unsigned char* st = smem + (t % A_STAGES) * STAGE + a_off;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int r = 8 * (NA * aw + j) + prow;
      const int sw = (r >> 1) & 7;
      const int g = ppos ^ sw;  // raw chunk held at this position: floats k = 4g .. 4g+3
      unsigned char* rowp = st + j * 1024 + prow * 128;
      const float x[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
      uint2 o[2];
      if (H) split4h(x, o); else split4<2>(x, o);
      const int q2 = g & ~1, half = g & 1;
      *reinterpret_cast<uint2*>(rowp + ((q2 ^ sw) * 16) + half * 8) = o[0];
      *reinterpret_cast<uint2*>(rowp + (((q2 + 1) ^ sw) * 16) + half * 8) = o[1];
    }
  };

  f32x16 acc[2][2];
  f32x16 accx[H ? 2 : 1][H ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.0f;
        if (H) accx[H ? i : 0][H ? j : 0][r] = 0.0f;
      }

  const int frow = lane & 31, fh = lane >> 5;
  auto compute = [&](int t) {
    const unsigned char* As = smem + (t % A_STAGES) * STAGE;
    const unsigned char* Bs = smem + B_OFF + (t % B_STAGES) * B_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2][2], bfr[2][2];
      const int q = 2 * ks + fh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + frow, sw = (r >> 1) & 7;
        af[0][i] = *reinterpret_cast<const bf16x8*>(As + r * 128 + (((2 * q) ^ sw) * 16));
        af[1][i] = *reinterpret_cast<const bf16x8*>(As + r * 128 + (((2 * q + 1) ^ sw) * 16));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + frow, sw = (r >> 1) & 7;
        bfr[0][j] = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + ((q ^ sw) * 16));
        bfr[1][j] = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + (((4 + q) ^ sw) * 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (H) {
            const int ii = H ? i : 0, jj = H ? j : 0;
            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][i]), __builtin_bit_cast(f16x8, bfr[1][j]), accx[ii][jj], 0, 0, 0);
            accx[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1][i]), __builtin_bit_cast(f16x8, bfr[0][j]), accx[ii][jj], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][i]), __builtin_bit_cast(f16x8, bfr[0][j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  {
    // prologue: queue = B0, A0, A1
    issue_b(0);
    issue_a(0);
    if (T > 1) { issue_a(1); vmcnt_wait<NA>(); } else { vmcnt_wait<0>(); }
    raw_load(0);
    split_store(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const bool n1 = t + 1 < T, n2 = t + 2 < T;
      if (n1) issue_b(t + 1);  // B stage (t+1)&1 and A stage (t+2)%3 were last read by the MFMAs of tile t-1 (barrier passed)
      if (n2) issue_a(t + 2);
      if (n1) {  // queue (oldest first): A(t+1), B(t+1), A(t+2); A(t+1) has been in flight for a whole tile
        if (n2) vmcnt_wait<NB + NA>(); else vmcnt_wait<NB>();
        raw_load(t + 1);
      }
      compute(t);
      if (n1) {
        split_store(t + 1);
        if (n2) vmcnt_wait<NA>(); else vmcnt_wait<0>();
      }
      __syncthreads();
    }
  }
  tile_epilogue<H, BN>(reinterpret_cast<float*>(smem), acc, accx, wm, wn, lane, tid, m0, n0, M, N, C, ldc, vec4, false, e);
}

// fp32 matrix -> two 16-bit planes [rows_pad][ldp] (zero padded): out(n, k) = W[n][k] (transpose = 0) or W[k][n] (1)
template <bool H>
__global__ void split_planes_kernel(const float* __restrict__ W, int n_rows, int n_k, int ldw, int transpose,
                                    uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int rows_pad, int ldp) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int a = ty + 8 * i;
    float v = 0.0f;
    if (transpose) {  // tile[k][n] read along n
      const int k = k0 + a, n = n0 + tx;
      if (k < n_k && n < n_rows) v = W[(long)k * ldw + n];
      tile[a][tx] = v;
    } else {  // tile[n][k] read along k
      const int n = n0 + a, k = k0 + tx;
      if (n < n_rows && k < n_k) v = W[(long)n * ldw + k];
      tile[a][tx] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nn = ty + 8 * i, kk = tx;
    const float v = transpose ? tile[kk][nn] : tile[nn][kk];
    const int n = n0 + nn, k = k0 + kk;
    if (n < rows_pad && k < ldp) {
      uint16_t h, l;
      if (H) {
        const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
        const _Float16 hh = (_Float16)c;
        const _Float16 ll = (_Float16)((c - (float)hh) * F16_RES_SCALE);
        h = __builtin_bit_cast(uint16_t, hh);
        l = __builtin_bit_cast(uint16_t, ll);
      } else {
        const __bf16 hh = (__bf16)v;
        const __bf16 ll = (__bf16)(v - (float)hh);
        h = __builtin_bit_cast(uint16_t, hh);
        l = __builtin_bit_cast(uint16_t, ll);
      }
      hi[(long)n * ldp + k] = h;
      lo[(long)n * ldp + k] = l;
    }
  }
}

template <int NS, bool H = false>
void launch_split(const nsky_gemm_desc* d, const EpiCtx& e, int splits, int k_split_len, int vec4, hipStream_t s) {
  dim3 grid(ceil_div(d->M, 128) * ceil_div(d->N, 128) * splits);
#define NSKY_SGEMM_LAUNCH(AK, BKC)                                                                          \
  hipLaunchKernelGGL((gemm_bf16s_kernel<NS, AK, BKC, H>), grid, dim3(256), 0, s, d->A, d->B, d->C, d->M, d->N, \
                     d->K, d->lda, d->ldb, d->ldc, k_split_len, vec4, d->a_rowsum, e)
  if (d->a_kcontig && d->b_kcontig) NSKY_SGEMM_LAUNCH(true, true);
  else if (d->a_kcontig && !d->b_kcontig) NSKY_SGEMM_LAUNCH(true, false);
  else if (!d->a_kcontig && d->b_kcontig) NSKY_SGEMM_LAUNCH(false, true);
  else NSKY_SGEMM_LAUNCH(false, false);
#undef NSKY_SGEMM_LAUNCH
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int OCC>
void launch(const nsky_gemm_desc* d, const EpiCtx& e, int splits, int k_split_len, int vec4, hipStream_t s) {
  dim3 grid(ceil_div(d->M, BM) * ceil_div(d->N, BN) * splits);
#define NSKY_GEMM_LAUNCH(AK, BKC)                                                                              \
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, AK, BKC, BK, OCC>), grid, dim3(256), 0, s, d->A, d->B, \
                     d->C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, k_split_len, vec4, d->a_rowsum, e)
  if (d->a_kcontig && d->b_kcontig) NSKY_GEMM_LAUNCH(true, true);
  else if (d->a_kcontig && !d->b_kcontig) NSKY_GEMM_LAUNCH(true, false);
  else if (!d->a_kcontig && d->b_kcontig) NSKY_GEMM_LAUNCH(false, true);
  else NSKY_GEMM_LAUNCH(false, false);
#undef NSKY_GEMM_LAUNCH
}

__global__ void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx, const float* __restrict__ w, int w_stride,
                              float* __restrict__ out, int rows_per_block) {
  // block: 256 threads = 64 columns x 4 row-phases; out[c] += sum_r (w ? w[r * w_stride] : 1) X[r][c]
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int phase = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.0f;
  if (col < N) {
    if (w)
      for (int r = r0 + phase; r < r1; r += 4) s = fmaf(w[(long)r * w_stride], X[(long)r * ldx + col], s);
    else
      for (int r = r0 + phase; r < r1; r += 4) s += X[(long)r * ldx + col];
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (phase == 0 && col < N) atomicAdd(out + col, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// float4 form (N % 4 == 0, ldx % 4 == 0, 16-byte aligned X): a wave reads one 1 KB row segment per instruction (64 lanes x 4
// columns), 4 row-phases per block, 4 independent accumulator chains per lane; ~3x the scalar form's HBM rate
__global__ void colsum4_kernel(const float* __restrict__ X, int M, int N, int ldx, const float* __restrict__ w, int w_stride,
                               float* __restrict__ out, int rows_per_block) {
  const int col = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
  const int phase = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
    for (int r = r0 + phase; r < r1; r += 4) {
      const float4 x = *reinterpret_cast<const float4*>(X + (long)r * ldx + col);
      const float wr = w ? w[(long)r * w_stride] : 1.0f;
      s.x = fmaf(wr, x.x, s.x); s.y = fmaf(wr, x.y, s.y); s.z = fmaf(wr, x.z, s.z); s.w = fmaf(wr, x.w, s.w);
    }
  }
  __shared__ float4 red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (phase == 0 && col < N) {
    const float4 a = red[threadIdx.x], b = red[threadIdx.x + 64], c = red[threadIdx.x + 128], d = red[threadIdx.x + 192];
    atomicAdd(out + col, (a.x + b.x) + (c.x + d.x));
    atomicAdd(out + col + 1, (a.y + b.y) + (c.y + d.y));
    atomicAdd(out + col + 2, (a.z + b.z) + (c.z + d.z));
    atomicAdd(out + col + 3, (a.w + b.w) + (c.w + d.w));
  }
}

}  // namespace

extern "C" int nsky_gemm_f32(const nsky_gemm_desc* d, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && d->A && d->B && d->C, "nsky_gemm_f32: null operand");
  NSKY_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "nsky_gemm_f32: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  NSKY_CHECK_ARG(d->lda % 4 == 0 && d->ldb % 4 == 0, "nsky_gemm_f32: lda/ldb must be multiples of 4 (got %d,%d)", d->lda, d->ldb);
  NSKY_CHECK_ARG(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "nsky_gemm_f32: A/B must be 16-byte aligned");
  if (d->a_kcontig || d->b_kcontig) NSKY_CHECK_ARG(d->K % 4 == 0, "nsky_gemm_f32: K=%d must be a multiple of 4 for k-contiguous operands", d->K);
  if (d->a_kcontig) NSKY_CHECK_ARG(d->lda >= d->K, "nsky_gemm_f32: lda < K"); else NSKY_CHECK_ARG(d->a_native_nt > 0 || d->lda >= d->M, "nsky_gemm_f32: lda < M");
  if (d->b_kcontig) NSKY_CHECK_ARG(d->ldb >= d->K, "nsky_gemm_f32: ldb < K"); else NSKY_CHECK_ARG(d->b_native_nt > 0 || d->ldb >= d->N, "nsky_gemm_f32: ldb < N");
  NSKY_CHECK_ARG(d->ldc >= d->N, "nsky_gemm_f32: ldc < N");
  int splits = d->k_splits > 1 ? d->k_splits : 1;
  int k_split_len = d->K;
  if (splits > 1) {
    NSKY_CHECK_ARG(d->epi == NSKY_EPI_NONE && !d->bias && d->beta == 0.0f, "nsky_gemm_f32: split-K needs a plain epilogue");
    k_split_len = ((d->K + splits - 1) / splits + 31) / 32 * 32;
    splits = (d->K + k_split_len - 1) / k_split_len;
  }
  switch (d->epi) {
    case NSKY_EPI_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1, "nsky_gemm_f32: FILM needs aux0 (freq) and aux1 (phase)"); break;
    case NSKY_EPI_MUL_AUX: case NSKY_EPI_BWD_RELU: case NSKY_EPI_BWD_LEAKY: NSKY_CHECK_ARG(d->aux0, "nsky_gemm_f32: epilogue needs aux0"); break;
    case NSKY_EPI_BWD_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1 && d->aux2 && d->out1 && d->out2, "nsky_gemm_f32: BWD_FILM needs aux0..2, out1, out2"); break;
    default: break;
  }
  EpiCtx e;
  e.bias = d->bias; e.epi = d->epi; e.p0 = d->p0; e.p1 = d->p1;
  e.aux0 = d->aux0; e.ldaux0 = d->ldaux0; e.aux1 = d->aux1; e.ldaux1 = d->ldaux1; e.aux2 = d->aux2; e.ldaux2 = d->ldaux2;
  e.out1 = d->out1; e.ldout1 = d->ldout1; e.out2 = d->out2; e.ldout2 = d->ldout2;
  e.row_mod = d->row_mod > 0 ? d->row_mod : d->M;
  e.beta = d->beta;
  e.rs_limit = d->rowsum_k_limit > 0 ? d->rowsum_k_limit : 0x7fffffff;
  e.a_nnt = d->a_native_nt; e.b_nnt = d->b_native_nt; e.a_scale_max = d->a_scale_max;
  if (d->a_native_nt > 0) NSKY_CHECK_ARG(!d->a_kcontig && d->M <= 32 * d->a_native_nt, "nsky_gemm_f32: a tile-native A is the m-contiguous (a_kcontig = 0) operand of a weight gradient");
  if (d->b_native_nt > 0) NSKY_CHECK_ARG(!d->b_kcontig && d->N <= 32 * d->b_native_nt, "nsky_gemm_f32: a tile-native B is the n-contiguous (b_kcontig = 0) operand of a weight gradient");
  if (d->a_scale_max) NSKY_CHECK_ARG(d->precision == NSKY_PREC_F16X2 && d->N > 64, "nsky_gemm_f32: a_scale_max is for the fp16-split contraction (N > 64)");
  if (d->a_rowsum && d->rowsum_k_limit > 0) NSKY_CHECK_ARG(d->rowsum_k_limit % 32 == 0, "nsky_gemm_f32: rowsum_k_limit=%d must be a multiple of 32", d->rowsum_k_limit);
  hipStream_t s = (hipStream_t)stream;
  // float4 epilogue when every row operand is 16-byte aligned with ld % 4 == 0 and N % 4 == 0
  auto ok = [](const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) % 16 == 0) && (ld % 4 == 0)); };
  const int vec4 = (d->N % 4 == 0) && ok(d->C, d->ldc) && ok(d->bias, 4) && ok(d->aux0, d->ldaux0) && ok(d->aux1, d->ldaux1) &&
                   ok(d->aux2, d->ldaux2) && ok(d->out1, d->ldout1) && ok(d->out2, d->ldout2);
  // (a contraction shorter than one 32-deep k-tile -- the input gradient of a 3- or 4-row head -- keeps the exact kernel, whose loads are
  // guarded element by element: the split kernels stage whole k-tiles)
  if (d->precision != NSKY_PREC_F32 && d->N > 64 && (d->K >= 32 || d->a_scale_max)) {
    NSKY_CHECK_ARG(d->precision == NSKY_PREC_BF16X2 || d->precision == NSKY_PREC_BF16X3 || d->precision == NSKY_PREC_F16X2,
                   "nsky_gemm_f32: unknown precision %d", d->precision);
    if (d->precision == NSKY_PREC_F16X2) launch_split<2, true>(d, e, splits, k_split_len, vec4, s);
    else if (d->precision == NSKY_PREC_BF16X2) launch_split<2>(d, e, splits, k_split_len, vec4, s);
    else launch_split<3>(d, e, splits, k_split_len, vec4, s);
    NSKY_CHECK_LAUNCH("nsky_gemm_f32(split-bf16)");
    return NSKY_OK;
  }
  if (d->N <= 32)
    launch<128, 32, 4, 1, 32, 2>(d, e, splits, k_split_len, vec4, s);
  else if (d->N <= 64)
    launch<128, 64, 2, 2, 32, 2>(d, e, splits, k_split_len, vec4, s);
  else
    launch<128, 128, 2, 2, 32, 2>(d, e, splits, k_split_len, vec4, s);
  NSKY_CHECK_LAUNCH("nsky_gemm_f32");
  return NSKY_OK;
}

extern "C" int nsky_split_planes(const float* W, int32_t n_rows, int32_t n_k, int32_t ldw, int32_t transpose, int32_t precision,
                                 uint16_t* hi, uint16_t* lo, int32_t rows_pad, int32_t ldp, nsky_stream_t stream) {
  NSKY_CHECK_ARG(W && hi && lo && n_rows > 0 && n_k > 0, "nsky_split_planes: bad arguments");
  NSKY_CHECK_ARG(rows_pad >= n_rows && rows_pad % 256 == 0 && ldp >= n_k && ldp % 32 == 0, "nsky_split_planes: planes must be padded to 256 rows x 32 k (rows_pad=%d ldp=%d)", rows_pad, ldp);
  NSKY_CHECK_ARG(ldw >= (transpose ? n_rows : n_k), "nsky_split_planes: ldw too small");
  NSKY_CHECK_ARG(precision == NSKY_PREC_F16X2 || precision == NSKY_PREC_BF16X2, "nsky_split_planes: precision must be F16X2 or BF16X2");
  dim3 grid(ldp / 32, rows_pad / 32);
  if (precision == NSKY_PREC_F16X2)
    hipLaunchKernelGGL((split_planes_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, W, n_rows, n_k, ldw, transpose, hi, lo, rows_pad, ldp);
  else
    hipLaunchKernelGGL((split_planes_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, W, n_rows, n_k, ldw, transpose, hi, lo, rows_pad, ldp);
  NSKY_CHECK_LAUNCH("nsky_split_planes");
  return NSKY_OK;
}

extern "C" int nsky_gemm_f32_planes(const nsky_gemm_desc* d, const uint16_t* B_hi, const uint16_t* B_lo, int32_t ldp, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && d->A && d->C && B_hi && B_lo, "nsky_gemm_f32_planes: null operand");
  NSKY_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->K % 4 == 0, "nsky_gemm_f32_planes: K=%d must be a positive multiple of 4", d->K);
  NSKY_CHECK_ARG(d->a_kcontig && d->lda >= d->K && d->lda % 4 == 0 && ((uintptr_t)d->A % 16) == 0, "nsky_gemm_f32_planes: A must be k-contiguous, 16-byte aligned, lda %% 4 == 0");
  NSKY_CHECK_ARG(ldp >= (d->K + 31) / 32 * 32 && ldp % 8 == 0 && ((uintptr_t)B_hi % 16) == 0 && ((uintptr_t)B_lo % 16) == 0, "nsky_gemm_f32_planes: planes must be 16-byte aligned with ldp %% 8 == 0 and cover K rounded up to 32");
  NSKY_CHECK_ARG(d->ldc >= d->N, "nsky_gemm_f32_planes: ldc < N");
  NSKY_CHECK_ARG(d->k_splits <= 1 && d->a_rowsum == nullptr, "nsky_gemm_f32_planes: no split-K / row sums");
  NSKY_CHECK_ARG(d->precision == NSKY_PREC_F16X2 || d->precision == NSKY_PREC_BF16X2, "nsky_gemm_f32_planes: precision must be F16X2 or BF16X2");
  switch (d->epi) {
    case NSKY_EPI_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1, "nsky_gemm_f32_planes: FILM needs aux0 (freq) and aux1 (phase)"); break;
    case NSKY_EPI_MUL_AUX: case NSKY_EPI_BWD_RELU: case NSKY_EPI_BWD_LEAKY: NSKY_CHECK_ARG(d->aux0, "nsky_gemm_f32_planes: epilogue needs aux0"); break;
    case NSKY_EPI_BWD_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1 && d->aux2 && d->out1 && d->out2, "nsky_gemm_f32_planes: BWD_FILM needs aux0..2, out1, out2"); break;
    default: break;
  }
  EpiCtx e;
  e.bias = d->bias; e.epi = d->epi; e.p0 = d->p0; e.p1 = d->p1;
  e.aux0 = d->aux0; e.ldaux0 = d->ldaux0; e.aux1 = d->aux1; e.ldaux1 = d->ldaux1; e.aux2 = d->aux2; e.ldaux2 = d->ldaux2;
  e.out1 = d->out1; e.ldout1 = d->ldout1; e.out2 = d->out2; e.ldout2 = d->ldout2;
  e.row_mod = d->row_mod > 0 ? d->row_mod : d->M;
  e.beta = d->beta;
  e.rs_limit = 0x7fffffff;
  e.a_nnt = 0; e.b_nnt = 0; e.a_scale_max = nullptr;
  NSKY_CHECK_ARG(d->a_native_nt == 0 && d->b_native_nt == 0 && d->a_scale_max == nullptr, "nsky_gemm_f32_planes: row-major operands only");
  auto ok = [](const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) % 16 == 0) && (ld % 4 == 0)); };
  const int vec4 = (d->N % 4 == 0) && ok(d->C, d->ldc) && ok(d->bias, 4) && ok(d->aux0, d->ldaux0) && ok(d->aux1, d->ldaux1) &&
                   ok(d->aux2, d->ldaux2) && ok(d->out1, d->ldout1) && ok(d->out2, d->ldout2);
  const dim3 grid(ceil_div(d->M, 128) * ceil_div(d->N, 128));
  if (d->precision == NSKY_PREC_F16X2)
    hipLaunchKernelGGL((gemm_planes_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, d->A, B_hi, B_lo, d->C, d->M, d->N, d->K, d->lda, ldp, d->ldc, vec4, e);
  else
    hipLaunchKernelGGL((gemm_planes_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, d->A, B_hi, B_lo, d->C, d->M, d->N, d->K, d->lda, ldp, d->ldc, vec4, e);
  NSKY_CHECK_LAUNCH("nsky_gemm_f32_planes");
  return NSKY_OK;
}

static int colsum_launch(const float* X, int32_t M, int32_t N, int32_t ldx, const float* w, int32_t w_stride, float* out,
                         hipStream_t stream) {
  if (N % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0) {
    const int rows_per_block = 256;  // >= 2 x 256 blocks for the [98 304 .. 393 216, 256] matrices of the step
    dim3 grid(ceil_div(N, 256), ceil_div(M, rows_per_block));
    hipLaunchKernelGGL(colsum4_kernel, grid, dim3(256), 0, stream, X, M, N, ldx, w, w_stride, out, rows_per_block);
    return 0;
  }
  const int rows_per_block = 512;
  dim3 grid(ceil_div(N, 64), ceil_div(M, rows_per_block));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, X, M, N, ldx, w, w_stride, out, rows_per_block);
  return 0;
}

extern "C" int nsky_colsum_f32(const float* X, int32_t M, int32_t N, int32_t ldx, float* out, nsky_stream_t stream) {
  NSKY_CHECK_ARG(X && out && M > 0 && N > 0 && ldx >= N, "nsky_colsum_f32: bad arguments");
  colsum_launch(X, M, N, ldx, nullptr, 0, out, (hipStream_t)stream);
  NSKY_CHECK_LAUNCH("nsky_colsum_f32");
  return NSKY_OK;
}

extern "C" int nsky_weighted_colsum_f32(const float* X, int32_t M, int32_t N, int32_t ldx, const float* w, int32_t w_stride,
                                        float* out, nsky_stream_t stream) {
  NSKY_CHECK_ARG(X && out && w && M > 0 && N > 0 && ldx >= N && w_stride >= 1, "nsky_weighted_colsum_f32: bad arguments");
  colsum_launch(X, M, N, ldx, w, w_stride, out, (hipStream_t)stream);
  NSKY_CHECK_LAUNCH("nsky_weighted_colsum_f32");
  return NSKY_OK;
}
