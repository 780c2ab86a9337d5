// Attention core of the RENI++ transformer decoder (the illumination model neusky/configs/neusky_config.py:78-95 configures:
// conditioning="Attention", VN invariance, SO2 about z, 8 heads x 6 layers, hidden 128; decoded at neusky/models/neusky_model.py:
// 488-506,535-549).  The `reni` package that holds the decoder is absent from the reference tree: this follows the published
// architecture as restated in oracle/neusky_oracle.py:reni_attention_decode and model_components/illumination.py:AttentionDecoder
// (PARITY UNPINNED).
//
// One head of one camera: the token t_n(d) = d_x A_n + d_y B_n + C_n is linear in (d_x, d_y), so with the per-camera key / value parts
// K~_n = [kA_n | kB_n | kC_n], V~_n = [vA_n | vB_n | vC_n] (3 x 16 = 48 wide) and the query q~ = [d_x q | d_y q | q]:
//     s_n = q~ . K~_n,   p = softmax_n(s),   o = d_x (p VA) + d_y (p VB) + (p VC)
// Exact fp32 arithmetic on the vector units (the products are [rows, 48] x [48, 100]: 4.8 k multiply-adds per row, head and product).
// Whatever is the same for every lane of a wave -- a token's K~ / V~ row in the row kernels, a row's q / dO in the token kernel -- is
// fetched by SCALAR loads (read-only kernel arguments, wave-uniform addresses) and enters the multiply-adds as the SGPR operand: no LDS
// traffic, no barriers (the first form read K~ / V~ from LDS by broadcast and was LDS-bandwidth bound: 1.2 / 1.4 / 2.1 ms per layer).
//   forward        : thread = (camera, direction) row; two passes over the tokens (row maximum, then exp / sum / value accumulation);
//                    saves the row maximum and sum
//   backward, rows : ds_n = p_n (dp_n - D), D = dO . O (the flash-attention identity, with the combined output), dq~ = sum_n ds_n K~_n
//   backward, tokens: thread = token of one (camera, head), K~_n, V~_n and their gradients in registers; the camera's rows are staged
//                    64 at a time in LDS by coalesced loads: dV~_n = sum_rows p_n dO~, dK~_n = sum_rows ds_n q~ -- no cross-lane reduction
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

constexpr int DH = 16, E = 3 * DH, ATT_THREADS = 128, ATT_LMAX = 128;

struct AttnArgs {
  const float* Q;      // [U, D, H]   H = n_heads * 16
  const float* dirs;   // [U, D, 3]
  const float* Kt;     // [U, n_heads, L, 48]
  const float* Vt;     // [U, n_heads, L, 48]
  float* O;            // [U, D, H]
  float* rmax;         // [U, n_heads, D]
  float* rsum;         // [U, n_heads, D]
  const float* dO;     // [U, D, H]
  float* dQ;           // [U, D, H]
  float* dKt;          // [U, n_heads, L, 48]
  float* dVt;
  int U, D, L, nh;
  float scale;
};

// q~ = [d_x q | d_y q | q] of this thread's row (q pre-multiplied by `scale`)
__device__ __forceinline__ void query_tilde(const AttnArgs& a, long row, int h, float (&qt)[E], float& dx, float& dy) {
  const float* qp = a.Q + row * (long)(a.nh * DH) + h * DH;
  float q[DH];
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const float4 v = reinterpret_cast<const float4*>(qp)[i];
    q[4 * i] = v.x * a.scale; q[4 * i + 1] = v.y * a.scale; q[4 * i + 2] = v.z * a.scale; q[4 * i + 3] = v.w * a.scale;
  }
  dx = a.dirs[row * 3];
  dy = a.dirs[row * 3 + 1];
#pragma unroll
  for (int j = 0; j < DH; ++j) { qt[j] = dx * q[j]; qt[DH + j] = dy * q[j]; qt[2 * DH + j] = q[j]; }
}

__device__ __forceinline__ float dot48(const float (&x)[E], const float* __restrict__ row) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int k = 0; k < E / 4; ++k) {
    s0 = fmaf(x[4 * k], row[4 * k], s0); s1 = fmaf(x[4 * k + 1], row[4 * k + 1], s1);
    s2 = fmaf(x[4 * k + 2], row[4 * k + 2], s2); s3 = fmaf(x[4 * k + 3], row[4 * k + 3], s3);
  }
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(ATT_THREADS) void attn_core_fwd_kernel(const AttnArgs a, const float* __restrict__ Kt, const float* __restrict__ Vt) {
  const int tid = threadIdx.x, h = blockIdx.y, u = blockIdx.z;
  const float* __restrict__ sK = Kt + ((long)u * a.nh + h) * a.L * E;
  const float* __restrict__ sV = Vt + ((long)u * a.nh + h) * a.L * E;
  const int d = blockIdx.x * ATT_THREADS + tid;
  const bool live = d < a.D;
  const long row = (long)u * a.D + (live ? d : a.D - 1);
  float qt[E], dx, dy;
  query_tilde(a, row, h, qt, dx, dy);
  float m = -3.0e38f;
  for (int n = 0; n < a.L; ++n) m = fmaxf(m, dot48(qt, sK + n * E));
  float l = 0.0f, o3[E];
#pragma unroll
  for (int k = 0; k < E; ++k) o3[k] = 0.0f;
  for (int n = 0; n < a.L; ++n) {
    const float p = expf(dot48(qt, sK + n * E) - m);
    l += p;
    const float* __restrict__ v = sV + n * E;
#pragma unroll
    for (int k = 0; k < E; ++k) o3[k] = fmaf(p, v[k], o3[k]);
  }
  if (!live) return;
  const float inv = 1.0f / l;
  float* op = a.O + row * (long)(a.nh * DH) + h * DH;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    float4 r;
    r.x = (dx * o3[4 * i] + dy * o3[DH + 4 * i] + o3[2 * DH + 4 * i]) * inv;
    r.y = (dx * o3[4 * i + 1] + dy * o3[DH + 4 * i + 1] + o3[2 * DH + 4 * i + 1]) * inv;
    r.z = (dx * o3[4 * i + 2] + dy * o3[DH + 4 * i + 2] + o3[2 * DH + 4 * i + 2]) * inv;
    r.w = (dx * o3[4 * i + 3] + dy * o3[DH + 4 * i + 3] + o3[2 * DH + 4 * i + 3]) * inv;
    reinterpret_cast<float4*>(op)[i] = r;
  }
  const long st = ((long)u * a.nh + h) * a.D + d;
  a.rmax[st] = m;
  a.rsum[st] = l;
}

__global__ __launch_bounds__(ATT_THREADS) void attn_core_bwd_rows_kernel(const AttnArgs a, const float* __restrict__ Kt, const float* __restrict__ Vt,
                                                                          float* __restrict__ drow) {
  const int tid = threadIdx.x, h = blockIdx.y, u = blockIdx.z;
  const float* __restrict__ sK = Kt + ((long)u * a.nh + h) * a.L * E;
  const float* __restrict__ sV = Vt + ((long)u * a.nh + h) * a.L * E;
  const int d = blockIdx.x * ATT_THREADS + tid;
  const bool live = d < a.D;
  const long row = (long)u * a.D + (live ? d : a.D - 1);
  float qt[E], dx, dy;
  query_tilde(a, row, h, qt, dx, dy);
  const long ho = row * (long)(a.nh * DH) + h * DH;
  float dO3[E], Drow = 0.0f;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const float4 g = reinterpret_cast<const float4*>(a.dO + ho)[i], o = reinterpret_cast<const float4*>(a.O + ho)[i];
    const float gg[4] = {g.x, g.y, g.z, g.w};
    Drow += g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) { dO3[4 * i + q] = dx * gg[q]; dO3[DH + 4 * i + q] = dy * gg[q]; dO3[2 * DH + 4 * i + q] = gg[q]; }
  }
  const long st = ((long)u * a.nh + h) * a.D + (live ? d : a.D - 1);
  const float m = a.rmax[st], inv = 1.0f / a.rsum[st];
  float dqt[E];
#pragma unroll
  for (int k = 0; k < E; ++k) dqt[k] = 0.0f;
  for (int n = 0; n < a.L; ++n) {
    const float* __restrict__ kr = sK + n * E;
    const float p = expf(dot48(qt, kr) - m) * inv;
    const float ds = p * (dot48(dO3, sV + n * E) - Drow);
#pragma unroll
    for (int k = 0; k < E; ++k) dqt[k] = fmaf(ds, kr[k], dqt[k]);
  }
  if (!live) return;
  drow[st] = Drow;  // D = dO . O of this row and head: the token kernel's ds = p (dp - D)
  float* qp = a.dQ + ho;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    float4 r;
    r.x = (dx * dqt[4 * i] + dy * dqt[DH + 4 * i] + dqt[2 * DH + 4 * i]) * a.scale;
    r.y = (dx * dqt[4 * i + 1] + dy * dqt[DH + 4 * i + 1] + dqt[2 * DH + 4 * i + 1]) * a.scale;
    r.z = (dx * dqt[4 * i + 2] + dy * dqt[DH + 4 * i + 2] + dqt[2 * DH + 4 * i + 2]) * a.scale;
    r.w = (dx * dqt[4 * i + 3] + dy * dqt[DH + 4 * i + 3] + dqt[2 * DH + 4 * i + 3]) * a.scale;
    reinterpret_cast<float4*>(qp)[i] = r;
  }
}

// thread = token n of (camera u, head h, row split z), K~_n, V~_n and their gradients in registers; the camera's rows come through LDS
// 64 at a time -- coalesced vector loads of q, dO (16 + 16 floats a row: the (d_x, d_y, 1) weights are applied to the 16-wide partial
// dot products, not to the operands) and of (d_x, d_y, maximum, 1 / sum, D), many in flight -- and are read back by broadcast:
//   s = d_x (q . kA) + d_y (q . kB) + q . kC,   dp = d_x (g . vA) + d_y (g . vB) + g . vC,   p = exp(s - max) / sum,   ds = p (dp - D)
//   dV~_n += p [d_x g | d_y g | g],   dK~_n += ds [d_x q | d_y q | q]           -- 9 LDS reads and 192 multiply-adds per row and lane
// (scalar loads of a row's q / dO straight from global memory, the second form, paid one HBM latency per row: 2.1 ms per layer)
constexpr int ATT_RB = 64;
__global__ __launch_bounds__(ATT_THREADS) void attn_core_bwd_tokens_kernel(const AttnArgs a, const float* __restrict__ drow, int splits) {
  __shared__ __attribute__((aligned(16))) float sQ[ATT_RB * DH];
  __shared__ __attribute__((aligned(16))) float sG[ATT_RB * DH];
  __shared__ __attribute__((aligned(16))) float sS[ATT_RB * 8];  // d_x, d_y, max, 1 / sum, D
  const int tid = threadIdx.x, h = blockIdx.x, u = blockIdx.y, z = blockIdx.z;
  const int n = tid;
  const bool tok = n < a.L;
  const long kv = (((long)u * a.nh + h) * a.L + (tok ? n : 0)) * E;
  float Kn[E], Vn[E], dK[E], dV[E];
#pragma unroll
  for (int k = 0; k < E / 4; ++k) {
    const float4 kk = reinterpret_cast<const float4*>(a.Kt + kv)[k], vv = reinterpret_cast<const float4*>(a.Vt + kv)[k];
    Kn[4 * k] = kk.x; Kn[4 * k + 1] = kk.y; Kn[4 * k + 2] = kk.z; Kn[4 * k + 3] = kk.w;
    Vn[4 * k] = vv.x; Vn[4 * k + 1] = vv.y; Vn[4 * k + 2] = vv.z; Vn[4 * k + 3] = vv.w;
  }
#pragma unroll
  for (int k = 0; k < E; ++k) { dK[k] = 0.0f; dV[k] = 0.0f; }
  const int per = (a.D + splits - 1) / splits;
  const int d_beg = z * per, d_end = min(a.D, d_beg + per);
  const int H = a.nh * DH;
  // staging role: row tid / 2 of the block, half tid % 2 of the 16 head features (two float4 each of q and dO)
  const int sr = tid >> 1, sh = tid & 1;
  for (int d0 = d_beg; d0 < d_end; d0 += ATT_RB) {
    __syncthreads();  // the previous block's rows are no longer read
    {
      const int d = d0 + sr;
      float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, g0 = q0, g1 = q0;
      float dx = 0.f, dy = 0.f, m = 0.f, inv = 0.f, Dr = 0.f;
      if (d < d_end) {
        const long row = (long)u * a.D + d;
        const long ho = row * H + h * DH + 8 * sh;
        q0 = *reinterpret_cast<const float4*>(a.Q + ho); q1 = *reinterpret_cast<const float4*>(a.Q + ho + 4);
        g0 = *reinterpret_cast<const float4*>(a.dO + ho); g1 = *reinterpret_cast<const float4*>(a.dO + ho + 4);
        if (sh == 0) {
          const long st = ((long)u * a.nh + h) * a.D + d;
          dx = a.dirs[row * 3]; dy = a.dirs[row * 3 + 1];
          m = a.rmax[st]; inv = 1.0f / a.rsum[st]; Dr = drow[st];
        }
      }
      *reinterpret_cast<float4*>(sQ + sr * DH + 8 * sh) = q0; *reinterpret_cast<float4*>(sQ + sr * DH + 8 * sh + 4) = q1;
      *reinterpret_cast<float4*>(sG + sr * DH + 8 * sh) = g0; *reinterpret_cast<float4*>(sG + sr * DH + 8 * sh + 4) = g1;
      if (sh == 0) {  // (a row beyond the range: 1 / sum = 0, so p = 0 and nothing is added)
        *reinterpret_cast<float4*>(sS + sr * 8) = make_float4(dx, dy, m, inv);
        sS[sr * 8 + 4] = Dr;
      }
    }
    __syncthreads();
    if (tok) {
      const int nr = min(ATT_RB, d_end - d0);
      for (int r = 0; r < nr; ++r) {
        float q[DH], g[DH];
#pragma unroll
        for (int i = 0; i < DH / 4; ++i) {
          const float4 qv = reinterpret_cast<const float4*>(sQ + r * DH)[i], gv = reinterpret_cast<const float4*>(sG + r * DH)[i];
          q[4 * i] = qv.x; q[4 * i + 1] = qv.y; q[4 * i + 2] = qv.z; q[4 * i + 3] = qv.w;
          g[4 * i] = gv.x; g[4 * i + 1] = gv.y; g[4 * i + 2] = gv.z; g[4 * i + 3] = gv.w;
        }
        const float4 st4 = *reinterpret_cast<const float4*>(sS + r * 8);
        const float dx = st4.x, dy = st4.y, m = st4.z, inv = st4.w, Dr = sS[r * 8 + 4];
        float sA = 0.f, sB = 0.f, sC = 0.f, pA = 0.f, pB = 0.f, pC = 0.f;
#pragma unroll
        for (int j = 0; j < DH; ++j) {
          sA = fmaf(q[j], Kn[j], sA); sB = fmaf(q[j], Kn[DH + j], sB); sC = fmaf(q[j], Kn[2 * DH + j], sC);
          pA = fmaf(g[j], Vn[j], pA); pB = fmaf(g[j], Vn[DH + j], pB); pC = fmaf(g[j], Vn[2 * DH + j], pC);
        }
        const float s = a.scale * fmaf(dx, sA, fmaf(dy, sB, sC));
        const float p = expf(s - m) * inv;
        const float ds = p * (fmaf(dx, pA, fmaf(dy, pB, pC)) - Dr) * a.scale;
        const float dsx = ds * dx, dsy = ds * dy, px = p * dx, py = p * dy;
#pragma unroll
        for (int j = 0; j < DH; ++j) {
          dK[j] = fmaf(dsx, q[j], dK[j]); dK[DH + j] = fmaf(dsy, q[j], dK[DH + j]); dK[2 * DH + j] = fmaf(ds, q[j], dK[2 * DH + j]);
          dV[j] = fmaf(px, g[j], dV[j]); dV[DH + j] = fmaf(py, g[j], dV[DH + j]); dV[2 * DH + j] = fmaf(p, g[j], dV[2 * DH + j]);
        }
      }
    }
  }
  if (tok) {
    if (splits == 1) {
#pragma unroll
      for (int k = 0; k < E / 4; ++k) {
        reinterpret_cast<float4*>(a.dKt + kv)[k] = make_float4(dK[4 * k], dK[4 * k + 1], dK[4 * k + 2], dK[4 * k + 3]);
        reinterpret_cast<float4*>(a.dVt + kv)[k] = make_float4(dV[4 * k], dV[4 * k + 1], dV[4 * k + 2], dV[4 * k + 3]);
      }
    } else {  // several workgroups share a (camera, head): the caller zero-filled dK~ / dV~
#pragma unroll
      for (int k = 0; k < E; ++k) { atomicAdd(a.dKt + kv + k, dK[k]); atomicAdd(a.dVt + kv + k, dV[k]); }
    }
  }
}

int check_attn(const char* who, int U, int D, int L, int nh, const void* const* ptrs, int np) {
  NSKY_CHECK_ARG(U >= 1 && D >= 1 && L >= 1 && L <= ATT_LMAX && L * E % 4 == 0 && nh >= 1 && nh <= 65535 && U <= 65535,
                 "%s: U %d, D %d, L %d (1..%d), heads %d", who, U, D, L, ATT_LMAX, nh);
  for (int i = 0; i < np; ++i) NSKY_CHECK_ARG(ptrs[i] && ((uintptr_t)ptrs[i] % 16) == 0, "%s: operand %d null / not 16-byte aligned", who, i);
  return NSKY_OK;
}

}  // namespace

extern "C" int nsky_attn_core_fwd(const float* Q, const float* dirs, const float* Kt, const float* Vt, int32_t U, int32_t D, int32_t L,
                                  int32_t n_heads, float scale, float* O, float* row_max, float* row_sum, nsky_stream_t stream) {
  const void* ptrs[] = {Q, Kt, Vt, O};
  if (int rc = check_attn("nsky_attn_core_fwd", U, D, L, n_heads, ptrs, 4)) return rc;
  NSKY_CHECK_ARG(dirs && row_max && row_sum, "nsky_attn_core_fwd: null operand");
  AttnArgs a{};
  a.Q = Q; a.dirs = dirs; a.Kt = Kt; a.Vt = Vt; a.O = O; a.rmax = row_max; a.rsum = row_sum;
  a.U = U; a.D = D; a.L = L; a.nh = n_heads; a.scale = scale;
  hipLaunchKernelGGL(attn_core_fwd_kernel, dim3(ceil_div(D, ATT_THREADS), n_heads, U), dim3(ATT_THREADS), 0, (hipStream_t)stream, a, Kt, Vt);
  NSKY_CHECK_LAUNCH("nsky_attn_core_fwd");
  return NSKY_OK;
}

extern "C" int nsky_attn_core_bwd(const float* Q, const float* dirs, const float* Kt, const float* Vt, const float* O, const float* row_max,
                                  const float* row_sum, const float* dO, int32_t U, int32_t D, int32_t L, int32_t n_heads, float scale,
                                  float* dQ, float* dKt, float* dVt, float* drow, nsky_stream_t stream) {
  const void* ptrs[] = {Q, Kt, Vt, O, dO, dQ, dKt, dVt};
  if (int rc = check_attn("nsky_attn_core_bwd", U, D, L, n_heads, ptrs, 8)) return rc;
  NSKY_CHECK_ARG(dirs && row_max && row_sum && drow, "nsky_attn_core_bwd: null operand");
  AttnArgs a{};
  a.Q = Q; a.dirs = dirs; a.Kt = Kt; a.Vt = Vt; a.O = const_cast<float*>(O); a.rmax = const_cast<float*>(row_max);
  a.rsum = const_cast<float*>(row_sum); a.dO = dO; a.dQ = dQ; a.dKt = dKt; a.dVt = dVt;
  a.U = U; a.D = D; a.L = L; a.nh = n_heads; a.scale = scale;
  // D = dO . O per row and head travels from the row kernel to the token kernel through the row_sum-shaped scratch behind dQ's last use:
  // it is stored in place of nothing the caller reads -- `drow` aliases no output: the caller passes it (see neusky_hip.h)
  hipLaunchKernelGGL(attn_core_bwd_rows_kernel, dim3(ceil_div(D, ATT_THREADS), n_heads, U), dim3(ATT_THREADS), 0, (hipStream_t)stream, a, Kt, Vt, drow);
  NSKY_CHECK_LAUNCH("nsky_attn_core_bwd (rows)");
  // one workgroup per (camera, head) leaves most of the chip idle when there are few cameras: split the rows, reduce by atomics
  int splits = 1;
  while (splits < 8 && (long)U * n_heads * splits < 2048 && D / (2 * splits) >= 64) splits *= 2;
  if (splits > 1) {
    (void)hipMemsetAsync(dKt, 0, (size_t)U * n_heads * L * E * sizeof(float), (hipStream_t)stream);
    (void)hipMemsetAsync(dVt, 0, (size_t)U * n_heads * L * E * sizeof(float), (hipStream_t)stream);
  }
  hipLaunchKernelGGL(attn_core_bwd_tokens_kernel, dim3(n_heads, U, splits), dim3(ATT_THREADS), 0, (hipStream_t)stream, a, (const float*)drow, splits);
  NSKY_CHECK_LAUNCH("nsky_attn_core_bwd (tokens)");
  return NSKY_OK;
}
