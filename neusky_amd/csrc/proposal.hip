// The proposal networks' density MLP (nerfstudio HashMLPDensityField: Linear(in, 16) + ReLU -> Linear(16, 1), called through
// ProposalNetworkSampler at neusky/models/neusky_model.py:561) on the hash-encoded rows of 1e5..3e5 sample points.  With 10 inputs, 16
// hidden units and one output a tensor-core GEMM per layer is all tile overhead (a 128 x 32 tile for N = 16 / N = 1, a [P, 16] hidden
// matrix through HBM, a split-K weight-gradient GEMM with K = P): here a lane carries a point through both layers in registers,
// forward (a lane per point, the weights from LDS as broadcast reads) and backward (sixteen lanes per point, below).
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

constexpr int PH = 16;   // hidden units
constexpr int PIN = 12;  // input columns held (in_dim <= 12)

struct ProposalMlp {
  const float* w0; int ldw0; const float* b0; const float* w1; const float* b1; int in_dim;
};

__device__ __forceinline__ void load_weights(const ProposalMlp& m, float (*w0)[PIN], float* b0, float* w1, float* b1, int tid, int nthreads) {
  for (int i = tid; i < PH * PIN; i += nthreads) {
    const int j = i / PIN, k = i % PIN;
    w0[j][k] = k < m.in_dim ? m.w0[(long)j * m.ldw0 + k] : 0.0f;
  }
  for (int i = tid; i < PH; i += nthreads) { b0[i] = m.b0[i]; w1[i] = m.w1[i]; }
  if (tid == 0) *b1 = m.b1[0];
}

__device__ __forceinline__ void load_row(const float* __restrict__ feat, int ldf, long p, int in_dim, float (&f)[PIN]) {
  const float* r = feat + p * ldf;
#pragma unroll
  for (int k = 0; k < PIN; ++k) f[k] = k < in_dim ? r[k] : 0.0f;
}

__global__ __launch_bounds__(256) void proposal_mlp_fwd_kernel(const float* __restrict__ feat, int ldf, long P, ProposalMlp m, float* __restrict__ raw) {
  __shared__ float w0[PH][PIN], b0[PH], w1[PH], b1;
  load_weights(m, w0, b0, w1, &b1, threadIdx.x, 256);
  __syncthreads();
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    float f[PIN];
    load_row(feat, ldf, p, m.in_dim, f);
    float out = b1;
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      float h = b0[j];
#pragma unroll
      for (int k = 0; k < PIN; ++k) h = fmaf(w0[j][k], f[k], h);
      out = fmaf(w1[j], fmaxf(h, 0.0f), out);
    }
    raw[p] = out;
  }
}

// Backward: SIXTEEN lanes per point (lane i of a group = hidden unit i), four points per wave step.  A lane keeps its unit's weight row
// in registers; the point's features (lane j of the group loads feature j: one coalesced row) reach every unit by 16-lane broadcasts;
// dW0 += dh^T f is then ONE v_mfma_f32_16x16x4_f32 per step with both operands already where the instruction wants them (A: unit i of
// point k in lane i + 16 k, B: feature j of point k in lane j + 16 k), exact fp32; d_feat = W0^T dh by four more of them.  The sums of a
// block meet in LDS and leave as one atomic per value and block.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void proposal_mlp_bwd_kernel(const float* __restrict__ feat, int ldf, long P, ProposalMlp m,
                                                               const float* __restrict__ d_raw, float* __restrict__ d_feat, float* __restrict__ dw0,
                                                               float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1) {
  constexpr int NV = PH * 16 + 2 * PH + 1;  // dW0 as a 16 x 16 tile | db0 | dW1 | db1
  __shared__ float red[NV];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, k = lane >> 4;
  for (int v = tid; v < NV; v += 256) red[v] = 0.0f;
  float wrow[PIN];
#pragma unroll
  for (int j = 0; j < PIN; ++j) wrow[j] = j < m.in_dim ? m.w0[(long)i * m.ldw0 + j] : 0.0f;
  const float bi = m.b0[i], w1i = m.w1[i];
  float w0t[4];  // A operand of the d_feat products: W0[4 mm + lane / 16][lane % 16]
#pragma unroll
  for (int mm = 0; mm < 4; ++mm) w0t[mm] = i < m.in_dim ? m.w0[(long)(4 * mm + k) * m.ldw0 + i] : 0.0f;
  __syncthreads();
  f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float ab0 = 0.0f, aw1 = 0.0f, ab1 = 0.0f;
  const long wave = (long)blockIdx.x * 4 + (tid >> 6), n_waves = (long)gridDim.x * 4;
  const long steps = (P + 3) / 4;
  // eight steps' loads are issued before their arithmetic (one wave per SIMD: a step is ~300 clocks of work behind a ~2000-clock load)
  constexpr int U = 8;
  for (long st0 = wave; st0 < steps; st0 += U * n_waves) {
    float f_[U], g_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long pn = 4 * (st0 + u * n_waves) + k;
      f_[u] = (pn < P && i < m.in_dim) ? feat[pn * ldf + i] : 0.0f;
      g_[u] = pn < P ? d_raw[pn] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long st = st0 + u * n_waves;
      if (st >= steps) break;
      const long p = 4 * st + k;
      const bool live = p < P;
      const float fval = f_[u], g = g_[u];
      float h = bi;
#pragma unroll
      for (int j = 0; j < PIN; ++j) h = fmaf(wrow[j], __shfl(fval, j, 16), h);
      const float dh = h > 0.0f ? g * w1i : 0.0f;
      ab0 += dh;
      aw1 = fmaf(g, fmaxf(h, 0.0f), aw1);
      if (i == 0) ab1 += g;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dh, fval, acc, 0, 0, 0);
      if (d_feat) {
        // d_feat^T [feature j][point n] = sum over units of W0[unit][j] dh_unit(point n): four more MFMAs (K = 4 units each), A = W0^T from
        // registers, B = the step's dh values moved to (lane % 16 = point, lane / 16 = unit within the K slice) by one permute each
        f32x4_t df = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          const float b = __shfl(dh, (4 * mm + k) + 16 * (i & 3), 64);
          df = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[mm], i < 4 ? b : 0.0f, df, 0, 0, 0);
        }
        // lane (c = i, q = k) holds features 4 q + r of point c (c < 4)
        const long pc = 4 * st + i;
        if (i < 4 && pc < P) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 4 * k + r;
            if (j < ldf) d_feat[pc * ldf + j] = j < m.in_dim ? df[r] : 0.0f;
          }
        }
      }
    }
  }
  // accumulator tile: lane (c = lane & 15, q = lane >> 4) holds rows 4 q + r, column c
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(&red[(4 * k + r) * 16 + i], acc[r]);
  atomicAdd(&red[PH * 16 + i], ab0);
  atomicAdd(&red[PH * 16 + PH + i], aw1);
  if (i == 0) atomicAdd(&red[NV - 1], ab1);
  __syncthreads();
  for (int v = tid; v < NV; v += 256) {
    const float sum = red[v];
    if (sum == 0.0f) continue;
    if (v < PH * 16) {
      const int j = v >> 4, c = v & 15;
      if (c < m.in_dim) atomicAdd(dw0 + (long)j * m.ldw0 + c, sum);
    } else if (v < PH * 16 + PH) {
      atomicAdd(db0 + (v - PH * 16), sum);
    } else if (v < PH * 16 + 2 * PH) {
      atomicAdd(dw1 + (v - PH * 16 - PH), sum);
    } else {
      atomicAdd(db1, sum);
    }
  }
}

int check_mlp(const float* feat, int32_t ldf, int64_t P, int32_t in_dim, int32_t hidden, const float* w0, int32_t ldw0, const float* b0,
              const float* w1, const float* b1, const char* who) {
  NSKY_CHECK_ARG(feat && w0 && b0 && w1 && b1 && P > 0, "%s: null argument", who);
  NSKY_CHECK_ARG(hidden == PH && in_dim >= 1 && in_dim <= PIN && ldf >= in_dim && ldw0 >= in_dim, "%s: hidden %d (16), in_dim %d (<= 12), ldf %d, ldw0 %d", who,
                 hidden, in_dim, ldf, ldw0);
  return NSKY_OK;
}

}  // namespace

extern "C" int nsky_proposal_mlp_fwd(const float* feat, int32_t ldf, int64_t P, int32_t in_dim, int32_t hidden, const float* w0, int32_t ldw0,
                                     const float* b0, const float* w1, const float* b1, float* raw, nsky_stream_t stream) {
  if (P == 0) return NSKY_OK;
  if (int rc = check_mlp(feat, ldf, P, in_dim, hidden, w0, ldw0, b0, w1, b1, "nsky_proposal_mlp_fwd")) return rc;
  NSKY_CHECK_ARG(raw, "nsky_proposal_mlp_fwd: null output");
  const ProposalMlp m{w0, ldw0, b0, w1, b1, in_dim};
  const long blocks = (P + 255) / 256;
  hipLaunchKernelGGL(proposal_mlp_fwd_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream, feat, ldf, (long)P, m, raw);
  NSKY_CHECK_LAUNCH("nsky_proposal_mlp_fwd");
  return NSKY_OK;
}

extern "C" int nsky_proposal_mlp_bwd(const float* feat, int32_t ldf, int64_t P, int32_t in_dim, int32_t hidden, const float* w0, int32_t ldw0,
                                     const float* b0, const float* w1, const float* b1, const float* d_raw, float* d_feat, float* dw0, float* db0,
                                     float* dw1, float* db1, nsky_stream_t stream) {
  if (P == 0) return NSKY_OK;
  if (int rc = check_mlp(feat, ldf, P, in_dim, hidden, w0, ldw0, b0, w1, b1, "nsky_proposal_mlp_bwd")) return rc;
  NSKY_CHECK_ARG(d_raw && dw0 && db0 && dw1 && db1, "nsky_proposal_mlp_bwd: null argument");
  const ProposalMlp m{w0, ldw0, b0, w1, b1, in_dim};
  NSKY_CHECK_ARG(!d_feat || ldf <= 16, "nsky_proposal_mlp_bwd: ldf %d (<= 16 with d_feat)", ldf);
  const long blocks = (P + 1023) / 1024;  // >= 64 steps of four points per wave; <= 512 blocks end with 289 atomics each
  hipLaunchKernelGGL(proposal_mlp_bwd_kernel, dim3((unsigned)(blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks))), dim3(256), 0, (hipStream_t)stream, feat,
                     ldf, (long)P, m, d_raw, d_feat, dw0, db0, dw1, db1);
  NSKY_CHECK_LAUNCH("nsky_proposal_mlp_bwd");
  return NSKY_OK;
}
