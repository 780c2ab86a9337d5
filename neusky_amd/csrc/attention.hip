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

// ---------------------------------------------------------------------------------------------------------------------
// The forward on the matrix cores (rows of a camera in blocks of 128: four waves x 32 rows; the VALU kernel above keeps the short
// blocks: a ray's own row, D < 32).  Both products of a head run as fp32-grade fp16 hi + residual splits (3 x v_mfma_f32_32x32x16_f16
// per product, the chain kernels' arithmetic), tokens on the M side:
//     S^T[token, row] = K~[token, 48] q~^T[48, row]      4 token tiles x 3 k-steps        A = K~ fragments (LDS), B = q~ planes (registers)
//     O~^T[dim, row]  = V~^T[dim, token] P^T[token, row]  2 dim tiles x 8 k-steps          A = V~^T fragments (LDS), B = the score
//                                                                                          accumulators themselves, exponentiated:
// the accumulator of a 32 x 32 tile holds, in lane (row c, half h) register 4 g + q, token 8 g + 4 h + q of row c -- eight consecutive
// registers are exactly the eight k-slots a B fragment of the next product wants (k-slot (h, j) <-> token 8 (j / 4) + 4 h + j % 4 of
// the k-step: the A fragments are packed with the same slot order), so the probabilities never leave the registers.  A row's maximum
// and sum are register reductions plus one exchange with lane c + 32.  A head's fragments are packed by the whole workgroup from the
// fp32 K~ / V~ (loaded one head ahead, into registers, under the previous head's products).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_WAVES = 4, MF_THREADS = 64 * MF_WAVES, MF_ROWS = 32 * MF_WAVES;
constexpr int MF_KF = 4 * 3, MF_VF = 2 * 8;  // fragments (hi + lo pairs) of K~ and of V~^T per head

__device__ __forceinline__ void split8h(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 xh = (_Float16)x[j];
    hi[j] = xh;
    lo[j] = (_Float16)(x[j] - (float)xh);
  }
}

// power-of-two scale s with m s in [2^14, 2^15) (the fp16 residual of every element within 2^-13 of the largest stays a normal
// number: the matrix cores flush fp16 subnormals), inv = 1 / s
__device__ __forceinline__ float pow2_scale(float m, float& inv) {
  if (!(m > 0.0f) || !(m < 3.0e38f)) { inv = 1.0f; return 1.0f; }
  int e;
  (void)frexpf(m, &e);
  e = max(-100, min(100, e));
  inv = ldexpf(1.0f, e - 15);
  return ldexpf(1.0f, 15 - e);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(MF_THREADS, 2) void attn_core_fwd_mfma_kernel(const AttnArgs a) {
  __shared__ f16x8 sK[MF_KF * 2 * 64];  // 24 KB: [token tile][k-step][hi, lo][lane]
  __shared__ f16x8 sV[MF_VF * 2 * 64];  // 32 KB: [dim tile][k-step][hi, lo][lane]
  extern __shared__ __attribute__((aligned(16))) float sR[];  // [16 ceil(L / 16)][48] (<= 24 KB): a head's V~ rows as they are in memory (transposed on the way into sV)
  __shared__ float sM[2][MF_WAVES][2];  // largest |K~|, |V~| of the head being packed, per wave (double-buffered by head parity)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h2 = lane >> 5;
  const int u = blockIdx.y;
  const int d = blockIdx.x * MF_ROWS + wave * 32 + c;
  const bool live = d < a.D;
  const long row = (long)u * a.D + (live ? d : a.D - 1);
  const int H = a.nh * DH;
  const float dx = a.dirs[row * 3], dy = a.dirs[row * 3 + 1];
  const int n_tt = (a.L + 31) >> 5, n_ks = (a.L + 15) >> 4;  // token tiles / value k-steps that hold a token
  const int nV4 = a.L * (E / 4);                             // float4 pieces of a head's V~ (<= 1536 = 6 per thread)
  // packing roles: K~ item i = tid + 256 i (i < 3): fragment item / 64 = (token tile, k-step), piece lane item % 64; V~: float4 piece
  // tid + 256 i (i < 6) of the head's rows.  Every load is unconditional, from a clamped address (a load inside a branch is waited for
  // on the spot); what lies beyond the tokens is zeroed when it is packed.
  float4 rk[3][2], rv[6];
  auto fetch = [&](int h) {
    const float* Ku = a.Kt + ((long)u * a.nh + h) * a.L * E;
    const float4* Vu = reinterpret_cast<const float4*>(a.Vt + ((long)u * a.nh + h) * a.L * E);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int item = tid + MF_THREADS * i, f = item >> 6, l = item & 63;
      const int token = 32 * (f / 3) + (l & 31), col = 16 * (f % 3) + 4 * (l >> 5);
      const float* kp = Ku + min(token, a.L - 1) * E + col;
      rk[i][0] = *reinterpret_cast<const float4*>(kp);
      rk[i][1] = *reinterpret_cast<const float4*>(kp + 8);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) rv[i] = Vu[min(tid + MF_THREADS * i, nV4 - 1)];
  };
  fetch(0);
  for (int h = 0; h < a.nh; ++h) {
    {  // largest magnitudes of this head's K~ and V~
      float mk = 0.0f, mv = 0.0f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int item = tid + MF_THREADS * i, f = item >> 6, l = item & 63;
        if (32 * (f / 3) + (l & 31) >= a.L) rk[i][0] = rk[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        mk = fmaxf(fmaxf(mk, fmaxf(fmaxf(fabsf(rk[i][0].x), fabsf(rk[i][0].y)), fmaxf(fabsf(rk[i][0].z), fabsf(rk[i][0].w)))),
                   fmaxf(fmaxf(fabsf(rk[i][1].x), fabsf(rk[i][1].y)), fmaxf(fabsf(rk[i][1].z), fabsf(rk[i][1].w))));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) mv = fmaxf(mv, fmaxf(fmaxf(fabsf(rv[i].x), fabsf(rv[i].y)), fmaxf(fabsf(rv[i].z), fabsf(rv[i].w))));
      mk = wave_max(mk); mv = wave_max(mv);
      if (lane == 0) { sM[h & 1][wave][0] = mk; sM[h & 1][wave][1] = mv; }
    }
    __syncthreads();  // the previous head's fragments and rows are no longer read; the maxima are in place
    float k_inv, v_inv;
    const float k_s = pow2_scale(fmaxf(fmaxf(sM[h & 1][0][0], sM[h & 1][1][0]), fmaxf(sM[h & 1][2][0], sM[h & 1][3][0])), k_inv);
    const float v_s = pow2_scale(fmaxf(fmaxf(sM[h & 1][0][1], sM[h & 1][1][1]), fmaxf(sM[h & 1][2][1], sM[h & 1][3][1])), v_inv);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int item = tid + MF_THREADS * i, f = item >> 6, l = item & 63;
      const float x[8] = {rk[i][0].x * k_s, rk[i][0].y * k_s, rk[i][0].z * k_s, rk[i][0].w * k_s, rk[i][1].x * k_s, rk[i][1].y * k_s, rk[i][1].z * k_s, rk[i][1].w * k_s};
      f16x8 hi, lo;
      split8h(x, hi, lo);
      sK[(2 * f) * 64 + l] = hi;
      sK[(2 * f + 1) * 64 + l] = lo;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int piece = tid + MF_THREADS * i;
      if (piece < 16 * n_ks * (E / 4)) reinterpret_cast<float4*>(sR)[piece] = piece < nV4 ? make_float4(rv[i].x * v_s, rv[i].y * v_s, rv[i].z * v_s, rv[i].w * v_s) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (h + 1 < a.nh) fetch(h + 1);  // in flight under this head's products
    // V~^T fragments: item = (dim tile, k-step) x lane (dim r, half hh): the eight tokens 16 ks + 8 (j / 4) + 4 hh + j % 4 of dim 32 dt + r
    // (lanes along the dims: consecutive LDS banks)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int item = tid + MF_THREADS * i, f = item >> 6, l = item & 63;
      const int dim = 32 * (f >> 3) + (l & 31), t0 = 16 * (f & 7) + 4 * (l >> 5);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (dim < E && (f & 7) < n_ks) ? sR[(t0 + 8 * (j >> 2) + (j & 3)) * E + dim] : 0.0f;
      f16x8 hi, lo;
      split8h(x, hi, lo);
      sV[(2 * f) * 64 + l] = hi;
      sV[(2 * f + 1) * 64 + l] = lo;
    }
    __syncthreads();
    // ---- q~ planes: k-step p = part p of [d_x q | d_y q | q]; k-slot (h2, j) <-> feature 8 (j / 4) + 4 h2 + j % 4 of the head
    f16x8 qh[3], ql[3];
    float q_inv;
    {
      const float* qp = a.Q + row * (long)H + h * DH + 4 * h2;
      const float4 qa = *reinterpret_cast<const float4*>(qp), qb = *reinterpret_cast<const float4*>(qp + 8);
      const float q8[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
      float qm = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) qm = fmaxf(qm, fabsf(q8[j]));
      qm = fmaxf(qm, __shfl_xor(qm, 32, 64));
      const float q_s = pow2_scale(qm, q_inv);  // |d_x|, |d_y| <= 1: the plain part is the largest
      const float coef[3] = {dx * q_s, dy * q_s, q_s};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // d_x q is not exact in fp32: the ROUNDED product is what gets split.  Left to itself the compiler forms the high half from
          // the rounded product (v_cvt) and the residual from the exact one (v_fma_mix), or the other way round: where the two
          // roundings disagree (one element in 2^14) the pair is off by an fp16 ulp, 2^-11 of the element
          x[j] = q8[j] * coef[p];
          asm volatile("" : "+v"(x[j]));
        }
        split8h(x, qh[p], ql[p]);
      }
    }
    // ---- scores (token tiles without a token are skipped: wave-uniform branches)
    const float s_inv = k_inv * q_inv * a.scale;  // accumulator -> score
    f32x16 s[4];
    float m = -3.0e38f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tt][r] = 0.0f;
      if (tt < n_tt) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const f16x8 ah = sK[(2 * (tt * 3 + ks)) * 64 + lane], al = sK[(2 * (tt * 3 + ks) + 1) * 64 + lane];
          s[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], s[tt], 0, 0, 0);
          s[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], s[tt], 0, 0, 0);
          s[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], s[tt], 0, 0, 0);
        }
        // register 4 g + q: token 32 tt + 8 g + 4 h2 + q
        if (32 * tt + 32 <= a.L) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[tt][r] *= s_inv; m = fmaxf(m, s[tt][r]); }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[tt][r] = 32 * tt + 8 * (r >> 2) + 4 * h2 + (r & 3) < a.L ? s[tt][r] * s_inv : -3.0e38f;
            m = fmaxf(m, s[tt][r]);
          }
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float lsum = 0.0f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      if (tt < n_tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __expf(s[tt][r] - m);  // (a masked score: exp(-3e38 - m) = 0)
          lsum += p;
          s[tt][r] = p * 16384.0f;  // p <= 1: the fp16 pair resolves 2^-24 of 2^14 p
        }
      }
    lsum += __shfl_xor(lsum, 32, 64);
    // ---- O~^T = V~^T P^T: the exponentiated accumulators are the B planes
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < n_ks) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = s[ks >> 1][8 * (ks & 1) + j];
        f16x8 ph, pl;
        split8h(x, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f16x8 ah = sV[(2 * (dt * 8 + ks)) * 64 + lane], al = sV[(2 * (dt * 8 + ks) + 1) * 64 + lane];
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph, o[dt], 0, 0, 0);
        }
      }
    // ---- combine the three parts: feature e = 8 g + 4 h2 + q (g = 0, 1) <- d_x O~[e] + d_y O~[16 + e] + O~[32 + e]
    if (live) {
      const float inv = v_inv / (16384.0f * lsum);
      float* op = a.O + row * (long)H + h * DH + 4 * h2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float4 r;
        r.x = (dx * o[0][4 * g] + dy * o[0][4 * (g + 2)] + o[1][4 * g]) * inv;
        r.y = (dx * o[0][4 * g + 1] + dy * o[0][4 * (g + 2) + 1] + o[1][4 * g + 1]) * inv;
        r.z = (dx * o[0][4 * g + 2] + dy * o[0][4 * (g + 2) + 2] + o[1][4 * g + 2]) * inv;
        r.w = (dx * o[0][4 * g + 3] + dy * o[0][4 * (g + 2) + 3] + o[1][4 * g + 3]) * inv;
        *reinterpret_cast<float4*>(op + 8 * g) = r;
      }
      if (h2 == 0) {
        const long st = ((long)u * a.nh + h) * a.D + d;
        a.rmax[st] = m;
        a.rsum[st] = lsum;
      }
    }
  }
}

// The row side of the backward on the matrix cores, 256 rows of a camera per workgroup (eight waves x 32 rows):
//     S^T  = K~ q~^T,   dP^T = V~ dO~^T          tokens on the M side, k = the 48 dims (dO~ = [d_x dO | d_y dO | dO])
//     dS   = P (dP - D) with P = exp(S - max) / sum from the forward's row statistics, D = dO . O
//     dq~^T = K~^T dS^T                           dims on the M side, k = tokens: dS goes from the accumulators into the B planes
// A head needs K~ twice (token-major and dim-major fragments) and V~ token-major: 80 KB of fragments + the rows of K~ being transposed.
constexpr int MB_WAVES = 8, MB_THREADS = 64 * MB_WAVES, MB_ROWS = 32 * MB_WAVES;

__device__ __forceinline__ void publish_absmax(float* slot, float v, int lane) {  // v >= 0: the bit patterns order like the values
  v = wave_max(v);
  if (lane == 0 && v > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}

__global__ __launch_bounds__(MB_THREADS, 1) void attn_core_bwd_rows_mfma_kernel(const AttnArgs a, float* __restrict__ drow, float* __restrict__ stat) {
  __shared__ f16x8 sKt[MF_KF * 2 * 64];  // 24 KB: K~ [token tile][k-step of dims][hi, lo][lane]
  __shared__ f16x8 sVt[MF_KF * 2 * 64];  // 24 KB: V~, the same form
  __shared__ f16x8 sKd[MF_VF * 2 * 64];  // 32 KB: K~^T [dim tile][k-step of tokens][hi, lo][lane]
  extern __shared__ __attribute__((aligned(16))) float sR[];  // [16 ceil(L / 16)][48]: K~ rows on their way into sKd
  __shared__ float sM[2][MB_WAVES][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h2 = lane >> 5;
  const int u = blockIdx.y;
  const int d = blockIdx.x * MB_ROWS + wave * 32 + c;
  const bool live = d < a.D;
  const long row = (long)u * a.D + (live ? d : a.D - 1);
  const int H = a.nh * DH;
  const float dx = a.dirs[row * 3], dy = a.dirs[row * 3 + 1];
  const int n_tt = (a.L + 31) >> 5, n_ks = (a.L + 15) >> 4;
  const int nK4 = a.L * (E / 4);
  // token-major items (12 fragments x 64 lanes = 768): item tid, and item 512 + tid for the first four waves
  float4 rk[2][2], rv[2][2], rr[3];
  auto fetch = [&](int h) {
    const float* Ku = a.Kt + ((long)u * a.nh + h) * a.L * E;
    const float* Vu = a.Vt + ((long)u * a.nh + h) * a.L * E;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int item = min(tid + MB_THREADS * i, MF_KF * 64 - 1), f = item >> 6, l = item & 63;
      const int token = 32 * (f / 3) + (l & 31), col = 16 * (f % 3) + 4 * (l >> 5);
      const long off = (long)min(token, a.L - 1) * E + col;
      rk[i][0] = *reinterpret_cast<const float4*>(Ku + off); rk[i][1] = *reinterpret_cast<const float4*>(Ku + off + 8);
      rv[i][0] = *reinterpret_cast<const float4*>(Vu + off); rv[i][1] = *reinterpret_cast<const float4*>(Vu + off + 8);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) rr[i] = reinterpret_cast<const float4*>(Ku)[min(tid + MB_THREADS * i, nK4 - 1)];
  };
  auto max4 = [](float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  fetch(0);
  for (int h = 0; h < a.nh; ++h) {
    {
      float mk = 0.0f, mv = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int item = tid + MB_THREADS * i, f = min(item, MF_KF * 64 - 1) >> 6;
        if (item >= MF_KF * 64 || 32 * (f / 3) + (item & 31) >= a.L) rk[i][0] = rk[i][1] = rv[i][0] = rv[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        mk = fmaxf(mk, fmaxf(max4(rk[i][0]), max4(rk[i][1])));
        mv = fmaxf(mv, fmaxf(max4(rv[i][0]), max4(rv[i][1])));
      }
      mk = wave_max(mk); mv = wave_max(mv);
      if (lane == 0) { sM[h & 1][wave][0] = mk; sM[h & 1][wave][1] = mv; }
    }
    __syncthreads();
    float k_inv, v_inv, mk = 0.0f, mv = 0.0f;
#pragma unroll
    for (int w = 0; w < MB_WAVES; ++w) { mk = fmaxf(mk, sM[h & 1][w][0]); mv = fmaxf(mv, sM[h & 1][w][1]); }
    const float k_s = pow2_scale(mk, k_inv), v_s = pow2_scale(mv, v_inv);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int item = tid + MB_THREADS * i, f = item >> 6, l = item & 63;
      if (item < MF_KF * 64) {
        const float xk[8] = {rk[i][0].x * k_s, rk[i][0].y * k_s, rk[i][0].z * k_s, rk[i][0].w * k_s, rk[i][1].x * k_s, rk[i][1].y * k_s, rk[i][1].z * k_s, rk[i][1].w * k_s};
        const float xv[8] = {rv[i][0].x * v_s, rv[i][0].y * v_s, rv[i][0].z * v_s, rv[i][0].w * v_s, rv[i][1].x * v_s, rv[i][1].y * v_s, rv[i][1].z * v_s, rv[i][1].w * v_s};
        f16x8 hi, lo;
        split8h(xk, hi, lo);
        sKt[(2 * f) * 64 + l] = hi; sKt[(2 * f + 1) * 64 + l] = lo;
        split8h(xv, hi, lo);
        sVt[(2 * f) * 64 + l] = hi; sVt[(2 * f + 1) * 64 + l] = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = tid + MB_THREADS * i;
      if (piece < 16 * n_ks * (E / 4)) reinterpret_cast<float4*>(sR)[piece] = piece < nK4 ? make_float4(rr[i].x * k_s, rr[i].y * k_s, rr[i].z * k_s, rr[i].w * k_s) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (h + 1 < a.nh) fetch(h + 1);
    // K~^T fragments: 16 x 64 = 1024 items, two per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int item = tid + MB_THREADS * i, f = item >> 6, l = item & 63;
      const int dim = 32 * (f >> 3) + (l & 31), t0 = 16 * (f & 7) + 4 * (l >> 5);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (dim < E && (f & 7) < n_ks) ? sR[(t0 + 8 * (j >> 2) + (j & 3)) * E + dim] : 0.0f;
      f16x8 hi, lo;
      split8h(x, hi, lo);
      sKd[(2 * f) * 64 + l] = hi; sKd[(2 * f + 1) * 64 + l] = lo;
    }
    __syncthreads();
    // ---- this row's operands: q~ and dO~ planes (k-slot (h2, j) <-> feature 8 (j / 4) + 4 h2 + j % 4), D = dO . O, the forward's statistics
    f16x8 qh[3], ql[3], gh[3], gl[3];
    float q_inv, g_inv, Drow;
    {
      const long ho = row * (long)H + h * DH + 4 * h2;
      const float4 qa = *reinterpret_cast<const float4*>(a.Q + ho), qb = *reinterpret_cast<const float4*>(a.Q + ho + 8);
      const float4 ga = *reinterpret_cast<const float4*>(a.dO + ho), gb = *reinterpret_cast<const float4*>(a.dO + ho + 8);
      const float4 oa = *reinterpret_cast<const float4*>(a.O + ho), ob = *reinterpret_cast<const float4*>(a.O + ho + 8);
      const float q8[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w}, g8[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
      Drow = (ga.x * oa.x + ga.y * oa.y + ga.z * oa.z + ga.w * oa.w) + (gb.x * ob.x + gb.y * ob.y + gb.z * ob.z + gb.w * ob.w);
      Drow += __shfl_xor(Drow, 32, 64);
      float qm = 0.0f, gm = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { qm = fmaxf(qm, fabsf(q8[j])); gm = fmaxf(gm, fabsf(g8[j])); }
      qm = fmaxf(qm, __shfl_xor(qm, 32, 64)); gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
      // largest |q|, |dO| (and below |dS|) of this (camera, head): the token kernel scales its operands with them
      publish_absmax(stat + ((long)u * a.nh + h) * 4, live ? qm : 0.0f, lane);
      publish_absmax(stat + ((long)u * a.nh + h) * 4 + 1, live ? gm : 0.0f, lane);
      const float q_s = pow2_scale(qm, q_inv), g_s = pow2_scale(gm, g_inv);
      const float cq[3] = {dx * q_s, dy * q_s, q_s}, cg[3] = {dx * g_s, dy * g_s, g_s};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // (the rounded products are what gets split: see the forward)
          x[j] = q8[j] * cq[p]; y[j] = g8[j] * cg[p];
          asm volatile("" : "+v"(x[j]), "+v"(y[j]));
        }
        split8h(x, qh[p], ql[p]);
        split8h(y, gh[p], gl[p]);
      }
    }
    const long st = ((long)u * a.nh + h) * a.D + (live ? d : a.D - 1);
    const float m = a.rmax[st], l_inv = 1.0f / a.rsum[st];
    const float s_inv = k_inv * q_inv * a.scale, dp_inv = v_inv * g_inv;
    // ---- dS, tile by tile (register 4 g + q: token 32 tt + 8 g + 4 h2 + q)
    f32x16 ds[4];
    float dsm = 0.0f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[tt][r] = 0.0f;
      if (tt < n_tt) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const f16x8 kh = sKt[(2 * (tt * 3 + ks)) * 64 + lane], kl = sKt[(2 * (tt * 3 + ks) + 1) * 64 + lane];
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
          const f16x8 vh = sVt[(2 * (tt * 3 + ks)) * 64 + lane], vl = sVt[(2 * (tt * 3 + ks) + 1) * 64 + lane];
          dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, gl[ks], dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, gh[ks], dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, gh[ks], dp, 0, 0, 0);
        }
        const bool whole = 32 * tt + 32 <= a.L;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __expf(s[r] * s_inv - m) * l_inv;
          float v = p * (dp[r] * dp_inv - Drow);
          if (!whole) v = 32 * tt + 8 * (r >> 2) + 4 * h2 + (r & 3) < a.L ? v : 0.0f;
          ds[tt][r] = v;
          dsm = fmaxf(dsm, fabsf(v));
        }
      }
    }
    dsm = fmaxf(dsm, __shfl_xor(dsm, 32, 64));
    publish_absmax(stat + ((long)u * a.nh + h) * 4 + 2, live ? dsm : 0.0f, lane);
    float ds_inv;
    const float ds_s = pow2_scale(dsm, ds_inv);
    // ---- dq~^T = K~^T dS^T
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < n_ks) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = ds[ks >> 1][8 * (ks & 1) + j] * ds_s;
        f16x8 ph, pl;
        split8h(x, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f16x8 ah = sKd[(2 * (dt * 8 + ks)) * 64 + lane], al = sKd[(2 * (dt * 8 + ks) + 1) * 64 + lane];
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph, o[dt], 0, 0, 0);
        }
      }
    if (live) {
      const float inv = k_inv * ds_inv * a.scale;
      float* qp = a.dQ + row * (long)H + h * DH + 4 * h2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float4 r;
        r.x = (dx * o[0][4 * g] + dy * o[0][4 * (g + 2)] + o[1][4 * g]) * inv;
        r.y = (dx * o[0][4 * g + 1] + dy * o[0][4 * (g + 2) + 1] + o[1][4 * g + 1]) * inv;
        r.z = (dx * o[0][4 * g + 2] + dy * o[0][4 * (g + 2) + 2] + o[1][4 * g + 2]) * inv;
        r.w = (dx * o[0][4 * g + 3] + dy * o[0][4 * (g + 2) + 3] + o[1][4 * g + 3]) * inv;
        *reinterpret_cast<float4*>(qp + 8 * g) = r;
      }
      if (h2 == 0) drow[st] = Drow;
    }
  }
}

// The token side on the matrix cores: one workgroup per (camera, head), wave w owns the token tile 32 w .. 32 w + 31 and walks the
// camera's rows 32 at a time.  Swapping the operands of the score product gives the TRANSPOSED accumulator for free (fragment layouts of
// the two MFMA inputs are the same: lane = the outer index, eight k-slots in registers):
//     S  = q~ K~^T,   dP = dO~ V~^T        A = the row tile's q~ / dO~ fragments (LDS), B = this wave's K~ / V~ fragments (registers):
//                                          lane = token, register 4 g + q = row 8 g + 4 h2 + q
//     dV~ += P^T dO~,   dK~ += dS^T q~     A = the accumulators themselves (lane = token, k-slots = rows), B = dO~ / q~ with the rows on k
// The sums over the camera's rows stay in accumulators (2 x 2 tiles per wave): no atomics, no cross-lane reduction, a plain store at the
// end.  Operand scales come from the row kernel's published maxima (uniform over the rows, as the accumulation requires).
constexpr int MT_WAVES = 4, MT_THREADS = 64 * MT_WAVES;

__global__ __launch_bounds__(MT_THREADS, 2) void attn_core_bwd_tokens_mfma_kernel(const AttnArgs a, const float* __restrict__ drow, const float* __restrict__ stat) {
  __shared__ f16x8 sQA[3 * 2 * 64], sGA[3 * 2 * 64];  // q~ / dO~ of the row tile, rows on the lanes: [part][hi, lo][lane]        6 KB each
  __shared__ f16x8 sQB[4 * 2 * 64], sGB[4 * 2 * 64];  // the same with the rows on k: [dim tile][16-row k-step][hi, lo][lane]     8 KB each
  // the row tile's q / dO rows and statistics, double-buffered: tile i + 1 is stored while tile i is being multiplied
  __shared__ __attribute__((aligned(16))) float sRawQ[2][32 * DH], sRawG[2][32 * DH];
  __shared__ __attribute__((aligned(16))) float sDx[2][32], sDy[2][32], sMx[2][32], sLi[2][32], sDr[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h2 = lane >> 5;
  const int h = blockIdx.x, u = blockIdx.y;
  const int H = a.nh * DH;
  const int token = 32 * wave + n;
  const bool has_tokens = 32 * wave < a.L;
  const float* st4 = stat + ((long)u * a.nh + h) * 4;
  float q_inv, g_inv, ds_inv;
  const float q_s = pow2_scale(st4[0], q_inv), g_s = pow2_scale(st4[1], g_inv), ds_s = pow2_scale(st4[2], ds_inv);
  // ---- this wave's K~ / V~ fragments (B operand: lane = token, k-slot (h2, j) <-> dim 16 p + 8 (j / 4) + 4 h2 + j % 4), scaled per wave
  f16x8 kh[3], kl[3], vh[3], vl[3];
  float k_inv, v_inv;
  {
    const long off = (((long)u * a.nh + h) * a.L + min(token, a.L - 1)) * E + 4 * h2;
    float4 kr[3][2], vr[3][2];
    float mk = 0.0f, mv = 0.0f;
    auto max4 = [](float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      kr[p][0] = *reinterpret_cast<const float4*>(a.Kt + off + 16 * p); kr[p][1] = *reinterpret_cast<const float4*>(a.Kt + off + 16 * p + 8);
      vr[p][0] = *reinterpret_cast<const float4*>(a.Vt + off + 16 * p); vr[p][1] = *reinterpret_cast<const float4*>(a.Vt + off + 16 * p + 8);
      if (token >= a.L) kr[p][0] = kr[p][1] = vr[p][0] = vr[p][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      mk = fmaxf(mk, fmaxf(max4(kr[p][0]), max4(kr[p][1])));
      mv = fmaxf(mv, fmaxf(max4(vr[p][0]), max4(vr[p][1])));
    }
    const float k_s = pow2_scale(wave_max(mk), k_inv), v_s = pow2_scale(wave_max(mv), v_inv);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const float xk[8] = {kr[p][0].x * k_s, kr[p][0].y * k_s, kr[p][0].z * k_s, kr[p][0].w * k_s, kr[p][1].x * k_s, kr[p][1].y * k_s, kr[p][1].z * k_s, kr[p][1].w * k_s};
      const float xv[8] = {vr[p][0].x * v_s, vr[p][0].y * v_s, vr[p][0].z * v_s, vr[p][0].w * v_s, vr[p][1].x * v_s, vr[p][1].y * v_s, vr[p][1].z * v_s, vr[p][1].w * v_s};
      split8h(xk, kh[p], kl[p]);
      split8h(xv, vh[p], vl[p]);
    }
  }
  const float s_inv = k_inv * q_inv * a.scale, dp_inv = v_inv * g_inv;
  f32x16 accK[2], accV[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accK[dt][r] = 0.0f; accV[dt][r] = 0.0f; }
  // staging roles: thread t < 128: float4 (t & 3) of row t >> 2 of Q; t >= 128: of dO; threads 0 .. 31 also the row's statistics
  const int s_row = (tid & 127) >> 2, s_q4 = tid & 3;
  const float* s_src = (tid < 128 ? a.Q : a.dO) + h * DH + 4 * s_q4;
  const int s_off = s_row * DH + 4 * s_q4;
  float4 raw;
  float r_dx, r_dy, r_m, r_li, r_dr;
  auto fetch = [&](int d0) {
    const int d = d0 + s_row;
    raw = *reinterpret_cast<const float4*>(s_src + ((long)u * a.D + min(d, a.D - 1)) * H);
    if (d >= a.D) raw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) {
      const int dd = d0 + tid;
      const long row = (long)u * a.D + min(dd, a.D - 1);
      const long st = ((long)u * a.nh + h) * a.D + min(dd, a.D - 1);
      r_dx = a.dirs[row * 3]; r_dy = a.dirs[row * 3 + 1]; r_m = a.rmax[st]; r_li = 1.0f / a.rsum[st]; r_dr = drow[st];
      if (dd >= a.D) { r_li = 0.0f; r_dr = 0.0f; }  // a row beyond the camera's: p = 0, dS = 0
    }
  };
  auto stash = [&](int b) {  // the fetched row tile into buffer b
    *reinterpret_cast<float4*>((tid < 128 ? sRawQ[b] : sRawG[b]) + s_off) = raw;
    if (tid < 32) { sDx[b][tid] = r_dx; sDy[b][tid] = r_dy; sMx[b][tid] = r_m; sLi[b][tid] = r_li; sDr[b][tid] = r_dr; }
  };
  fetch(0);
  stash(0);
  if (32 < a.D) fetch(32);
  for (int d0 = 0, b = 0; d0 < a.D; d0 += 32, b ^= 1) {
    __syncthreads();  // the previous row tile's fragments are no longer read; buffer b is complete
    // ---- fragments of the row tile: 14 x 64 lane-items (3 + 3 with the rows on the lanes, 4 + 4 with the rows on k), waves take them in turn
    for (int f = wave; f < 14; f += MT_WAVES) {
      const bool isq = f < 3 || (f >= 6 && f < 10);
      const float* rawp = isq ? sRawQ[b] : sRawG[b];
      const float sc = isq ? q_s : g_s;
      float x[8];
      f16x8* dst;
      if (f < 6) {  // rows on the lanes: lane (row n, half h2), part p: features 8 (j / 4) + 4 h2 + j % 4 of the row, times (d_x, d_y, 1)
        const int p = f < 3 ? f : f - 3;
        const float cf = (p == 0 ? sDx[b][n] : (p == 1 ? sDy[b][n] : 1.0f)) * sc;
        const float4 v0 = *reinterpret_cast<const float4*>(rawp + n * DH + 4 * h2), v1 = *reinterpret_cast<const float4*>(rawp + n * DH + 8 + 4 * h2);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v[j] * cf;
        dst = (f < 3 ? sQA : sGA) + (2 * p) * 64;
      } else {  // rows on k: lane (dim n of tile dt, half h2), k-step kr: rows 16 kr + 8 (j / 4) + 4 h2 + j % 4
        const int g = f < 10 ? f - 6 : f - 10, dt = g >> 1, kr = g & 1;
        const int dim = 32 * dt + n, p = dim >> 4, e = dim & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 16 * kr + 8 * (j >> 2) + 4 * h2 + (j & 3);
          const float cf = (p == 0 ? sDx[b][r] : (p == 1 ? sDy[b][r] : 1.0f)) * sc;
          x[j] = dim < E ? rawp[r * DH + e] * cf : 0.0f;
        }
        dst = (f < 10 ? sQB : sGB) + (2 * g) * 64;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));  // the rounded product is what gets split (see the forward)
      f16x8 hi, lo;
      split8h(x, hi, lo);
      dst[lane] = hi;
      dst[64 + lane] = lo;
    }
    __syncthreads();  // the fragments are in place; nobody reads buffer b ^ 1 any more (its tile was packed an iteration ago)
    if (d0 + 32 < a.D) {
      stash(b ^ 1);
      if (d0 + 64 < a.D) fetch(d0 + 64);
    }
    if (has_tokens) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const f16x8 qh = sQA[(2 * p) * 64 + lane], ql = sQA[(2 * p + 1) * 64 + lane];
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kl[p], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql, kh[p], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kh[p], s, 0, 0, 0);
        const f16x8 gh = sGA[(2 * p) * 64 + lane], gl = sGA[(2 * p + 1) * 64 + lane];
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, vl[p], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, vh[p], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, vh[p], dp, 0, 0, 0);
      }
      // register 4 g + q: row 8 g + 4 h2 + q of the tile; this lane's token
      const float tok_live = token < a.L ? 1.0f : 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = *reinterpret_cast<const float4*>(sMx[b] + 8 * g + 4 * h2), l4 = *reinterpret_cast<const float4*>(sLi[b] + 8 * g + 4 * h2),
                     d4 = *reinterpret_cast<const float4*>(sDr[b] + 8 * g + 4 * h2);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float p = __expf(s[4 * g + q] * s_inv - mm[q]) * (ll[q] * tok_live);
          dp[4 * g + q] = p * (dp[4 * g + q] * dp_inv - dd[q]) * ds_s;  // dS, scaled
          s[4 * g + q] = p * 16384.0f;                                  // P, scaled
        }
      }
#pragma unroll
      for (int kr = 0; kr < 2; ++kr) {
        float xp[8], xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xp[j] = s[8 * kr + j]; xs[j] = dp[8 * kr + j]; }
        f16x8 ph, pl, dh_, dl_;
        split8h(xp, ph, pl);
        split8h(xs, dh_, dl_);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f16x8 bh = sGB[(2 * (2 * dt + kr)) * 64 + lane], bl = sGB[(2 * (2 * dt + kr) + 1) * 64 + lane];
          accV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, bl, accV[dt], 0, 0, 0);
          accV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, bh, accV[dt], 0, 0, 0);
          accV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, bh, accV[dt], 0, 0, 0);
          const f16x8 ch = sQB[(2 * (2 * dt + kr)) * 64 + lane], cl = sQB[(2 * (2 * dt + kr) + 1) * 64 + lane];
          accK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh_, cl, accK[dt], 0, 0, 0);
          accK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl_, ch, accK[dt], 0, 0, 0);
          accK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh_, ch, accK[dt], 0, 0, 0);
        }
      }
    }
  }
  // ---- accumulator of dim tile dt: lane = dim 32 dt + n, register 4 g + q = token 32 wave + 8 g + 4 h2 + q
  if (has_tokens) {
    const float v_out = g_inv / 16384.0f, k_out = q_inv * ds_inv * a.scale;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dim = 32 * dt + n;
      if (dim < E) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tk = 32 * wave + 8 * (r >> 2) + 4 * h2 + (r & 3);
          if (tk < a.L) {
            const long o = (((long)u * a.nh + h) * a.L + tk) * E + dim;
            a.dKt[o] = accK[dt][r] * k_out;
            a.dVt[o] = accV[dt][r] * v_out;
          }
        }
      }
    }
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// The rays' own rows (neusky_model.py:535-549: the background radiance of a ray along its own direction): a ray is one more direction
// of ITS camera.  The rows arrive sorted by camera (perm[i] = the ray at sorted position i, seg[u] .. seg[u + 1] = camera u's rays); one
// workgroup per (camera, head) keeps K~_n (and, backward, V~_n and both gradients) of token n in thread n's registers -- the layout of
// the token kernel above -- and walks the camera's handful of rays:
//   forward : s_n by thread n, block maximum / sum through LDS, then thread e < 48 sums p_n V~[n][e] over the tokens (V~ read along e)
//   backward: ds_n, dK~_n += ds_n q~, dV~_n += p_n dO~ in registers; dq~[e] = sum_n ds_n K~[n][e] by thread e; at the end the camera's
//             dK~ / dV~ (already holding the grid rows' sums: same stream, earlier launch) take the rays' part by a plain read-add-write.
struct RayArgs {
  const float* Q; const float* dirs; const float* Kt; const float* Vt;   // Q [R, H], dirs [R, 3]
  const int* perm; const int* seg;                                       // [R], [U + 1]
  float* O; float* rmax; float* rsum;                                    // [R, H], [R, heads] x 2
  const float* dO; float* dQ; float* dKt; float* dVt;
  int U, R, L, nh;
  float scale;
};

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red, int tid) {  // 128 threads = 2 waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();  // (red is free again)
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(red[0], red[1]) : red[0] + red[1];
}

template <bool BWD>
__global__ __launch_bounds__(ATT_THREADS) void attn_rays_kernel(const RayArgs a) {
  __shared__ float red[2];
  __shared__ float sP[ATT_LMAX];   // p_n (forward) / ds_n (backward)
  __shared__ float sO[E];          // the 48-wide sums before the (d_x, d_y, 1) combination
  __shared__ float sT[ATT_LMAX * E];  // this (camera, head)'s V~ (forward) / K~ (backward), [token][48]: every ray's 48 token sums read it (round 6:
                                      // they walked it in global memory, 100 dependent-latency loads per ray, the kernel's time)
  // blockIdx.z: the 32-ray slice of the camera's rays this workgroup walks (a batch drawn from ONE image -- the eval-latent fit -- puts
  // all its rays on one camera: its slices run side by side; a camera with up to 32 rays has one workgroup and the rest leave at once)
  const int tid = threadIdx.x, h = blockIdx.x, u = blockIdx.y;
  const int seg_beg = a.seg[u], seg_end = a.seg[u + 1];
  if (seg_beg + 32 * (int)blockIdx.z >= seg_end) return;
  const bool shared_camera = seg_end - seg_beg > 32;  // several workgroups add into this camera's dK~ / dV~: atomics
  const int n = tid;
  const bool tok = n < a.L;
  const int H = a.nh * DH;
  const long kv0 = ((long)u * a.nh + h) * a.L * E;
  const long kv = kv0 + (long)(tok ? n : 0) * E;
  float Kn[E], Vn[BWD ? E : 1], dK[BWD ? E : 1], dV[BWD ? E : 1];
#pragma unroll
  for (int k = 0; k < E / 4; ++k) {
    const float4 kk = reinterpret_cast<const float4*>(a.Kt + kv)[k];
    Kn[4 * k] = kk.x; Kn[4 * k + 1] = kk.y; Kn[4 * k + 2] = kk.z; Kn[4 * k + 3] = kk.w;
    if (BWD) {
      const float4 vv = reinterpret_cast<const float4*>(a.Vt + kv)[k];
      Vn[4 * k] = vv.x; Vn[4 * k + 1] = vv.y; Vn[4 * k + 2] = vv.z; Vn[4 * k + 3] = vv.w;
      dK[4 * k] = dK[4 * k + 1] = dK[4 * k + 2] = dK[4 * k + 3] = 0.0f;
      dV[4 * k] = dV[4 * k + 1] = dV[4 * k + 2] = dV[4 * k + 3] = 0.0f;
    }
  }
  {
    const float* src = (BWD ? a.Kt : a.Vt) + kv0;
    for (int i = tid; i < a.L * E / 4; i += ATT_THREADS) reinterpret_cast<float4*>(sT)[i] = reinterpret_cast<const float4*>(src)[i];
  }
  __syncthreads();
  for (int s0 = seg_beg + 32 * blockIdx.z; s0 < seg_end; s0 += 32 * gridDim.z)  // (slices z, z + Z, ..: more than 32 Z rays on one camera)
  for (int i = s0; i < min(seg_end, s0 + 32); ++i) {
    const int r = a.perm[i];                       // block-uniform
    const float* qp = a.Q + (long)r * H + h * DH;
    const float dx = a.dirs[(long)r * 3], dy = a.dirs[(long)r * 3 + 1];
    float q[DH], g[BWD ? DH : 1];
    float Drow = 0.0f;
#pragma unroll
    for (int j = 0; j < DH; ++j) {
      q[j] = qp[j];
      if (BWD) {
        g[j] = a.dO[(long)r * H + h * DH + j];
        Drow = fmaf(g[j], a.O[(long)r * H + h * DH + j], Drow);
      }
    }
    float sA = 0.f, sB = 0.f, sC = 0.f;
#pragma unroll
    for (int j = 0; j < DH; ++j) { sA = fmaf(q[j], Kn[j], sA); sB = fmaf(q[j], Kn[DH + j], sB); sC = fmaf(q[j], Kn[2 * DH + j], sC); }
    const float s = tok ? a.scale * fmaf(dx, sA, fmaf(dy, sB, sC)) : -3.0e38f;
    const long st = (long)r * a.nh + h;
    if (!BWD) {
      const float m = block_reduce(s, true, red, tid);
      const float p = tok ? __expf(s - m) : 0.0f;
      const float l = block_reduce(p, false, red, tid);
      if (tok) sP[n] = p;
      __syncthreads();
      if (tid < E) {  // thread e: sum over the tokens, V~ read along e
        float acc = 0.0f;
        for (int t2 = 0; t2 < a.L; ++t2) acc = fmaf(sP[t2], sT[t2 * E + tid], acc);
        sO[tid] = acc;
      }
      __syncthreads();
      if (tid < DH) a.O[(long)r * H + h * DH + tid] = (dx * sO[tid] + dy * sO[DH + tid] + sO[2 * DH + tid]) / l;
      if (tid == 0) { a.rmax[st] = m; a.rsum[st] = l; }
    } else {
      const float m = a.rmax[st], inv = 1.0f / a.rsum[st];
      float pA = 0.f, pB = 0.f, pC = 0.f;
#pragma unroll
      for (int j = 0; j < DH; ++j) { pA = fmaf(g[j], Vn[j], pA); pB = fmaf(g[j], Vn[DH + j], pB); pC = fmaf(g[j], Vn[2 * DH + j], pC); }
      const float p = tok ? __expf(s - m) * inv : 0.0f;
      const float ds = p * (fmaf(dx, pA, fmaf(dy, pB, pC)) - Drow) * a.scale;
      const float dsx = ds * dx, dsy = ds * dy, px = p * dx, py = p * dy;
#pragma unroll
      for (int j = 0; j < DH; ++j) {
        dK[j] = fmaf(dsx, q[j], dK[j]); dK[DH + j] = fmaf(dsy, q[j], dK[DH + j]); dK[2 * DH + j] = fmaf(ds, q[j], dK[2 * DH + j]);
        dV[j] = fmaf(px, g[j], dV[j]); dV[DH + j] = fmaf(py, g[j], dV[DH + j]); dV[2 * DH + j] = fmaf(p, g[j], dV[2 * DH + j]);
      }
      __syncthreads();  // (the previous ray's sP / sO are no longer read)
      if (tok) sP[n] = ds;
      __syncthreads();
      if (tid < E) {
        float acc = 0.0f;
        for (int t2 = 0; t2 < a.L; ++t2) acc = fmaf(sP[t2], sT[t2 * E + tid], acc);
        sO[tid] = acc;
      }
      __syncthreads();
      if (tid < DH) a.dQ[(long)r * H + h * DH + tid] = dx * sO[tid] + dy * sO[DH + tid] + sO[2 * DH + tid];  // (ds carries the scale)
    }
    __syncthreads();
  }
  if (BWD && tok && shared_camera) {
#pragma unroll
    for (int k = 0; k < E; ++k) { atomicAdd(a.dKt + kv + k, dK[k]); atomicAdd(a.dVt + kv + k, dV[k]); }
  } else if (BWD && tok) {
#pragma unroll
    for (int k = 0; k < E / 4; ++k) {
      float4 x = reinterpret_cast<float4*>(a.dKt + kv)[k], y = reinterpret_cast<float4*>(a.dVt + kv)[k];
      x.x += dK[4 * k]; x.y += dK[4 * k + 1]; x.z += dK[4 * k + 2]; x.w += dK[4 * k + 3];
      y.x += dV[4 * k]; y.y += dV[4 * k + 1]; y.z += dV[4 * k + 2]; y.w += dV[4 * k + 3];
      reinterpret_cast<float4*>(a.dKt + kv)[k] = x;
      reinterpret_cast<float4*>(a.dVt + kv)[k] = y;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Residual add + layer norm of the decoder's row-local stream, one wave per row (width W = 64 V, V = 1, 2, 4 or 8 consecutive features per
// lane): s = x + r (stored when r is given), y = (s - mean) rstd gamma + beta, the row's (mean, rstd) saved.  Backward, for frozen
// gamma / beta: ds = rstd (g - mean(g) - xhat mean(g xhat)) + ds_in, g = dy gamma -- the gradient of both x and r.
struct LnArgs {
  const float* x; const float* r; const float* gamma; const float* beta;
  float* s; float* y; float* stats;            // stats [M, 2]
  const float* dy; const float* ds_in; float* ds;
  long M; float eps;
};

template <int V>
__device__ __forceinline__ void ln_load(const float* p, float (&v)[V]) {
  if (V == 1) v[0] = p[0];
  if (V == 2) { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[V > 1 ? 1 : 0] = a.y; }
  if (V >= 4) {
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
      const float4 a = reinterpret_cast<const float4*>(p)[i];
      v[4 * i] = a.x; v[V > 1 ? 4 * i + 1 : 0] = a.y; v[V > 2 ? 4 * i + 2 : 0] = a.z; v[V > 3 ? 4 * i + 3 : 0] = a.w;
    }
  }
}
template <int V>
__device__ __forceinline__ void ln_store(float* p, const float (&v)[V]) {
  if (V == 1) p[0] = v[0];
  if (V == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[V > 1 ? 1 : 0]);
  if (V >= 4) {
#pragma unroll
    for (int i = 0; i < V / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[V > 1 ? 4 * i + 1 : 0], v[V > 2 ? 4 * i + 2 : 0], v[V > 3 ? 4 * i + 3 : 0]);
  }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int V, bool BWD>
__global__ __launch_bounds__(256) void add_layer_norm_kernel(const LnArgs a) {
  constexpr int W = 64 * V;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const long off = row * W + lane * V;
  float gm[V];
  ln_load<V>(a.gamma + lane * V, gm);
  if (!BWD) {
    float v[V], bt[V];
    ln_load<V>(a.x + off, v);
    if (a.r) {
      float r[V];
      ln_load<V>(a.r + off, r);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += r[j];
      ln_store<V>(a.s + off, v);
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < V; ++j) sum += v[j];
    const float mean = wave_sum(sum) * (1.0f / W);
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < V; ++j) sq = fmaf(v[j] - mean, v[j] - mean, sq);
    const float rstd = rsqrtf(wave_sum(sq) * (1.0f / W) + a.eps);
    ln_load<V>(a.beta + lane * V, bt);
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = fmaf((v[j] - mean) * rstd, gm[j], bt[j]);
    ln_store<V>(a.y + off, v);
    if (lane == 0) { a.stats[2 * row] = mean; a.stats[2 * row + 1] = rstd; }
  } else {
    float s[V], g[V];
    ln_load<V>(a.x + off, s);      // the saved sum s = x + r
    ln_load<V>(a.dy + off, g);
    const float mean = a.stats[2 * row], rstd = a.stats[2 * row + 1];
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      g[j] *= gm[j];
      s[j] = (s[j] - mean) * rstd;
      c1 += g[j];
      c2 = fmaf(g[j], s[j], c2);
    }
    c1 = wave_sum(c1) * (1.0f / W);
    c2 = wave_sum(c2) * (1.0f / W);
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = rstd * (g[j] - c1 - s[j] * c2);
    if (a.ds_in) {
      float u[V];
      ln_load<V>(a.ds_in + off, u);
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] += u[j];
    }
    ln_store<V>(a.ds + off, g);
  }
}

template <bool BWD>
int launch_add_ln(const LnArgs& a, int W, hipStream_t stream) {
  const dim3 grid((unsigned)((a.M + 3) / 4));
  switch (W) {
    case 64: hipLaunchKernelGGL((add_layer_norm_kernel<1, BWD>), grid, dim3(256), 0, stream, a); break;
    case 128: hipLaunchKernelGGL((add_layer_norm_kernel<2, BWD>), grid, dim3(256), 0, stream, a); break;
    case 256: hipLaunchKernelGGL((add_layer_norm_kernel<4, BWD>), grid, dim3(256), 0, stream, a); break;
    case 512: hipLaunchKernelGGL((add_layer_norm_kernel<8, BWD>), grid, dim3(256), 0, stream, a); break;
    default: return NSKY_ERR_ARG;
  }
  return NSKY_OK;
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
  if (D >= 32)  // matrix-core form: 128-row blocks of a camera
    hipLaunchKernelGGL(attn_core_fwd_mfma_kernel, dim3(ceil_div(D, MF_ROWS), U), dim3(MF_THREADS), (size_t)16 * ceil_div(L, 16) * E * sizeof(float), (hipStream_t)stream, a);
  else
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
  if (D >= 32) {  // matrix-core forms; the row kernel publishes the operand maxima of every (camera, head) behind the D = dO . O scratch
    float* stat = drow + (size_t)U * n_heads * D;
    (void)hipMemsetAsync(stat, 0, (size_t)U * n_heads * 4 * sizeof(float), (hipStream_t)stream);
    hipLaunchKernelGGL(attn_core_bwd_rows_mfma_kernel, dim3(ceil_div(D, MB_ROWS), U), dim3(MB_THREADS), (size_t)16 * ceil_div(L, 16) * E * sizeof(float), (hipStream_t)stream, a, drow, stat);
    NSKY_CHECK_LAUNCH("nsky_attn_core_bwd (rows)");
    hipLaunchKernelGGL(attn_core_bwd_tokens_mfma_kernel, dim3(n_heads, U), dim3(MT_THREADS), 0, (hipStream_t)stream, a, (const float*)drow, (const float*)stat);
    NSKY_CHECK_LAUNCH("nsky_attn_core_bwd (tokens)");
    return NSKY_OK;
  }
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

// workgroups per (head, camera) of the ray kernels: a workgroup walks 32-ray slices z, z + Z, ... of its camera's rays.  Z = what the
// rays need when they spread evenly over the cameras, at most 32: a training batch (1 024 rays over 300 cameras: a few rays each) gets
// ONE workgroup per (head, camera) -- rounds 4-5 launched 32 and 31 of them left at once: 76 800 workgroups whose dispatch was the
// kernel's time --; a batch drawn from one image (the eval-latent fit) still gets its 32 slices side by side.
static inline int ray_slices(int R, int U) { return max(1, min(32, ceil_div(R, 32 * max(U, 1)))); }

extern "C" int nsky_attn_core_rays_fwd(const float* Q, const float* dirs, const int32_t* perm, const int32_t* seg, const float* Kt, const float* Vt,
                                       int32_t U, int32_t R, int32_t L, int32_t n_heads, float scale, float* O, float* row_max, float* row_sum,
                                       nsky_stream_t stream) {
  const void* ptrs[] = {Kt, Vt};
  if (int rc = check_attn("nsky_attn_core_rays_fwd", U, 1, L, n_heads, ptrs, 2)) return rc;
  NSKY_CHECK_ARG(Q && dirs && perm && seg && O && row_max && row_sum && R >= 1, "nsky_attn_core_rays_fwd: null operand / no rays");
  RayArgs a{};
  a.Q = Q; a.dirs = dirs; a.Kt = Kt; a.Vt = Vt; a.perm = perm; a.seg = seg; a.O = O; a.rmax = row_max; a.rsum = row_sum;
  a.U = U; a.R = R; a.L = L; a.nh = n_heads; a.scale = scale;
  hipLaunchKernelGGL(attn_rays_kernel<false>, dim3(n_heads, U, ray_slices(R, U)), dim3(ATT_THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_attn_core_rays_fwd");
  return NSKY_OK;
}

extern "C" int nsky_attn_core_rays_bwd(const float* Q, const float* dirs, const int32_t* perm, const int32_t* seg, const float* Kt, const float* Vt,
                                       const float* O, const float* row_max, const float* row_sum, const float* dO, int32_t U, int32_t R, int32_t L,
                                       int32_t n_heads, float scale, float* dQ, float* dKt, float* dVt, nsky_stream_t stream) {
  const void* ptrs[] = {Kt, Vt, dKt, dVt};
  if (int rc = check_attn("nsky_attn_core_rays_bwd", U, 1, L, n_heads, ptrs, 4)) return rc;
  NSKY_CHECK_ARG(Q && dirs && perm && seg && O && row_max && row_sum && dO && dQ && R >= 1, "nsky_attn_core_rays_bwd: null operand / no rays");
  RayArgs a{};
  a.Q = Q; a.dirs = dirs; a.Kt = Kt; a.Vt = Vt; a.perm = perm; a.seg = seg; a.O = const_cast<float*>(O); a.rmax = const_cast<float*>(row_max);
  a.rsum = const_cast<float*>(row_sum); a.dO = dO; a.dQ = dQ; a.dKt = dKt; a.dVt = dVt;
  a.U = U; a.R = R; a.L = L; a.nh = n_heads; a.scale = scale;
  hipLaunchKernelGGL(attn_rays_kernel<true>, dim3(n_heads, U, ray_slices(R, U)), dim3(ATT_THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_attn_core_rays_bwd");
  return NSKY_OK;
}

#define NSKY_AL16(p) (((uintptr_t)(p) % 16) == 0)
extern "C" int nsky_add_layer_norm_fwd(const float* x, const float* r, const float* gamma, const float* beta, int64_t M, int32_t W, float eps, float* s,
                                       float* y, float* stats, nsky_stream_t stream) {
  NSKY_CHECK_ARG(x && gamma && beta && y && stats && M >= 1 && (r == nullptr || s != nullptr), "nsky_add_layer_norm_fwd: null operand / empty batch");
  NSKY_CHECK_ARG(NSKY_AL16(x) && NSKY_AL16(y) && (!r || (NSKY_AL16(r) && NSKY_AL16(s))) && NSKY_AL16(gamma) && NSKY_AL16(beta), "nsky_add_layer_norm_fwd: alignment");
  LnArgs a{};
  a.x = x; a.r = r; a.gamma = gamma; a.beta = beta; a.s = s; a.y = y; a.stats = stats; a.M = M; a.eps = eps;
  NSKY_CHECK_ARG(launch_add_ln<false>(a, W, (hipStream_t)stream) == NSKY_OK, "nsky_add_layer_norm_fwd: width %d (64, 128, 256 or 512)", W);
  NSKY_CHECK_LAUNCH("nsky_add_layer_norm_fwd");
  return NSKY_OK;
}

extern "C" int nsky_add_layer_norm_bwd(const float* s, const float* stats, const float* gamma, const float* dy, const float* ds_in, int64_t M, int32_t W,
                                       float* ds, nsky_stream_t stream) {
  NSKY_CHECK_ARG(s && stats && gamma && dy && ds && M >= 1, "nsky_add_layer_norm_bwd: null operand / empty batch");
  NSKY_CHECK_ARG(NSKY_AL16(s) && NSKY_AL16(dy) && NSKY_AL16(ds) && (!ds_in || NSKY_AL16(ds_in)) && NSKY_AL16(gamma), "nsky_add_layer_norm_bwd: alignment");
  LnArgs a{};
  a.x = s; a.stats = const_cast<float*>(stats); a.gamma = gamma; a.dy = dy; a.ds_in = ds_in; a.ds = ds; a.M = M;
  NSKY_CHECK_ARG(launch_add_ln<true>(a, W, (hipStream_t)stream) == NSKY_OK, "nsky_add_layer_norm_bwd: width %d (64, 128, 256 or 512)", W);
  NSKY_CHECK_LAUNCH("nsky_add_layer_norm_bwd");
  return NSKY_OK;
}
