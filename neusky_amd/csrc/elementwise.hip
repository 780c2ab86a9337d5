// Small bandwidth-bound helpers of the train step:
//   * reverse-over-forward step of a Softplus layer carrying three input-tangents (normals / eikonal path)
//   * proposal PDF re-sampling (nerfstudio PDFSampler semantics, sequential fp32 CDF -> bit-exact indices)
//   * fused Adam update over a flat parameter slab
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

// For a layer a = softplus_beta(z), tangents ta_k = u_k * s with s = sigmoid(beta z), u_k = W ta_prev_k:
//   dz   = da * s + sum_k dta_k * ta_k * beta * (1 - s)          (u s s' / s = ta * beta (1-s))
//   du_k = dta_k * s
// dta_k is either a matrix block or the outer product ggrad[n,k] * wvec[c] (last hidden layer).
__global__ void softplus_tangent_bwd_kernel(const float* __restrict__ da, const float* __restrict__ s,
                                            const float* __restrict__ ta, const float* __restrict__ dta,
                                            const float* __restrict__ ggrad, const float* __restrict__ wvec, float beta, int N,
                                            int Cc, int ld, float* __restrict__ dz, float* __restrict__ du) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * Cc) return;
  const int n = (int)(i / Cc), c = (int)(i % Cc);
  const long o = (long)n * ld + c;
  const float sv = s[o];
  float acc = da ? da[o] * sv : 0.0f;
  const float k1 = beta * (1.0f - sv);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long ok = ((long)k * N + n) * ld + c;
    const float g = dta ? dta[ok] : ggrad[(long)n * 3 + k] * wvec[c];
    acc = fmaf(g * ta[ok], k1, acc);
    du[ok] = g * sv;
  }
  dz[o] = acc;
}

// one wave per ray
__global__ __launch_bounds__(256) void pdf_sample_kernel(const float* __restrict__ weights, const float* __restrict__ bins,
                                                         const float* __restrict__ u_base, const float* __restrict__ jitter,
                                                         int R, int n0, int nb, float hist_pad, float eps,
                                                         float* __restrict__ new_bins, int* __restrict__ inds_out) {
  extern __shared__ float sm[];  // per wave: cdf[n0+1], bins[n0+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  float* cdf = sm + wave * 2 * (n0 + 1);
  float* eb = cdf + (n0 + 1);
  for (int i = lane; i <= n0; i += 64) eb[i] = bins[(long)r * (n0 + 1) + i];
  {
    // The CDF a CPU float cumsum gives, bit for bit: sums accumulated in double and rounded to float at every step.  The terms are
    // floats within a few binades of each other (every weight carries the histogram padding), so their double partial sums are EXACT
    // and therefore independent of the order of the additions: the wave adds them in parallel (a strided sum for the total, an
    // inclusive scan per block of 64 bins for the CDF) and gets what the sequential loop got.
    const float* w = weights + (long)r * n0;
    double part = 0.0;
    for (int i = lane; i < n0; i += 64) part += (double)(w[i] + hist_pad);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    float wsum = (float)part;
    const float padding = fmaxf(eps - wsum, 0.0f);
    const float padn = padding / (float)n0;
    wsum += padding;
    if (lane == 0) cdf[0] = 0.0f;
    double carry = 0.0;
    for (int base = 0; base < n0; base += 64) {
      const int i = base + lane;
      double incl = i < n0 ? (double)(((w[i] + hist_pad) + padn) / wsum) : 0.0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      if (i < n0) cdf[i + 1] = fminf(1.0f, (float)(carry + incl));
      carry += __shfl(incl, 63, 64);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const float jit = jitter ? jitter[r] / (float)nb : 0.0f;
  for (int j = lane; j < nb; j += 64) {
    const float u = u_base[j] + jit;
    // searchsorted(cdf, u, side="right"): first index with cdf[idx] > u
    int lo = 0, hi = n0 + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = min(max(lo - 1, 0), n0), above = min(max(lo, 0), n0);
    const float c0 = cdf[below], c1 = cdf[above], b0 = eb[below], b1 = eb[above];
    float t = (u - c0) / (c1 - c0);
    if (!(t == t) ) t = 0.0f;            // nan_to_num(nan=0)
    if (isinf(t)) t = t > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    new_bins[(long)r * nb + j] = b0 + t * (b1 - b0);
    if (inds_out) inds_out[(long)r * nb + j] = lo;
  }
}

// torch.optim.Adam's update with torch's own scalar handling: every scalar is formed in double on the host (1 - beta, lr / (1 - beta1^t),
// sqrt(1 - beta2^t)) and rounded to float once, as torch rounds a Python scalar where it meets a float tensor:
//   m.lerp_(g, 1 - beta1);  v.mul_(beta2).addcmul_(g, g, value = 1 - beta2);  p.addcdiv_(m, v.sqrt() / sqrt(bc2) + eps, value = -lr / bc1)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float b2, float om1, float om2, float eps, float step_size, float bc2_sqrt, float grad_scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  const float m0 = m[i];
  const float mi = m0 + om1 * (gi - m0);          // lerp, weight 1 - beta1 < 0.5
  const float vi = b2 * v[i] + om2 * (gi * gi);
  m[i] = mi; v[i] = vi;
  p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}


// Weight-norm preparation of one dense layer, laid out as the consuming kernel wants it:
//   out[r][c] = g[sr] * v[sr][sc] / ||v[sr]||   with sr = row_map[r], sc = col_map[c]  (-1 = structural zero)
// One workgroup (256 threads) per output row; inv_norm[sr] is kept for the backward pass.
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, int I, int ldv,
                                                              const int* __restrict__ row_map, const int* __restrict__ col_map,
                                                              int Cp, float* __restrict__ out, int ldo, float* __restrict__ inv_norm) {
  const int r = blockIdx.x, sr = row_map[r];
  float* o = out + (long)r * ldo;
  if (sr < 0) {
    for (int c = threadIdx.x; c < Cp; c += 256) o[c] = 0.0f;
    return;
  }
  const float* vr = v + (long)sr * ldv;
  float ss = 0.0f;
  for (int c = threadIdx.x; c < I; c += 256) ss = fmaf(vr[c], vr[c], ss);
  __shared__ float red[256];
  red[threadIdx.x] = ss;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float inv = 1.0f / sqrtf(red[0]);
  if (threadIdx.x == 0) inv_norm[sr] = inv;
  const float sc = g[sr] * inv;
  for (int c = threadIdx.x; c < Cp; c += 256) {
    const int scn = col_map[c];
    o[c] = scn >= 0 ? sc * vr[scn] : 0.0f;
  }
}

// dW = gathered d_out row;  dot = <dW, v_r>;  dg = dot / ||v||;  dv = g/||v|| (dW - v dot / ||v||^2)
// inverse_col[sc] = the output column that reads source column sc (or -1: that column of v is not used -> dv = the
// projection term only)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ d_out, int ldo, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ inv_norm,
                                                              int I, int ldv, const int* __restrict__ row_map,
                                                              const int* __restrict__ inverse_col, float* __restrict__ dv,
                                                              float* __restrict__ dg) {
  const int r = blockIdx.x, sr = row_map[r];
  if (sr < 0) return;
  const float* d = d_out + (long)r * ldo;
  const float* vr = v + (long)sr * ldv;
  float dot = 0.0f;
  for (int c = threadIdx.x; c < I; c += 256) {
    const int oc = inverse_col[c];
    if (oc >= 0) dot = fmaf(d[oc], vr[c], dot);
  }
  __shared__ float red[256];
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  dot = red[0];
  const float inv = inv_norm[sr];
  if (threadIdx.x == 0) dg[sr] = dot * inv;
  const float a = g[sr] * inv, b = dot * inv * inv;
  float* dvr = dv + (long)sr * ldv;
  for (int c = threadIdx.x; c < I; c += 256) {
    const int oc = inverse_col[c];
    dvr[c] = a * ((oc >= 0 ? d[oc] : 0.0f) - vr[c] * b);
  }
}


// ---- ray set-up of the proposal sampler (no gradients flow through any of it) ------------------------------------
// nerfstudio SphereCollider(center 0, radius, near_plane): near / far of every ray against the scene sphere
__global__ void sphere_collider_kernel(const float* __restrict__ o, const float* __restrict__ d, int R, float radius,
                                       float near_plane, float* __restrict__ nears, float* __restrict__ fars) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2];
  const float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
  const float a = dx * dx + dy * dy + dz * dz;
  const float b = 2.0f * (ox * dx + oy * dy + oz * dz);
  const float c = ox * ox + oy * oy + oz * oz - radius * radius;
  const float disc = b * b - 4.0f * a * c;
  const bool ok = disc > 0.0f;
  const float sq = sqrtf(ok ? disc : 0.0f);
  const float t0 = (-b - sq) / (2.0f * a), t1 = (-b + sq) / (2.0f * a);
  const float nr = fmaxf(ok ? t0 : 0.0f, near_plane);
  nears[r] = nr;
  fars[r] = fmaxf(ok ? t1 : 0.0f, nr + 1e-6f);
}

// spacing bin s in [0,1] -> euclidean distance along the ray; ONE explicit rounding sequence shared by every kernel
__device__ __forceinline__ float spacing_to_euclid(float s, float nr, float fr) { return fmaf(s, fr, (1.0f - s) * nr); }

__device__ __forceinline__ float linspace01(int i, int steps) {  // torch.linspace(0, 1, steps)[i] as the GPU kernel forms it
  const float step = 1.0f / (float)(steps - 1);
  return i < steps / 2 ? step * (float)i : 1.0f - step * (float)(steps - i - 1);
}

// UniformSampler(single_jitter): spacing bins [R,n+1] (one jitter per ray, or the plain lattice) and euclidean bins
__global__ void uniform_bins_kernel(const float* __restrict__ nears, const float* __restrict__ fars,
                                    const float* __restrict__ jitter, int R, int n, float* __restrict__ sbins,
                                    float* __restrict__ ebins) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * (n + 1)) return;
  const int r = (int)(idx / (n + 1)), i = (int)(idx % (n + 1));
  float s = linspace01(i, n + 1);
  if (jitter) {
    const float lo = i == 0 ? s : (s + linspace01(i - 1, n + 1)) / 2.0f;
    const float hi = i == n ? s : (linspace01(i + 1, n + 1) + s) / 2.0f;
    s = lo + (hi - lo) * jitter[r];
  }
  sbins[idx] = s;
  ebins[idx] = spacing_to_euclid(s, nears[r], fars[r]);
}

// spacing bins -> euclidean bins (optional) and the bin mid-points along the ray, pos = o + d (e_i + e_{i+1}) / 2
__global__ void bins_to_samples_kernel(const float* __restrict__ sbins, const float* __restrict__ nears,
                                       const float* __restrict__ fars, const float* __restrict__ o, const float* __restrict__ d,
                                       int R, int n, float* __restrict__ ebins, float* __restrict__ pos) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * (n + 1)) return;
  const int r = (int)(idx / (n + 1)), i = (int)(idx % (n + 1));
  const float nr = nears[r], fr = fars[r];
  const float s0 = sbins[idx];
  const float e0 = spacing_to_euclid(s0, nr, fr);
  if (ebins) ebins[idx] = e0;
  if (pos && i < n) {
    const float s1 = sbins[idx + 1];
    const float mid = (e0 + spacing_to_euclid(s1, nr, fr)) / 2.0f;
    float* p = pos + ((long)r * n + i) * 3;
    p[0] = o[3 * r] + d[3 * r] * mid;
    p[1] = o[3 * r + 1] + d[3 * r + 1] * mid;
    p[2] = o[3 * r + 2] + d[3 * r + 2] * mid;
  }
}

// float4 form (C % 4 == 0, ld % 4 == 0, 16-byte aligned matrices): a wave reads one 1 KB row segment per instruction, the four
// waves of a block take every fourth row of the block's row range.  In the outer-product mode the same pass also forms
// wsum[c] += sum_{n,k} ggrad[n,k] ta_k[n,c] (the gradient of wvec: the sdf row of the geometry network's last layer), which
// otherwise costs a second read of the three tangent matrices.
__global__ __launch_bounds__(256) void softplus_tangent_bwd4_kernel(const float* __restrict__ da, const float* __restrict__ s,
                                                                    const float* __restrict__ ta, const float* __restrict__ dta,
                                                                    const float* __restrict__ ggrad, const float* __restrict__ wvec, float beta,
                                                                    int N, int Cc, int ld, float* __restrict__ dz, float* __restrict__ du,
                                                                    float* __restrict__ wsum, int rows_per_block) {
  const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
  const int phase = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(N, r0 + rows_per_block);
  float4 wacc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < Cc) {
    const float4 wv = wvec ? *reinterpret_cast<const float4*>(wvec + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = r0 + phase; n < r1; n += 4) {
      const long o = (long)n * ld + c;
      const float4 sv = *reinterpret_cast<const float4*>(s + o);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (da) {
        const float4 d = *reinterpret_cast<const float4*>(da + o);
        acc = make_float4(d.x * sv.x, d.y * sv.y, d.z * sv.z, d.w * sv.w);
      }
      const float4 k1 = make_float4(beta * (1.0f - sv.x), beta * (1.0f - sv.y), beta * (1.0f - sv.z), beta * (1.0f - sv.w));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const long ok = ((long)k * N + n) * ld + c;
        const float4 t = *reinterpret_cast<const float4*>(ta + ok);
        float4 g;
        if (dta) {
          g = *reinterpret_cast<const float4*>(dta + ok);
        } else {
          const float gg = ggrad[(long)n * 3 + k];
          g = make_float4(gg * wv.x, gg * wv.y, gg * wv.z, gg * wv.w);
          wacc.x = fmaf(gg, t.x, wacc.x); wacc.y = fmaf(gg, t.y, wacc.y); wacc.z = fmaf(gg, t.z, wacc.z); wacc.w = fmaf(gg, t.w, wacc.w);
        }
        acc.x = fmaf(g.x * t.x, k1.x, acc.x); acc.y = fmaf(g.y * t.y, k1.y, acc.y);
        acc.z = fmaf(g.z * t.z, k1.z, acc.z); acc.w = fmaf(g.w * t.w, k1.w, acc.w);
        *reinterpret_cast<float4*>(du + ok) = make_float4(g.x * sv.x, g.y * sv.y, g.z * sv.z, g.w * sv.w);
      }
      *reinterpret_cast<float4*>(dz + o) = acc;
    }
  }
  if (!wsum) return;
  __shared__ float4 red[256];
  red[threadIdx.x] = wacc;
  __syncthreads();
  if (phase == 0 && c < Cc) {
    const float4 a = red[threadIdx.x], b = red[threadIdx.x + 64], cc = red[threadIdx.x + 128], d = red[threadIdx.x + 192];
    atomicAdd(wsum + c, (a.x + b.x) + (cc.x + d.x));
    atomicAdd(wsum + c + 1, (a.y + b.y) + (cc.y + d.y));
    atomicAdd(wsum + c + 2, (a.z + b.z) + (cc.z + d.z));
    atomicAdd(wsum + c + 3, (a.w + b.w) + (cc.w + d.w));
  }
}

// Many small device-to-device copies as ONE launch (byte runs): the parameter gradients autograd left in tensors of its own -> their
// places in the optimizer slab; the next step's input tensors -> the static buffers a captured graph reads.
constexpr int GATHER_MAX = 64;
struct GatherArgs {
  const unsigned char* src[GATHER_MAX];
  unsigned char* dst[GATHER_MAX];
  long n[GATHER_MAX];  // bytes
};
__global__ __launch_bounds__(256) void gather_segments_kernel(const GatherArgs a) {
  const int s = blockIdx.y;
  const unsigned char* __restrict__ src = a.src[s];
  unsigned char* __restrict__ dst = a.dst[s];
  const long n = a.n[s];
  const long stride = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
  const uintptr_t both = (uintptr_t)src | (uintptr_t)dst;
  if ((both & 15) == 0) {
    const long n16 = n >> 4;
    for (long i = t0; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (long i = 16 * n16 + t0; i < n; i += stride) dst[i] = src[i];
  } else if ((both & 3) == 0) {
    const long n4 = n >> 2;
    for (long i = t0; i < n4; i += stride) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    for (long i = 4 * n4 + t0; i < n; i += stride) dst[i] = src[i];
  } else {
    for (long i = t0; i < n; i += stride) dst[i] = src[i];
  }
}

static int copy_segments_launch(const void* const* src, void* const* dst, const int64_t* nbytes, int count, hipStream_t stream, const char* who) {
  for (int s0 = 0; s0 < count; s0 += GATHER_MAX) {
    const int cnt = count - s0 < GATHER_MAX ? count - s0 : GATHER_MAX;
    GatherArgs a;
    long longest = 0;
    for (int i = 0; i < cnt; ++i) {
      NSKY_CHECK_ARG(nbytes[s0 + i] >= 0 && (nbytes[s0 + i] == 0 || (src[s0 + i] && dst[s0 + i])), "%s: segment %d", who, s0 + i);
      a.src[i] = static_cast<const unsigned char*>(src[s0 + i]); a.dst[i] = static_cast<unsigned char*>(dst[s0 + i]); a.n[i] = nbytes[s0 + i];
      longest = a.n[i] > longest ? a.n[i] : longest;
    }
    if (longest == 0) continue;
    const long bx = (longest / 16 + 2047) / 2048;  // ~8 x 16 bytes per thread
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)(bx < 1 ? 1 : (bx > 256 ? 256 : bx)), cnt), dim3(256), 0, stream, a);
  }
  return NSKY_OK;
}

}  // namespace

extern "C" int nsky_softplus_tangent_bwd(const float* da, const float* s, const float* ta, const float* dta, const float* ggrad,
                                         const float* wvec, float beta, int32_t N, int32_t C, int32_t ld, float* dz, float* du,
                                         float* wsum, nsky_stream_t stream) {
  if ((long)N * C == 0) return NSKY_OK;
  NSKY_CHECK_ARG(s && ta && dz && du && (dta || (ggrad && wvec)) && ld >= C, "nsky_softplus_tangent_bwd: bad argument");
  NSKY_CHECK_ARG(!wsum || (!dta && ggrad), "nsky_softplus_tangent_bwd: wsum goes with the outer-product form (ggrad x wvec)");
  auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (C % 4 == 0 && ld % 4 == 0 && al(da) && al(s) && al(ta) && al(dta) && al(wvec) && al(dz) && al(du)) {
    const int rows_per_block = 192;  // 512 blocks at the step's 98 304 rows
    hipLaunchKernelGGL(softplus_tangent_bwd4_kernel, dim3(ceil_div(C, 256), ceil_div(N, rows_per_block)), dim3(256), 0, (hipStream_t)stream, da, s,
                       ta, dta, ggrad, wvec, beta, N, C, ld, dz, du, wsum, rows_per_block);
    NSKY_CHECK_LAUNCH("nsky_softplus_tangent_bwd");
    return NSKY_OK;
  }
  const long n = (long)N * C;
  hipLaunchKernelGGL(softplus_tangent_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, da, s, ta, dta, ggrad,
                     wvec, beta, N, C, ld, dz, du);
  if (wsum) {  // (unaligned operands: the weighted column sum as its own pass)
    for (int k = 0; k < 3; ++k)
      if (int rc = nsky_weighted_colsum_f32(ta + (long)k * N * ld, N, C, ld, ggrad + k, 3, wsum, stream)) return rc;
  }
  NSKY_CHECK_LAUNCH("nsky_softplus_tangent_bwd");
  return NSKY_OK;
}

extern "C" int nsky_pdf_sample(const float* weights, const float* bins, const float* u_base, const float* jitter, int32_t R,
                               int32_t n0, int32_t nb, float histogram_padding, float eps, float* new_bins, int32_t* inds,
                               nsky_stream_t stream) {
  if (R == 0) return NSKY_OK;
  NSKY_CHECK_ARG(weights && bins && u_base && new_bins && R > 0 && n0 > 0 && nb > 0 && n0 <= 4096, "nsky_pdf_sample: bad argument");
  const size_t smem = 4 * 2 * (size_t)(n0 + 1) * sizeof(float);
  hipLaunchKernelGGL(pdf_sample_kernel, dim3(ceil_div(R, 4)), dim3(256), smem, (hipStream_t)stream, weights, bins, u_base, jitter, R,
                     n0, nb, histogram_padding, eps, new_bins, inds);
  NSKY_CHECK_LAUNCH("nsky_pdf_sample");
  return NSKY_OK;
}

extern "C" int nsky_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                              double eps, int32_t step, float grad_scale, nsky_stream_t stream) {
  if (n == 0) return NSKY_OK;
  NSKY_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "nsky_adam_step: bad argument");
  NSKY_CHECK_ARG(beta1 >= 0.5 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "nsky_adam_step: betas (%g, %g): beta1 in [0.5, 1) (torch's lerp form), beta2 in [0, 1)", beta1, beta2);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2), grad_scale);
  NSKY_CHECK_LAUNCH("nsky_adam_step");
  return NSKY_OK;
}

extern "C" int nsky_weight_norm_fwd(const float* v, const float* g, int32_t n_rows_out, int32_t n_cols_out, int32_t in_features,
                                    int32_t ldv, const int32_t* row_map, const int32_t* col_map, float* out, int32_t ldo,
                                    float* inv_norm, nsky_stream_t stream) {
  NSKY_CHECK_ARG(v && g && row_map && col_map && out && inv_norm, "nsky_weight_norm_fwd: null argument");
  NSKY_CHECK_ARG(n_rows_out > 0 && n_cols_out > 0 && in_features > 0 && ldv >= in_features && ldo >= n_cols_out,
                 "nsky_weight_norm_fwd: bad sizes");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(n_rows_out), dim3(256), 0, (hipStream_t)stream, v, g, in_features, ldv, row_map,
                     col_map, n_cols_out, out, ldo, inv_norm);
  NSKY_CHECK_LAUNCH("nsky_weight_norm_fwd");
  return NSKY_OK;
}

extern "C" int nsky_weight_norm_bwd(const float* d_out, int32_t ldo, const float* v, const float* g, const float* inv_norm,
                                    int32_t n_rows_out, int32_t in_features, int32_t ldv, const int32_t* row_map,
                                    const int32_t* inverse_col, float* dv, float* dg, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d_out && v && g && inv_norm && row_map && inverse_col && dv && dg, "nsky_weight_norm_bwd: null argument");
  NSKY_CHECK_ARG(n_rows_out > 0 && in_features > 0 && ldv >= in_features, "nsky_weight_norm_bwd: bad sizes");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(n_rows_out), dim3(256), 0, (hipStream_t)stream, d_out, ldo, v, g, inv_norm,
                     in_features, ldv, row_map, inverse_col, dv, dg);
  NSKY_CHECK_LAUNCH("nsky_weight_norm_bwd");
  return NSKY_OK;
}

extern "C" int nsky_sphere_collider(const float* origins, const float* directions, int32_t R, float radius, float near_plane,
                                    float* nears, float* fars, nsky_stream_t stream) {
  if (R == 0) return NSKY_OK;
  NSKY_CHECK_ARG(origins && directions && nears && fars && R > 0 && radius > 0.0f, "nsky_sphere_collider: bad argument");
  hipLaunchKernelGGL(sphere_collider_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, origins, directions, R,
                     radius, near_plane, nears, fars);
  NSKY_CHECK_LAUNCH("nsky_sphere_collider");
  return NSKY_OK;
}

extern "C" int nsky_uniform_bins(const float* nears, const float* fars, const float* jitter, int32_t R, int32_t n, float* sbins,
                                 float* ebins, nsky_stream_t stream) {
  if (R == 0) return NSKY_OK;
  NSKY_CHECK_ARG(nears && fars && sbins && ebins && R > 0 && n > 0, "nsky_uniform_bins: bad argument");
  hipLaunchKernelGGL(uniform_bins_kernel, dim3(ceil_div((long)R * (n + 1), 256)), dim3(256), 0, (hipStream_t)stream, nears, fars,
                     jitter, R, n, sbins, ebins);
  NSKY_CHECK_LAUNCH("nsky_uniform_bins");
  return NSKY_OK;
}

extern "C" int nsky_bins_to_samples(const float* sbins, const float* nears, const float* fars, const float* origins,
                                    const float* directions, int32_t R, int32_t n, float* ebins, float* positions,
                                    nsky_stream_t stream) {
  if (R == 0) return NSKY_OK;
  NSKY_CHECK_ARG(sbins && nears && fars && R > 0 && n > 0 && (ebins || positions), "nsky_bins_to_samples: bad argument");
  NSKY_CHECK_ARG(!positions || (origins && directions), "nsky_bins_to_samples: positions need origins and directions");
  hipLaunchKernelGGL(bins_to_samples_kernel, dim3(ceil_div((long)R * (n + 1), 256)), dim3(256), 0, (hipStream_t)stream, sbins,
                     nears, fars, origins, directions, R, n, ebins, positions);
  NSKY_CHECK_LAUNCH("nsky_bins_to_samples");
  return NSKY_OK;
}

extern "C" int nsky_gather_segments(const nsky_segment* segments, int32_t n_segments, nsky_stream_t stream) {
  NSKY_CHECK_ARG(n_segments >= 0 && n_segments <= 4096 && (n_segments == 0 || segments), "nsky_gather_segments: bad argument");
  if (n_segments == 0) return NSKY_OK;
  const void* src[4096]; void* dst[4096]; int64_t nb[4096];
  for (int i = 0; i < n_segments; ++i) { src[i] = segments[i].src; dst[i] = segments[i].dst; nb[i] = segments[i].n * (int64_t)sizeof(float); }
  if (int rc = copy_segments_launch(src, dst, nb, n_segments, (hipStream_t)stream, "nsky_gather_segments")) return rc;
  NSKY_CHECK_LAUNCH("nsky_gather_segments");
  return NSKY_OK;
}

extern "C" int nsky_copy_segments(const void* const* src, void* const* dst, const int64_t* nbytes, int32_t n_segments, nsky_stream_t stream) {
  NSKY_CHECK_ARG(n_segments >= 0 && (n_segments == 0 || (src && dst && nbytes)), "nsky_copy_segments: bad argument");
  if (n_segments == 0) return NSKY_OK;
  if (int rc = copy_segments_launch(src, dst, nbytes, n_segments, (hipStream_t)stream, "nsky_copy_segments")) return rc;
  NSKY_CHECK_LAUNCH("nsky_copy_segments");
  return NSKY_OK;
}
