// Closed-form loss terms of the train step, fused: ONE launch each way for the NeuSky model's terms
// (neusky/models/neusky_model.py:933-1035) and one for the DDF model's (neusky/models/ddf_model.py:407-493), instead of ~100
// small torch launches forward and ~200 backward on [R]-sized tensors (2.3 ms of wall time between the last forward and the
// first backward dense layer).  Every formula keeps the torch semantics the reference gets from autograd: sgn(0) = 0 for |.|,
// clamp gradients pass inside the closed interval, nan_to_num passes where the input is a number, BCE's log is clamped at
// -100 and its gradient denominator at 1e-12, F.normalize / cosine_similarity clamp their norms at eps.
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

constexpr int NT_MAIN = 8, NT_DDF = 5;

template <int NT>
__device__ __forceinline__ void block_accumulate(float (&loc)[NT], float* __restrict__ out) {
  __shared__ float red[NT][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float v = wave_sum(loc[t]);
    if (lane == 0) red[t][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < NT) {
    const float v = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (v != 0.0f) atomicAdd(out + threadIdx.x, v);
  }
}

__device__ __forceinline__ float sgnf(float x) { return (x > 0.0f) - (x < 0.0f); }

__device__ __forceinline__ float srgb_raw(float c) {  // utils.py:25-30 before the clamp
  return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(fabsf(c), 1.0f / 2.4f) - 0.055f;
}

struct SkyRow {  // masked sRGB background vs masked image of one ray (losses.py:44-58)
  float a[3], b[3], yraw[3], na, nb, an[3], bn[3], sim;
};
__device__ __forceinline__ SkyRow sky_row(const float* hdr, const float* img, float sm) {
  SkyRow s;
  float aa = 0.f, bb = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    s.yraw[i] = srgb_raw(hdr[i]);
    s.a[i] = fminf(fmaxf(s.yraw[i], 0.0f), 1.0f) * sm;
    s.b[i] = img[i] * sm;
    aa += s.a[i] * s.a[i];
    bb += s.b[i] * s.b[i];
  }
  s.na = sqrtf(aa);
  s.nb = sqrtf(bb);
  const float da = fmaxf(s.na, 1e-20f), db = fmaxf(s.nb, 1e-20f);
  s.sim = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    s.an[i] = s.a[i] / da;
    s.bn[i] = s.b[i] / db;
    s.sim += s.an[i] * s.bn[i];
  }
  return s;
}

__global__ __launch_bounds__(256) void main_losses_fwd_kernel(nsky_main_losses_desc d, float* __restrict__ terms, float* __restrict__ wsum) {
  float loc[NT_MAIN] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long N = (long)d.R * d.S;
  const long n_grid = 3l * d.P;
  const long total = d.R + (d.eik ? N : 0) + (d.grid ? n_grid : 0) + (d.sdf_term ? d.M : 0);
  const float invR = 1.0f / d.R;
  for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
    long i = w;
    if (i < d.R) {
      const int r = (int)i;
      const float fg = d.mask[4 * r + 1], gm = d.mask[4 * r + 2], sky = d.mask[4 * r + 3];
      if (d.rgb) {  // :947-950
        const float keep = 1.0f - sky;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) s += fabsf(d.image[3 * r + c] * keep - d.rgb[3 * r + c] * keep);
        loc[0] += s * (invR / 3.0f);
      }
      if (d.weights) {  // :963-967
        float s = 0.f;
        for (int k = 0; k < d.S; ++k) s += d.weights[(long)r * d.S + k];
        wsum[r] = s;
        float ws = fminf(fmaxf(s, 1e-3f), 1.0f - 1e-3f);
        if (s != s) ws = 0.5f;
        const float l = -(fg * fmaxf(logf(ws), -100.0f) + (1.0f - fg) * fmaxf(logf(1.0f - ws), -100.0f));
        loc[2] += l * invR;
      }
      if (d.normal) {  // :995-1000 + monosdf_normal_loss
        const float p[3] = {d.normal[3 * r] * gm, d.normal[3 * r + 1] * gm, d.normal[3 * r + 2] * gm};
        const float np = fmaxf(sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), 1e-12f);
        const float gz = gm / fmaxf(fabsf(gm), 1e-12f);
        const float ph[3] = {p[0] / np, p[1] / np, p[2] / np};
        loc[4] += (fabsf(ph[0]) + fabsf(ph[1]) + fabsf(ph[2] - gz) + (1.0f - ph[2] * gz)) * invR;
      }
      if (d.hdr_bg) {  // :1002-1009
        const SkyRow s = sky_row(d.hdr_bg + 3 * r, d.image + 3 * r, sky);
        float mse = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) mse += (s.a[c] - s.b[c]) * (s.a[c] - s.b[c]);
        loc[5] += mse * (invR / 3.0f) + d.sky_alpha * (1.0f - s.sim) * invR;
      }
      if (r == 0 && d.vis_thr) {  // :1011-1030
        const float e = d.vis_thr[0] - d.vis_target;
        loc[6] += e * e;
      }
      continue;
    }
    i -= d.R;
    if (d.eik) {
      if (i < N) {  // :958-960
        const float gx = d.eik[3 * i], gy = d.eik[3 * i + 1], gz = d.eik[3 * i + 2];
        const float e = sqrtf(gx * gx + gy * gy + gz * gz) - 1.0f;
        loc[1] += e * e * (1.0f / N);
        continue;
      }
      i -= N;
    }
    if (d.grid) {
      if (i < n_grid) {  // :990-993
        loc[3] += fabsf(d.grid[i]) * (1.0f / n_grid);
        continue;
      }
      i -= n_grid;
    }
    if (d.sdf_term) {  // :1032-1035
      const float v = d.sdf_term[i];
      loc[7] += v * v * (1.0f / d.M);
    }
  }
  block_accumulate<NT_MAIN>(loc, terms);
}

__global__ __launch_bounds__(256) void main_losses_bwd_kernel(nsky_main_losses_desc d, const float* __restrict__ wsum,
                                                              const float* __restrict__ g, float* __restrict__ d_rgb,
                                                              float* __restrict__ d_eik, float* __restrict__ d_weights,
                                                              float* __restrict__ d_normal, float* __restrict__ d_hdr,
                                                              float* __restrict__ d_grid, float* __restrict__ d_sdf,
                                                              float* __restrict__ d_thr) {
  const long N = (long)d.R * d.S;
  const long n_grid = 3l * d.P;
  const long total = d.R + (d_weights ? N : 0) + (d_eik ? N : 0) + (d_grid ? n_grid : 0) + (d_sdf ? d.M : 0);
  const float invR = 1.0f / d.R;
  for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
    long i = w;
    if (i < d.R) {
      const int r = (int)i;
      const float gm = d.mask[4 * r + 2], sky = d.mask[4 * r + 3];
      if (d_rgb) {
        const float keep = 1.0f - sky;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          d_rgb[3 * r + c] = g[0] * keep * sgnf(d.rgb[3 * r + c] * keep - d.image[3 * r + c] * keep) * (invR / 3.0f);
      }
      if (d_normal) {
        const float p[3] = {d.normal[3 * r] * gm, d.normal[3 * r + 1] * gm, d.normal[3 * r + 2] * gm};
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        const float np = fmaxf(nrm, 1e-12f);
        const float gz = gm / fmaxf(fabsf(gm), 1e-12f);
        const float ph[3] = {p[0] / np, p[1] / np, p[2] / np};
        const float u[3] = {sgnf(ph[0]), sgnf(ph[1]), sgnf(ph[2] - gz) - gz};
        float dp[3];
        if (nrm > 1e-12f) {
          const float pu = ph[0] * u[0] + ph[1] * u[1] + ph[2] * u[2];
#pragma unroll
          for (int c = 0; c < 3; ++c) dp[c] = (u[c] - ph[c] * pu) / nrm;
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) dp[c] = u[c] / 1e-12f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) d_normal[3 * r + c] = g[4] * invR * dp[c] * gm;
      }
      if (d_hdr) {
        const SkyRow s = sky_row(d.hdr_bg + 3 * r, d.image + 3 * r, sky);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float dsim = s.na > 1e-20f ? (s.bn[c] - s.sim * s.an[c]) / s.na : s.bn[c] / 1e-20f;
          const float da = 2.0f * (s.a[c] - s.b[c]) * (invR / 3.0f) - d.sky_alpha * invR * dsim;
          const float x = d.hdr_bg[3 * r + c];
          const float pass = (s.yraw[c] >= 0.0f && s.yraw[c] <= 1.0f) ? 1.0f : 0.0f;
          const float dy = x <= 0.0031308f ? 12.92f : (1.055f / 2.4f) * powf(fabsf(x), 1.0f / 2.4f - 1.0f) * sgnf(x);
          d_hdr[3 * r + c] = g[5] * da * sky * pass * dy;
        }
      }
      if (r == 0 && d_thr) d_thr[0] = g[6] * 2.0f * (d.vis_thr[0] - d.vis_target);
      continue;
    }
    i -= d.R;
    if (d_weights) {
      if (i < N) {
        const int r = (int)(i / d.S);
        const float s = wsum[r], fg = d.mask[4 * r + 1];
        float gw = 0.f;
        if (s == s && s >= 1e-3f && s <= 1.0f - 1e-3f) gw = (s - fg) / fmaxf((1.0f - s) * s, 1e-12f) * invR;
        d_weights[i] = g[2] * gw;
        continue;
      }
      i -= N;
    }
    if (d_eik) {
      if (i < N) {
        const float gx = d.eik[3 * i], gy = d.eik[3 * i + 1], gz = d.eik[3 * i + 2];
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        const float k = nrm > 0.0f ? g[1] * 2.0f * (nrm - 1.0f) / nrm * (1.0f / N) : 0.0f;
        d_eik[3 * i] = k * gx; d_eik[3 * i + 1] = k * gy; d_eik[3 * i + 2] = k * gz;
        continue;
      }
      i -= N;
    }
    if (d_grid) {
      if (i < n_grid) {
        d_grid[i] = g[3] * sgnf(d.grid[i]) * (1.0f / n_grid);
        continue;
      }
      i -= n_grid;
    }
    if (d_sdf) d_sdf[i] = g[7] * 2.0f * d.sdf_term[i] * (1.0f / d.M);
  }
}

// ------------------------------------------------------------------------------------------ DDF model
__device__ __forceinline__ void depth_pair(const nsky_ddf_losses_desc& d, int i, float& e, float& gt, float& me) {
  const float m = d.mask[i];
  if (d.mask_to_circumference) { e = d.expected[i]; gt = m == 0.0f ? 2.0f * d.radius : d.term[i]; me = 1.0f; }
  else { e = d.expected[i] * m; gt = d.term[i] * m; me = m; }
}

__global__ __launch_bounds__(256) void ddf_losses_fwd_kernel(nsky_ddf_losses_desc d, float* __restrict__ terms) {
  float loc[NT_DDF] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const long total = (long)d.Mr + d.Mm + d.Ms;
  for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
    long i = w;
    if (i < d.Mr) {
      const float m = d.mask[i];
      if (d.want_depth) {  // :427-433
        float e, gt, me;
        depth_pair(d, (int)i, e, gt, me);
        const float iw = d.inverse_depth_weight ? 1.0f / (gt + 1e-6f) : 1.0f;
        const float dw = d.dist_weight ? d.dist_weight[i] : 1.0f;
        loc[0] += fabsf(e - gt) * dw * iw * (1.0f / d.Mr);
      }
      if (d.sdf) {
        const float v = d.sdf[i] * m;
        if (d.want_sdf_l2) loc[1] += v * v * (1.0f / d.Mr);
        if (d.want_sdf_l1) loc[2] += fabsf(v) * (1.0f / d.Mr);
      }
      continue;
    }
    i -= d.Mr;
    if (i < d.Mm) {  // :475-483, the [M] - [M,1] broadcast: mean over i, j of relu(a_j - b_i)^2
      const float a = d.mv_expected[i];
      float s = 0.f;
      for (int k = 0; k < d.Mm; ++k) {
        const float h = fmaxf(a - d.mv_term[k], 0.0f);
        s = fmaf(h, h, s);
      }
      loc[3] += s * (1.0f / ((float)d.Mm * (float)d.Mm));
      continue;
    }
    i -= d.Mm;
    loc[4] += fabsf(d.sky_expected[i] - d.sky_term[i]) * (1.0f / d.Ms);  // :485-490
  }
  block_accumulate<NT_DDF>(loc, terms);
}

__global__ __launch_bounds__(256) void ddf_losses_bwd_kernel(nsky_ddf_losses_desc d, const float* __restrict__ g,
                                                             float* __restrict__ d_expected, float* __restrict__ d_sdf,
                                                             float* __restrict__ d_mv, float* __restrict__ d_sky,
                                                             float* __restrict__ d_term, float* __restrict__ d_mv_term) {
  const long total = (long)d.Mr + d.Mm + d.Ms;
  for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
    long i = w;
    if (i < d.Mr) {
      const float m = d.mask[i];
      if (d_expected || d_term) {
        float v = 0.f, vt = 0.f;
        if (d.want_depth) {
          float e, gt, me;
          depth_pair(d, (int)i, e, gt, me);
          const float iw = d.inverse_depth_weight ? 1.0f / (gt + 1e-6f) : 1.0f;
          const float dw = d.dist_weight ? d.dist_weight[i] : 1.0f;
          const float k = g[0] * dw * (1.0f / d.Mr);
          v = k * me * sgnf(e - gt) * iw;
          // the target is the field's own rendering of the fit ray (ddf_model.py:411-419; gradients flow unless the pipeline
          // detaches it): d/d gt of |e - gt| / (gt + 1e-6), gt = term * mask or the un-masked term
          const float dgt = k * (-sgnf(e - gt) * iw - (d.inverse_depth_weight ? fabsf(e - gt) * iw * iw : 0.0f));
          vt = dgt * (d.mask_to_circumference ? (m == 0.0f ? 0.0f : 1.0f) : m);
        }
        if (d_expected) d_expected[i] = v;
        if (d_term) d_term[i] = vt;
      }
      if (d_sdf) {
        const float v = d.sdf[i] * m;
        float gs = 0.f;
        if (d.want_sdf_l2) gs += g[1] * 2.0f * v * m * (1.0f / d.Mr);
        if (d.want_sdf_l1) gs += g[2] * sgnf(v) * m * (1.0f / d.Mr);
        d_sdf[i] = gs;
      }
      continue;
    }
    i -= d.Mr;
    if (i < d.Mm) {
      const float c = g[3] * 2.0f * (1.0f / ((float)d.Mm * (float)d.Mm));
      if (d_mv) {
        const float a = d.mv_expected[i];
        float s = 0.f;
        for (int k = 0; k < d.Mm; ++k) s += fmaxf(a - d.mv_term[k], 0.0f);
        d_mv[i] = c * s;
      }
      if (d_mv_term) {  // row i of the [M,M] broadcast: -sum_j relu(a_j - b_i)
        const float b = d.mv_term[i];
        float s = 0.f;
        for (int k = 0; k < d.Mm; ++k) s += fmaxf(d.mv_expected[k] - b, 0.0f);
        d_mv_term[i] = -c * s;
      }
      continue;
    }
    i -= d.Mm;
    if (d_sky) d_sky[i] = g[4] * sgnf(d.sky_expected[i] - d.sky_term[i]) * (1.0f / d.Ms);
  }
}

int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// The scalar training metrics of one model in ONE launch (one workgroup): PSNR = 10 log10(peak^2 / mean(((pred - gt) mask)^2))
// (neusky_model.py:1066-1068 with peak = 1, no mask; ddf_model.py:381-405 with peak = the DDF radius and the batch mask), and, when the
// NeuS variance parameter is handed in, s_val = clip(exp(10 v), 1e-6, 1e6) and 1 / s_val (neusky_model.py:1071-1072).
__global__ __launch_bounds__(1024) void train_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             const float* __restrict__ mask, long n, float peak_sq,
                                                             const float* __restrict__ variance, float* __restrict__ out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float m = mask ? mask[i] : 1.0f;
    const float d = pred[i] * m - gt[i] * m;
    acc += (double)(d * d);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    const float mse = (float)(t / (double)n);
    out[0] = 10.0f * log10f(peak_sq / mse);
    if (variance) {
      const float sv = fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);
      out[1] = sv;
      out[2] = 1.0f / sv;
    }
  }
}

struct TotalArgs { nsky_total_segment s[NSKY_TOTAL_MAX_SEGMENTS]; int n; };

__global__ __launch_bounds__(1024) void weighted_total_fwd_kernel(TotalArgs a, float* __restrict__ total) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int k = 0; k < a.n; ++k) {
    const nsky_total_segment& g = a.s[k];
    float part = 0.0f;
    for (int i = threadIdx.x; i < g.n; i += 1024) part += g.coef ? g.coef[i] * g.x[i] : g.x[i];
    acc += (double)g.scale * (double)part;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    total[0] = (float)t;
  }
}

__global__ __launch_bounds__(256) void weighted_total_bwd_kernel(TotalArgs a, const float* __restrict__ g) {
  const nsky_total_segment& s = a.s[blockIdx.y];
  if (!s.grad) return;
  const float gs = g[0] * s.scale;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < s.n; i += gridDim.x * 256) s.grad[i] = s.coef ? gs * s.coef[i] : gs;
}

}  // namespace

extern "C" int nsky_weighted_total_fwd(const nsky_total_segment* segments, int32_t n_segments, float* total, nsky_stream_t stream) {
  NSKY_CHECK_ARG(segments && total && n_segments > 0 && n_segments <= NSKY_TOTAL_MAX_SEGMENTS, "nsky_weighted_total_fwd: bad argument");
  TotalArgs a;
  a.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    NSKY_CHECK_ARG(segments[i].x && segments[i].n >= 0, "nsky_weighted_total_fwd: bad segment");
    a.s[i] = segments[i];
  }
  hipLaunchKernelGGL(weighted_total_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, total);
  NSKY_CHECK_LAUNCH("nsky_weighted_total_fwd");
  return NSKY_OK;
}

extern "C" int nsky_weighted_total_bwd(const nsky_total_segment* segments, int32_t n_segments, const float* g, nsky_stream_t stream) {
  NSKY_CHECK_ARG(segments && g && n_segments > 0 && n_segments <= NSKY_TOTAL_MAX_SEGMENTS, "nsky_weighted_total_bwd: bad argument");
  TotalArgs a;
  a.n = n_segments;
  int nmax = 1;
  for (int i = 0; i < n_segments; ++i) { a.s[i] = segments[i]; nmax = segments[i].n > nmax ? segments[i].n : nmax; }
  hipLaunchKernelGGL(weighted_total_bwd_kernel, dim3((nmax + 255) / 256 > 64 ? 64 : (nmax + 255) / 256, n_segments), dim3(256), 0, (hipStream_t)stream, a, g);
  NSKY_CHECK_LAUNCH("nsky_weighted_total_bwd");
  return NSKY_OK;
}

extern "C" int nsky_main_losses_fwd(const nsky_main_losses_desc* d, float* terms, float* wsum, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && terms && d->R > 0 && d->mask && d->image, "nsky_main_losses_fwd: null argument");
  NSKY_CHECK_ARG(!d->weights || (wsum && d->S > 0), "nsky_main_losses_fwd: weights need S and the wsum side output");
  NSKY_CHECK_ARG(!d->eik || d->S > 0, "nsky_main_losses_fwd: eik needs S");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(terms, 0, NT_MAIN * sizeof(float), s) != hipSuccess) { nsky_set_error("nsky_main_losses_fwd: memset failed"); return NSKY_ERR_LAUNCH; }
  const long total = d->R + (d->eik ? (long)d->R * d->S : 0) + (d->grid ? 3l * d->P : 0) + (d->sdf_term ? d->M : 0);
  hipLaunchKernelGGL(main_losses_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, *d, terms, wsum);
  NSKY_CHECK_LAUNCH("nsky_main_losses_fwd");
  return NSKY_OK;
}

extern "C" int nsky_main_losses_bwd(const nsky_main_losses_desc* d, const float* wsum, const float* d_terms, float* d_rgb, float* d_eik,
                                    float* d_weights, float* d_normal, float* d_hdr_bg, float* d_grid, float* d_sdf_term,
                                    float* d_vis_thr, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && d_terms && d->R > 0 && d->mask && d->image, "nsky_main_losses_bwd: null argument");
  NSKY_CHECK_ARG((!d_rgb || d->rgb) && (!d_eik || d->eik) && (!d_weights || (d->weights && wsum)) && (!d_normal || d->normal) &&
                     (!d_hdr_bg || d->hdr_bg) && (!d_grid || d->grid) && (!d_sdf_term || d->sdf_term) && (!d_vis_thr || d->vis_thr),
                 "nsky_main_losses_bwd: a gradient was requested for an absent input");
  const long N = (long)d->R * d->S;
  const long total = d->R + (d_weights ? N : 0) + (d_eik ? N : 0) + (d_grid ? 3l * d->P : 0) + (d_sdf_term ? d->M : 0);
  hipLaunchKernelGGL(main_losses_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d, wsum, d_terms, d_rgb, d_eik,
                     d_weights, d_normal, d_hdr_bg, d_grid, d_sdf_term, d_vis_thr);
  NSKY_CHECK_LAUNCH("nsky_main_losses_bwd");
  return NSKY_OK;
}

extern "C" int nsky_ddf_losses_fwd(const nsky_ddf_losses_desc* d, float* terms, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && terms && d->Mr >= 0 && d->Mm >= 0 && d->Ms >= 0, "nsky_ddf_losses_fwd: null argument");
  NSKY_CHECK_ARG(d->Mr == 0 || (d->mask && (!d->want_depth || (d->expected && d->term))), "nsky_ddf_losses_fwd: depth term needs expected / term / mask");
  NSKY_CHECK_ARG(d->Mm == 0 || (d->mv_expected && d->mv_term), "nsky_ddf_losses_fwd: multi-view term needs both distance vectors");
  NSKY_CHECK_ARG(d->Ms == 0 || (d->sky_expected && d->sky_term), "nsky_ddf_losses_fwd: sky-ray term needs both distance vectors");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(terms, 0, NT_DDF * sizeof(float), s) != hipSuccess) { nsky_set_error("nsky_ddf_losses_fwd: memset failed"); return NSKY_ERR_LAUNCH; }
  const long total = (long)d->Mr + d->Mm + d->Ms;
  if (total == 0) return NSKY_OK;
  hipLaunchKernelGGL(ddf_losses_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, *d, terms);
  NSKY_CHECK_LAUNCH("nsky_ddf_losses_fwd");
  return NSKY_OK;
}

extern "C" int nsky_ddf_losses_bwd(const nsky_ddf_losses_desc* d, const float* d_terms, float* d_expected, float* d_sdf, float* d_mv_expected,
                                   float* d_sky_expected, float* d_term, float* d_mv_term, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && d_terms, "nsky_ddf_losses_bwd: null argument");
  NSKY_CHECK_ARG((!d_sdf || d->sdf) && (!d_expected || d->expected), "nsky_ddf_losses_bwd: a gradient was requested for an absent input");
  const long total = (long)d->Mr + d->Mm + d->Ms;
  if (total == 0) return NSKY_OK;
  hipLaunchKernelGGL(ddf_losses_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *d, d_terms, d_expected, d_sdf,
                     d_mv_expected, d_sky_expected, d_term, d_mv_term);
  NSKY_CHECK_LAUNCH("nsky_ddf_losses_bwd");
  return NSKY_OK;
}

extern "C" int nsky_train_metrics(const float* pred, const float* gt, const float* mask, int64_t n, float peak_sq, const float* variance, float* out,
                                  nsky_stream_t stream) {
  NSKY_CHECK_ARG(pred && gt && out && n > 0 && peak_sq > 0.0f, "nsky_train_metrics: bad argument");
  hipLaunchKernelGGL(train_metrics_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, mask, (long)n, peak_sq, variance, out);
  NSKY_CHECK_LAUNCH("nsky_train_metrics");
  return NSKY_OK;
}
