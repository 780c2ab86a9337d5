// Sample generators of the train step that the reference draws on the host with torch's RNG (and uploads): here each is ONE
// kernel with a counter-based generator (Philox4x32-10, Salmon et al. 2011), keyed by (seed, call number) so that a
// replayed HIP graph draws fresh numbers: the call number lives in device memory and is advanced by a one-thread kernel
// behind every draw.  The RNG streams are not the reference's (they are not portable anyway): the DISTRIBUTIONS are, and
// the tests check those.
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

constexpr float TWO_PI = 6.283185307179586f;

struct Philox {
  uint32_t k0, k1;
  __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // four 32-bit words for counter (a, b, call number)
  __device__ __forceinline__ void draw(uint32_t a, uint32_t b, uint64_t call, uint32_t (&out)[4]) const {
    uint32_t c[4] = {a, b, (uint32_t)call, (uint32_t)(call >> 32)};
    uint32_t x = k0, y = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, x, y);
      x += 0x9E3779B9u;
      y += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
  }
};

// uniform in (0, 1), both ends excluded IN FLOAT: 23 bits + 1/2 is exact in fp32 (24 bits + 1/2 rounds the top value up to 1.0, which as
// cos(phi) = 2 u - 1 = 1 puts a multi-view point exactly on the pole, where the local frame of ddf_model.py:158-181 divides 0 by 0: one NaN
// per ~10 k training steps)
__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }

__global__ void advance_counter_kernel(uint64_t* counter) { *counter += 1; }

// VMFDDFSampler (neusky/model_components/ddf_sampler.py:205-286): num_positions points on the unit sphere (upper hemisphere),
// per point num_directions directions from a von Mises-Fisher-like lobe around the inward normal (Wood's rejection sampler with
// the reference's acceptance test `>= -e`, :220), flipped into the inward half space.  One thread per (position, direction).
__global__ __launch_bounds__(256) void ddf_vmf_samples_kernel(int n_pos, int n_dir, float kappa, float radius, int upper, uint64_t seed,
                                                              const uint64_t* __restrict__ counter, float* __restrict__ origins,
                                                              float* __restrict__ directions) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pos * n_dir) return;
  const int n = i / n_dir;
  const uint64_t call = *counter;
  Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  // the position of sample n: the same words in every thread of that position (stream 0)
  rng.draw((uint32_t)n, 0u, call, w);
  const float theta = TWO_PI * u01(w[0]);
  const float cphi = 2.0f * u01(w[1]) - 1.0f;
  const float sphi = sqrtf(fmaxf(0.0f, 1.0f - cphi * cphi));
  float p[3] = {sphi * cosf(theta), sphi * sinf(theta), cphi};
  if (upper && p[2] < 0.0f) { p[0] = -p[0]; p[1] = -p[1]; p[2] = -p[2]; }
  const float nrm[3] = {-p[0], -p[1], -p[2]};
  // tangent direction: a normal 3-vector, projected off the normal (stream 1, counter = sample)
  rng.draw((uint32_t)i, 1u, call, w);
  const float r0 = sqrtf(-2.0f * logf(u01(w[0]))), r1 = sqrtf(-2.0f * logf(u01(w[2])));
  float z[3] = {r0 * cosf(TWO_PI * u01(w[1])), r0 * sinf(TWO_PI * u01(w[1])), r1 * cosf(TWO_PI * u01(w[3]))};
  float inv = rsqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  z[0] *= inv; z[1] *= inv; z[2] *= inv;
  const float zn = z[0] * nrm[0] + z[1] * nrm[1] + z[2] * nrm[2];
  z[0] -= zn * nrm[0]; z[1] -= zn * nrm[1]; z[2] -= zn * nrm[2];
  inv = rsqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  z[0] *= inv; z[1] *= inv; z[2] *= inv;
  // cosine to the normal: Wood (1994), d = 3 (Beta(1, 1) = U(0, 1)); streams 2.. until accepted
  const float dm1 = 2.0f;
  const float b = dm1 / (2.0f * kappa + sqrtf(4.0f * kappa * kappa + dm1 * dm1));
  const float x0 = (1.0f - b) / (1.0f + b);
  const float c = kappa * x0 + dm1 * logf(1.0f - x0 * x0);
  float t = x0;
  bool done = false;
  for (uint32_t s = 2; s < 10 && !done; ++s) {
    rng.draw((uint32_t)i, s, call, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (done) break;
      const float u = u01(w[k]);
      const float tt = (1.0f - (1.0f + b) * u) / (1.0f - (1.0f - b) * u);
      if (kappa * tt + dm1 * logf(1.0f - x0 * tt) - c >= -2.718281828459045f) { t = tt; done = true; }
    }
  }
  const float st = sqrtf(fmaxf(0.0f, 1.0f - t * t));
  float x[3] = {z[0] * st + t * nrm[0], z[1] * st + t * nrm[1], z[2] * st + t * nrm[2]};
  inv = rsqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (x[0] * nrm[0] + x[1] * nrm[1] + x[2] * nrm[2] < 0.0f) inv = -inv;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    origins[(long)i * 3 + a] = p[a] * radius;
    directions[(long)i * 3 + a] = x[a] * inv;
  }
}

// ---- query rows of the DDF-fit losses ------------------------------------------------------------------------------------
// DDFModel.get_outputs (neusky/models/ddf_model.py:193-219, :279-321, :324-360) evaluates the DDF on three sets of rays: the fit
// rays themselves, one multi-view ray per fit ray (from a random point of the upper hemisphere towards the fit ray's ground-truth
// termination point) and the sky rays traced backwards from where they leave the sphere.  This kernel writes, one thread per
// evaluation, the sphere position and the encoded local direction row [d_loc | NeRF2(d_loc) | 0] (get_localised_transforms
// :158-181, directional_distance_field.py:188-191,270-271) of all of them, plus the by-products the losses use.
__device__ __forceinline__ void local_frame(const float pos[3], float x[3], float y[3], float z[3]) {
  y[0] = -pos[0]; y[1] = -pos[1]; y[2] = -pos[2];
  x[0] = -y[1]; x[1] = y[0]; x[2] = 0.0f;  // cross(up = (0, 0, 1), y)
  const float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (xn > 0.0f) { x[0] /= xn; x[1] /= xn; x[2] /= xn; }
  else { x[0] = 1.0f; x[1] = 0.0f; x[2] = 0.0f; }  // a position ON the pole (the reference divides 0 by 0 there): any unit vector across the axis
  z[0] = y[1] * x[2] - y[2] * x[1]; z[1] = y[2] * x[0] - y[0] * x[2]; z[2] = y[0] * x[1] - y[1] * x[0];  // cross(y, x)
  const float zn = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  z[0] /= zn; z[1] /= zn; z[2] /= zn;
}

struct FitRowsArgs {
  const float* positions; const float* directions; const float* term_dist;  // fit rays [N,3] [N,3] [N]
  const float* mv_in;     // [N,3] or null: drawn here
  const float* sky_o; const float* sky_d;  // [Ns,3]
  const uint64_t* counter;
  uint64_t seed;
  float* q_pos; float* xrow; float* mv_out; float* sky_gt; float* dist_weight;
  int N, Ns, ldx, want_mv, include_z;
  float radius, weight_exp;
};

__device__ __forceinline__ void mv_point(const FitRowsArgs& a, int j, float pts[3]) {
  if (a.mv_in) {
    pts[0] = a.mv_in[j * 3]; pts[1] = a.mv_in[j * 3 + 1]; pts[2] = a.mv_in[j * 3 + 2];
  } else {  // random_points_on_unit_sphere (:290)
    Philox rng{(uint32_t)a.seed, (uint32_t)(a.seed >> 32)};
    uint32_t w[4];
    rng.draw((uint32_t)j, 0x6d76u, *a.counter, w);
    const float theta = TWO_PI * u01(w[0]);
    const float cphi = 2.0f * u01(w[1]) - 1.0f;
    const float sphi = sqrtf(fmaxf(0.0f, 1.0f - cphi * cphi));
    pts[0] = sphi * cosf(theta); pts[1] = sphi * sinf(theta); pts[2] = cphi;
  }
  pts[2] = fabsf(pts[2]);  // :295
}

__global__ __launch_bounds__(256) void ddf_fit_rows_fwd_kernel(FitRowsArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int n_mv = a.want_mv ? a.N : 0;
  if (e >= a.N + n_mv + a.Ns) return;
  float pos[3], wd[3];
  if (e < a.N) {
    for (int c = 0; c < 3; ++c) { pos[c] = a.positions[e * 3 + c]; wd[c] = a.directions[e * 3 + c]; }
    if (a.dist_weight) {  // :224-238
      const float d2 = pos[0] * pos[0] + pos[1] * pos[1] + (a.include_z ? pos[2] * pos[2] : 0.0f);
      a.dist_weight[e] = 1.0f - powf(sqrtf(d2) / a.radius, a.weight_exp);
    }
  } else if (e < a.N + n_mv) {
    const int j = e - a.N;
    float pts[3];
    mv_point(a, j, pts);
    const float t = a.term_dist[j];
    float dv[3];
    for (int c = 0; c < 3; ++c) dv[c] = a.positions[j * 3 + c] + a.directions[j * 3 + c] * t - pts[c];  // :287, :298
    const float len = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
    for (int c = 0; c < 3; ++c) { pos[c] = pts[c]; wd[c] = dv[c] / len; a.mv_out[j * 3 + c] = pts[c]; }
  } else {
    const int j = e - a.N - n_mv;
    float o[3], d[3];
    for (int c = 0; c < 3; ++c) { o[c] = a.sky_o[j * 3 + c]; d[c] = a.sky_d[j * 3 + c]; }
    // ray_sphere_intersection (neusky/utils/utils.py:68-93): unit directions, far root, no clamping
    const float b = 2.0f * (d[0] * o[0] + d[1] * o[1] + d[2] * o[2]);
    const float cc = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - a.radius * a.radius;
    const float sq = sqrtf(b * b - 4.0f * cc);
    const float t = fmaxf((-b - sq) * 0.5f, (-b + sq) * 0.5f);
    float g2 = 0.0f;
    for (int c = 0; c < 3; ++c) { pos[c] = o[c] + t * d[c]; wd[c] = -d[c]; g2 += (o[c] - pos[c]) * (o[c] - pos[c]); }
    a.sky_gt[j] = sqrtf(g2);  // :343
  }
  float x[3], y[3], z[3];
  local_frame(pos, x, y, z);
  const float dl[3] = {x[0] * wd[0] + x[1] * wd[1] + x[2] * wd[2], y[0] * wd[0] + y[1] * wd[1] + y[2] * wd[2],
                       z[0] * wd[0] + z[1] * wd[1] + z[2] * wd[2]};
  float* row = a.xrow + (long)e * a.ldx;
  for (int c = 0; c < 3; ++c) { a.q_pos[(long)e * 3 + c] = pos[c]; row[c] = dl[c]; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float arg = TWO_PI * dl[i] * (f == 0 ? 1.0f : 4.0f);
      row[3 + i * 2 + f] = sinf(arg);
      row[9 + i * 2 + f] = sinf(arg + 1.5707963267948966f);
    }
  for (int c = 15; c < a.ldx; ++c) row[c] = 0.0f;
}

// gradient of the multi-view rows w.r.t. the fit rays' termination distance (the only differentiable input: the ground truth is
// rendered from the SDF field and, with stop_sdf_gradients = False (neusky_config.py:45), trains it)
__global__ __launch_bounds__(256) void ddf_fit_rows_bwd_kernel(const float* __restrict__ positions, const float* __restrict__ directions,
                                                               const float* __restrict__ term_dist, const float* __restrict__ mv_points,
                                                               const float* __restrict__ d_xrow, int ldx, int N,
                                                               float* __restrict__ d_term_dist) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const float pts[3] = {mv_points[j * 3], mv_points[j * 3 + 1], mv_points[j * 3 + 2]};
  const float t = term_dist[j];
  float dir[3], dv[3];
  for (int c = 0; c < 3; ++c) { dir[c] = directions[j * 3 + c]; dv[c] = positions[j * 3 + c] + dir[c] * t - pts[c]; }
  const float len = sqrtf(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
  const float u[3] = {dv[0] / len, dv[1] / len, dv[2] / len};
  float x[3], y[3], z[3];
  local_frame(pts, x, y, z);
  const float dl[3] = {x[0] * u[0] + x[1] * u[1] + x[2] * u[2], y[0] * u[0] + y[1] * u[1] + y[2] * u[2], z[0] * u[0] + z[1] * u[1] + z[2] * u[2]};
  const float* g = d_xrow + (long)j * ldx;
  float gl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = g[i];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float w = TWO_PI * (f == 0 ? 1.0f : 4.0f);
      const float arg = w * dl[i];
      acc += w * (g[3 + i * 2 + f] * cosf(arg) + g[9 + i * 2 + f] * cosf(arg + 1.5707963267948966f));
    }
    gl[i] = acc;
  }
  float gu[3];
  for (int c = 0; c < 3; ++c) gu[c] = x[c] * gl[0] + y[c] * gl[1] + z[c] * gl[2];
  const float ug = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
  float gt = 0.0f;
  for (int c = 0; c < 3; ++c) gt += (gu[c] - u[c] * ug) / len * dir[c];
  d_term_dist[j] = gt;
}

// ---- RENI++ decoder inputs for every (latent set, direction) pair -------------------------------------------------------
// RENIField's rotation-invariant conditioning (reni_field.py invariant representation, used at neusky_model.py:1207-1252): for
// latent codes Z [U,L,3] and directions d [D,3], row u D + d of the mapping network's input is [|Z_xy|, Z_z, Z_xy . d_xy] per
// latent (3 L columns) and the FiLM input row is [|d_xy|, d_z | NeRF2 of those] (10 columns).  One thread per (row, latent);
// the backward sums a row gradient back onto the latent codes (the decoder is frozen, the codes are trained).
__global__ __launch_bounds__(256) void reni_grid_inputs_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ dirs, int U, int L,
                                                                   int D, const float* __restrict__ ray_dirs,
                                                                   const long* __restrict__ ray_latent, int R, float* __restrict__ cond,
                                                                   int ldc, float* __restrict__ xrow, int ldx) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ((long)U * D + R) * L) return;
  const int l = (int)(t % L);
  const long row = t / L;
  int u;
  const float* dp;
  if (row < (long)U * D) { u = (int)(row / D); dp = dirs + (row % D) * 3; }
  else { const long r = row - (long)U * D; u = (int)ray_latent[r]; dp = ray_dirs + r * 3; }  // the rays' own rows (:535-549)
  const float zx = Z[((long)u * L + l) * 3], zy = Z[((long)u * L + l) * 3 + 1], zz = Z[((long)u * L + l) * 3 + 2];
  const float dx = dp[0], dy = dp[1];
  float* c = cond + row * ldc + 3 * l;
  c[0] = sqrtf(zx * zx + zy * zy);
  c[1] = zz;
  c[2] = zx * dx + zy * dy;
  if (l == 0) {
    for (int k = 3 * L; k < ldc; ++k) cond[row * ldc + k] = 0.0f;
    float* x = xrow + row * ldx;
    const float v[2] = {sqrtf(dx * dx + dy * dy), dp[2]};
    x[0] = v[0]; x[1] = v[1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float arg = TWO_PI * v[i] * (f == 0 ? 1.0f : 4.0f);
        x[2 + i * 2 + f] = sinf(arg);
        x[6 + i * 2 + f] = sinf(arg + 1.5707963267948966f);
      }
    for (int k = 10; k < ldx; ++k) x[k] = 0.0f;
  }
}

// d_Z (zero-filled by the launcher) += the grid rows' gradient.  Workgroup = (latent set u, 32 consecutive directions), thread =
// one of the 3 L columns of the row gradient: 32 independent 4-byte loads per thread (whole 1200-byte rows per wave group), then
// one or two float atomics per thread.
constexpr int RENI_BWD_ROWS = 32;
__global__ __launch_bounds__(320) void reni_grid_inputs_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ dirs, int U, int L,
                                                                   int D, const float* __restrict__ d_cond, int ldc,
                                                                   float* __restrict__ d_Z) {
  const int u = blockIdx.y, d0 = blockIdx.x * RENI_BWD_ROWS;
  for (int col = threadIdx.x; col < 3 * L; col += blockDim.x) {
    const int l = col / 3, k = col % 3;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int j = 0; j < RENI_BWD_ROWS; ++j) {
      const int d = d0 + j;
      if (d < D) {
        const float g = d_cond[((long)u * D + d) * ldc + col];
        if (k == 2) { s0 = fmaf(g, dirs[d * 3], s0); s1 = fmaf(g, dirs[d * 3 + 1], s1); }
        else s0 += g;
      }
    }
    float* o = d_Z + ((long)u * L + l) * 3;
    if (k == 0) {
      const float zx = Z[((long)u * L + l) * 3], zy = Z[((long)u * L + l) * 3 + 1];
      const float n = sqrtf(zx * zx + zy * zy);
      if (n > 0.0f) { atomicAdd(o, s0 * zx / n); atomicAdd(o + 1, s0 * zy / n); }
    } else if (k == 1) {
      atomicAdd(o + 2, s0);
    } else {
      atomicAdd(o, s0); atomicAdd(o + 1, s1);
    }
  }
}

// the rays' rows add onto their camera's latent gradient (after the kernel above has written it)
__global__ __launch_bounds__(256) void reni_ray_inputs_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ ray_dirs,
                                                                  const long* __restrict__ ray_latent, int R, int L,
                                                                  const float* __restrict__ d_cond_rays, int ldc, float* __restrict__ d_Z) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= R * L) return;
  const int l = t % L, r = t / L;
  const long u = ray_latent[r];
  const float zx = Z[(u * L + l) * 3], zy = Z[(u * L + l) * 3 + 1];
  const float* g = d_cond_rays + (long)r * ldc + 3 * l;
  float gx = g[2] * ray_dirs[r * 3], gy = g[2] * ray_dirs[r * 3 + 1];
  const float n = sqrtf(zx * zx + zy * zy);
  if (n > 0.0f) { gx += g[0] * zx / n; gy += g[0] * zy / n; }
  float* o = d_Z + (u * L + l) * 3;
  atomicAdd(o, gx); atomicAdd(o + 1, gy); atomicAdd(o + 2, g[1]);
}

// Probe points of the hash-grid density loss (neusky_model.py:704-724): every lattice point jittered uniformly inside its cell
// (lattice + (u gap - gap / 2), u ~ U(0,1)^3) with a uniformly random unit direction (a normalised normal 3-vector).
__global__ __launch_bounds__(256) void grid_probe_points_kernel(const float* __restrict__ lattice, float gx, float gy, float gz, int P, uint64_t seed,
                                                                const uint64_t* __restrict__ counter, float* __restrict__ pos,
                                                                float* __restrict__ dirs) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const Philox g{(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t a[4], b[4];
  g.draw(0x47524944u, (uint32_t)p, *counter, a);
  g.draw(0x47524945u, (uint32_t)p, *counter, b);
  pos[3 * p] = lattice[3 * p] + (u01(a[0]) * gx - 0.5f * gx);
  pos[3 * p + 1] = lattice[3 * p + 1] + (u01(a[1]) * gy - 0.5f * gy);
  pos[3 * p + 2] = lattice[3 * p + 2] + (u01(a[2]) * gz - 0.5f * gz);
  const float r0 = sqrtf(-2.0f * logf(u01(b[0]))), r1 = sqrtf(-2.0f * logf(u01(b[2])));
  const float x = r0 * cosf(TWO_PI * u01(b[1])), y = r0 * sinf(TWO_PI * u01(b[1])), z = r1 * cosf(TWO_PI * u01(b[3]));
  const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
  dirs[3 * p] = x * inv; dirs[3 * p + 1] = y * inv; dirs[3 * p + 2] = z * inv;
}

// The decoder's HDR output for the direction grid and the batch's own rays (neusky_model.py:488-549: exp output activation of the
// RENI++ head, unnormalised by the per-image scale): grid[u, d] = exp(raw[u D + d]) scale[u], rays[r] = exp(raw[U D + r]) scale[ray_latent[r]].
__global__ __launch_bounds__(256) void reni_output_fwd_kernel(const float* __restrict__ raw, int ldr, const float* __restrict__ scale,
                                                              const long* __restrict__ ray_latent, int U, int D, int R,
                                                              float* __restrict__ grid, float* __restrict__ rays) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= U * D + R) return;
  const float s = scale[i < U * D ? i / D : ray_latent[i - U * D]];
  float* o = i < U * D ? grid + 3l * i : rays + 3l * (i - U * D);
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = expf(raw[(long)i * ldr + c]) * s;
}

// d_raw = d_out exp(raw) scale (pad columns zero), d_scale[u] += sum d_out exp(raw): one atomic per wave and image for the grid rows
// (consecutive rows share their image), one per row for the rays
__global__ __launch_bounds__(256) void reni_output_bwd_kernel(const float* __restrict__ raw, int ldr, const float* __restrict__ scale,
                                                              const long* __restrict__ ray_latent, int U, int D, int R,
                                                              const float* __restrict__ d_grid, const float* __restrict__ d_rays,
                                                              float* __restrict__ d_raw, float* __restrict__ d_scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = U * D + R;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const bool is_grid = ic < U * D;
  const int u = is_grid ? ic / D : (int)ray_latent[ic - U * D];
  const float* g = is_grid ? (d_grid ? d_grid + 3l * ic : nullptr) : (d_rays ? d_rays + 3l * (ic - U * D) : nullptr);
  const float s = scale[u];
  float acc = 0.0f;
  float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (live && g) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e = expf(raw[(long)ic * ldr + c]);
      out[c] = g[c] * e * s;
      acc += g[c] * e;
    }
  }
  if (live) {
    for (int c = 0; c < ldr; ++c) d_raw[(long)ic * ldr + c] = c < 3 ? out[c] : 0.0f;
  }
  if (!d_scale) return;
  acc = live ? acc : 0.0f;
  // segmented wave reduction over runs of equal u (sorted for the grid rows; arbitrary for the rays: those fall back to per-lane atomics)
  const int lane = threadIdx.x & 63;
  const int u0 = __shfl(u, 0, 64);
  const bool uniform = __all(!live || (is_grid && u == u0));
  if (uniform) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0 && acc != 0.0f) atomicAdd(d_scale + u0, acc);
  } else if (live && acc != 0.0f) {
    atomicAdd(d_scale + u, acc);
  }
}

// IcosahedronSampler with apply_random_rotation (neusky/model_components/illumination_samplers.py:75-110 -> neusky_model.py:456-458)
// and the upper-hemisphere subset of the rotated set (neusky_model.py:1650-1657) in ONE workgroup: a uniform random rotation from a
// unit quaternion of four normals (or the caller's matrix), dirs = base R^T, and sel = the indices of the D / 2 directions with the
// largest z in ascending order (for a centrally symmetric set this IS the z > 0 subset, with a size a replayed graph can rely on).
// The call counter is advanced by the kernel itself (one workgroup: every reader is behind the barrier).
__global__ __launch_bounds__(1024) void illumination_directions_kernel(const float* __restrict__ base, int D, const float* __restrict__ rotation_in,
                                                                       uint64_t seed, uint64_t* __restrict__ counter, float* __restrict__ dirs,
                                                                       int32_t* __restrict__ sel, float* __restrict__ rot_out) {
  __shared__ float R[9];
  __shared__ float zs[1024];
  __shared__ int wave_total[16];
  const int t = threadIdx.x;
  if (t == 0) {
    if (rotation_in) {
      for (int i = 0; i < 9; ++i) R[i] = rotation_in[i];
    } else {
      const Philox g{(uint32_t)seed, (uint32_t)(seed >> 32)};
      uint32_t w[4];
      g.draw(0x4c494748u, 0u, *counter, w);
      const float r0 = sqrtf(-2.0f * logf(u01(w[0]))), r1 = sqrtf(-2.0f * logf(u01(w[2])));
      float q[4] = {r0 * cosf(TWO_PI * u01(w[1])), r0 * sinf(TWO_PI * u01(w[1])), r1 * cosf(TWO_PI * u01(w[3])), r1 * sinf(TWO_PI * u01(w[3]))};
      const float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      const float qw = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
      const float c = 2.0f * qw * qw - 1.0f;  // R = (2 w^2 - 1) I + 2 (v v^T + w [v]x)
      R[0] = c + 2.0f * x * x;        R[1] = 2.0f * (x * y - qw * z); R[2] = 2.0f * (x * z + qw * y);
      R[3] = 2.0f * (y * x + qw * z); R[4] = c + 2.0f * y * y;        R[5] = 2.0f * (y * z - qw * x);
      R[6] = 2.0f * (z * x - qw * y); R[7] = 2.0f * (z * y + qw * x); R[8] = c + 2.0f * z * z;
    }
    if (rot_out)
      for (int i = 0; i < 9; ++i) rot_out[i] = R[i];
  }
  __syncthreads();
  if (t == 0 && !rotation_in) *counter += 1;
  float z = -INFINITY;
  if (t < D) {
    const float bx = base[3 * t], by = base[3 * t + 1], bz = base[3 * t + 2];
    dirs[3 * t] = R[0] * bx + R[1] * by + R[2] * bz;
    dirs[3 * t + 1] = R[3] * bx + R[4] * by + R[5] * bz;
    z = R[6] * bx + R[7] * by + R[8] * bz;
    dirs[3 * t + 2] = z;
  }
  zs[t] = z;
  __syncthreads();
  int rank = 0;
  for (int j = 0; j < D; ++j) {
    const float zj = zs[j];
    rank += (zj > z || (zj == z && j < t)) ? 1 : 0;
  }
  const bool chosen = t < D && rank < D / 2;
  const unsigned long long mask = __ballot(chosen);
  const int lane = t & 63, wave = t >> 6;
  if (lane == 0) wave_total[wave] = __popcll(mask);
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wave_total[w];
  if (chosen) sel[before + __popcll(mask & ((1ull << lane) - 1ull))] = t;
}

}  // namespace

extern "C" int nsky_ddf_vmf_samples(int32_t n_positions, int32_t n_directions, float kappa, float radius, int32_t upper_hemisphere,
                                    uint64_t seed, uint64_t* counter, float* origins, float* directions, nsky_stream_t stream) {
  NSKY_CHECK_ARG(n_positions >= 0 && n_directions >= 0 && counter && kappa > 0.0f, "nsky_ddf_vmf_samples: bad argument");
  const long n = (long)n_positions * n_directions;
  if (n == 0) return NSKY_OK;
  NSKY_CHECK_ARG(origins && directions, "nsky_ddf_vmf_samples: null output");
  hipLaunchKernelGGL(ddf_vmf_samples_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n_positions, n_directions, kappa,
                     radius, upper_hemisphere, seed, counter, origins, directions);
  hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  NSKY_CHECK_LAUNCH("nsky_ddf_vmf_samples");
  return NSKY_OK;
}

extern "C" int nsky_ddf_fit_rows_fwd(const float* positions, const float* directions, const float* term_dist, int32_t N,
                                     const float* mv_points_in, uint64_t seed, uint64_t* counter, const float* sky_o, const float* sky_d,
                                     int32_t Ns, float radius, int32_t want_mv, float weight_exp, int32_t weight_include_z, float* q_pos,
                                     float* xrow, int32_t ldx, float* mv_points_out, float* sky_gt, float* distance_weight,
                                     nsky_stream_t stream) {
  NSKY_CHECK_ARG(N >= 0 && Ns >= 0 && ldx >= 15 && radius > 0.0f, "nsky_ddf_fit_rows_fwd: bad sizes");
  const int n_mv = want_mv ? N : 0;
  const int E = N + n_mv + Ns;
  if (E == 0) return NSKY_OK;
  NSKY_CHECK_ARG(q_pos && xrow && (N == 0 || (positions && directions)) && (n_mv == 0 || (term_dist && mv_points_out)) &&
                     (Ns == 0 || (sky_o && sky_d && sky_gt)), "nsky_ddf_fit_rows_fwd: null argument");
  NSKY_CHECK_ARG(n_mv == 0 || mv_points_in || counter, "nsky_ddf_fit_rows_fwd: multi-view points are neither given nor can they be drawn (no counter)");
  FitRowsArgs a;
  a.positions = positions; a.directions = directions; a.term_dist = term_dist; a.mv_in = mv_points_in; a.sky_o = sky_o; a.sky_d = sky_d;
  a.counter = counter; a.seed = seed; a.q_pos = q_pos; a.xrow = xrow; a.mv_out = mv_points_out; a.sky_gt = sky_gt;
  a.dist_weight = distance_weight; a.N = N; a.Ns = Ns; a.ldx = ldx; a.want_mv = want_mv; a.include_z = weight_include_z;
  a.radius = radius; a.weight_exp = weight_exp;
  hipLaunchKernelGGL(ddf_fit_rows_fwd_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, (hipStream_t)stream, a);
  if (n_mv > 0 && !mv_points_in) hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  NSKY_CHECK_LAUNCH("nsky_ddf_fit_rows_fwd");
  return NSKY_OK;
}

extern "C" int nsky_ddf_fit_rows_bwd(const float* positions, const float* directions, const float* term_dist, const float* mv_points,
                                     int32_t N, const float* d_xrow_mv, int32_t ldx, float* d_term_dist, nsky_stream_t stream) {
  if (N == 0) return NSKY_OK;
  NSKY_CHECK_ARG(positions && directions && term_dist && mv_points && d_xrow_mv && d_term_dist && N > 0 && ldx >= 15,
                 "nsky_ddf_fit_rows_bwd: bad argument");
  hipLaunchKernelGGL(ddf_fit_rows_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, positions, directions, term_dist,
                     mv_points, d_xrow_mv, ldx, N, d_term_dist);
  NSKY_CHECK_LAUNCH("nsky_ddf_fit_rows_bwd");
  return NSKY_OK;
}

extern "C" int nsky_reni_grid_inputs_fwd(const float* latents, const float* directions, int32_t U, int32_t L, int32_t D,
                                         const float* ray_dirs, const int64_t* ray_latent, int32_t R, float* cond, int32_t ldcond,
                                         float* xrow, int32_t ldx, nsky_stream_t stream) {
  const long n = ((long)U * D + R) * L;
  if (n == 0) return NSKY_OK;
  NSKY_CHECK_ARG(latents && cond && xrow && ldcond >= 3 * L && ldx >= 10 && (D == 0 || directions) && (R == 0 || (ray_dirs && ray_latent)),
                 "nsky_reni_grid_inputs_fwd: bad argument");
  hipLaunchKernelGGL(reni_grid_inputs_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, latents, directions, U, L, D,
                     ray_dirs, reinterpret_cast<const long*>(ray_latent), R, cond, ldcond, xrow, ldx);
  NSKY_CHECK_LAUNCH("nsky_reni_grid_inputs_fwd");
  return NSKY_OK;
}

extern "C" int nsky_reni_grid_inputs_bwd(const float* latents, const float* directions, int32_t U, int32_t L, int32_t D,
                                         const float* ray_dirs, const int64_t* ray_latent, int32_t R, const float* d_cond, int32_t ldcond,
                                         float* d_latents, nsky_stream_t stream) {
  if ((long)U * L == 0) return NSKY_OK;
  NSKY_CHECK_ARG(latents && d_cond && d_latents && ldcond >= 3 * L && (D == 0 || directions) && (R == 0 || (ray_dirs && ray_latent)),
                 "nsky_reni_grid_inputs_bwd: bad argument");
  if (hipMemsetAsync(d_latents, 0, sizeof(float) * 3 * (size_t)U * L, (hipStream_t)stream) != hipSuccess) {
    nsky_set_error("nsky_reni_grid_inputs_bwd: memset failed");
    return NSKY_ERR_LAUNCH;
  }
  if (D > 0)
    hipLaunchKernelGGL(reni_grid_inputs_bwd_kernel, dim3(ceil_div(D, RENI_BWD_ROWS), U), dim3(320), 0, (hipStream_t)stream, latents,
                       directions, U, L, D, d_cond, ldcond, d_latents);
  if (R > 0)
    hipLaunchKernelGGL(reni_ray_inputs_bwd_kernel, dim3(ceil_div((long)R * L, 256)), dim3(256), 0, (hipStream_t)stream, latents, ray_dirs,
                       reinterpret_cast<const long*>(ray_latent), R, L, d_cond + (long)U * D * ldcond, ldcond, d_latents);
  NSKY_CHECK_LAUNCH("nsky_reni_grid_inputs_bwd");
  return NSKY_OK;
}

extern "C" int nsky_illumination_directions(const float* base, int32_t D, const float* rotation_in, uint64_t seed, uint64_t* counter, float* dirs,
                                            int32_t* sel, float* rot_out, nsky_stream_t stream) {
  NSKY_CHECK_ARG(base && dirs && sel && D >= 2 && D <= 1024 && D % 2 == 0 && (rotation_in || counter),
                 "nsky_illumination_directions: bad argument (D %d: even, <= 1024)", D);
  hipLaunchKernelGGL(illumination_directions_kernel, dim3(1), dim3((D + 63) / 64 * 64), 0, (hipStream_t)stream, base, D, rotation_in, seed, counter,
                     dirs, sel, rot_out);
  NSKY_CHECK_LAUNCH("nsky_illumination_directions");
  return NSKY_OK;
}

extern "C" int nsky_reni_output_fwd(const float* raw, int32_t ldr, const float* scale, const int64_t* ray_latent, int32_t U, int32_t D, int32_t R,
                                    float* grid, float* rays, nsky_stream_t stream) {
  NSKY_CHECK_ARG(raw && scale && ldr >= 3 && U >= 0 && D >= 0 && R >= 0 && (U * D == 0 || grid) && (R == 0 || (rays && ray_latent)),
                 "nsky_reni_output_fwd: bad argument");
  const int n = U * D + R;
  if (n == 0) return NSKY_OK;
  hipLaunchKernelGGL(reni_output_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, ldr, scale, (const long*)ray_latent, U, D,
                     R, grid, rays);
  NSKY_CHECK_LAUNCH("nsky_reni_output_fwd");
  return NSKY_OK;
}

extern "C" int nsky_reni_output_bwd(const float* raw, int32_t ldr, const float* scale, const int64_t* ray_latent, int32_t U, int32_t D, int32_t R,
                                    const float* d_grid, const float* d_rays, float* d_raw, float* d_scale, nsky_stream_t stream) {
  NSKY_CHECK_ARG(raw && scale && d_raw && ldr >= 3 && ldr <= 4 && U >= 0 && D >= 0 && R >= 0 && (R == 0 || ray_latent), "nsky_reni_output_bwd: bad argument");
  const int n = U * D + R;
  if (n == 0) return NSKY_OK;
  hipLaunchKernelGGL(reni_output_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, ldr, scale, (const long*)ray_latent, U, D,
                     R, d_grid, d_rays, d_raw, d_scale);
  NSKY_CHECK_LAUNCH("nsky_reni_output_bwd");
  return NSKY_OK;
}

extern "C" int nsky_grid_probe_points(const float* lattice, const float* gap3_host, int32_t P, uint64_t seed, uint64_t* counter, float* positions,
                                      float* directions, nsky_stream_t stream) {
  if (P == 0) return NSKY_OK;
  NSKY_CHECK_ARG(lattice && gap3_host && counter && positions && directions && P > 0, "nsky_grid_probe_points: bad argument");
  hipLaunchKernelGGL(grid_probe_points_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, (hipStream_t)stream, lattice, gap3_host[0], gap3_host[1],
                     gap3_host[2], P, seed, counter, positions, directions);
  hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  NSKY_CHECK_LAUNCH("nsky_grid_probe_points");
  return NSKY_OK;
}
