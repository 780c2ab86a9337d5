// Multiresolution hash-grid encode (tiny-cuda-nn HashGrid semantics, fp32 table) fused with the rest
// of the MLP input row (raw position, NeRF positional encoding) and, optionally, the forward-mode
// input Jacobian (three tangent rows per point).
//
// Work decomposition: a workgroup = 4 waves owns 64 consecutive points; lane = point, wave w walks
// levels w, w+4, w+8, w+12 so that a wave-instruction's 64 gathers all hit ONE level's slab of the
// table (coarse levels live in L2, fine ones in the 256 MiB Infinity Cache; the 8 corner gathers x 4
// levels per thread are independent loads in flight).  The finished row (up to 72 floats) is staged
// in LDS and written back row-major, whole 128-B lines at a time, instead of 8-byte scattered stores.
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

// two consecutive floats of a gradient row: the hash columns start at column feat0 of the row, which is odd for the SDF field
// ([x | PE6 | hash]: feat0 = 39), so the pair is only 4-byte aligned -- no float2 dereference
__device__ __forceinline__ float2 load2(const float* p) {
  float2 v;
  __builtin_memcpy(&v, p, sizeof(v));
  return v;
}


constexpr int PB = 64;         // points per workgroup
constexpr int MAXW = 72;       // widest row: 3 + 36 + 32 = 71 -> 72
constexpr int LDS_LD = MAXW + 1;
constexpr float TWO_PI = 6.283185307179586f;
constexpr float HALF_PI = 1.5707963267948966f;

struct Grid {
  const float* table;
  int n_levels, smoothstep;
  float scale[16];
  int resolution[16];
  uint32_t offset[17];
};

__device__ __forceinline__ uint32_t grid_index(uint32_t size, uint32_t res, uint32_t px, uint32_t py, uint32_t pz) {
  uint32_t stride = 1, index = 0;
  if (stride <= size) { index += px * stride; stride *= res; }
  if (stride <= size) { index += py * stride; stride *= res; }
  if (stride <= size) { index += pz * stride; stride *= res; }
  if (size < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
  return index % size;
}

// position fed to the grid + its Jacobian w.r.t. the world position (3x3, row a = d pos_a / d x_b)
__device__ __forceinline__ void grid_position(const float x[3], int mode, float pos[3], float J[3][3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) J[a][b] = (a == b) ? 1.0f : 0.0f;
  if (mode == 0) {
    pos[0] = x[0]; pos[1] = x[1]; pos[2] = x[2];
    return;
  }
  float c[3] = {x[0], x[1], x[2]};
  if (mode == 1) {  // L-infinity contraction
    float ax = fabsf(x[0]), ay = fabsf(x[1]), az = fabsf(x[2]);
    float m = fmaxf(ax, fmaxf(ay, az));
    if (!(m < 1.0f)) {
      int im = (ax >= ay && ax >= az) ? 0 : ((ay >= az) ? 1 : 2);
      float sgn = x[im] >= 0.0f ? 1.0f : -1.0f;
      float inv = 1.0f / m;
      float k = (2.0f - inv) * inv;             // f(x) = k x
      float dk = (-2.0f + 2.0f * inv) * inv * inv;  // dk/dm
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        c[a] = k * x[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? k : 0.0f) + (b == im ? x[a] * dk * sgn : 0.0f);
      }
    }
  } else {  // L2 contraction
    float m = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (!(m < 1.0f)) {
      float inv = 1.0f / m;
      float k = (2.0f - inv) * inv;
      float dk = (-2.0f + 2.0f * inv) * inv * inv;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        c[a] = k * x[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? k : 0.0f) + x[a] * dk * x[b] * inv;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    pos[a] = (c[a] + 2.0f) * 0.25f;
#pragma unroll
    for (int b = 0; b < 3; ++b) J[a][b] *= 0.25f;
  }
}

struct Cell {
  uint32_t idx[8];
  float w[3], dw[3];  // interpolation weight per axis and d w / d pos (includes the level scale)
};

__device__ __forceinline__ void locate(const Grid& g, int level, const float pos[3], Cell& c) {
  const float scale = g.scale[level];
  const uint32_t res = (uint32_t)g.resolution[level];
  const uint32_t size = g.offset[level + 1] - g.offset[level];
  uint32_t pg[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p = fmaf(scale, pos[a], 0.5f);
    float fl = floorf(p);
    pg[a] = (uint32_t)(int)fl;
    float t = p - fl;
    if (g.smoothstep) {
      c.w[a] = t * t * (3.0f - 2.0f * t);
      c.dw[a] = 6.0f * t * (1.0f - t) * scale;
    } else {
      c.w[a] = t;
      c.dw[a] = scale;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    c.idx[k] = g.offset[level] + grid_index(size, res, pg[0] + (k & 1), pg[1] + ((k >> 1) & 1), pg[2] + ((k >> 2) & 1));
}

__host__ __device__ __forceinline__ int row_width(int include_x, int pe_freqs, int n_levels) {
  return (include_x ? 3 : 0) + 6 * pe_freqs + 2 * n_levels;
}

template <bool TANGENTS>
__global__ __launch_bounds__(256) void encode_fwd_kernel(Grid g, const float* __restrict__ x, int P, int mode, int include_x,
                                                         int pe_freqs, float pe_max_exp, float* __restrict__ Y, int ldy, float* __restrict__ T) {
  constexpr int NT = TANGENTS ? 4 : 1;
  __shared__ float S[NT][PB][LDS_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * PB + lane;
  const bool valid = p < P;
  const int width = row_width(include_x, pe_freqs, g.n_levels);
  const int feat0 = (include_x ? 3 : 0) + 6 * pe_freqs;

  float xv[3] = {0.f, 0.f, 0.f};
  if (valid) { xv[0] = x[(long)p * 3]; xv[1] = x[(long)p * 3 + 1]; xv[2] = x[(long)p * 3 + 2]; }
  float pos[3], J[3][3];
  grid_position(xv, mode, pos, J);

  // zero the pad columns and (wave 0) fill x / PE
  for (int c = width + wave; c < ldy && c < MAXW; c += 4)
#pragma unroll
    for (int t = 0; t < NT; ++t) S[t][lane][c] = 0.0f;
  if (include_x && wave == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      S[0][lane][a] = xv[a];
      if (TANGENTS)
#pragma unroll
        for (int k = 0; k < 3; ++k) S[1 + k][lane][a] = (a == k) ? 1.0f : 0.0f;
    }
  }
  if (pe_freqs > 0) {
    const int pe0 = include_x ? 3 : 0;
    const int npe = 3 * pe_freqs;
    // nerfstudio NeRFEncoding: column i*F+f = sin(2 pi x_i 2^f), second half + pi/2
    for (int c = wave; c < npe; c += 4) {
      const int i = c / pe_freqs, f = c % pe_freqs;
      const float freq = exp2f(pe_freqs > 1 ? (float)f * pe_max_exp / (float)(pe_freqs - 1) : 0.0f);
      const float arg = TWO_PI * xv[i] * freq;
      const float arg2 = arg + HALF_PI;
      S[0][lane][pe0 + c] = sinf(arg);
      S[0][lane][pe0 + npe + c] = sinf(arg2);
      if (TANGENTS) {
        const float d1 = cosf(arg) * TWO_PI * freq, d2 = cosf(arg2) * TWO_PI * freq;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          S[1 + k][lane][pe0 + c] = (i == k) ? d1 : 0.0f;
          S[1 + k][lane][pe0 + npe + c] = (i == k) ? d2 : 0.0f;
        }
      }
    }
  }

  // A wave owns levels wave, wave + 4, ... (at most ML of them: n_levels <= 16).  ALL their corner rows are requested before the first
  // one is used -- 8 ML gathers of 8 bytes in flight per lane, finest (hashed: Infinity-Cache latency) levels first -- instead of one
  // level's eight at a time behind the previous level's arithmetic.
  constexpr int ML = 4;
  Cell cs[ML];
  float2 v[ML][8];
#pragma unroll
  for (int i = ML - 1; i >= 0; --i) {
    const int level = wave + 4 * i;
    if (level < g.n_levels) {
      locate(g, level, pos, cs[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[i][k] = reinterpret_cast<const float2*>(g.table)[cs[i].idx[k]];
    }
  }
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    const int level = wave + 4 * i;
    if (level >= g.n_levels) continue;
    const Cell& c = cs[i];
    float f0 = 0.f, f1 = 0.f;
    float d0[3] = {0.f, 0.f, 0.f}, d1[3] = {0.f, 0.f, 0.f};  // d feat / d pos_a
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
      const float wy = (k & 2) ? c.w[1] : 1.0f - c.w[1];
      const float wz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
      const float w = wx * wy * wz;
      f0 = fmaf(w, v[i][k].x, f0);
      f1 = fmaf(w, v[i][k].y, f1);
      if (TANGENTS) {
        const float gx = ((k & 1) ? c.dw[0] : -c.dw[0]) * wy * wz;
        const float gy = ((k & 2) ? c.dw[1] : -c.dw[1]) * wx * wz;
        const float gz = ((k & 4) ? c.dw[2] : -c.dw[2]) * wx * wy;
        d0[0] = fmaf(gx, v[i][k].x, d0[0]); d1[0] = fmaf(gx, v[i][k].y, d1[0]);
        d0[1] = fmaf(gy, v[i][k].x, d0[1]); d1[1] = fmaf(gy, v[i][k].y, d1[1]);
        d0[2] = fmaf(gz, v[i][k].x, d0[2]); d1[2] = fmaf(gz, v[i][k].y, d1[2]);
      }
    }
    S[0][lane][feat0 + 2 * level] = f0;
    S[0][lane][feat0 + 2 * level + 1] = f1;
    if (TANGENTS) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S[1 + k][lane][feat0 + 2 * level] = d0[0] * J[0][k] + d0[1] * J[1][k] + d0[2] * J[2][k];
        S[1 + k][lane][feat0 + 2 * level + 1] = d1[0] * J[0][k] + d1[1] * J[1][k] + d1[2] * J[2][k];
      }
    }
  }
  __syncthreads();
  const int wcols = min(ldy, MAXW);
  const int p0 = blockIdx.x * PB;
  const int nrows = min(PB, P - p0);
  for (int idx = threadIdx.x; idx < nrows * wcols; idx += 256) {
    const int r = idx / wcols, c = idx % wcols;
    Y[(long)(p0 + r) * ldy + c] = S[0][r][c];
    if (TANGENTS)
#pragma unroll
      for (int k = 0; k < 3; ++k) T[((long)k * P + p0 + r) * ldy + c] = S[1 + k][r][c];
  }
}

template <bool TANGENTS>
__global__ __launch_bounds__(256) void encode_bwd_kernel(Grid g, const float* __restrict__ x, int P, int mode, int include_x,
                                                         int pe_freqs, float pe_max_exp, const float* __restrict__ dY, int lddy,
                                                         const float* __restrict__ dT, float* __restrict__ dtable,
                                                         float* __restrict__ dx) {
  __shared__ float red[4][PB][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * PB + lane;
  const bool valid = p < P;
  const int feat0 = (include_x ? 3 : 0) + 6 * pe_freqs;

  float xv[3] = {0.f, 0.f, 0.f};
  if (valid) { xv[0] = x[(long)p * 3]; xv[1] = x[(long)p * 3 + 1]; xv[2] = x[(long)p * 3 + 2]; }
  float pos[3], J[3][3];
  grid_position(xv, mode, pos, J);
  float gx_acc[3] = {0.f, 0.f, 0.f};  // dL/dx through dY (first-order input gradient)

  if (dx && valid) {
    const float* row = dY + (long)p * lddy;
    if (include_x && wave == 0)
#pragma unroll
      for (int a = 0; a < 3; ++a) gx_acc[a] += row[a];
    if (pe_freqs > 0) {
      const int pe0 = include_x ? 3 : 0, npe = 3 * pe_freqs;
      for (int c = wave; c < npe; c += 4) {
        const int i = c / pe_freqs, f = c % pe_freqs;
        const float freq = exp2f(pe_freqs > 1 ? (float)f * pe_max_exp / (float)(pe_freqs - 1) : 0.0f);
        const float arg = TWO_PI * xv[i] * freq;
        const float dsum = row[pe0 + c] * cosf(arg) + row[pe0 + npe + c] * cosf(arg + HALF_PI);
        gx_acc[i] += dsum * TWO_PI * freq;
      }
    }
  }

  for (int level = wave; level < g.n_levels; level += 4) {
    if (!valid) continue;
    Cell c;
    locate(g, level, pos, c);
    const float2 gy = load2(dY + (long)p * lddy + feat0 + 2 * level);
    // gradient w.r.t. d feat / d pos_a, pulled back through J from the tangent-row gradients
    float2 gpa[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    if (TANGENTS) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float2 gt = load2(dT + ((long)k * P + p) * lddy + feat0 + 2 * level);
#pragma unroll
        for (int a = 0; a < 3; ++a) { gpa[a].x = fmaf(gt.x, J[a][k], gpa[a].x); gpa[a].y = fmaf(gt.y, J[a][k], gpa[a].y); }
      }
    }
    float2 v[8];
    if (dx)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const float2*>(g.table)[c.idx[k]];
    float dpos[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
      const float wy = (k & 2) ? c.w[1] : 1.0f - c.w[1];
      const float wz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
      const float cx = ((k & 1) ? c.dw[0] : -c.dw[0]) * wy * wz;
      const float cy = ((k & 2) ? c.dw[1] : -c.dw[1]) * wx * wz;
      const float cz = ((k & 4) ? c.dw[2] : -c.dw[2]) * wx * wy;
      float a0 = wx * wy * wz * gy.x, a1 = wx * wy * wz * gy.y;
      if (TANGENTS) {
        a0 += cx * gpa[0].x + cy * gpa[1].x + cz * gpa[2].x;
        a1 += cx * gpa[0].y + cy * gpa[1].y + cz * gpa[2].y;
      }
      if (dtable) {
        atomicAdd(dtable + 2l * c.idx[k], a0);
        atomicAdd(dtable + 2l * c.idx[k] + 1, a1);
      }
      if (dx) {
        const float s = v[k].x * gy.x + v[k].y * gy.y;
        dpos[0] = fmaf(cx, s, dpos[0]); dpos[1] = fmaf(cy, s, dpos[1]); dpos[2] = fmaf(cz, s, dpos[2]);
      }
    }
    if (dx)
#pragma unroll
      for (int b = 0; b < 3; ++b) gx_acc[b] += dpos[0] * J[0][b] + dpos[1] * J[1][b] + dpos[2] * J[2][b];
  }
  if (dx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) red[wave][lane][a] = gx_acc[a];
    __syncthreads();
    if (wave == 0 && valid)
#pragma unroll
      for (int a = 0; a < 3; ++a) dx[(long)p * 3 + a] = red[0][lane][a] + red[1][lane][a] + red[2][lane][a] + red[3][lane][a];
  }
}

// ---- table-gradient scatter --------------------------------------------------------------------
// Float atomics leave L2 as one request per touched 64-B line per wave-instruction, and the chip-wide rate
// for lines-scattered requests is ~17x below the contiguous rate.  So the scatter is laid out to (a) put
// the two features of a corner on adjacent lanes (one line, one request), and (b) keep the small dense
// levels, where thousands of points collide on a few KB, in LDS and flush them once per workgroup.
struct CornerTerm {
  uint32_t idx;
  float a;  // value to add to table[idx][f]
};

template <bool TANGENTS>
__device__ __forceinline__ CornerTerm corner_term(const Grid& g, int level, int k, int f, const float pos[3], const float J[3][3],
                                                  const float* __restrict__ dY, const float* __restrict__ dT, int P, int p,
                                                  int lddy, int feat0) {
  const float scale = g.scale[level];
  const uint32_t res = (uint32_t)g.resolution[level];
  const uint32_t size = g.offset[level + 1] - g.offset[level];
  uint32_t pg[3];
  float w[3], dw[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q = fmaf(scale, pos[a], 0.5f);
    const float fl = floorf(q);
    pg[a] = (uint32_t)(int)fl;
    const float t = q - fl;
    if (g.smoothstep) { w[a] = t * t * (3.0f - 2.0f * t); dw[a] = 6.0f * t * (1.0f - t) * scale; }
    else { w[a] = t; dw[a] = scale; }
  }
  CornerTerm ct;
  ct.idx = g.offset[level] + grid_index(size, res, pg[0] + (k & 1), pg[1] + ((k >> 1) & 1), pg[2] + ((k >> 2) & 1));
  const float wx = (k & 1) ? w[0] : 1.0f - w[0];
  const float wy = (k & 2) ? w[1] : 1.0f - w[1];
  const float wz = (k & 4) ? w[2] : 1.0f - w[2];
  const int col = feat0 + 2 * level + f;
  float a = wx * wy * wz * dY[(long)p * lddy + col];
  if (TANGENTS) {
    const float cx = ((k & 1) ? dw[0] : -dw[0]) * wy * wz;
    const float cy = ((k & 2) ? dw[1] : -dw[1]) * wx * wz;
    const float cz = ((k & 4) ? dw[2] : -dw[2]) * wx * wy;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float gt = dT[((long)t * P + p) * lddy + col];
      a = fmaf(gt, cx * J[0][t] + cy * J[1][t] + cz * J[2][t], a);
    }
  }
  ct.a = a;
  return ct;
}

// fine / hashed levels: block = 16 points x (8 corners x 2 features), blockIdx.y = level - level0
template <bool TANGENTS>
__global__ __launch_bounds__(256) void encode_bwd_scatter_kernel(Grid g, const float* __restrict__ x, int P, int mode, int feat0,
                                                                 const float* __restrict__ dY, int lddy,
                                                                 const float* __restrict__ dT, float* __restrict__ dtable,
                                                                 int level0) {
  const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (p >= P) return;
  const int k = (threadIdx.x >> 1) & 7, f = threadIdx.x & 1;
  const int level = level0 + blockIdx.y;
  const float xv[3] = {x[(long)p * 3], x[(long)p * 3 + 1], x[(long)p * 3 + 2]};
  float pos[3], J[3][3];
  grid_position(xv, mode, pos, J);
  const CornerTerm ct = corner_term<TANGENTS>(g, level, k, f, pos, J, dY, dT, P, p, lddy, feat0);
  atomicAdd(dtable + 2l * ct.idx + f, ct.a);
}

// coarse dense levels [0, n_coarse): accumulated in LDS over a long run of points, flushed with contiguous atomics
template <bool TANGENTS>
__global__ __launch_bounds__(256) void encode_bwd_coarse_kernel(Grid g, const float* __restrict__ x, int P, int mode, int feat0,
                                                                const float* __restrict__ dY, int lddy,
                                                                const float* __restrict__ dT, float* __restrict__ dtable,
                                                                int n_coarse, int points_per_block) {
  extern __shared__ float acc[];  // 2 * offset[n_coarse] floats
  const int nflt = 2 * (int)g.offset[n_coarse];
  for (int i = threadIdx.x; i < nflt; i += 256) acc[i] = 0.0f;
  __syncthreads();
  const int k = (threadIdx.x >> 1) & 7, f = threadIdx.x & 1;
  const int p_beg = blockIdx.x * points_per_block;
  const int p_end = min(P, p_beg + points_per_block);
  for (int p = p_beg + (threadIdx.x >> 4); p < p_end; p += 16) {
    const float xv[3] = {x[(long)p * 3], x[(long)p * 3 + 1], x[(long)p * 3 + 2]};
    float pos[3], J[3][3];
    grid_position(xv, mode, pos, J);
    for (int level = 0; level < n_coarse; ++level) {
      const CornerTerm ct = corner_term<TANGENTS>(g, level, k, f, pos, J, dY, dT, P, p, lddy, feat0);
      atomicAdd(acc + 2 * ct.idx + f, ct.a);  // ds_add_f32
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nflt; i += 256) {
    const float v = acc[i];
    if (v != 0.0f) atomicAdd(dtable + i, v);
  }
}

// ---- table gradient, owner-computes form --------------------------------------------------------------------------------
// The scatter above is bound by the L2 atomic units: one request per touched 64-B line and wave instruction, ~3.4e10 requests
// per second chip-wide, and a hashed level sends every corner of every point to a different line.  Here a workgroup OWNS a
// 16384-entry chunk of one level's slab (128 KB of float2 accumulators in LDS) and adds every corner that lands in it with
// ds_add_f32; the chunk is written back once, as whole lines.  Three things keep the owners cheap:
//   * a pre-pass writes, per (level, chunk), a BITMAP over the points: bit p is set when point p can have a corner in that chunk
//     (hashed level: the chunk numbers of its four (y, z) corner pairs -- x and x + 1 differ below bit 14, the chunk bits start
//     there, so the two x-corners of a pair share a chunk; dense level: the chunks its corner-index range touches).  A workgroup
//     of the pre-pass assembles the 32 x 1024-point bitmap tile of its level in LDS and writes whole words: no global atomics.
//     An owner then reads 1/32 of a bit per point instead of walking all points;
//   * the set bits (1/8 of the points on a hashed level) are queued in LDS and worked through by full waves -- positions,
//     weights and the exact corner indices are computed only for those;
//   * ds_add_f32 retires about one lane per clock and CU, so a level's adds must be spread over ~32 workgroups: levels with
//     few chunks (the dense ones) split their POINTS over several workgroups per chunk, whose write-backs are then atomic.
constexpr int OWN_CH = 16384;            // entries per chunk (a power of two: chunk number = index >> 14)
constexpr int OWN_SHIFT = 14;
constexpr int OWN_THREADS = 1024;
constexpr int OWN_QCAP = 16384 - 64;      // queue entries (2 B each; with the scan's 32 wave totals: 32 KB beside the 128 KB chunk = all 160 KB)
constexpr int OWN_WG_PER_LEVEL = 32;
constexpr int OWN_DENSE_SPLITS = 32;    // workgroups per chunk of a dense level
constexpr int OWN_MAX_CHUNKS = 32;       // per level: slabs of at most 2^19 entries
constexpr int BM_POINTS = 1024;          // points per workgroup of the bitmap pre-pass

struct OwnerArgs {
  Grid g;
  const float* x;
  const float* dY;
  const float* dT;
  float* dtable;
  uint32_t* bitmaps;       // [n_levels_owned][32 chunks][words], words = 32 * ceil(P / 1024): bit p % 32 of word p / 32
  int words;
  const float* packed;     // tangent calls: [n_levels][P][8] = dY and the three dT column pairs of a level side by side (owner_pack_kernel)
  int P, mode, feat0, lddy;
  int n_levels_owned;      // scatter levels
  int level[16];           // their level numbers
  int splits[16];          // workgroups sharing a chunk of that level (each takes 1/splits of the points; > 1: atomic write-back)
  int wg0[17];             // prefix sums of workgroups per owned level
};

__host__ __device__ __forceinline__ bool level_is_dense(uint32_t size, uint32_t res) {
  return (unsigned long long)res * res * res <= (unsigned long long)size;
}

struct F3 { float v[3]; };
constexpr uint32_t HASH_P1 = 2654435761u, HASH_P2 = 805459861u;

// Tangent calls: an owner needs dY and the three dT column pairs of its level for 1 point in 8 -- four 8-byte pieces from four rows
// 288 bytes apart and a batch apart (measured: 2.2 GB of HBM / Infinity-Cache fetches per 98 304-point call, the kernel's bound).  One
// pass packs them level-major, 32 bytes per (level, point), so that an entry is one aligned 32-byte read.
__global__ __launch_bounds__(256) void owner_pack_kernel(const float* __restrict__ dY, const float* __restrict__ dT, int P, int lddy, int feat0,
                                                         int n_levels, float* __restrict__ packed) {
  // a wave: 4 points x 16 levels (lanes = (point, level): the 16 level pairs of a row are 128 contiguous bytes)
  const int lane = threadIdx.x & 63;
  const long p = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int l = lane & 15;
  if (p >= P || l >= n_levels) return;
  const long o = p * lddy + feat0 + 2 * l;
  const float2 gy = load2(dY + o);
  const float2 g0 = load2(dT + o);
  const float2 g1 = load2(dT + (long)P * lddy + o);
  const float2 g2 = load2(dT + 2l * P * lddy + o);
  float4* dst = reinterpret_cast<float4*>(packed + ((long)l * P + p) * 8);
  dst[0] = make_float4(gy.x, gy.y, g0.x, g0.y);
  dst[1] = make_float4(g1.x, g1.y, g2.x, g2.y);
}

__global__ __launch_bounds__(BM_POINTS) void owner_bitmaps_kernel(OwnerArgs a) {
  __shared__ uint32_t bm[OWN_MAX_CHUNKS][BM_POINTS / 32];
  const int tid = threadIdx.x, li = blockIdx.y;
  const int p = blockIdx.x * BM_POINTS + tid;
  reinterpret_cast<uint32_t*>(bm)[tid] = 0u;  // 32 x 32 words = 1024
  __syncthreads();
  const int level = a.level[li];
  const uint32_t res = (uint32_t)a.g.resolution[level];
  const uint32_t size = a.g.offset[level + 1] - a.g.offset[level];
  const int nch = (int)((size + OWN_CH - 1) / OWN_CH);
  if (p < a.P) {
    const F3 xv = reinterpret_cast<const F3*>(a.x)[p];
    float pos[3], J[3][3];
    grid_position(xv.v, a.mode, pos, J);
    const float scale = a.g.scale[level];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pg[d] = (uint32_t)(int)floorf(fmaf(scale, pos[d], 0.5f));
    uint32_t chunks = 0;  // bit c: chunk c may receive a corner of this point
    if (level_is_dense(size, res)) {
      // inside the grid the eight corner indices lie in [corner 0, corner 0 + 1 + res + res^2]; a point whose cell is not inside
      // (index arithmetic wraps, grid_index's modulo bites; a negative cell is a huge unsigned number) is looked at by every owner
      const bool inside = pg[0] < res - 1u && pg[1] < res - 1u && pg[2] < res - 1u;
      if (inside) {
        const uint32_t i0 = pg[0] + pg[1] * res + pg[2] * res * res, i1 = i0 + 1u + res + res * res;
        for (uint32_t c = i0 >> OWN_SHIFT; c <= min(i1 >> OWN_SHIFT, (uint32_t)nch - 1u); ++c) chunks |= 1u << c;
      } else {
        chunks = nch >= 32 ? 0xFFFFFFFFu : (1u << nch) - 1u;
      }
    } else {
      const uint32_t smask = size - 1, Y0 = pg[1] * HASH_P1, Z0 = pg[2] * HASH_P2;
      if ((pg[0] & (uint32_t)(OWN_CH - 1)) == (uint32_t)(OWN_CH - 1)) {
        chunks = nch >= 32 ? 0xFFFFFFFFu : (1u << nch) - 1u;  // x + 1 carries into the chunk bits (cell -1 is the common case): rare
      } else {
        const uint32_t xhi = pg[0] & ~(uint32_t)(OWN_CH - 1);
        const uint32_t yz[4] = {Y0 ^ Z0, (Y0 + HASH_P1) ^ Z0, Y0 ^ (Z0 + HASH_P2), (Y0 + HASH_P1) ^ (Z0 + HASH_P2)};
#pragma unroll
        for (int bc = 0; bc < 4; ++bc) chunks |= 1u << (((yz[bc] ^ xhi) & smask) >> OWN_SHIFT);
      }
    }
    while (chunks) {
      const int c = __ffs(chunks) - 1;
      chunks &= chunks - 1;
      atomicOr(&bm[c][tid >> 5], 1u << (tid & 31));
    }
  }
  __syncthreads();
  const int c = tid >> 5, w = tid & 31;
  if (c < nch) a.bitmaps[((long)li * OWN_MAX_CHUNKS + c) * a.words + blockIdx.x * (BM_POINTS / 32) + w] = bm[c][w];
}

// one queued point of an owner: exact corner indices, interpolation weights, ds_add_f32 of the corners inside the chunk
template <bool TANGENTS>
__device__ __forceinline__ void owner_point(const OwnerArgs& a, float* acc, const float (&xv)[3], const float2 gy, const float2 (&gt)[3],
                                            float scale, uint32_t res, uint32_t size, bool dense, bool smooth, uint32_t chunk, uint32_t c_beg) {
  const uint32_t smask = size - 1;  // hashed: size is a power of two (checked on the host), grid_index's modulo is this mask
  float pos[3], J[3][3];
  grid_position(xv, a.mode, pos, J);
  uint32_t pg[3];
  float w[3], dw[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float q = fmaf(scale, pos[d], 0.5f);
    const float fl = floorf(q);
    pg[d] = (uint32_t)(int)fl;
    const float td = q - fl;
    w[d] = smooth ? td * td * (3.0f - 2.0f * td) : td;
    dw[d] = smooth ? 6.0f * td * (1.0f - td) * scale : scale;
  }
  float2 gpa[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};  // gradient w.r.t. d feat / d pos_a, pulled back through J
  if (TANGENTS)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) { gpa[d].x = fmaf(gt[k].x, J[d][k], gpa[d].x); gpa[d].y = fmaf(gt[k].y, J[d][k], gpa[d].y); }
  // y / z terms of the index in wrapping 32-bit arithmetic, as grid_index forms them: hashed y P1 ^ z P2, dense y res + z res^2
  const uint32_t YM = dense ? res : HASH_P1, ZM = dense ? res * res : HASH_P2;
  const uint32_t Y0 = pg[1] * YM, Z0 = pg[2] * ZM;
  if (!dense) {
    // Hashed level: a point's eight corners are four (y, z) pairs of two x-neighbours, and typically ONE pair lies in this chunk.
    // An LDS atomic costs its ~64 clocks whatever its lane mask, so the wave does not walk all eight corners under masks: every lane
    // lists its pairs that touch the chunk and the wave walks list positions (one or two, rarely more: the ballot decides).
    uint32_t pairs = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t yz = (Y0 + ((q & 1) ? YM : 0u)) ^ (Z0 + ((q & 2) ? ZM : 0u));
      const uint32_t i0 = (pg[0] ^ yz) & smask, i1 = ((pg[0] + 1u) ^ yz) & smask;
      if ((i0 >> OWN_SHIFT) == chunk || (i1 >> OWN_SHIFT) == chunk) pairs |= 1u << q;
    }
    while (__ballot(pairs != 0)) {
      if (pairs) {
        const int q = __ffs(pairs) - 1;
        pairs &= pairs - 1;
        const uint32_t yz = (Y0 + ((q & 1) ? YM : 0u)) ^ (Z0 + ((q & 2) ? ZM : 0u));
        const float wy = (q & 1) ? w[1] : 1.0f - w[1];
        const float wz = (q & 2) ? w[2] : 1.0f - w[2];
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          const uint32_t idx = ((pg[0] + (uint32_t)xb) ^ yz) & smask;
          if ((idx >> OWN_SHIFT) != chunk) continue;
          const float wx = xb ? w[0] : 1.0f - w[0];
          float a0 = wx * wy * wz * gy.x, a1 = wx * wy * wz * gy.y;
          if (TANGENTS) {
            const float cx = (xb ? dw[0] : -dw[0]) * wy * wz;
            const float cy = ((q & 1) ? dw[1] : -dw[1]) * wx * wz;
            const float cz = ((q & 2) ? dw[2] : -dw[2]) * wx * wy;
            a0 += cx * gpa[0].x + cy * gpa[1].x + cz * gpa[2].x;
            a1 += cx * gpa[0].y + cy * gpa[1].y + cz * gpa[2].y;
          }
          const int loc = (int)(idx - c_beg);
          atomicAdd(acc + 2 * loc, a0);      // ds_add_f32
          atomicAdd(acc + 2 * loc + 1, a1);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t xk = pg[0] + (k & 1), yk = Y0 + ((k & 2) ? YM : 0u), zk = Z0 + ((k & 4) ? ZM : 0u);
    uint32_t idx = xk + yk + zk;
    if (idx >= size) idx %= size;
    if ((idx >> OWN_SHIFT) != chunk) continue;
    const float wx = (k & 1) ? w[0] : 1.0f - w[0];
    const float wy = (k & 2) ? w[1] : 1.0f - w[1];
    const float wz = (k & 4) ? w[2] : 1.0f - w[2];
    float a0 = wx * wy * wz * gy.x, a1 = wx * wy * wz * gy.y;
    if (TANGENTS) {
      const float cx = ((k & 1) ? dw[0] : -dw[0]) * wy * wz;
      const float cy = ((k & 2) ? dw[1] : -dw[1]) * wx * wz;
      const float cz = ((k & 4) ? dw[2] : -dw[2]) * wx * wy;
      a0 += cx * gpa[0].x + cy * gpa[1].x + cz * gpa[2].x;
      a1 += cx * gpa[0].y + cy * gpa[1].y + cz * gpa[2].y;
    }
    const int loc = (int)(idx - c_beg);
    atomicAdd(acc + 2 * loc, a0);      // ds_add_f32
    atomicAdd(acc + 2 * loc + 1, a1);
  }
}

// An owner works through its share of the points in PHASES: up to OWN_PHASE_WORDS bitmap words (65 536 points: the span a 16-bit
// queue entry can address) are read two per thread, a block-wide prefix sum of their popcounts places every set bit in the queue
// without atomics, and the queue is then worked off with OWN_MLP points per thread in flight.  (With one 8 192-point trip per
// barrier pair -- the first form -- a hashed level's owner had ONE point per thread per trip: 33 serial memory round trips.)
// The phase shrinks to 1024 / 512 / 256 words when the queue could not hold its bits (dense levels: most bits are set).
template <bool TANGENTS>
__global__ __launch_bounds__(OWN_THREADS) void encode_bwd_owner_kernel(OwnerArgs a) {
  extern __shared__ float acc[];  // 2 * OWN_CH accumulators, then the queue (point number - phase start) and the scan's wave totals
  uint16_t* queue = reinterpret_cast<uint16_t*>(acc + 2 * OWN_CH);
  int* wtot = reinterpret_cast<int*>(queue + OWN_QCAP);  // [2][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int li = 0;
  while (li + 1 < a.n_levels_owned && (int)blockIdx.x >= a.wg0[li + 1]) ++li;
  const int level = a.level[li], splits = a.splits[li];
  const int item = (int)blockIdx.x - a.wg0[li];
  const uint32_t chunk = (uint32_t)(item / splits);
  const int ps = item % splits;
  const Grid& g = a.g;
  const float scale = g.scale[level];
  const uint32_t res = (uint32_t)g.resolution[level];
  const uint32_t size = g.offset[level + 1] - g.offset[level];
  const bool dense = level_is_dense(size, res);
  const uint32_t c_beg = chunk * OWN_CH;
  const int n_own = (int)min((uint32_t)OWN_CH, size - c_beg);
  // this workgroup's share of the points: whole bitmap words
  const int words_per = (a.words + splits - 1) / splits;
  const int w_beg = ps * words_per, w_end = min(a.words, w_beg + words_per);
  const uint32_t* bits = a.bitmaps + ((long)li * OWN_MAX_CHUNKS + chunk) * a.words;
  if (splits > 1) {  // (a dense level's chunk none of this share's points touches: nothing to add)
    uint32_t any = 0;
    for (int w = w_beg + tid; w < w_end; w += OWN_THREADS) any |= bits[w];
    if (tid == 0) wtot[0] = 0;  // (no __syncthreads_or: its static LDS word would not fit beside 160 KB of dynamic LDS)
    __syncthreads();
    if (any) wtot[0] = 1;
    __syncthreads();
    if (!wtot[0]) return;
  }
  for (int i = tid; i < OWN_CH / 2; i += OWN_THREADS) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int col = a.feat0 + 2 * level;
  const bool smooth = g.smoothstep != 0;
  const F3* xs = reinterpret_cast<const F3*>(a.x);
  constexpr int MLP = TANGENTS ? 2 : 4;
  for (int wb = w_beg; wb < w_end;) {
    uint32_t word0 = wb + tid < w_end ? bits[wb + tid] : 0u;
    uint32_t word1 = wb + OWN_THREADS + tid < w_end ? bits[wb + OWN_THREADS + tid] : 0u;
    const int c0 = __popc(word0), c1 = __popc(word1);
    int i0 = c0, i1 = c1;  // inclusive scans over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v0 = __shfl_up(i0, off, 64), v1 = __shfl_up(i1, off, 64);
      if (lane >= off) { i0 += v0; i1 += v1; }
    }
    __syncthreads();  // (the previous phase's queue and totals are no longer read; the first pass: acc is zero)
    if (lane == 63) { wtot[wave] = i0; wtot[16 + wave] = i1; }
    __syncthreads();
    int base0 = 0, base1 = 0, half0 = 0, quarter0 = 0, total1 = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t0 = wtot[w], t1 = wtot[16 + w];
      if (w < wave) { base0 += t0; base1 += t1; }
      if (w < 8) quarter0 += t0;   // words [0, 512)
      half0 += t0;                 // words [0, 1024)
      total1 += t1;
    }
    int nw, nq;  // words taken by this phase, their set bits (uniform)
    if (half0 + total1 <= OWN_QCAP) { nw = 2 * OWN_THREADS; nq = half0 + total1; }
    else if (half0 <= OWN_QCAP) { nw = OWN_THREADS; nq = half0; }
    else if (quarter0 <= OWN_QCAP) { nw = 512; nq = quarter0; }
    else { nw = 256; nq = wtot[0] + wtot[1] + wtot[2] + wtot[3]; }  // <= 8192 bits
    if (tid < nw) {
      int at = base0 + i0 - c0;
      while (word0) {
        const int b = __ffs(word0) - 1;
        word0 &= word0 - 1;
        queue[at++] = (uint16_t)(tid * 32 + b);
      }
    }
    if (OWN_THREADS + tid < nw) {
      int at = half0 + base1 + i1 - c1;
      while (word1) {
        const int b = __ffs(word1) - 1;
        word1 &= word1 - 1;
        queue[at++] = (uint16_t)((OWN_THREADS + tid) * 32 + b);
      }
    }
    __syncthreads();
    // ---- the queued points, by full waves, MLP of them per thread in flight
    const int p0 = wb * 32;
    for (int e0 = tid; e0 < nq; e0 += MLP * OWN_THREADS) {
      F3 xv[MLP];
      float2 gy[MLP], gt[MLP][3];
#pragma unroll
      for (int j = 0; j < MLP; ++j) {
        const int e = e0 + j * OWN_THREADS;
        const int p = p0 + (int)queue[e < nq ? e : e0];
        xv[j] = xs[p];
        if (TANGENTS) {
          const float4* rec = reinterpret_cast<const float4*>(a.packed + ((long)level * a.P + p) * 8);
          const float4 r0 = rec[0], r1 = rec[1];
          gy[j] = make_float2(r0.x, r0.y);
          gt[j][0] = make_float2(r0.z, r0.w); gt[j][1] = make_float2(r1.x, r1.y); gt[j][2] = make_float2(r1.z, r1.w);
        } else {
          gy[j] = load2(a.dY + (long)p * a.lddy + col);
#pragma unroll
          for (int k = 0; k < 3; ++k) gt[j][k] = make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int j = 0; j < MLP; ++j)
        if (e0 + j * OWN_THREADS < nq) owner_point<TANGENTS>(a, acc, xv[j].v, gy[j], gt[j], scale, res, size, dense, smooth, chunk, c_beg);
    }
    wb += nw;
  }
  __syncthreads();
  float* out = a.dtable + 2l * (g.offset[level] + c_beg);
  if (splits == 1 && (n_own & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    // this workgroup is the only writer of these entries in the launch: read-modify-write of whole lines, four 16-byte pieces per
    // thread in flight (an `if (v != 0) out[i] += v` loop is one memory round trip per iteration)
    const int n4 = n_own >> 1;
    float4* o4 = reinterpret_cast<float4*>(out);
    const float4* a4 = reinterpret_cast<const float4*>(acc);
    for (int i0 = tid; i0 < n4; i0 += 4 * OWN_THREADS) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i0 + j * OWN_THREADS < n4 ? o4[i0 + j * OWN_THREADS] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * OWN_THREADS;
        if (i < n4) {
          const float4 d = a4[i];
          o4[i] = make_float4(v[j].x + d.x, v[j].y + d.y, v[j].z + d.z, v[j].w + d.w);
        }
      }
    }
  } else if (splits == 1) {
    for (int i = tid; i < 2 * n_own; i += OWN_THREADS) {
      const float v = acc[i];
      if (v != 0.0f) out[i] += v;
    }
  } else {
    for (int i = tid; i < 2 * n_own; i += OWN_THREADS) {
      const float v = acc[i];
      if (v != 0.0f) atomicAdd(out + i, v);
    }
  }
}

__global__ void hash_indices_kernel(Grid g, const float* __restrict__ x, int P, int mode, uint32_t* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)P * g.n_levels) return;
  const int p = (int)(t / g.n_levels), level = (int)(t % g.n_levels);
  float xv[3] = {x[(long)p * 3], x[(long)p * 3 + 1], x[(long)p * 3 + 2]};
  float pos[3], J[3][3];
  grid_position(xv, mode, pos, J);
  Cell c;
  locate(g, level, pos, c);
#pragma unroll
  for (int k = 0; k < 8; ++k) out[t * 8 + k] = c.idx[k];
}

int make_grid(const nsky_hashgrid_desc* d, Grid& g, const char* who) {
  NSKY_CHECK_ARG(d && d->table, "%s: null grid", who);
  NSKY_CHECK_ARG(d->n_levels >= 1 && d->n_levels <= 16, "%s: n_levels=%d out of range [1,16]", who, d->n_levels);
  g.table = d->table; g.n_levels = d->n_levels; g.smoothstep = d->smoothstep;
  for (int i = 0; i < 16; ++i) { g.scale[i] = d->scale[i]; g.resolution[i] = d->resolution[i]; }
  for (int i = 0; i < 17; ++i) g.offset[i] = d->offset[i];
  for (int i = 0; i < d->n_levels; ++i)
    NSKY_CHECK_ARG(d->offset[i + 1] > d->offset[i] && d->resolution[i] > 0, "%s: bad level %d geometry", who, i);
  return NSKY_OK;
}

}  // namespace

extern "C" int nsky_encode_fwd(const nsky_hashgrid_desc* d, const float* x, int32_t P, int32_t mode, int32_t include_x,
                               int32_t pe_freqs, float pe_max_exp, float* Y, int32_t ldy, float* T, nsky_stream_t stream) {
  Grid g;
  if (int rc = make_grid(d, g, "nsky_encode_fwd")) return rc;
  if (P == 0) return NSKY_OK;
  NSKY_CHECK_ARG(x && Y && P > 0, "nsky_encode_fwd: null x/Y or negative P");
  NSKY_CHECK_ARG(mode >= 0 && mode <= 2 && pe_freqs >= 0 && pe_freqs <= 6, "nsky_encode_fwd: bad mode/pe_freqs");
  const int width = row_width(include_x, pe_freqs, g.n_levels);
  NSKY_CHECK_ARG(width <= MAXW && ldy >= width && ldy <= MAXW, "nsky_encode_fwd: row width %d / ldy %d unsupported (max %d)", width, ldy, MAXW);
  dim3 grid(ceil_div(P, PB));
  if (T)
    hipLaunchKernelGGL(encode_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g, x, P, mode, include_x, pe_freqs, pe_max_exp, Y, ldy, T);
  else
    hipLaunchKernelGGL(encode_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g, x, P, mode, include_x, pe_freqs, pe_max_exp, Y, ldy, T);
  NSKY_CHECK_LAUNCH("nsky_encode_fwd");
  return NSKY_OK;
}

static bool owner_geometry_ok(const Grid& g) {  // every level: at most 32 chunks; hashed slabs of 2^T entries (tcnn's geometry)
  for (int l = 0; l < g.n_levels; ++l) {
    const uint32_t sz = g.offset[l + 1] - g.offset[l];
    if (sz > (uint32_t)OWN_MAX_CHUNKS * OWN_CH) return false;
    if (!level_is_dense(sz, (uint32_t)g.resolution[l]) && (sz & (sz - 1)) != 0) return false;
  }
  return true;
}

static int64_t owner_bitmap_bytes(const Grid& g, int P) {
  const int64_t b = (int64_t)g.n_levels * OWN_MAX_CHUNKS * ((int64_t)ceil_div(P, BM_POINTS) * (BM_POINTS / 32)) * (int64_t)sizeof(uint32_t);
  return (b + 255) / 256 * 256;
}

extern "C" int64_t nsky_encode_bwd_workspace_bytes(const nsky_hashgrid_desc* d, int32_t P, int32_t tangents) {
  Grid g;
  if (make_grid(d, g, "nsky_encode_bwd_workspace_bytes") != NSKY_OK || P < NSKY_ENCODE_BWD_OWNER_MIN_POINTS || !owner_geometry_ok(g)) return 0;
  return owner_bitmap_bytes(g, P) + (tangents ? (int64_t)g.n_levels * P * 8 * (int64_t)sizeof(float) : 0);  // bitmaps | packed gradients
}

extern "C" int nsky_encode_bwd(const nsky_hashgrid_desc* d, const float* x, int32_t P, int32_t mode, int32_t include_x,
                               int32_t pe_freqs, float pe_max_exp, const float* dY, int32_t lddy, const float* dT, float* dtable,
                               float* dx, void* workspace, nsky_stream_t stream) {
  Grid g;
  if (int rc = make_grid(d, g, "nsky_encode_bwd")) return rc;
  if (P == 0) return NSKY_OK;
  NSKY_CHECK_ARG(x && dY && (dtable || dx) && P > 0, "nsky_encode_bwd: null argument");
  NSKY_CHECK_ARG(mode >= 0 && mode <= 2 && pe_freqs >= 0 && pe_freqs <= 6, "nsky_encode_bwd: bad mode/pe_freqs");
  hipStream_t s = (hipStream_t)stream;
  const int feat0 = (include_x ? 3 : 0) + 6 * pe_freqs;
  if (!dtable) {
    // a frozen table: the input gradient alone
  } else if (workspace != nullptr && P >= NSKY_ENCODE_BWD_OWNER_MIN_POINTS && owner_geometry_ok(g)) {
    // Many points: chunk owners for EVERY level (LDS accumulation, above).
    OwnerArgs oa;
    oa.g = g; oa.x = x; oa.dY = dY; oa.dT = dT; oa.dtable = dtable; oa.P = P; oa.mode = mode; oa.feat0 = feat0; oa.lddy = lddy;
    oa.bitmaps = reinterpret_cast<uint32_t*>(workspace);
    oa.packed = nullptr;
    if (dT) {
      float* packed = reinterpret_cast<float*>(static_cast<char*>(workspace) + owner_bitmap_bytes(g, P));
      hipLaunchKernelGGL(owner_pack_kernel, dim3(ceil_div(P, 16)), dim3(256), 0, s, dY, dT, P, lddy, feat0, g.n_levels, packed);
      oa.packed = packed;
    }
    oa.words = ceil_div(P, BM_POINTS) * (BM_POINTS / 32);
    oa.n_levels_owned = 0;
    int wgs = 0;
    for (int l = g.n_levels - 1; l >= 0; --l) {  // finest first: the hashed levels' owners are the long tasks, the dense levels' short ones fill in behind
      const int nch = ceil_div((long)(g.offset[l + 1] - g.offset[l]), OWN_CH);
      // hashed: the hash spreads a level's adds evenly over its 32 chunks; dense: the points' spatial distribution decides, and a
      // scene's points crowd a few slabs (a chunk of a dense level is a few z-planes: measured on a train step's termination points,
      // four workgroups of a 14-chunk level had all the work) -- every chunk's points are split over OWN_DENSE_SPLITS workgroups
      // (64 for a one-chunk level); a workgroup whose share of the bitmap is empty leaves before it touches its accumulators
      const bool dense_l = level_is_dense(g.offset[l + 1] - g.offset[l], (uint32_t)g.resolution[l]);
      int sp = dense_l ? (2 * OWN_WG_PER_LEVEL / nch > OWN_DENSE_SPLITS ? 2 * OWN_WG_PER_LEVEL / nch : OWN_DENSE_SPLITS) : OWN_WG_PER_LEVEL / nch;
      // the coarsest hashed levels: a crowded cell sends all its points to the same four chunks (measured on a train step's termination
      // points: the first hashed level's busiest owner alone took 0.15 ms) -- their chunks' points are shared by up to four workgroups
      if (!dense_l) {
        const int crowd = 512 / g.resolution[l];
        sp *= crowd > 4 ? 4 : (crowd < 1 ? 1 : crowd);
      }
      if (sp < 1) sp = 1;
      const int max_sp = oa.words / 32 > 1 ? oa.words / 32 : 1;  // at least 32 bitmap words (1024 points) per workgroup
      if (sp > max_sp) sp = max_sp;
      oa.level[oa.n_levels_owned] = l;
      oa.splits[oa.n_levels_owned] = sp;
      oa.wg0[oa.n_levels_owned++] = wgs;
      wgs += nch * sp;
    }
    oa.wg0[oa.n_levels_owned] = wgs;
    static bool own_attr = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_bwd_owner_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_bwd_owner_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      return true;
    }();
    (void)own_attr;
    hipLaunchKernelGGL(owner_bitmaps_kernel, dim3(ceil_div(P, BM_POINTS), oa.n_levels_owned), dim3(BM_POINTS), 0, s, oa);
    const size_t smem = 2 * OWN_CH * sizeof(float) + 16384 * sizeof(uint16_t);
    if (dT)
      hipLaunchKernelGGL(encode_bwd_owner_kernel<true>, dim3(wgs), dim3(OWN_THREADS), smem, s, oa);
    else
      hipLaunchKernelGGL(encode_bwd_owner_kernel<false>, dim3(wgs), dim3(OWN_THREADS), smem, s, oa);
    NSKY_CHECK_LAUNCH("nsky_encode_bwd(owner)");
  } else {
    // Few points (an owner's fixed costs -- zero and write back 128 KB per chunk -- only pay once the atomics they replace outnumber
    // them): coarse levels whose slabs fit together in LDS (<= 144 KiB) are privatised per workgroup, the rest scatter directly
    int n_coarse = 0;
    while (n_coarse < g.n_levels && (size_t)g.offset[n_coarse + 1] * 2 * sizeof(float) <= 144 * 1024) ++n_coarse;
    if (n_coarse > 0) {
      const int blocks = 256;
      const int ppb = ((P + blocks - 1) / blocks + 15) / 16 * 16;
      const size_t smem = (size_t)g.offset[n_coarse] * 2 * sizeof(float);
      dim3 grid(ceil_div(P, ppb));
      // raise the dynamic-LDS ceiling once per process (not a stream operation; kept out of captured regions)
      static bool attr_set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_bwd_coarse_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_bwd_coarse_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)attr_set;
      if (dT) {
        hipLaunchKernelGGL(encode_bwd_coarse_kernel<true>, grid, dim3(256), smem, s, g, x, P, mode, feat0, dY, lddy, dT, dtable, n_coarse, ppb);
      } else {
        hipLaunchKernelGGL(encode_bwd_coarse_kernel<false>, grid, dim3(256), smem, s, g, x, P, mode, feat0, dY, lddy, dT, dtable, n_coarse, ppb);
      }
      NSKY_CHECK_LAUNCH("nsky_encode_bwd(coarse)");
    }
    if (n_coarse < g.n_levels) {
      dim3 grid(ceil_div(P, 16), g.n_levels - n_coarse);
      if (dT)
        hipLaunchKernelGGL(encode_bwd_scatter_kernel<true>, grid, dim3(256), 0, s, g, x, P, mode, feat0, dY, lddy, dT, dtable, n_coarse);
      else
        hipLaunchKernelGGL(encode_bwd_scatter_kernel<false>, grid, dim3(256), 0, s, g, x, P, mode, feat0, dY, lddy, dT, dtable, n_coarse);
      NSKY_CHECK_LAUNCH("nsky_encode_bwd(scatter)");
    }
  }
  if (dx) {  // first-order input gradient (lane = point kernel, no table traffic)
    dim3 grid(ceil_div(P, PB));
    hipLaunchKernelGGL(encode_bwd_kernel<false>, grid, dim3(256), 0, s, g, x, P, mode, include_x, pe_freqs, pe_max_exp, dY, lddy,
                       (const float*)nullptr, (float*)nullptr, dx);
    NSKY_CHECK_LAUNCH("nsky_encode_bwd(dx)");
  }
  return NSKY_OK;
}

extern "C" int nsky_hash_indices(const nsky_hashgrid_desc* d, const float* x, int32_t P, int32_t mode, uint32_t* idx,
                                 nsky_stream_t stream) {
  Grid g;
  if (int rc = make_grid(d, g, "nsky_hash_indices")) return rc;
  if (P == 0) return NSKY_OK;
  NSKY_CHECK_ARG(x && idx && P > 0, "nsky_hash_indices: null argument");
  const long n = (long)P * g.n_levels;
  hipLaunchKernelGGL(hash_indices_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, g, x, P, mode, idx);
  NSKY_CHECK_LAUNCH("nsky_hash_indices");
  return NSKY_OK;
}
