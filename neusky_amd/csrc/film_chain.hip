// FiLM-SIREN chain (neusky/utils/siren.py:108-208; call sites neusky/fields/directional_distance_field.py:233-299 and the
// RENI-shaped illumination decode, neusky/models/neusky_model.py:488-506) as ONE kernel per row tile, forward and backward.
//
// Transposed formulation: a wave owns 32 batch rows and carries them through the whole network.  A layer is
//   X_out[feature, row] = W[feature, k] X_in[k, row]:   weights = MFMA A operand, activations = MFMA B operand,
// so the 32x32 fp32 accumulator of v_mfma_f32_32x32x16_f16 (feature on the register index, batch row on the lane) IS, after
// the fp16 hi/lo split, the B operand of the next layer's product (the product sums over the accumulator's ROW index: no lane
// movement, no LDS; cdna_hip_programming.md section 3).  The k order this imposes (element j of lane half h of k-step s is
// feature 16 s + 8 (j >> 2) + 4 h + (j & 3)) is baked into the packed weight stream.  The mapping network's hidden state
// never leaves the registers; the 2 n_film H wide frequency / phase matrix is never formed: F, phase and z of one 32-feature
// tile are three accumulators that meet in the tile's epilogue (arg = (15 F + 30) z + phase).
//
// Arithmetic: fp32-grade products from three fp16 MFMAs into ONE fp32 accumulator, a b ~ ah bh + ah bl + al bh with
// a = ah + al exactly to 2^-22: both operands are pre-scaled by powers of two (weights per 32-row tile, activations per batch
// row, which sits on the lane, so the scale is a per-lane scalar) so that hi AND the unscaled residual stay inside fp16's
// normal range for every element within 2^-16 of its row's maximum (smaller elements keep an absolute error of 2^-39 of the
// maximum).  The scales are undone in the epilogue (exact: powers of two).
//
// Weights: packed once per optimisation step (film_pack_kernel) into the exact byte order the kernel consumes, and streamed
// by every workgroup through a 128 KB LDS ring with global_load_lds_dwordx4 (lane-linear 1 KB pieces): slab = one k-step of
// one 32-feature tile = 1 KB hi plane + 1 KB lo plane, [lane][8 fp16] so the A fragment is one conflict-free ds_read_b128;
// group = 8 slabs = 16 KB = the DMA / hand-shake unit (one counted s_waitcnt + one s_barrier per group, 7 groups in flight).
#include "chain.h"

namespace {

// fragment buffers of the four-wave forward kernel's products (chain.h product<.., NB>): it has the registers for a fourth pair (LDS reads
// three k-steps ahead instead of two), measured same-box at 2.515-2.540 ms against 2.503-2.532 with three: the kernel does not wait on LDS
constexpr int FWD_FRAG_BUFFERS = 3;


// ---------------------------------------------------------------------------------------------------------------------
// stream layout.  Forward order (tile = 32 output features x K):
//   mapping layer 0: NT tiles (K = cond_dim) | mapping layers 1..: NT tiles each (K = H)
//   FiLM layer i, feature tile t: F tile (rows i H + 32 t of the mapping head), phase tile (rows (n_film + i) H + 32 t),
//                                 z tile (FiLM weight rows 32 t; K = x_dim for i = 0, else H)
//   head: one tile (rows 0..out_dim-1, zero padded to 32)
// Backward order (direction 1): for FiLM layer i = n_film-1 .. 0: for t: F tile, phase tile (as above; recomputed, never
//   stored) ; then for input-feature tile u: transposed FiLM weight tile (rows = input features 32 u.., k = output features)
//   (skipped for i = 0); then the mapping head transposed, k-group outer: for kg (128 of the 2 n_film H head rows): for u:
//   tile (rows = hidden features 32 u.., k = head rows 128 kg ..); then mapping layers n_map-1 .. 1 transposed (for u: tile,
//   K = H); then mapping layer 0 transposed: tiles over ceil(cond_dim / 32) input-feature tiles, K = H.
// Every tile occupies a whole number of groups.
struct Layout {
  int NT, Gc, Gx, Gh;
  long base_map, base_film0, base_film, base_head, total_groups;
  int n_tiles;
};

__host__ __device__ inline Layout fwd_layout(const nsky_film_net& n) {
  Layout L;
  L.NT = n.hidden / 32;
  L.Gc = groups_of(n.cond_dim);
  L.Gx = groups_of(n.x_dim);
  L.Gh = groups_of(n.hidden);
  L.base_map = (long)L.NT * L.Gc;
  L.base_film0 = L.base_map + (long)(n.n_map - 1) * L.NT * L.Gh;
  L.base_film = L.base_film0 + (long)L.NT * (2 * L.Gh + L.Gx);
  L.base_head = L.base_film + (long)(n.n_film - 1) * L.NT * 3 * L.Gh;
  L.total_groups = L.base_head + L.Gh;
  L.n_tiles = n.n_map * L.NT + n.n_film * L.NT * 3 + 1;
  return L;
}


__device__ inline TileDesc fwd_tile(const nsky_film_net& n, const Layout& L, int idx) {
  TileDesc d;
  d.transposed = 0;
  d.k0 = 0;
  const int NT = L.NT, H = n.hidden;
  if (idx < NT) {
    d.W = n.map_w[0]; d.ld = n.map_ld[0]; d.row0 = 32 * idx; d.nrows = 32; d.K = n.cond_dim; d.group = (long)idx * L.Gc;
    return d;
  }
  idx -= NT;
  if (idx < (n.n_map - 1) * NT) {
    const int l = 1 + idx / NT, t = idx % NT;
    d.W = n.map_w[l]; d.ld = n.map_ld[l]; d.row0 = 32 * t; d.nrows = 32; d.K = H; d.group = L.base_map + (long)idx * L.Gh;
    return d;
  }
  idx -= (n.n_map - 1) * NT;
  if (idx < n.n_film * NT * 3) {
    const int i = idx / (3 * NT), t = (idx / 3) % NT, which = idx % 3;
    const long tb = i == 0 ? L.base_film0 + (long)t * (2 * L.Gh + L.Gx) : L.base_film + ((long)(i - 1) * NT + t) * 3 * L.Gh;
    d.nrows = 32;
    if (which < 2) {
      d.W = n.mo_w; d.ld = n.mo_ld; d.row0 = (which == 0 ? i : n.n_film + i) * H + 32 * t; d.K = H; d.group = tb + which * L.Gh;
    } else {
      d.W = n.film_w[i]; d.ld = n.film_ld[i]; d.row0 = 32 * t; d.K = i == 0 ? n.x_dim : H; d.group = tb + 2 * L.Gh;
    }
    return d;
  }
  d.W = n.out_w; d.ld = n.out_ld; d.row0 = 0; d.nrows = n.out_dim; d.K = H; d.group = L.base_head;
  return d;
}

// direction 1 (FiLM backward): for i = n_film-1 .. 0: for t: F tile, phase tile; if i > 0: for u: transposed FiLM weight tile
//   (rows = input features 32 u .., k = output features).
// direction 2 (mapping backward): mapping head transposed, k-block outer: for kb: for u: tile (rows = hidden features 32 u ..,
//   k = head rows 64 kb .. 64 kb + 63; four slabs: two tiles per group); layers n_map-1 .. 1 transposed: for u; layer 0 transposed:
//   ceil(cond_dim / 32) tiles.
__host__ __device__ inline void bwd_film_layout(const nsky_film_net& n, long& total_groups, int& n_tiles) {
  const int NT = n.hidden / 32, Gh = groups_of(n.hidden);
  n_tiles = n.n_film * NT * 2 + (n.n_film - 1) * NT + 1;  // last: first FiLM layer transposed (rows = x features) -> d_x
  total_groups = (long)n_tiles * Gh;
}
__host__ __device__ inline void bwd_map_layout(const nsky_film_net& n, long& total_groups, int& n_tiles) {
  const int NT = n.hidden / 32, Gh = groups_of(n.hidden);
  const int nkb = 2 * n.n_film * n.hidden / 64, ct = (n.cond_dim + 31) / 32;
  n_tiles = nkb * NT + (n.n_map - 1) * NT + ct;
  total_groups = (long)nkb * (NT / 2) + (long)((n.n_map - 1) * NT + ct) * Gh;
}

__device__ inline TileDesc bwd_film_tile(const nsky_film_net& n, int idx) {
  TileDesc d;
  const int NT = n.hidden / 32, H = n.hidden, Gh = groups_of(H);
  d.group = (long)idx * Gh;
  d.nrows = 32; d.K = H; d.k0 = 0; d.transposed = 0;
  if (idx == n.n_film * NT * 2 + (n.n_film - 1) * NT) {
    d.W = n.film_w[0]; d.ld = n.film_ld[0]; d.row0 = 0; d.nrows = n.x_dim; d.transposed = 1;
    return d;
  }
  // layers i = n_film-1 .. 1 own 3 NT tiles each, layer 0 owns 2 NT
  int i = n.n_film - 1, rem = idx;
  while (i > 0 && rem >= 3 * NT) { rem -= 3 * NT; --i; }
  if (rem < 2 * NT) {
    const int t = rem / 2, which = rem % 2;
    d.W = n.mo_w; d.ld = n.mo_ld; d.row0 = (which == 0 ? i : n.n_film + i) * H + 32 * t;
  } else {
    const int u = rem - 2 * NT;
    d.W = n.film_w[i]; d.ld = n.film_ld[i]; d.row0 = 32 * u; d.transposed = 1;
  }
  return d;
}

__device__ inline TileDesc bwd_map_tile(const nsky_film_net& n, int idx) {
  TileDesc d;
  const int NT = n.hidden / 32, H = n.hidden, Gh = groups_of(H);
  const int nkb = 2 * n.n_film * H / 64;
  d.nrows = 32; d.transposed = 1; d.k0 = 0;
  if (idx < nkb * NT) {  // k-block (64 head rows) outer, output tile inner; tiles 2 v and 2 v + 1 of a k-block share group kb NT / 2 + v
    const int kb = idx / NT, u = idx % NT;
    d.W = n.mo_w; d.ld = n.mo_ld; d.row0 = 32 * u; d.K = 64; d.k0 = 64 * kb; d.group = idx / 2; d.slab0 = 4 * (u & 1);
    return d;
  }
  idx -= nkb * NT;
  d.group = (long)nkb * (NT / 2) + (long)idx * Gh;
  d.K = H;
  if (idx < (n.n_map - 1) * NT) {
    const int l = n.n_map - 1 - idx / NT, u = idx % NT;
    d.W = n.map_w[l]; d.ld = n.map_ld[l]; d.row0 = 32 * u;
    return d;
  }
  idx -= (n.n_map - 1) * NT;
  d.W = n.map_w[0]; d.ld = n.map_ld[0]; d.row0 = 32 * idx; d.nrows = min(32, n.cond_dim - 32 * idx);
  return d;
}

__host__ __device__ inline void dir_layout(const nsky_film_net& n, int direction, long& total_groups, int& n_tiles) {
  if (direction == 0) { const Layout L = fwd_layout(n); total_groups = L.total_groups; n_tiles = L.n_tiles; }
  else if (direction == 1) bwd_film_layout(n, total_groups, n_tiles);
  else bwd_map_layout(n, total_groups, n_tiles);
}

// bias table: [mapping layer l: H][mapping head: 2 n_film H][FiLM layer i: H][head: 32]
__device__ inline void write_bias_table(const nsky_film_net& n, float* bl, int tid) {
  const int H = n.hidden;
  int off = 0;
  for (int l = 0; l < n.n_map; ++l, off += H)
    for (int i = tid; i < H; i += 256) bl[off + i] = n.map_b[l] ? n.map_b[l][i] : 0.0f;
  for (int i = tid; i < 2 * n.n_film * H; i += 256) bl[off + i] = n.mo_b ? n.mo_b[i] : 0.0f;
  off += 2 * n.n_film * H;
  for (int l = 0; l < n.n_film; ++l, off += H)
    for (int i = tid; i < H; i += 256) bl[off + i] = n.film_b[l] ? n.film_b[l][i] : 0.0f;
  for (int i = tid; i < 32; i += 256) bl[off + i] = (n.out_b && i < n.out_dim) ? n.out_b[i] : 0.0f;
}


__global__ __launch_bounds__(256) void film_pack_kernel(nsky_film_net net, int direction, unsigned char* __restrict__ stream,
                                                        float* __restrict__ table) {
  __shared__ float w[32][PACK_KMAX + 1];
  __shared__ float red[256];
  const int tid = threadIdx.x;
  long total_groups;
  int n_tiles;
  dir_layout(net, direction, total_groups, n_tiles);
  if ((int)blockIdx.x == n_tiles) {  // the extra block: every bias of the network, in the order the chain kernels index them
    write_bias_table(net, table, tid);
    return;
  }
  const TileDesc d = direction == 0 ? fwd_tile(net, fwd_layout(net), blockIdx.x)
                                    : (direction == 1 ? bwd_film_tile(net, blockIdx.x) : bwd_map_tile(net, blockIdx.x));
  pack_tile(d, stream, table + BIAS_FLOATS, w, red);
}



struct FwdArgs {
  nsky_film_net net;
  const unsigned char* stream;
  const float* table;  // [BIAS_FLOATS biases | SCALE_FLOATS reciprocal tile scales] (film_pack_kernel)
  const float* cond; int ldcond;
  const float* x; int ldx;
  int M;
  float* h_save[MAXL];  // mapping activations (after LeakyReLU), native [ceil32(M), H]; NULL = not kept
  float* z_save[MAXL];  // FiLM pre-activations W y + b, native [ceil32(M), H]; NULL = not kept
  float* y_save[MAXL];  // FiLM outputs, native [ceil32(M), H]; never NULL: also the hand-off to the next layer
  float* res; int ldres;
};


template <int H, int KSC>
__global__ __launch_bounds__(256, 1) void film_fwd_kernel(const FwdArgs a) {
  constexpr int NT = H / 32, KS = H / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + (BIAS_FLOATS + SCALE_FLOATS) * 4];
  float* bl = reinterpret_cast<float*>(smem + RING_BYTES);
  const nsky_film_net& net = a.net;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  float* sl = bl + BIAS_FLOATS;  // reciprocal tile scales, stream order
  {  // biases + tile scales: one table written by the pack kernel; all loads in flight at once
    constexpr int N4 = (BIAS_FLOATS + SCALE_FLOATS) / 4;
    float4 q[(N4 + 255) / 256];
#pragma unroll
    for (int i = 0; i < (N4 + 255) / 256; ++i)
      if (i * 256 + tid < N4) q[i] = ldg4(a.table + 4 * (i * 256 + tid));
#pragma unroll
    for (int i = 0; i < (N4 + 255) / 256; ++i)
      if (i * 256 + tid < N4) *reinterpret_cast<float4*>(bl + 4 * (i * 256 + tid)) = q[i];
  }
  __syncthreads();
  WStream ws;
  ws.src = a.stream + wave * 4096 + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * 4096;
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin(ws);

  const long rt = (long)blockIdx.x * 4 + wave;  // 32-row tile of this wave
  const long row = rt * 32 + c;
  const bool live = row < a.M;
  const long rowc = live ? row : a.M - 1;
  const bool wave_live = rt * 32 < a.M;  // wave-uniform: the tile holds at least one real row (dead rows of it are stored too)
  const int n_map = net.n_map, n_film = net.n_film;
  int tile = 0;  // index into scales[], stream order

  // ------------------------------------------------------------------ mapping network
  f16x8 hh[KS], hl[KS];  // hidden state planes
  float h_inv;           // 1 / row scale of the planes
  {
    f16x8 ch[KSC], cl[KSC];
    const int ksc = ksteps_of(net.cond_dim);
    const float c_inv = load_planes<KSC>(a.cond + rowc * a.ldcond, (net.cond_dim + 3) & ~3, ksc, h, ch, cl);
    float hn[NT][16];
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      product_dyn<KSC>(ws, ksc, ch, cl, acc);
      const float inv = c_inv * sl[tile++];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bl + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = fmaf(acc[4 * g + q], inv, bb[q]);
          const float o = v > 0.0f ? v : 0.2f * v;
          hn[t][4 * g + q] = o;
          m = fmaxf(m, fabsf(o));
        }
      }
      if (a.h_save[0] && wave_live) store_tile_nt(a.h_save[0] + (rt * NT + t) * 1024, lane, hn[t]);
    }
    const float s = row_scale(m, h_inv);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = hn[t][8 * u + j] * s;
        split8(x8, hh[2 * t + u], hl[2 * t + u]);
      }
  }
  for (int l = 1; l < n_map; ++l) {
    float hn[NT][16];
    float m = 0.0f;
    const float* bias = bl + l * H;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      product<KS, 4, false, RING_GROUPS, FWD_FRAG_BUFFERS>(ws, hh, hl, acc);
      const float inv = h_inv * sl[tile++];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = fmaf(acc[4 * g + q], inv, bb[q]);
          const float o = v > 0.0f ? v : 0.2f * v;
          hn[t][4 * g + q] = o;
          m = fmaxf(m, fabsf(o));
        }
      }
      if (a.h_save[l] && wave_live) store_tile_nt(a.h_save[l] + (rt * NT + t) * 1024, lane, hn[t]);
    }
    const float s = row_scale(m, h_inv);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = hn[t][8 * u + j] * s;
        split8(x8, hh[2 * t + u], hl[2 * t + u]);
      }
  }
  // ------------------------------------------------------------------ FiLM layers
  f16x8 xh[1], xl[1];
  const float x_inv = load_planes<1>(a.x + rowc * a.ldx, (net.x_dim + 3) & ~3, 1, h, xh, xl);
  f16x8 yh[KS], yl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) { yh[ks][j] = (_Float16)0.0f; yl[ks][j] = (_Float16)0.0f; }
  const float* bias_mo = bl + n_map * H;
  const float* bias_film = bias_mo + 2 * n_film * H;
  for (int i = 0; i < n_film; ++i) {
    const float* bF = bias_mo + i * H;
    const float* bP = bias_mo + (n_film + i) * H;
    const float* bZ = bias_film + i * H;
    // a wave whose tile lies wholly beyond M stores nothing and reads tile 0 back (in bounds; its results are never stored)
    float* zblk = a.z_save[i] ? a.z_save[i] + (wave_live ? rt : 0) * NT * 1024 : nullptr;
    float* yblk = a.y_save[i] + (wave_live ? rt : 0) * NT * 1024;
    for (int t = 0; t < NT; ++t) {
      f32x16 aF, aP, aZ;
#pragma unroll
      for (int r = 0; r < 16; ++r) { aF[r] = 0.0f; aP[r] = 0.0f; aZ[r] = 0.0f; }
      product<KS, 4, false, RING_GROUPS, FWD_FRAG_BUFFERS>(ws, hh, hl, aF);
      product<KS, 4, false, RING_GROUPS, FWD_FRAG_BUFFERS>(ws, hh, hl, aP);
      float z_unscale;
      if (i == 0) {
        product<1>(ws, xh, xl, aZ);
        z_unscale = x_inv;
      } else {
        product<KS, 4, false, RING_GROUPS, FWD_FRAG_BUFFERS>(ws, yh, yl, aZ);
        z_unscale = 1.0f / Y_SCALE;
      }
      const float iF = h_inv * sl[tile], iP = h_inv * sl[tile + 1], iZ = z_unscale * sl[tile + 2];
      tile += 3;
      float zz[16], yy[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * t + 8 * g + 4 * h;
        const float4 b4F = *reinterpret_cast<const float4*>(bF + fo);
        const float4 b4P = *reinterpret_cast<const float4*>(bP + fo);
        const float4 b4Z = *reinterpret_cast<const float4*>(bZ + fo);
        const float bf[4] = {b4F.x, b4F.y, b4F.z, b4F.w}, bp[4] = {b4P.x, b4P.y, b4P.z, b4P.w}, bz[4] = {b4Z.x, b4Z.y, b4Z.z, b4Z.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * g + q;
          const float F = fmaf(aF[r], iF, bf[q]), P = fmaf(aP[r], iP, bp[q]), z = fmaf(aZ[r], iZ, bz[q]);
          zz[r] = z;
          yy[r] = sin_cw(fmaf(fmaf(15.0f, F, 30.0f), z, P));
        }
      }
      if (wave_live) {
        if (zblk) store_tile_nt(zblk + t * 1024, lane, zz);
        store_tile(yblk + t * 1024, lane, yy);
      }
    }
    // hand-off: this lane reads back exactly the 16-byte pieces it stored (k-step ks, half u = register group 2 (ks & 1) + u of
    // tile ks / 2)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float x8[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 q = ldg4_nt(yblk + (ks >> 1) * 1024 + (2 * (ks & 1) + u) * 256 + lane * 4);
        x8[4 * u] = q.x * Y_SCALE; x8[4 * u + 1] = q.y * Y_SCALE; x8[4 * u + 2] = q.z * Y_SCALE; x8[4 * u + 3] = q.w * Y_SCALE;
      }
      split8(x8, yh[ks], yl[ks]);
    }
  }

  // ------------------------------------------------------------------ head
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    product<KS, 4, false, RING_GROUPS, FWD_FRAG_BUFFERS>(ws, yh, yl, acc);
    const float inv = sl[tile] / Y_SCALE;
    const float* bO = bias_film + n_film * H;
    if (live && h == 0)
      stg4(a.res + row * a.ldres, make_float4(fmaf(acc[0], inv, bO[0]), fmaf(acc[1], inv, bO[1]), fmaf(acc[2], inv, bO[2]),
                                              fmaf(acc[3], inv, bO[3])));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the tail groups of the ring are still landing
}

// =====================================================================================================================
// Backward, FiLM part.  Per 32-row tile of a wave, walking the FiLM layers from the last to the first:
//   F, phase of a 32-feature tile are RE-FORMED from the last mapping activation (two products, as in the forward: the
//   frequency / phase matrix is never stored), z is read back (tile-native), then
//     g = dY cos(arg), dz = g f, dF = 15 g z, dphase = g        (arg = f z + phase, f = 15 F + 30)
//   dz (for the weight gradient), dF and dphase (for the mapping head) are stored tile-native; dz stays in registers, is
//   pre-scaled per batch row (a per-lane scalar) and split into fp16 hi / residual planes, and dY of the layer below is
//   W^T dz on the transposed weight tiles.  Gradients therefore get the same fp32-grade products as the forward.
// A vector-memory load the compiler knows about would make it wait for every LDS-DMA piece issued before it (one in-order
// counter), so z is fetched with hidden loads one tile ahead and waited for with a counted vmcnt.
// Rounds 2-4 ran this as an eight-wave kernel (256 registers per wave: the incoming gradient parked in the layer's dz buffer, dz read
// back for the planes: 8 KB per row and layer, 12.3 GB per DDF launch); round 5's film_bwd4_kernel below keeps the gradient in the
// registers of a four-wave workgroup (5 KB per row and layer) and replaced it: -0.5 ms per step on a same-box A/B.

struct BwdFilmArgs {
  nsky_film_net net;
  const unsigned char* stream;
  const float* table;
  int M;
  const float* d_res; int ldres;   // [M, ldres] gradient of the raw head output (first out_dim columns)
  const float* h_last;             // native [ceil32(M), H]
  const float* z_save[MAXL];       // native
  float* dz_save[MAXL];            // native [ceil32(M), H]
  float* dfp;                      // native [ceil32(M), 2 n_film H]: dF of layer i in columns i H .., dphase in (n_film + i) H ..
  float* dfp_rowmax;               // [ceil32(M)] max |dfp| per batch row
  float* gmax;                     // zero-initialised by the caller: [i] = max |dz_save[i]|, [n_film] = max |dfp|
  float* d_x; int ldx;             // optional [M, ldx]: gradient w.r.t. the FiLM input rows (pad columns zeroed)
  int full_wgs, tail_wgs, tail_k;  // (unused by the four-wave kernel)
};


// Tail workgroups of the EIGHT-wave kernels (the mapping backward, the sdf chain: two waves per SIMD, 256 batch rows share one weight
// stream).  A launch of n wave tiles (32 rows each) on C CUs runs F = floor(n / (8 C)) C full workgroups of eight tiles; the
// remaining R = n - 8 F tiles (less than one workgroup round) do not get a round of full workgroups on a few CUs: they are dealt k =
// ceil(R / C) to a workgroup (T = ceil(R / k) "tail" workgroups, FIRST in the grid), whose other waves only take part in the ring's
// hand-shakes (product_skip).  A tail workgroup is done sooner than a full one and the dispatcher hands its CU the next workgroup, so
// the remainder costs its share of a round instead of a whole one (263 456 DDF rows = 8233 tiles: 1024 full + 41 one-tile workgroups
// instead of 1030 full ones = five rounds of time for 4.02 of work).
struct TailPlan { int full_wgs, tail_wgs, tail_k; };
inline TailPlan tail_plan(long n_tiles, int cus) {
  TailPlan p;
  p.full_wgs = (int)(n_tiles / (8L * cus)) * cus;
  const long rem = n_tiles - 8L * p.full_wgs;
  if (rem >= 8L * cus - cus) {  // nearly a whole round: full workgroups
    p.full_wgs += (int)((rem + 7) / 8); p.tail_wgs = 0; p.tail_k = 0;
    return p;
  }
  p.tail_k = (int)((rem + cus - 1) / cus);
  p.tail_wgs = p.tail_k ? (int)((rem + p.tail_k - 1) / p.tail_k) : 0;
  return p;
}
// wave tile of this wave, or -1 (a wave of a tail workgroup without one)
__device__ __forceinline__ long tail_tile(int full_wgs, int tail_wgs, int tail_k, long n_tiles, int wave) {
  const int b = blockIdx.x;
  if (b >= tail_wgs) return (long)(b - tail_wgs) * 8 + wave;
  const long t = 8L * full_wgs + (long)b * tail_k + wave;
  return (wave < tail_k && t < n_tiles) ? t : -1;
}

// product of a wave WITH a row tile, or (ACTIVE = false) its share of the ring's hand-shakes
template <int KSN, int PWN, bool ACTIVE>
__device__ __forceinline__ void prodw(WStream& ws, const f16x8 (&bh)[KSN], const f16x8 (&bl)[KSN], f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  if (ACTIVE) product<KSN, PWN>(ws, bh, bl, acc);
  else product_skip<KSN, PWN, false>(ws);
}

// ---------------------------------------------------------------------------------------------------------------------
// The FiLM backward with the gradient IN REGISTERS (round 5).  Four waves per workgroup, one per SIMD, 512 registers each -- the forward
// kernel's shape -- so that a wave can hold, beside the 128 registers of h planes, the layer's incoming gradient dY as eight accumulator
// tiles: pass 1 turns dY into dz IN PLACE (tile by tile, while dz, dF and dphase go out to memory for the weight gradients and the mapping
// backward), the planes of W^T dz are split from those registers once the row maximum is known, and the W^T products write the next
// layer's dY back into the same tiles.  What the eight-wave kernel moves through memory because 256 registers cannot hold two of the
// three matrices -- the parked dY (1 KB per row and layer out and in) and the dz read-back (1 KB) -- never leaves the wave: 5 KB per row
// and layer (h_last, z in; dz, dF, dphase out) instead of 8.  The price is the forward kernel's: the weight stream is walked once per
// 128 rows instead of per 256, and one wave per SIMD hides nothing behind a second wave.
// a hidden 16-byte load through a wave-uniform base (SGPR pair) and a per-lane byte offset: no 64-bit address per lane
template <int OFF>
__device__ __forceinline__ void hidden_load4_s(f32x4& q, const float* sbase, int voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(q) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

template <int H, bool ACTIVE>
__device__ __forceinline__ void film_bwd4_tile(const BwdFilmArgs& a, WStream& ws, const float* bl, const float* sl, const float* wo, long rt, int lane) {
  constexpr int NT = H / 32, KS = H / 16, GH = (KS + GSLABS - 1) / GSLABS, PW = 4;
  const nsky_film_net& net = a.net;
  const int c = lane & 31, h = lane >> 5;
  const long row = rt * 32 + c;
  const bool live = ACTIVE && row < a.M;
  const long rowc = row < a.M ? row : a.M - 1;
  const int n_film = net.n_film;
  f32x16 dY[NT];  // gradient w.r.t. the layer's sine outputs on entry to a layer, its dz after pass 1
  float h_inv = 1.0f, h_scale = 1.0f;
  if (ACTIVE) {
    float m = 0.0f;
    for (int t = 0; t < NT; ++t) {
      float hv[16];
      load_tile(a.h_last + (rt * NT + t) * 1024, lane, hv);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(hv[r]));
    }
    h_scale = row_scale(m, h_inv);
    // head gradient -> dY of the last FiLM layer (register 4 g + q of tile t = feature 32 t + 8 g + 4 h + q)
    const float4 dr = ldg4(a.d_res + rowc * a.ldres);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * t + 8 * g + 4 * h;
        const float4 w0 = *reinterpret_cast<const float4*>(wo + fo), w1 = *reinterpret_cast<const float4*>(wo + H + fo);
        const float4 w2 = *reinterpret_cast<const float4*>(wo + 2 * H + fo), w3 = *reinterpret_cast<const float4*>(wo + 3 * H + fo);
        dY[t][4 * g] = dr.x * w0.x + dr.y * w1.x + dr.z * w2.x + dr.w * w3.x;
        dY[t][4 * g + 1] = dr.x * w0.y + dr.y * w1.y + dr.z * w2.y + dr.w * w3.y;
        dY[t][4 * g + 2] = dr.x * w0.z + dr.y * w1.z + dr.z * w2.z + dr.w * w3.z;
        dY[t][4 * g + 3] = dr.x * w0.w + dr.y * w1.w + dr.z * w2.w + dr.w * w3.w;
      }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dY[t][r] = 0.0f;
  }
  const float* bias_mo = bl + net.n_map * H;
  float fp_max = 0.0f;
  int tile = 0;
  // the planes of the last mapping activation (every layer's F / phase products read them): at H = 128 formed ONCE and resident through
  // all layers; at H = 256 they, the gradient tiles and pass 2's dz planes are 384 registers of 512 and the allocator gives up (357
  // spills): re-formed per layer from h_last, dead during pass 2
  constexpr bool H_RESIDENT = H <= 128;
  f16x8 hh[KS], hl[KS];
  auto load_h = [&]() {
    int hoff = 0;
    asm volatile("" : "+s"(hoff));  // (the 32 tile addresses are the same in every layer: left visible, they are hoisted out of the layer loop and live in scratch)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float hv[16];
      load_tile(a.h_last + ((rt * NT + t) * 1024 + hoff), lane, hv);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x8[j] = hv[8 * u + j] * h_scale;
        split8(x8, hh[2 * t + u], hl[2 * t + u]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // compiler-visible loads: none pending when the hidden loads are counted
  };
  if (ACTIVE && H_RESIDENT) load_h();
  for (int i = n_film - 1; i >= 0; --i) {
    const float* bF = bias_mo + i * H;
    const float* bP = bias_mo + (n_film + i) * H;
    // wave-uniform bases (scalar registers) + ONE per-lane offset: eight 64-bit per-lane pointers were what the loop kept reloading from scratch
    const float* zp = a.z_save[i] + rt * NT * 1024;
    float* dzp = a.dz_save[i] + rt * NT * 1024;
    float* dFp = a.dfp + (rt * (2 * n_film * NT) + (long)i * NT) * 1024;
    float* dPp = a.dfp + (rt * (2 * n_film * NT) + (long)(n_film + i) * NT) * 1024;
    float dz_max = 0.0f;
    {
      if (ACTIVE && !H_RESIDENT) load_h();
      // ---- pass 1: dY -> dz in place, dF, dphase, tile by tile
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 zq[4];
        int toff = t * 1024;
        asm volatile("" : "+s"(toff));  // (an opaque offset, not opaque pointers: those would lose their address space and turn into flat_* accesses)
        const float* zpt = zp + toff;
        float *dzt = dzp + toff, *dFt = dFp + toff, *dPt = dPp + toff;
        if (ACTIVE) {
#pragma unroll
          for (int g = 0; g < 4; ++g) zq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
          hidden_load4_s<0>(zq[0], zpt, lane * 16);
          hidden_load4_s<1024>(zq[1], zpt, lane * 16);
          hidden_load4_s<2048>(zq[2], zpt, lane * 16);
          hidden_load4_s<3072>(zq[3], zpt, lane * 16);
        }
        f32x16 aF, aP;
        prodw<KS, PW, ACTIVE>(ws, hh, hl, aF);
        prodw<KS, PW, ACTIVE>(ws, hh, hl, aP);
        if (ACTIVE) {
          hidden_wait<2 * GH * PW>(zq);  // requested just before the two products: only their 2 GH transitions x PW pieces are younger
          const float iF = h_inv * sl[tile], iP = h_inv * sl[tile + 1];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g + 4 * h;
            const float4 b4F = *reinterpret_cast<const float4*>(bF + fo);
            const float4 b4P = *reinterpret_cast<const float4*>(bP + fo);
            const float bf[4] = {b4F.x, b4F.y, b4F.z, b4F.w}, bp[4] = {b4P.x, b4P.y, b4P.z, b4P.w};
            float dzv[4], dFv[4], dPv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = 4 * g + q;
              const float F = fmaf(aF[r], iF, bf[q]), P = fmaf(aP[r], iP, bp[q]), z = zq[g][q];
              const float f = fmaf(15.0f, F, 30.0f);
              const float gc = dY[t][r] * cos_cw(fmaf(f, z, P));
              dzv[q] = gc * f;
              dFv[q] = 15.0f * gc * z;
              dPv[q] = gc;
              dY[t][r] = dzv[q];
              dz_max = fmaxf(dz_max, fabsf(dzv[q]));
              fp_max = fmaxf(fp_max, fmaxf(fabsf(dFv[q]), fabsf(gc)));
            }
            stg4(dzt + (g * 256 + lane * 4), make_float4(dzv[0], dzv[1], dzv[2], dzv[3]));
            stg4(dFt + (g * 256 + lane * 4), make_float4(dFv[0], dFv[1], dFv[2], dFv[3]));
            stg4(dPt + (g * 256 + lane * 4), make_float4(dPv[0], dPv[1], dPv[2], dPv[3]));
            asm volatile("" : "+v"(dz_max), "+v"(fp_max));
          }
        }
        tile += 2;
      }
    }
    // ---- pass 2: dY of the layer below = W_i^T dz from the registers (i = 0: the gradient w.r.t. the input rows, one tile)
    {
      float dz_inv = 1.0f;
      f16x8 dh_[KS], dl_[KS];
      if (ACTIVE) {
        publish_max(a.gmax + i, dz_max, live, true, lane);
        const float s = row_scale(dz_max, dz_inv);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          float x8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x8[j] = dY[ks >> 1][8 * (ks & 1) + j] * s;
          split8(x8, dh_[ks], dl_[ks]);
        }
      }
      if (i > 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          f32x16 acc;
          prodw<KS, PW, ACTIVE>(ws, dh_, dl_, acc);
          if (ACTIVE) {
            const float inv = dz_inv * sl[tile];
#pragma unroll
            for (int r = 0; r < 16; ++r) dY[u][r] = acc[r] * inv;
          }
          ++tile;
        }
      } else {
        f32x16 acc;
        prodw<KS, PW, ACTIVE>(ws, dh_, dl_, acc);
        if (ACTIVE && a.d_x && live) {
          const float inv = dz_inv * sl[tile];
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (8 * g + 4 * h < a.ldx)
              stg4(a.d_x + row * a.ldx + 8 * g + 4 * h, make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv));
        }
        ++tile;
      }
    }
  }
  if (ACTIVE) {
    fp_max = fmaxf(fp_max, __shfl_xor(fp_max, 32, 64));
    if (h == 0) a.dfp_rowmax[row] = fp_max;
    publish_max(a.gmax + n_film, fp_max, live, true, lane);
  }
}

template <int H>
__global__ __launch_bounds__(256, 1) void film_bwd4_kernel(const BwdFilmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + (BIAS_FLOATS + SCALE_FLOATS) * 4 + 4 * H * 4];
  float* bl = reinterpret_cast<float*>(smem + RING_BYTES);
  float* sl = bl + BIAS_FLOATS;
  float* wo = sl + SCALE_FLOATS;  // head weights [4][H] (rows >= out_dim zero)
  const nsky_film_net& net = a.net;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    constexpr int N4 = (BIAS_FLOATS + SCALE_FLOATS) / 4;
    float4 q[(N4 + 255) / 256];
#pragma unroll
    for (int i = 0; i < (N4 + 255) / 256; ++i)
      if (i * 256 + tid < N4) q[i] = ldg4(a.table + 4 * (i * 256 + tid));
#pragma unroll
    for (int i = 0; i < (N4 + 255) / 256; ++i)
      if (i * 256 + tid < N4) *reinterpret_cast<float4*>(bl + 4 * (i * 256 + tid)) = q[i];
    for (int i = tid; i < 4 * H; i += 256) wo[i] = (i / H) < net.out_dim ? net.out_w[(long)(i / H) * net.out_ld + (i % H)] : 0.0f;
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every compiler-visible memory operation is done before the DMA stream starts
  WStream ws;
  ws.src = a.stream + wave * 4096 + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * 4096;
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin(ws);
  const long n_tiles = (a.M + 31) / 32;
  const long rt = (long)blockIdx.x * 4 + wave;
  if (rt < n_tiles) film_bwd4_tile<H, true>(a, ws, bl, sl, wo, rt, lane);
  else film_bwd4_tile<H, false>(a, ws, bl, sl, wo, 0, lane);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// Backward, mapping network.  dh = sum over the 2 n_film H head rows of Wmo^T dfp (k-block outer: 64 head rows of dfp are
// fetched tile-native one block ahead with hidden loads, pre-scaled by the row's maximum over ALL of dfp and split; dfp is read
// ONCE), then the
// mapping layers backwards, hidden state in registers as in the forward: dpre = dh * leaky'(h) is stored tile-native (weight
// gradients), split, and multiplied by the transposed weight tiles; the last product yields d_cond (row-major).
struct BwdMapArgs {
  nsky_film_net net;
  const unsigned char* stream;
  const float* table;
  int M;
  const float* dfp;          // native [ceil32(M), 2 n_film H]
  const float* dfp_rowmax;   // [ceil32(M)]
  const float* h_save[MAXL]; // native
  float* dpre_save[MAXL];    // native [ceil32(M), H]
  float* d_cond; int ldcond; // [M, ldcond] or NULL
  float* gmax;               // zero-initialised by the caller: [l] = max |dpre_save[l]|
  int full_wgs, tail_wgs, tail_k;  // tail_plan of the launch
};

template <int H, bool ACTIVE>
__device__ __forceinline__ void film_bwd_map_tile(const BwdMapArgs& a, WStream& ws, const float* sl, long rt, int lane) {
  constexpr int NT = H / 32, KS = H / 16, PW = 2;
  const nsky_film_net& net = a.net;
  const int c = lane & 31, h = lane >> 5;
  const long row = rt * 32 + c;
  const bool live = ACTIVE && row < a.M;
  const int nkb = 2 * net.n_film * H / 64, ntot = 2 * net.n_film * NT;
  float f_inv = 1.0f, f_scale = 1.0f;
  if (ACTIVE) f_scale = row_scale(a.dfp_rowmax[rt * 32 + c], f_inv);  // both lane halves read the same row: the shuffle is a no-op
  int tile = 0;
  const float* fblk = a.dfp + rt * ntot * 1024;
  const int top = net.n_map - 1;
  float m = 0.0f;  // largest |dpre_top| of this lane's row half
  // ONE pass over the 2 n_film H head rows: all NT accumulator tiles of dh stay in registers (NT x 16), the head rows come 64 at a
  // time (two native tiles = 8 pieces, one k-block ahead).  A k-block's weight tiles carry their own power-of-two scales; the
  // accumulators are kept in units of the current tile's scale (an exact rescale by the ratio of two powers of two when it changes),
  // so the MFMAs add straight into them.
  f32x4 fq[8];
#define NSKY_FQ_WAIT(N)                                                                                                         \
  asm volatile("s_waitcnt vmcnt(%8)"                                                                                            \
               : "+v"(fq[0]), "+v"(fq[1]), "+v"(fq[2]), "+v"(fq[3]), "+v"(fq[4]), "+v"(fq[5]), "+v"(fq[6]), "+v"(fq[7])           \
               : "n"(N)                                                                                                         \
               : "memory")
  f32x16 dh[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[u][r] = 0.0f;
  if (ACTIVE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) hidden_load4(fq[j], fblk + (j >> 2) * 1024 + (j & 3) * 256 + lane * 4);
    NSKY_FQ_WAIT(0);  // first block (everything older lands with it)
  }
  for (int kb = 0; kb < nkb; ++kb) {
    f16x8 ph[4], pl[4];
    if (ACTIVE) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float x8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 q = fq[(ks >> 1) * 4 + 2 * (ks & 1) + u];
          x8[4 * u] = q[0] * f_scale; x8[4 * u + 1] = q[1] * f_scale; x8[4 * u + 2] = q[2] * f_scale; x8[4 * u + 3] = q[3] * f_scale;
        }
        split8(x8, ph[ks], pl[ks]);
      }
      const int kn = kb + 1 < nkb ? kb + 1 : kb;  // the last block re-requests itself (same count of operations in flight)
#pragma unroll
      for (int j = 0; j < 8; ++j) hidden_load4(fq[j], fblk + (long)(2 * kn + (j >> 2)) * 1024 + (j & 3) * 256 + lane * 4);
    }
#pragma unroll
    for (int v = 0; v < NT / 2; ++v) {
      if (ACTIVE) {
        if (kb > 0) {  // units of the previous block's tile scale -> units of this block's (reciprocal scales: exact powers of two)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float ratio = __int_as_float(__float_as_int(sl[tile + e - NT]) - __float_as_int(sl[tile + e]) + 0x3f800000);
            if (ratio != 1.0f)
#pragma unroll
              for (int r = 0; r < 16; ++r) dh[2 * v + e][r] *= ratio;
          }
        }
        product_pair<PW>(ws, ph, pl, dh[2 * v], dh[2 * v + 1]);
      } else {
        product_skip<8, PW, false>(ws);
      }
      tile += 2;
    }
    // the next block's 8 pieces were requested before this block's NT / 2 pair products (one transition x PW DMA pieces each)
    if (ACTIVE) NSKY_FQ_WAIT(PW * (NT / 2));
  }
  // dpre of the top mapping layer: dh * leaky'(h_top) (the activation keeps the sign of the pre-activation), stored tile by tile
  if (ACTIVE) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float inv = f_inv * sl[tile - NT + u];
      float hv[16], dv[16];
      load_tile(a.h_save[top] + (rt * NT + u) * 1024, lane, hv);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float g = dh[u][r] * inv;
        dv[r] = hv[r] > 0.0f ? g : 0.2f * g;
        m = fmaxf(m, fabsf(dv[r]));
      }
      store_tile(a.dpre_save[top] + (rt * NT + u) * 1024, lane, dv);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef NSKY_FQ_WAIT

  for (int l = top; l >= 0; --l) {
    float d_inv = 1.0f;
    f16x8 ph[KS], pl[KS];
    if (ACTIVE) {
      publish_max(a.gmax + l, m, live, true, lane);
      const float s = row_scale(m, d_inv);
      // B planes of dpre_l: its NT tiles come back from where this wave stored them
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float dv[16];
        load_tile(a.dpre_save[l] + (rt * NT + u) * 1024, lane, dv);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          float x8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x8[j] = dv[8 * v + j] * s;
          split8(x8, ph[2 * u + v], pl[2 * u + v]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // compiler-visible loads: none may be pending across the products
    }
    m = 0.0f;
    if (l > 0) {
      for (int u = 0; u < NT; ++u) {
        f32x16 acc;
        prodw<KS, PW, ACTIVE>(ws, ph, pl, acc);
        if (ACTIVE) {
          const float inv = d_inv * sl[tile];
          float hv[16], dv[16];
          load_tile(a.h_save[l - 1] + (rt * NT + u) * 1024, lane, hv);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float g = acc[r] * inv;
            dv[r] = hv[r] > 0.0f ? g : 0.2f * g;
            m = fmaxf(m, fabsf(dv[r]));
          }
          store_tile(a.dpre_save[l - 1] + (rt * NT + u) * 1024, lane, dv);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ++tile;
      }
    } else {
      const int ct = (net.cond_dim + 31) / 32;
      for (int u = 0; u < ct; ++u) {
        f32x16 acc;
        prodw<KS, PW, ACTIVE>(ws, ph, pl, acc);
        if (ACTIVE && a.d_cond && live) {
          const float inv = d_inv * sl[tile];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int fo = 32 * u + 8 * g + 4 * h;
            if (fo < a.ldcond)
              stg4(a.d_cond + row * a.ldcond + fo, make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv));
          }
        }
        ++tile;
      }
    }
  }
}

template <int H>
__global__ __launch_bounds__(512, 2) void film_bwd_map_kernel(const BwdMapArgs a) {
  // Eight waves (two per SIMD, 256 registers each) share one weight stream over 256 batch rows.  The register budget is met as in
  // the FiLM backward: the head product keeps its NT accumulator tiles and only 64 head rows of operand planes at a time, and no
  // layer's matrix stays in registers -- a finished tile of dpre is stored (tile-native, also the weight gradient's operand) and the
  // layer below reads the tiles back (the lane that stored a piece loads it).  The remainder of the last round runs in tail workgroups.
  constexpr int PW = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + SCALE_FLOATS * 4];
  float* sl = reinterpret_cast<float*>(smem + RING_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < SCALE_FLOATS; i += 512) sl[i] = a.table[BIAS_FLOATS + i];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws.src = a.stream + wave * (PW * 1024) + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * (PW * 1024);
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin<PW>(ws);
  const long n_tiles = (a.M + 31) / 32;
  const long rt = tail_tile(a.full_wgs, a.tail_wgs, a.tail_k, n_tiles, wave);
  if (rt >= 0 && rt < n_tiles) film_bwd_map_tile<H, true>(a, ws, sl, rt, lane);
  else film_bwd_map_tile<H, false>(a, ws, sl, 0, lane);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// SDF value chain: the geometry network evaluated for its signed distance only (SDFAlbedoField.get_sdf_at_pos,
// sdf_albedo_field.py:169-174: encode row -> Linear + Softplus(beta) -> Linear + Softplus(beta) -> the sdf row of the last Linear),
// forward and backward as one kernel each on the machinery above.  Used at the DDF termination points (2.6e5 rows per step).
// Eight waves per workgroup share the (small: 0.4 MB) weight stream; the hidden activations are written once, tile-native, for the
// backward and for the weight gradients, and read back by the wave that wrote them to form the next layer's operand planes once
// their row maximum is known.  sigmoid(beta z) is recovered from the saved softplus output: 1 - exp(-beta a).
struct SdfLayout { int NT, G0, Gh, KS0; };
__host__ __device__ inline SdfLayout sdf_layout(const nsky_sdf_net& n) {
  SdfLayout L;
  L.NT = n.hidden / 32; L.KS0 = ksteps_of(n.in_dim); L.G0 = groups_of(n.in_dim); L.Gh = groups_of(n.hidden);
  return L;
}
__host__ __device__ inline void sdf_dir_layout(const nsky_sdf_net& n, int direction, long& total_groups, int& n_tiles) {
  const SdfLayout L = sdf_layout(n);
  if (direction == 0) { n_tiles = 2 * L.NT; total_groups = (long)L.NT * (L.G0 + L.Gh); }
  else { const int ct = (n.in_dim + 31) / 32; n_tiles = L.NT + ct; total_groups = (long)(L.NT + ct) * L.Gh; }
}
__device__ inline TileDesc sdf_tile(const nsky_sdf_net& n, int direction, int idx) {
  const SdfLayout L = sdf_layout(n);
  TileDesc d;
  d.k0 = 0; d.nrows = 32;
  if (direction == 0) {
    d.transposed = 0;
    if (idx < L.NT) { d.W = n.w0; d.ld = n.ld0; d.row0 = 32 * idx; d.K = n.in_dim; d.group = (long)idx * L.G0; }
    else { d.W = n.w1; d.ld = n.ld1; d.row0 = 32 * (idx - L.NT); d.K = n.hidden; d.group = (long)L.NT * L.G0 + (long)(idx - L.NT) * L.Gh; }
  } else {
    d.transposed = 1; d.K = n.hidden; d.group = (long)idx * L.Gh;
    if (idx < L.NT) { d.W = n.w1; d.ld = n.ld1; d.row0 = 32 * idx; }
    else { d.W = n.w0; d.ld = n.ld0; d.row0 = 32 * (idx - L.NT); d.nrows = min(32, n.in_dim - 32 * (idx - L.NT)); }
  }
  return d;
}
// table: [b0: H][b1: H][w2 (the sdf row): H][b2: 1] at 0, reciprocal tile scales at BIAS_FLOATS
__global__ __launch_bounds__(256) void sdf_pack_kernel(nsky_sdf_net net, int direction, unsigned char* __restrict__ stream,
                                                       float* __restrict__ table) {
  __shared__ float w[32][PACK_KMAX + 1];
  __shared__ float red[256];
  long total_groups;
  int n_tiles;
  sdf_dir_layout(net, direction, total_groups, n_tiles);
  if ((int)blockIdx.x == n_tiles) {
    const int H = net.hidden;
    for (int i = threadIdx.x; i < H; i += 256) {
      table[i] = net.b0 ? net.b0[i] : 0.0f;
      table[H + i] = net.b1 ? net.b1[i] : 0.0f;
      table[2 * H + i] = net.w2[i];
    }
    if (threadIdx.x == 0) table[3 * H] = net.b2 ? net.b2[0] : 0.0f;
    return;
  }
  pack_tile(sdf_tile(net, direction, blockIdx.x), stream, table + BIAS_FLOATS, w, red);
}

struct SdfFwdArgs {
  nsky_sdf_net net;
  const unsigned char* stream;
  const float* table;
  const float* E; int ldE;   // [M, ldE] encode rows
  int M;
  float* a0; float* a1;      // native [ceil32(M), H] softplus outputs of the two hidden layers
  float* sdf;                // [M]
  int full_wgs, tail_wgs, tail_k;  // tail_plan of the launch
};


template <int H, int KS0>
__global__ __launch_bounds__(512, 2) void sdf_fwd_kernel(const SdfFwdArgs a) {
  constexpr int NT = H / 32, KS = H / 16, PW = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + (3 * H + 4 + 64) * 4];
  float* bl = reinterpret_cast<float*>(smem + RING_BYTES);  // b0 | b1 | w2 | b2
  float* sl = bl + 3 * H + 4;                               // tile scales (2 NT <= 64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  for (int i = tid; i < 3 * H + 1; i += 512) bl[i] = a.table[i];
  for (int i = tid; i < 2 * NT; i += 512) sl[i] = a.table[BIAS_FLOATS + i];
  // the remainder of the last round runs in tail workgroups (tail_plan): a wave of one without a row tile only takes part in the
  // ring's hand-shakes (product_skip)
  const long n_tiles = (a.M + 31) / 32;
  const long rt0 = tail_tile(a.full_wgs, a.tail_wgs, a.tail_k, n_tiles, wave);
  const bool wave_live = __builtin_amdgcn_readfirstlane((int)(rt0 >= 0 && rt0 < n_tiles)) != 0;
  const long rt = wave_live ? rt0 : 0, rts = rt;
  const long row = rt * 32 + c;
  const bool live = wave_live && row < a.M;
  const long rowc = row < a.M ? row : a.M - 1;
  const float beta = a.net.beta, inv_beta = 1.0f / beta;
  f16x8 eh[KS0], el[KS0];
  const float e_inv = load_planes<KS0>(a.E + rowc * a.ldE, a.net.in_dim, KS0, h, eh, el);
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws.src = a.stream + wave * (PW * 1024) + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * (PW * 1024);
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin<PW>(ws);
  int tile = 0;
  float* a0blk = a.a0 + rts * NT * 1024;
  float* a1blk = a.a1 + rts * NT * 1024;
  float m = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (wave_live) product<KS0, PW>(ws, eh, el, acc);
    else product_skip<KS0, PW, false>(ws);
    const float inv = e_inv * sl[tile++];
    if (wave_live) {
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bl + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * g + q] = softplus_b(fmaf(acc[4 * g + q], inv, bb[q]), beta, inv_beta);
          m = fmaxf(m, v[4 * g + q]);
        }
      }
      store_tile(a0blk + t * 1024, lane, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's a0 stores have left before it reads them back
  f16x8 ah[KS], al[KS];
  float a_inv = 1.0f;
  if (wave_live) {
    a_inv = planes_from_tiles<NT>(a0blk, lane, m, ah, al);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ah[ks][j] = (_Float16)0.0f; al[ks][j] = (_Float16)0.0f; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float part = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (wave_live) product<KS, PW>(ws, ah, al, acc);
    else product_skip<KS, PW, false>(ws);
    const float inv = a_inv * sl[tile++];
    if (wave_live) {
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bl + H + 32 * t + 8 * g + 4 * h);
        const float4 w4 = *reinterpret_cast<const float4*>(bl + 2 * H + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * g + q] = softplus_b(fmaf(acc[4 * g + q], inv, bb[q]), beta, inv_beta);
          part = fmaf(v[4 * g + q], ww[q], part);
        }
      }
      store_tile_nt(a1blk + t * 1024, lane, v);
    }
  }
  part += __shfl_xor(part, 32, 64);  // the two lane halves hold different features of the same row
  if (live && h == 0) a.sdf[row] = part + bl[3 * H];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

struct SdfBwdArgs {
  nsky_sdf_net net;
  const unsigned char* stream;
  const float* table;
  int M;
  const float* g;            // [M] gradient of the sdf
  const float* a0; const float* a1;
  float* dz1; float* dz0;    // native [ceil32(M), H]: pre-activation gradients (also the weight gradients' operands)
  float* dE; int ldE;        // [M, ldE] or NULL
  float* dw2;                // [H] += sum_rows g a1 (or NULL)
  float* db2;                // [1] += sum_rows g (with dw2)
  float* gmax;               // [2]: max |dz1|, max |dz0| (zero-initialised by the caller)
  int full_wgs, tail_wgs, tail_k;  // tail_plan of the launch
};

template <int H>
__global__ __launch_bounds__(512, 2) void sdf_bwd_kernel(const SdfBwdArgs a) {
  constexpr int NT = H / 32, KS = H / 16, PW = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES + (2 * H + 4 + 64) * 4];
  float* w2 = reinterpret_cast<float*>(smem + RING_BYTES);  // the sdf row
  float* dw2s = w2 + H;                                      // this workgroup's sum_rows g a1, then sum_rows g
  float* sl = dw2s + H + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int ct = (a.net.in_dim + 31) / 32;
  for (int i = tid; i < H; i += 512) { w2[i] = a.table[2 * H + i]; dw2s[i] = 0.0f; }
  if (tid < 4) dw2s[H + tid] = 0.0f;
  for (int i = tid; i < NT + ct; i += 512) sl[i] = a.table[BIAS_FLOATS + i];
  const long n_tiles = (a.M + 31) / 32;
  const long rt0 = tail_tile(a.full_wgs, a.tail_wgs, a.tail_k, n_tiles, wave);  // tail workgroups: see sdf_fwd_kernel
  const bool wave_live = __builtin_amdgcn_readfirstlane((int)(rt0 >= 0 && rt0 < n_tiles)) != 0;
  const long rt = wave_live ? rt0 : 0, rts = rt;
  const long row = rt * 32 + c;
  const bool live = wave_live && row < a.M;
  const float beta = a.net.beta;
  const float g = live ? a.g[row] : 0.0f;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws.src = a.stream + wave * (PW * 1024) + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * (PW * 1024);
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin<PW>(ws);
  if (a.dw2) {
    float p = h == 0 ? g : 0.0f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (lane == 0) atomicAdd(dw2s + H, p);
  }
  int tile = 0;
  float* dz1blk = a.dz1 + rts * NT * 1024;
  float* dz0blk = a.dz0 + rts * NT * 1024;
  // ---- dz1 = g w2 sigmoid(beta z1); dw2 += g a1
  float m = 0.0f;
  for (int t = 0; wave_live && t < NT; ++t) {
    float av[16], dv[16];
    load_tile(a.a1 + (rts * NT + t) * 1024, lane, av);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 w4 = *reinterpret_cast<const float4*>(w2 + 32 * t + 8 * gq + 4 * h);
      const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * gq + q;
        dv[r] = g * ww[q] * -expm1f(-beta * av[r]);
        m = fmaxf(m, fabsf(dv[r]));
      }
    }
    if (a.dw2) {
      // dw2[feature] += sum over the 32 rows of this lane half of g a1: a halving butterfly (at every step a lane keeps the half of
      // its values its bit selects and adds the partner's: 8 + 4 + 2 + 1 + 1 = 16 exchanges instead of 16 x 5), after which lane
      // (c, h) holds the row sum of accumulator register 8 c4 + 4 c3 + 2 c2 + c1 and the even lanes add theirs: one LDS atomic
      float v8[8], v4[4], v2[2];
      const bool b4 = (c & 16) != 0, b3 = (c & 8) != 0, b2 = (c & 4) != 0, b1 = (c & 2) != 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float lo = g * av[j], hi = g * av[8 + j];
        v8[j] = (b4 ? hi : lo) + __shfl_xor(b4 ? lo : hi, 16, 64);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v4[j] = (b3 ? v8[4 + j] : v8[j]) + __shfl_xor(b3 ? v8[j] : v8[4 + j], 8, 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) v2[j] = (b2 ? v4[2 + j] : v4[j]) + __shfl_xor(b2 ? v4[j] : v4[2 + j], 4, 64);
      float v1 = (b1 ? v2[1] : v2[0]) + __shfl_xor(b1 ? v2[0] : v2[1], 2, 64);
      v1 += __shfl_xor(v1, 1, 64);
      const int r = (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
      if ((c & 1) == 0) atomicAdd(dw2s + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3), v1);
    }
    store_tile(dz1blk + t * 1024, lane, dv);
  }
  publish_max(a.gmax, m, live, wave_live, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f16x8 ph[KS], pl[KS];
  float p_inv = 1.0f;
  if (wave_live) {
    p_inv = planes_from_tiles<NT>(dz1blk, lane, m, ph, pl);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ph[ks][j] = (_Float16)0.0f; pl[ks][j] = (_Float16)0.0f; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- dz0 = (W1^T dz1) sigmoid(beta z0)
  m = 0.0f;
  for (int u = 0; u < NT; ++u) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (wave_live) product<KS, PW>(ws, ph, pl, acc);
    else product_skip<KS, PW, false>(ws);
    const float inv = p_inv * sl[tile++];
    if (wave_live) {
      float av[16], dv[16];
      load_tile(a.a0 + (rts * NT + u) * 1024, lane, av);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dv[r] = acc[r] * inv * -expm1f(-beta * av[r]);
        m = fmaxf(m, fabsf(dv[r]));
      }
      store_tile(dz0blk + u * 1024, lane, dv);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // compiler-visible loads: none pending across the next product
  }
  publish_max(a.gmax + 1, m, live, wave_live, lane);
  // ---- dE = W0^T dz0
  if (a.dE) {
    if (wave_live) p_inv = planes_from_tiles<NT>(dz0blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int u = 0; u < ct; ++u) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      if (wave_live) product<KS, PW>(ws, ph, pl, acc);
      else product_skip<KS, PW, false>(ws);
      const float inv = p_inv * sl[tile++];
      if (live) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int fo = 32 * u + 8 * gq + 4 * h;
          if (fo < a.ldE)
            stg4(a.dE + row * a.ldE + fo, make_float4(acc[4 * gq] * inv, acc[4 * gq + 1] * inv, acc[4 * gq + 2] * inv, acc[4 * gq + 3] * inv));
        }
      }
    }
  }
  if (a.dw2) {
    __syncthreads();
    for (int i = tid; i < H; i += 512)
      if (dw2s[i] != 0.0f) atomicAdd(a.dw2 + i, dw2s[i]);
    if (tid == 0 && a.db2 && dw2s[H] != 0.0f) atomicAdd(a.db2, dw2s[H]);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

int device_cus() {
  static int cus = [] {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }();
  return cus;
}

int check_sdf_net(const nsky_sdf_net* n, const char* who) {
  NSKY_CHECK_ARG(n, "%s: null network", who);
  NSKY_CHECK_ARG(n->hidden == 256 && n->in_dim >= 4 && n->in_dim <= 80 && n->in_dim % 4 == 0, "%s: hidden %d (256) / in_dim %d (4..80, multiple of 4)", who,
                 n->hidden, n->in_dim);
  NSKY_CHECK_ARG(n->w0 && n->w1 && n->w2 && n->ld0 >= n->in_dim && n->ld1 >= n->hidden && n->beta > 0.0f, "%s: weights", who);
  return NSKY_OK;
}

int check_net(const nsky_film_net* n, const char* who) {
  NSKY_CHECK_ARG(n, "%s: null network", who);
  NSKY_CHECK_ARG(n->hidden == 128 || n->hidden == 256, "%s: hidden width %d (128 or 256)", who, n->hidden);
  NSKY_CHECK_ARG(n->n_map >= 1 && n->n_map <= MAXL && n->n_film >= 1 && n->n_film <= MAXL, "%s: layer counts %d / %d", who, n->n_map, n->n_film);
  NSKY_CHECK_ARG(n->cond_dim >= 1 && n->cond_dim <= PACK_KMAX && n->x_dim >= 1 && n->x_dim <= 16 && n->out_dim >= 1 && n->out_dim <= 4,
                 "%s: cond_dim %d (<= %d), x_dim %d (<= 16), out_dim %d (<= 4)", who, n->cond_dim, PACK_KMAX, n->x_dim, n->out_dim);
  const int nb = n->n_map * n->hidden + 3 * n->n_film * n->hidden + 32;
  NSKY_CHECK_ARG(nb <= BIAS_FLOATS, "%s: %d bias values exceed the LDS table (%d)", who, nb, BIAS_FLOATS);
  NSKY_CHECK_ARG(fwd_layout(*n).n_tiles <= SCALE_FLOATS, "%s: %d weight tiles exceed the LDS scale table (%d)", who, fwd_layout(*n).n_tiles, SCALE_FLOATS);
  for (int l = 0; l < n->n_map; ++l) NSKY_CHECK_ARG(n->map_w[l] && n->map_ld[l] >= (l == 0 ? n->cond_dim : n->hidden), "%s: mapping layer %d", who, l);
  for (int l = 0; l < n->n_film; ++l) NSKY_CHECK_ARG(n->film_w[l] && n->film_ld[l] >= (l == 0 ? n->x_dim : n->hidden), "%s: FiLM layer %d", who, l);
  NSKY_CHECK_ARG(n->mo_w && n->mo_ld >= n->hidden && n->out_w && n->out_ld >= n->hidden, "%s: head weights", who);
  return NSKY_OK;
}

}  // namespace

extern "C" int nsky_film_stream_layout(const nsky_film_net* net, int32_t direction, int64_t* stream_bytes, int32_t* n_tiles) {
  if (int rc = check_net(net, "nsky_film_stream_layout")) return rc;
  NSKY_CHECK_ARG(direction >= 0 && direction <= 2, "nsky_film_stream_layout: direction %d", direction);
  long groups;
  int tiles;
  dir_layout(*net, direction, groups, tiles);
  NSKY_CHECK_ARG(tiles <= SCALE_FLOATS, "nsky_film_stream_layout: %d weight tiles exceed the scale table (%d)", tiles, SCALE_FLOATS);
  if (stream_bytes) *stream_bytes = (groups + RING_GROUPS + 2) * (int64_t)GROUP;
  if (n_tiles) *n_tiles = tiles;
  return NSKY_OK;
}

extern "C" int nsky_film_pack(const nsky_film_net* net, int32_t direction, void* stream_buf, float* table, nsky_stream_t stream) {
  if (int rc = check_net(net, "nsky_film_pack")) return rc;
  NSKY_CHECK_ARG(direction >= 0 && direction <= 2 && stream_buf && table && ((uintptr_t)stream_buf % 16) == 0 && ((uintptr_t)table % 16) == 0, "nsky_film_pack: bad arguments");
  long groups;
  int tiles;
  dir_layout(*net, direction, groups, tiles);
  NSKY_CHECK_ARG(tiles <= SCALE_FLOATS, "nsky_film_pack: %d weight tiles exceed the scale table (%d)", tiles, SCALE_FLOATS);
  if (direction != 0) NSKY_CHECK_ARG(net->hidden % 128 == 0, "nsky_film_pack: backward streams need hidden %% 128 == 0");
  hipLaunchKernelGGL(film_pack_kernel, dim3(tiles + 1), dim3(256), 0, (hipStream_t)stream, *net, direction, (unsigned char*)stream_buf, table);
  NSKY_CHECK_LAUNCH("nsky_film_pack");
  return NSKY_OK;
}

extern "C" int nsky_film_chain_fwd(const nsky_film_net* net, const void* stream_buf, const float* table, const float* cond,
                                   int32_t ldcond, const float* x, int32_t ldx, int32_t M, float* const* h_save, float* const* z_save,
                                   float* const* y_save, float* res, int32_t ldres, nsky_stream_t stream) {
  if (int rc = check_net(net, "nsky_film_chain_fwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && table && cond && x && res && y_save && M > 0, "nsky_film_chain_fwd: null operand / empty batch");
  NSKY_CHECK_ARG(ldcond % 4 == 0 && ldcond >= ((net->cond_dim + 3) & ~3) && ldx % 4 == 0 && ldx >= ((net->x_dim + 3) & ~3) && ldres >= 4 && ldres % 4 == 0,
                 "nsky_film_chain_fwd: leading dimensions (cond %d, x %d, res %d) must be multiples of 4 covering the padded widths", ldcond, ldx, ldres);
  NSKY_CHECK_ARG(((uintptr_t)cond % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)res % 16) == 0 && ((uintptr_t)stream_buf % 16) == 0 && ((uintptr_t)table % 16) == 0,
                 "nsky_film_chain_fwd: operands must be 16-byte aligned");
  FwdArgs a;
  a.net = *net;
  a.stream = (const unsigned char*)stream_buf; a.table = table;
  a.cond = cond; a.ldcond = ldcond; a.x = x; a.ldx = ldx; a.M = M; a.res = res; a.ldres = ldres;
  for (int l = 0; l < MAXL; ++l) {
    a.h_save[l] = (h_save && l < net->n_map) ? h_save[l] : nullptr;
    a.z_save[l] = (z_save && l < net->n_film) ? z_save[l] : nullptr;
    a.y_save[l] = l < net->n_film ? y_save[l] : nullptr;
    if (l < net->n_film) NSKY_CHECK_ARG(a.y_save[l] && ((uintptr_t)a.y_save[l] % 16) == 0, "nsky_film_chain_fwd: y_save[%d] missing / unaligned", l);
  }
  const dim3 grid(ceil_div(M, 128));
  const int ksc = ksteps_of(net->cond_dim);
#define NSKY_FILM_FWD(HH, KK) hipLaunchKernelGGL((film_fwd_kernel<HH, KK>), grid, dim3(256), 0, (hipStream_t)stream, a)
  if (net->hidden == 256) { if (ksc <= 4) NSKY_FILM_FWD(256, 4); else NSKY_FILM_FWD(256, 20); }
  else { if (ksc <= 4) NSKY_FILM_FWD(128, 4); else NSKY_FILM_FWD(128, 20); }
#undef NSKY_FILM_FWD
  NSKY_CHECK_LAUNCH("nsky_film_chain_fwd");
  return NSKY_OK;
}


extern "C" int nsky_film_chain_bwd_film(const nsky_film_net* net, const void* stream_buf, const float* table, int32_t M, const float* d_res,
                                        int32_t ldres, const float* h_last, const float* const* z_save, float* const* dz_save, float* dfp,
                                        float* dfp_rowmax, float* gmax, float* d_x, int32_t ldx, nsky_stream_t stream) {
  if (int rc = check_net(net, "nsky_film_chain_bwd_film")) return rc;
  NSKY_CHECK_ARG(stream_buf && table && d_res && h_last && z_save && dz_save && dfp && dfp_rowmax && gmax && M > 0, "nsky_film_chain_bwd_film: null operand / empty batch");
  NSKY_CHECK_ARG(ldres >= 4 && ldres % 4 == 0 && ((uintptr_t)d_res % 16) == 0 && ((uintptr_t)h_last % 16) == 0 && ((uintptr_t)dfp % 16) == 0 &&
                     ((uintptr_t)stream_buf % 16) == 0 && ((uintptr_t)table % 16) == 0, "nsky_film_chain_bwd_film: alignment / ldres");
  NSKY_CHECK_ARG(net->hidden % 128 == 0, "nsky_film_chain_bwd_film: hidden %% 128");
  BwdFilmArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.table = table; a.M = M; a.d_res = d_res; a.ldres = ldres; a.h_last = h_last;
  a.dfp = dfp; a.dfp_rowmax = dfp_rowmax; a.gmax = gmax; a.d_x = d_x; a.ldx = ldx;
  if (d_x) NSKY_CHECK_ARG(ldx % 4 == 0 && ldx >= ((net->x_dim + 3) & ~3) && ldx <= 16 && ((uintptr_t)d_x % 16) == 0, "nsky_film_chain_bwd_film: d_x layout (ldx %d)", ldx);
  for (int l = 0; l < MAXL; ++l) {
    a.z_save[l] = l < net->n_film ? z_save[l] : nullptr;
    a.dz_save[l] = l < net->n_film ? dz_save[l] : nullptr;
    if (l < net->n_film) NSKY_CHECK_ARG(a.z_save[l] && a.dz_save[l] && ((uintptr_t)a.z_save[l] % 16) == 0 && ((uintptr_t)a.dz_save[l] % 16) == 0, "nsky_film_chain_bwd_film: z_save / dz_save[%d]", l);
  }
  a.full_wgs = a.tail_wgs = a.tail_k = 0;  // (four row tiles per workgroup, one per wave: no tail plan)
  const dim3 grid(ceil_div(ceil_div(M, 32), 4));
  if (net->hidden == 256) hipLaunchKernelGGL((film_bwd4_kernel<256>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((film_bwd4_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_film_chain_bwd_film");
  return NSKY_OK;
}

extern "C" int nsky_film_chain_bwd_map(const nsky_film_net* net, const void* stream_buf, const float* table, int32_t M, const float* dfp,
                                       const float* dfp_rowmax, const float* const* h_save, float* const* dpre_save, float* d_cond,
                                       int32_t ldcond, float* gmax, nsky_stream_t stream) {
  if (int rc = check_net(net, "nsky_film_chain_bwd_map")) return rc;
  NSKY_CHECK_ARG(stream_buf && table && dfp && dfp_rowmax && h_save && dpre_save && gmax && M > 0, "nsky_film_chain_bwd_map: null operand / empty batch");
  NSKY_CHECK_ARG(((uintptr_t)dfp % 16) == 0 && ((uintptr_t)stream_buf % 16) == 0 && ((uintptr_t)table % 16) == 0, "nsky_film_chain_bwd_map: alignment");
  if (d_cond) NSKY_CHECK_ARG(ldcond % 4 == 0 && ldcond >= ((net->cond_dim + 3) & ~3) && ((uintptr_t)d_cond % 16) == 0, "nsky_film_chain_bwd_map: d_cond layout");
  NSKY_CHECK_ARG(net->hidden % 128 == 0, "nsky_film_chain_bwd_map: hidden %% 128");
  BwdMapArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.table = table; a.M = M; a.dfp = dfp; a.dfp_rowmax = dfp_rowmax;
  a.d_cond = d_cond; a.ldcond = ldcond; a.gmax = gmax;
  for (int l = 0; l < MAXL; ++l) {
    a.h_save[l] = l < net->n_map ? h_save[l] : nullptr;
    a.dpre_save[l] = l < net->n_map ? dpre_save[l] : nullptr;
    if (l < net->n_map) NSKY_CHECK_ARG(a.h_save[l] && a.dpre_save[l] && ((uintptr_t)a.h_save[l] % 16) == 0 && ((uintptr_t)a.dpre_save[l] % 16) == 0, "nsky_film_chain_bwd_map: h_save / dpre_save[%d]", l);
  }
  const TailPlan tp = tail_plan(ceil_div(M, 32), device_cus());
  a.full_wgs = tp.full_wgs; a.tail_wgs = tp.tail_wgs; a.tail_k = tp.tail_k;
  const dim3 grid(tp.full_wgs + tp.tail_wgs);
  if (net->hidden == 256) hipLaunchKernelGGL((film_bwd_map_kernel<256>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((film_bwd_map_kernel<128>), grid, dim3(512), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_film_chain_bwd_map");
  return NSKY_OK;
}

extern "C" int nsky_sdf_stream_layout(const nsky_sdf_net* net, int32_t direction, int64_t* stream_bytes, int32_t* n_tiles) {
  if (int rc = check_sdf_net(net, "nsky_sdf_stream_layout")) return rc;
  NSKY_CHECK_ARG(direction == 0 || direction == 1, "nsky_sdf_stream_layout: direction %d", direction);
  long groups; int tiles;
  sdf_dir_layout(*net, direction, groups, tiles);
  if (stream_bytes) *stream_bytes = (groups + RING_GROUPS + 2) * (int64_t)GROUP;
  if (n_tiles) *n_tiles = tiles;
  return NSKY_OK;
}

extern "C" int nsky_sdf_pack(const nsky_sdf_net* net, int32_t direction, void* stream_buf, float* table, nsky_stream_t stream) {
  if (int rc = check_sdf_net(net, "nsky_sdf_pack")) return rc;
  NSKY_CHECK_ARG((direction == 0 || direction == 1) && stream_buf && table, "nsky_sdf_pack: bad argument");
  long groups; int tiles;
  sdf_dir_layout(*net, direction, groups, tiles);
  hipLaunchKernelGGL(sdf_pack_kernel, dim3(tiles + 1), dim3(256), 0, (hipStream_t)stream, *net, direction, (unsigned char*)stream_buf, table);
  NSKY_CHECK_LAUNCH("nsky_sdf_pack");
  return NSKY_OK;
}

extern "C" int nsky_sdf_chain_fwd(const nsky_sdf_net* net, const void* stream_buf, const float* table, const float* E, int32_t ldE,
                                  int32_t M, float* a0_save, float* a1_save, float* sdf, nsky_stream_t stream) {
  if (int rc = check_sdf_net(net, "nsky_sdf_chain_fwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && table && E && a0_save && a1_save && sdf && M > 0 && ldE >= net->in_dim && ldE % 4 == 0, "nsky_sdf_chain_fwd: bad argument");
  NSKY_CHECK_ARG(((uintptr_t)E % 16) == 0 && ((uintptr_t)a0_save % 16) == 0 && ((uintptr_t)a1_save % 16) == 0 && ((uintptr_t)stream_buf % 16) == 0,
                 "nsky_sdf_chain_fwd: alignment");
  SdfFwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.table = table; a.E = E; a.ldE = ldE; a.M = M; a.a0 = a0_save; a.a1 = a1_save;
  a.sdf = sdf;
  const TailPlan tp = tail_plan((M + 31) / 32, device_cus());
  a.full_wgs = tp.full_wgs; a.tail_wgs = tp.tail_wgs; a.tail_k = tp.tail_k;
  const dim3 grid(tp.full_wgs + tp.tail_wgs);
  const int ks0 = ksteps_of(net->in_dim);
#define NSKY_SDF_FWD(KK) hipLaunchKernelGGL((sdf_fwd_kernel<256, KK>), grid, dim3(512), 0, (hipStream_t)stream, a)
  switch (ks0) {
    case 1: NSKY_SDF_FWD(1); break;
    case 2: NSKY_SDF_FWD(2); break;
    case 3: NSKY_SDF_FWD(3); break;
    case 4: NSKY_SDF_FWD(4); break;
    default: NSKY_SDF_FWD(5); break;
  }
#undef NSKY_SDF_FWD
  NSKY_CHECK_LAUNCH("nsky_sdf_chain_fwd");
  return NSKY_OK;
}

extern "C" int nsky_sdf_chain_bwd(const nsky_sdf_net* net, const void* stream_buf, const float* table, int32_t M, const float* g_sdf,
                                  const float* a0_save, const float* a1_save, float* dz1, float* dz0, float* dE, int32_t ldE, float* dw2,
                                  float* db2, float* gmax, nsky_stream_t stream) {
  if (int rc = check_sdf_net(net, "nsky_sdf_chain_bwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && table && g_sdf && a0_save && a1_save && dz1 && dz0 && gmax && M > 0, "nsky_sdf_chain_bwd: bad argument");
  if (dE) NSKY_CHECK_ARG(ldE % 4 == 0 && ldE >= net->in_dim && ((uintptr_t)dE % 16) == 0, "nsky_sdf_chain_bwd: dE layout");
  SdfBwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.table = table; a.M = M; a.g = g_sdf; a.a0 = a0_save; a.a1 = a1_save;
  a.dz1 = dz1; a.dz0 = dz0; a.dE = dE; a.ldE = ldE; a.dw2 = dw2; a.db2 = db2; a.gmax = gmax;
  const TailPlan tp = tail_plan((M + 31) / 32, device_cus());
  a.full_wgs = tp.full_wgs; a.tail_wgs = tp.tail_wgs; a.tail_k = tp.tail_k;
  hipLaunchKernelGGL((sdf_bwd_kernel<256>), dim3(tp.full_wgs + tp.tail_wgs), dim3(512), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_sdf_chain_bwd");
  return NSKY_OK;
}
