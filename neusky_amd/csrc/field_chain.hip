// SDF / albedo field as chain kernels: SDFAlbedoField.get_outputs + get_colors (neusky/fields/sdf_albedo_field.py:185-269) with the
// geometry network it inherits from nerfstudio's SDFField (forward_geonetwork: [x | PE6(x) | hash(x)] -> Linear + Softplus(100) ->
// Linear + Softplus(100) -> Linear -> [sdf | feat]) and the analytic replacement of torch.autograd.grad(sdf, x, create_graph=True)
// (:231-238): the encode row's Jacobian (three tangent rows per point) rides through the geometry layers in forward mode.
//
// Four kernels on the machinery of chain.h (weights streamed through the LDS ring as fp16 hi + residual planes, activations as the
// MFMA B operand in registers, fp32-grade products, tile-native saves):
//
//   field_geo_fwd    QUAD layout: a wave's 32 B-operand columns are 8 points x {value, d/dx, d/dy, d/dz}; column 4 p + j of a wave
//                    is row-set j of point p, so the four row-sets of a point sit in one hardware quad of lanes.  One product
//                    W a serves value and tangent columns alike; the epilogue gives the value lane a = softplus(z) and the tangent
//                    lanes ta_k = tz_k sigmoid(beta z), the sigmoid arriving from the quad's lane 0 by a DPP quad broadcast.  The
//                    last geometry layer's sdf row is a VALU dot with the second layer's output: sdf from the value lane, d sdf / dx_k
//                    from the tangent lanes.  Saves a0, a1 ("quad-native": tile-native matrices whose row 4 n + j is row-set j of
//                    point n) and a tile-native copy of the encode rows for the first layer's weight gradient.
//   field_colour_fwd 32 points per wave: feat = W2f a1 + b (the 256 feature rows of the last geometry layer), colour network
//                    [feat | x PE] -> 256 ReLU -> 256 ReLU -> 3 sigmoid; saves the colour-net input, c0, c1 tile-native.
//   field_colour_bwd the reverse of it down to d a1 (value rows).
//   field_geo_bwd    QUAD layout again: the hand-derived reverse of the forward-mode pass (the reference's double backward):
//                        dtz_k = dta_k s,   dz = da s + sum_k dta_k ta_k beta (1 - s),   s = sigmoid(beta z) = 1 - exp(-beta a)
//                    with the sum over a point's tangent lanes taken by two DPP quad permutes; writes the pre-activation gradients
//                    quad-native (weight-gradient operands) and d(encode rows) in the stacked [E; T0; T1; T2] layout of the encode
//                    backward.
//
// Workgroups are PERSISTENT (WAVES = 4 waves, two workgroups per CU with a 64 KB ring each): the grid is at most twice the CU count, a
// workgroup walks the weight stream cyclically, round r gives wave w the row tile r * WAVES G + w * G + b (G workgroups, b =
// blockIdx.x), so the tiles of the last, partial round are spread over the first waves of all workgroups and a wave without a tile
// only takes part in the ring's hand-shakes (product_skip): no workgroup round is paid for a handful of tail tiles.
#include "chain.h"

namespace {

constexpr int H = 256, NT = 8, KS = 16;
// Workgroup shape: WAVES waves share one weight stream through a ring of RG groups; 8 / WAVES workgroups are resident per CU (two waves
// per SIMD either way).  With two 4-wave workgroups per CU the two waves of a SIMD belong to different workgroups: no barrier couples
// them, so one's epilogue arithmetic runs beside the other's MFMAs instead of both alternating in lockstep.
constexpr int WAVES = 4, PW = 16 / WAVES, RG = WAVES == 8 ? 8 : 4, RING = RG * GROUP, THREADS = 64 * WAVES, WGS_PER_CU = 8 / WAVES;

// ---------------------------------------------------------------------------------------------------------------------
// generic packed stream: a list of layers, each ceil(rows / 32) tiles of groups_of(K) groups
struct ChainLayers { nsky_chain_layer l[NSKY_CHAIN_MAX_LAYERS]; int n; };

__host__ __device__ inline void chain_layout(const nsky_chain_layer* l, int n, long& groups, int& tiles) {
  groups = 0; tiles = 0;
  for (int i = 0; i < n; ++i) {
    const int t = (l[i].rows + 31) / 32;
    tiles += t;
    groups += (long)t * groups_of(l[i].K);
  }
}

__global__ __launch_bounds__(256) void chain_pack_kernel(ChainLayers L, unsigned char* __restrict__ stream, float* __restrict__ scales) {
  __shared__ float w[32][PACK_KMAX + 1];
  __shared__ float red[256];
  int tile = blockIdx.x, i = 0;
  long group = 0;
  for (; i < L.n; ++i) {
    const int t = (L.l[i].rows + 31) / 32;
    if (tile < t) break;
    tile -= t;
    group += (long)t * groups_of(L.l[i].K);
  }
  const nsky_chain_layer& y = L.l[i];
  TileDesc d;
  d.W = y.W; d.ld = y.ld; d.row0 = 32 * tile; d.nrows = min(32, y.rows - 32 * tile); d.K = y.K; d.transposed = y.transposed; d.k0 = 0;
  d.group = group + (long)tile * groups_of(y.K);
  pack_tile(d, stream, scales, w, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// Cross-lane moves must execute with the whole quad enabled: never inside an arm of a `c ? a : b` whose condition differs between the
// lanes of a quad (the arm runs under its own EXEC mask and the DPP move reads zeros from the disabled lanes): every quad move below
// is a statement of its own, ahead of the selects that use it.
template <int G>
__device__ __forceinline__ float quad_bcast(float v) {  // the value of the quad's lane G in all four lanes
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), G * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_bcast0(float v) { return quad_bcast<0>(v); }
__device__ __forceinline__ float quad_sum(float v) {  // sum over the quad, in all four lanes
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)));
  return v;
}
// registers 4 j .. 4 j + 3 of the quad's lane 0 (the value lane) in lane j: the value row's sixteen features of a tile are worked on by
// the four lanes of its quad, four each (the transcendental work of a softplus layer is needed for the value row only)
__device__ __forceinline__ void quad_scatter_value(const float (&x)[16], int j, float (&mine)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float b0 = quad_bcast0(x[q]), b1 = quad_bcast0(x[4 + q]), b2 = quad_bcast0(x[8 + q]), b3 = quad_bcast0(x[12 + q]);
    mine[q] = j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3));
  }
}
// the reverse: element q of lane g -> register 4 g + q of every lane
__device__ __forceinline__ void quad_gather_all(const float (&mine)[4], float (&all)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    all[q] = quad_bcast<0>(mine[q]); all[4 + q] = quad_bcast<1>(mine[q]); all[8 + q] = quad_bcast<2>(mine[q]); all[12 + q] = quad_bcast<3>(mine[q]);
  }
}

// softplus(beta; threshold 20) and sigmoid(beta v) from one exp / rcp / log
__device__ __forceinline__ void softplus_sig(float v, float beta, float inv_beta, float& sp, float& sg) {
  const float bv = beta * v;
  const float t = __expf(-fabsf(bv));
  const float u = 1.0f + t, um1 = u - 1.0f;
  const float r = __builtin_amdgcn_rcpf(u);
  const float l = um1 == 0.0f ? t : __logf(u) * (t * __builtin_amdgcn_rcpf(um1));
  sp = bv > 20.0f ? v : (fmaxf(bv, 0.0f) + l) * inv_beta;
  sg = bv >= 0.0f ? r : t * r;
}

// planes of 2 NT k-steps from NT tiles this wave stored, scaled by s (the caller's power of two)
template <int NTT>
__device__ __forceinline__ void planes_scaled(const float* blk, int lane, float s, f16x8 (&ph)[2 * NTT], f16x8 (&pl)[2 * NTT]) {
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 q = ldg4_nt(blk + t * 1024 + g * 256 + lane * 4);
      v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float x8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x8[j] = v[8 * u + j] * s;
      split8(x8, ph[2 * t + u], pl[2 * t + u]);
    }
  }
}

template <int KSN, bool ACTIVE, bool ZERO = true>
__device__ __forceinline__ void prod(WStream& ws, const f16x8 (&bh)[KSN], const f16x8 (&bl)[KSN], f32x16& acc) {
  if (ZERO) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  }
  if (ACTIVE) product<KSN, PW, true, RG>(ws, bh, bl, acc);
  else product_skip<KSN, PW, true, RG>(ws);
}

__device__ __forceinline__ void ws_setup(WStream& ws, const unsigned char* stream, unsigned char* smem, int wave, int lane, int total_groups) {
  ws.src = stream + wave * (PW * 1024) + lane * 16;
  ws.dst = (uint32_t)(uintptr_t)smem + wave * (PW * 1024);
  ws.lds_lane = (uint32_t)(uintptr_t)smem + lane * 16;
  ws_begin_wrap<PW, RG>(ws, total_groups);
}

// =====================================================================================================================
// geometry network forward, quad layout.  Stream: W0 (NT tiles, K = in_dim), W1 (NT tiles, K = H).
struct GeoFwdArgs {
  nsky_field_net net;
  const unsigned char* stream; const float* scales; int total_groups;
  const float* ET; int ldE;   // stacked encode rows [4 N, ldE]: row j N + n = row-set j of point n
  int N, n_tiles;             // points; wave tiles of 8 points
  float* a0q; float* a1q;     // quad-native [ceil32(4 N), H]
  float* Eq;                  // quad-native [ceil32(4 N), 128] copy of the encode rows (optional)
  float* a1max;               // [N] largest |a1| of the value row (optional)
  float* sdf; float* grad;    // [N], [N, 3]
  float* qmax;                // [2] (optional, caller zero-fills): max |Eq|, max |a0q| over value and tangent rows alike
};

// Epilogue of one 32-feature tile of a forward-mode softplus layer in the quad layout.  acc: this column's accumulator; inv: its
// un-scale; bias_t / w_t: the layer's bias and (DOT) the sdf row, at the tile's first feature.  The value row's pre-activations are
// spread over the quad (lane g: features 8 g + 4 h ..), softplus and sigmoid are evaluated there, the value row's outputs are stored
// by the lanes that computed them, the sigmoids return to every lane for the tangent rows ta_k = tz_k sigmoid(beta z).
// m: running row maximum of this lane's column; part_v / part_t: running dots with the sdf row (value row: quad partials).
template <bool DOT>
__device__ __forceinline__ void geo_fwd_epilogue(const f32x16& acc, float inv, const float* bias_t, const float* w_t, float beta, float inv_beta, int lane,
                                                 float* blk_t, float& m, float& part_v, float& part_t) {
  const int c = lane & 31, h = lane >> 5, j = c & 3;
  float x[16], zv[4], sp[4], sg[4], S[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = acc[r] * inv;
  quad_scatter_value(x, j, zv);
  const float4 b4 = *reinterpret_cast<const float4*>(bias_t + 8 * j + 4 * h);
  const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
  float mv = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    softplus_sig(zv[q] + bb[q], beta, inv_beta, sp[q], sg[q]);
    mv = fmaxf(mv, sp[q]);
  }
  quad_gather_all(sg, S);
  mv = quad_max(mv);
  float mt = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    x[r] *= S[r];  // tangent rows (the value lane's own x is not used past this point)
    mt = fmaxf(mt, fabsf(x[r]));
  }
  m = fmaxf(m, j == 0 ? mv : mt);
  if (DOT) {
    const float4 wv = *reinterpret_cast<const float4*>(w_t + 8 * j + 4 * h);
    part_v = fmaf(sp[0], wv.x, fmaf(sp[1], wv.y, fmaf(sp[2], wv.z, fmaf(sp[3], wv.w, part_v))));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 w4 = *reinterpret_cast<const float4*>(w_t + 8 * g + 4 * h);
      part_t = fmaf(x[4 * g], w4.x, fmaf(x[4 * g + 1], w4.y, fmaf(x[4 * g + 2], w4.z, fmaf(x[4 * g + 3], w4.w, part_t))));
    }
  }
  // the value row's piece g (features 8 g + 4 h ..) from lane g of the quad; the tangent rows from their own lanes
  stg4(blk_t + j * 256 + ((c & ~3) + 32 * h) * 4, make_float4(sp[0], sp[1], sp[2], sp[3]));
  if (j != 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) stg4(blk_t + g * 256 + lane * 4, make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]));
  }
}

template <int KS0, bool ACTIVE>
__device__ __forceinline__ void geo_fwd_tile(const GeoFwdArgs& a, WStream& ws, const float* bl, const float* sl, long tile, int lane) {
  const int c = lane & 31, h = lane >> 5, p = c >> 2, j = c & 3;
  const long n = tile * 8 + p;
  const bool live = ACTIVE && n < a.N;
  const long nc = n < a.N ? n : a.N - 1;
  const float beta = a.net.beta, inv_beta = 1.0f / beta;
  f16x8 eh[KS0], el[KS0];
  float e_inv = 1.0f;
  if (ACTIVE) {
    const float* rowp = a.ET + ((long)j * a.N + nc) * a.ldE;
    float v[KS0][8];
    float m = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int feat = 16 * ks + 8 * u + 4 * h;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (feat < a.net.in_dim) q = ldg4(rowp + feat);
        v[ks][4 * u] = q.x; v[ks][4 * u + 1] = q.y; v[ks][4 * u + 2] = q.z; v[ks][4 * u + 3] = q.w;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
      }
    if (a.Eq) {  // tile-native copy (4 tiles of 32 features, zero beyond in_dim): the X operand of the first layer's weight gradient
      float* eb = a.Eq + tile * 4 * 1024 + lane * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ks = 2 * t + (g >> 1), u = g & 1;
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ks < KS0) q = make_float4(v[ks < KS0 ? ks : 0][4 * u], v[ks < KS0 ? ks : 0][4 * u + 1], v[ks < KS0 ? ks : 0][4 * u + 2], v[ks < KS0 ? ks : 0][4 * u + 3]);
          stg4(eb + t * 1024 + g * 256, q);
        }
    }
    if (a.qmax) publish_max(a.qmax, m, live, true, lane);
    const float s = row_scale(m, e_inv);
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = v[ks][q] * s;
      split8(x, eh[ks], el[ks]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float* a0blk = a.a0q + tile * NT * 1024;
  float* a1blk = a.a1q + tile * NT * 1024;
  // ---- layer 0
  float m = 0.0f, part_v = 0.0f, part_t = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
    prod<KS0, ACTIVE>(ws, eh, el, acc);
    if (ACTIVE) geo_fwd_epilogue<false>(acc, e_inv * sl[t], bl + 32 * t, nullptr, beta, inv_beta, lane, a0blk + t * 1024, m, part_v, part_t);
  }
  f16x8 ah[KS], al[KS];
  float a_inv = 1.0f;
  if (ACTIVE) {
    if (a.qmax) publish_max(a.qmax + 1, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's a0 stores have left before it reads them back
    a_inv = planes_from_tiles<NT>(a0blk, lane, m, ah, al);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- layer 1 + the sdf row of layer 2
  m = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ah, al, acc);
    if (ACTIVE) geo_fwd_epilogue<true>(acc, a_inv * sl[NT + t], bl + H + 32 * t, bl + 2 * H + 32 * t, beta, inv_beta, lane, a1blk + t * 1024, m, part_v, part_t);
  }
  if (ACTIVE) {
    const float pq = quad_sum(part_v);  // the value row's dot: partials in the four lanes of the quad
    float part = j == 0 ? pq : part_t;
    part += __shfl_xor(part, 32, 64);  // the two lane halves hold different features of the same column
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (live && h == 0) {
      if (j == 0) {
        a.sdf[n] = part + bl[3 * H];
        if (a.a1max) a.a1max[n] = m;
      } else {
        a.grad[n * 3 + (j - 1)] = part;
      }
    }
  }
}

template <int KS0>
__global__ __launch_bounds__(THREADS, 2) void field_geo_fwd_kernel(const GeoFwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING + (3 * H + 4 + 32) * 4];
  float* bl = reinterpret_cast<float*>(smem + RING);  // b0 | b1 | w_sdf | b_sdf
  float* sl = bl + 3 * H + 4;                               // 2 NT tile scales
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < H; i += THREADS) { bl[i] = a.net.b0[i]; bl[H + i] = a.net.b1[i]; bl[2 * H + i] = a.net.w_sdf[i]; }
  if (tid == 0) bl[3 * H] = a.net.b_sdf ? a.net.b_sdf[0] : 0.0f;
  for (int i = tid; i < 2 * NT; i += THREADS) sl[i] = a.scales[i];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws_setup(ws, a.stream, smem, wave, lane, a.total_groups);
  const long G = gridDim.x;
  for (long base = 0; base < a.n_tiles; base += WAVES * G) {
    const long tile = base + wave * G + blockIdx.x;
    if (tile < a.n_tiles) geo_fwd_tile<KS0, true>(a, ws, bl, sl, tile, lane);
    else geo_fwd_tile<KS0, false>(a, ws, bl, sl, 0, lane);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// colour path forward, 32 points per wave.  Stream: W2f (NT tiles, K = H), Wc0 (NT tiles, K = 300: [feat | 0 0 0 0 | x PE | 0]),
// Wc1 (NT tiles, K = H).
constexpr int KSX = 3;         // k-steps of columns 256 .. 303 of the colour net's input (behind the 16 of the feature columns)
constexpr int XPE_TILES = 4;   // tiles per row block of the saved columns 256.. of the colour-net input (the weight gradient walks 128 features)
constexpr int XPE_COL0 = 260;  // first x / PE column of the colour-net input (fields/sdf_albedo_field.py: [feat | sdf slot, 3 pad | x PE | pad])

struct ColFwdArgs {
  nsky_field_net net;
  const unsigned char* stream; const float* scales; int total_groups;
  const float* ET; int ldE;
  int N, n_tiles;              // points; wave tiles of 32 points
  const float* a1q; const float* a1max;
  float* a1v;                  // native [ceil32(N), H]: the value rows of a1 (X operand of the feature rows' weight gradient; optional)
  float* feat;                 // native [ceil32(N), H]: the geometric features = columns 0..255 of the colour net's input
  float* xpe;                  // native [ceil32(N), 128]: its columns 256..303 (rest unwritten: the weight gradient drops them)
  float* c0; float* c1;        // native [ceil32(N), H]
  float* alb;                  // [N, 4]
};

template <bool ACTIVE>
__device__ __forceinline__ void col_fwd_tile(const ColFwdArgs& a, WStream& ws, const float* bl, const float* sl, long tile, int lane) {
  const int c = lane & 31, h = lane >> 5;
  const long n = tile * 32 + c;
  const bool live = ACTIVE && n < a.N;
  const long nc = n < a.N ? n : a.N - 1;
  f16x8 ph[KS], pl[KS], xh[KSX], xl[KSX];
  float p_inv = 1.0f;
  if (ACTIVE) {
    const float s = row_scale(a.a1max[nc], p_inv);
    const float* src = a.a1q + ((nc >> 3) * NT) * 1024 + (4 * (int)(nc & 7) + 32 * h) * 4;
    float* dst = a.a1v ? a.a1v + tile * NT * 1024 + lane * 4 : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x8[8];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int g = 2 * u + g2;
          const float4 q = ldg4(src + t * 1024 + g * 256);
          if (dst) stg4(dst + t * 1024 + g * 256, q);
          x8[4 * g2] = q.x * s; x8[4 * g2 + 1] = q.y * s; x8[4 * g2 + 2] = q.z * s; x8[4 * g2 + 3] = q.w * s;
        }
        split8(x8, ph[2 * t + u], pl[2 * t + u]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float* cinblk = a.feat + tile * NT * 1024;
  float* xpeblk = a.xpe + tile * XPE_TILES * 1024;
  float* c0blk = a.c0 + tile * NT * 1024;
  float* c1blk = a.c1 + tile * NT * 1024;
  // ---- feature rows of the last geometry layer
  float m = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      const float inv = p_inv * sl[t];
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bl + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * g + q] = fmaf(acc[4 * g + q], inv, bb[q]);
          m = fmaxf(m, fabsf(v[4 * g + q]));
        }
      }
      store_tile(cinblk + t * 1024, lane, v);
    }
  }
  if (ACTIVE) {
    // columns 256 .. 303 of the colour-net input: [0 0 0 0 | x PE (npe) | 0 ..] from the value row of the encode matrix
    const float* erow = a.ET + nc * a.ldE;
    float xv[KSX][8];
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 16 * (KS + ks) + 8 * u + 4 * h - XPE_COL0;  // column of the encode row
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col >= 0 && col < a.net.npe + 1) q = ldg4(erow + col);
        if (col + 3 >= a.net.npe) {  // the piece that straddles the end of x PE: what follows belongs to the hash features
          if (col + 0 >= a.net.npe) q.x = 0.0f;
          if (col + 1 >= a.net.npe) q.y = 0.0f;
          if (col + 2 >= a.net.npe) q.z = 0.0f;
          if (col + 3 >= a.net.npe) q.w = 0.0f;
        }
        xv[ks][4 * u] = q.x; xv[ks][4 * u + 1] = q.y; xv[ks][4 * u + 2] = q.z; xv[ks][4 * u + 3] = q.w;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
        // tiles 0, 1 of the saved columns 256.. (tile 1's second half and tiles 2, 3 stay unwritten)
        stg4(xpeblk + ((2 * ks + u) >> 2) * 1024 + ((2 * ks + u) & 3) * 256 + lane * 4, q);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float s = row_scale(m, p_inv);
    planes_scaled<NT>(cinblk, lane, s, ph, pl);
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) {
      float x8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x8[q] = xv[ks][q] * s;
      split8(x8, xh[ks], xl[ks]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- colour layer 0
  m = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);            // the tile's first two groups: the 256 feature columns
    prod<KSX, ACTIVE, false>(ws, xh, xl, acc);    // its third group: columns 256 .. 303 (same row scale, same accumulator)
    if (ACTIVE) {
      const float inv = p_inv * sl[NT + t];
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bl + H + 32 * t + 8 * g + 4 * h);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * g + q] = fmaxf(fmaf(acc[4 * g + q], inv, bb[q]), 0.0f);
          m = fmaxf(m, v[4 * g + q]);
        }
      }
      store_tile(c0blk + t * 1024, lane, v);
    }
  }
  if (ACTIVE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p_inv = planes_from_tiles<NT>(c0blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- colour layer 1 + the three output rows
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      const float inv = p_inv * sl[2 * NT + t];
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * t + 8 * g + 4 * h;
        const float4 b4 = *reinterpret_cast<const float4*>(bl + 2 * H + fo);
        const float4 w0 = *reinterpret_cast<const float4*>(bl + 3 * H + fo);
        const float4 w1 = *reinterpret_cast<const float4*>(bl + 4 * H + fo);
        const float4 w2 = *reinterpret_cast<const float4*>(bl + 5 * H + fo);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, w0a[4] = {w0.x, w0.y, w0.z, w0.w}, w1a[4] = {w1.x, w1.y, w1.z, w1.w}, w2a[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float o = fmaxf(fmaf(acc[4 * g + q], inv, bb[q]), 0.0f);
          v[4 * g + q] = o;
          o0 = fmaf(o, w0a[q], o0); o1 = fmaf(o, w1a[q], o1); o2 = fmaf(o, w2a[q], o2);
        }
      }
      store_tile_nt(c1blk + t * 1024, lane, v);
    }
  }
  if (ACTIVE) {
    o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64); o2 += __shfl_xor(o2, 32, 64);
    if (live && h == 0) {
      const float* bc2 = bl + 6 * H;
      stg4(a.alb + n * 4, make_float4(sigmoidf_(o0 + bc2[0]), sigmoidf_(o1 + bc2[1]), sigmoidf_(o2 + bc2[2]), 0.0f));
    }
  }
}

__global__ __launch_bounds__(THREADS, 2) void field_colour_fwd_kernel(const ColFwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING + (6 * H + 4 + 32) * 4];
  float* bl = reinterpret_cast<float*>(smem + RING);  // b2f | bc0 | bc1 | wc2[0..2] | bc2
  float* sl = bl + 6 * H + 4;                               // 3 NT tile scales
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < H; i += THREADS) {
    bl[i] = a.net.b2f[i]; bl[H + i] = a.net.bc0[i]; bl[2 * H + i] = a.net.bc1[i];
    for (int k = 0; k < 3; ++k) bl[(3 + k) * H + i] = a.net.wc2[(long)k * a.net.ldc2 + i];
  }
  if (tid < 3) bl[6 * H + tid] = a.net.bc2[tid];
  for (int i = tid; i < 3 * NT; i += THREADS) sl[i] = a.scales[i];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws_setup(ws, a.stream, smem, wave, lane, a.total_groups);
  const long G = gridDim.x;
  for (long base = 0; base < a.n_tiles; base += WAVES * G) {
    const long tile = base + wave * G + blockIdx.x;
    if (tile < a.n_tiles) col_fwd_tile<true>(a, ws, bl, sl, tile, lane);
    else col_fwd_tile<false>(a, ws, bl, sl, 0, lane);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// colour path backward, 32 points per wave.  Stream: Wc1^T (NT tiles), Wc0^T (10 tiles: rows = the 300 input columns), W2f^T (NT tiles);
// every K = H.
struct ColBwdArgs {
  nsky_field_net net;
  const unsigned char* stream; const float* scales; int total_groups;
  int N, n_tiles;
  const float* g_alb;          // [N, 3]
  const float* alb;            // [N, 4]
  const float* c0; const float* c1;
  float* dpc2;                 // [N, 4]: gradient of the output layer's pre-activation (weights of its weight-gradient column sums)
  float* dpc1; float* dpc0;    // native [ceil32(N), H]
  float* dfeat;                // native [ceil32(N), H]
  float* dxpe;                 // [N, 40]: gradient of the x / PE columns of the encode row (value rows)
  float* da1v;                 // native [ceil32(N), H]
  float* gmax;                 // [3]: max |dpc1|, |dpc0|, |dfeat| (caller zero-fills)
};

template <bool ACTIVE>
__device__ __forceinline__ void col_bwd_tile(const ColBwdArgs& a, WStream& ws, const float* wl, const float* sl, long tile, int lane) {
  const int c = lane & 31, h = lane >> 5;
  const long n = tile * 32 + c;
  const bool live = ACTIVE && n < a.N;
  const long nc = n < a.N ? n : a.N - 1;
  float* d1blk = a.dpc1 + tile * NT * 1024;
  float* d0blk = a.dpc0 + tile * NT * 1024;
  float* dfblk = a.dfeat + tile * NT * 1024;
  float* dablk = a.da1v + tile * NT * 1024;
  f16x8 ph[KS], pl[KS];
  float p_inv = 1.0f, m = 0.0f;
  if (ACTIVE) {
    float dp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float al = a.alb[nc * 4 + k];
      dp[k] = live ? a.g_alb[nc * 3 + k] * al * (1.0f - al) : 0.0f;
    }
    if (live && h == 0) stg4(a.dpc2 + n * 4, make_float4(dp[0], dp[1], dp[2], 0.0f));
    float cvn[16];
    load_tile(a.c1 + (tile * NT) * 1024, lane, cvn);
#pragma unroll 2
    for (int t = 0; t < NT; ++t) {
      float cv[16], dv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cv[r] = cvn[r];
      if (t + 1 < NT) load_tile(a.c1 + (tile * NT + t + 1) * 1024, lane, cvn);  // one tile ahead
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * t + 8 * g + 4 * h;
        const float4 w0 = *reinterpret_cast<const float4*>(wl + fo);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + H + fo);
        const float4 w2 = *reinterpret_cast<const float4*>(wl + 2 * H + fo);
        const float w0a[4] = {w0.x, w0.y, w0.z, w0.w}, w1a[4] = {w1.x, w1.y, w1.z, w1.w}, w2a[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * g + q;
          const float d = dp[0] * w0a[q] + dp[1] * w1a[q] + dp[2] * w2a[q];
          dv[r] = cv[r] > 0.0f ? d : 0.0f;
          m = fmaxf(m, fabsf(dv[r]));
        }
      }
      store_tile(d1blk + t * 1024, lane, dv);
    }
    publish_max(a.gmax, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p_inv = planes_from_tiles<NT>(d1blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- dpc0 = (Wc1^T dpc1) relu'(c0)
  m = 0.0f;
  for (int u = 0; u < NT; ++u) {
    f32x16 acc;
    f32x4 cq[4];
    if (ACTIVE) {  // the saved c0 tile, requested ahead of the product with hidden loads (counted wait behind it)
#pragma unroll
      for (int g = 0; g < 4; ++g) cq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) hidden_load4(cq[g], a.c0 + (tile * NT + u) * 1024 + g * 256 + lane * 4);
    }
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      hidden_wait<((KS + GSLABS - 1) / GSLABS) * PW>(cq);
      const float inv = p_inv * sl[u];
      float dv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dv[r] = cq[r >> 2][r & 3] > 0.0f ? acc[r] * inv : 0.0f;
        m = fmaxf(m, fabsf(dv[r]));
      }
      store_tile(d0blk + u * 1024, lane, dv);
    }
  }
  if (ACTIVE) {
    publish_max(a.gmax + 1, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's dpc0 stores have left before it reads them back
    p_inv = planes_from_tiles<NT>(d0blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- d(colour-net input) = Wc0^T dpc0: tiles 0..7 = dfeat, tiles 8, 9 = the x / PE columns
  m = 0.0f;
  for (int u = 0; u < NT + 2; ++u) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      const float inv = p_inv * sl[NT + u];
      if (u < NT) {
        float dv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dv[r] = acc[r] * inv;
          m = fmaxf(m, fabsf(dv[r]));
        }
        store_tile(dfblk + u * 1024, lane, dv);
      } else if (live && a.dxpe) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * u + 8 * g + 4 * h - XPE_COL0;
          if (col >= 0 && col < 40) stg4(a.dxpe + n * 40 + col, make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv));
        }
      }
    }
  }
  if (ACTIVE) {
    publish_max(a.gmax + 2, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p_inv = planes_from_tiles<NT>(dfblk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- d a1 (value rows) = W2f^T dfeat
  for (int u = 0; u < NT; ++u) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      const float inv = p_inv * sl[2 * NT + 2 + u];
      float dv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dv[r] = acc[r] * inv;
      store_tile(dablk + u * 1024, lane, dv);
    }
  }
}

__global__ __launch_bounds__(THREADS, 2) void field_colour_bwd_kernel(const ColBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING + (3 * H + 32) * 4];
  float* wl = reinterpret_cast<float*>(smem + RING);  // wc2[0..2]
  float* sl = wl + 3 * H;                                   // 3 NT + 2 tile scales
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < H; i += THREADS)
    for (int k = 0; k < 3; ++k) wl[k * H + i] = a.net.wc2[(long)k * a.net.ldc2 + i];
  for (int i = tid; i < 3 * NT + 2; i += THREADS) sl[i] = a.scales[i];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws_setup(ws, a.stream, smem, wave, lane, a.total_groups);
  const long G = gridDim.x;
  for (long base = 0; base < a.n_tiles; base += WAVES * G) {
    const long tile = base + wave * G + blockIdx.x;
    if (tile < a.n_tiles) col_bwd_tile<true>(a, ws, wl, sl, tile, lane);
    else col_bwd_tile<false>(a, ws, wl, sl, 0, lane);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// geometry network backward, quad layout.  Stream: W1^T (NT tiles), W0^T (ceil(in_dim / 32) tiles); every K = H.
struct GeoBwdArgs {
  nsky_field_net net;
  const unsigned char* stream; const float* scales; int total_groups;
  int N, n_tiles;
  const float* g_sdf;          // [N] or null
  const float* g_grad;         // [N, 3] or null
  const float* da1v;           // native [ceil32(N), H] or null: gradient of the value rows of a1 from the colour path
  const float* dxpe;           // [N, 40] or null: added to the x / PE columns of the value rows of dET
  const float* a0q; const float* a1q;
  float* d1q; float* d0q;      // quad-native [ceil32(4 N), H]
  float* dET; int ldE;         // stacked [4 N, ldE]
  float* gmax;                 // [2]: max |d1q|, |d0q| (caller zero-fills)
};

// reverse of one forward-mode softplus layer on a tile.  av: the saved layer outputs of this lane's column (value lane: a, tangent lanes:
// ta_k); gin: the incoming gradient of those outputs -> dv: the pre-activation gradients (value lane: dz = da s + sum_k dta_k ta_k
// beta (1 - s), tangent lanes: dtz_k = dta_k s), s = sigmoid(beta z) = 1 - exp(-beta a) recovered from the value row's saved output:
// the exponentials are taken by the four lanes of the quad, four features each, and handed back to every lane.
__device__ __forceinline__ void softplus_rev_tile(const float (&av)[16], const float (&gin)[16], float beta, int j, float (&dv)[16]) {
  float a4[4], e4[4], E[16];
  quad_scatter_value(av, j, a4);
#pragma unroll
  for (int q = 0; q < 4; ++q) e4[q] = __expf(-beta * a4[q]);  // 1 - sigmoid(beta z)
  quad_gather_all(e4, E);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float cross = quad_sum(j == 0 ? 0.0f : gin[r] * av[r] * (beta * E[r]));
    const float s = 1.0f - E[r];
    dv[r] = j == 0 ? fmaf(gin[r], s, cross) : gin[r] * s;
  }
}

template <bool ACTIVE>
__device__ __forceinline__ void geo_bwd_tile(const GeoBwdArgs& a, WStream& ws, const float* wl, const float* sl, long tile, int lane, int ct) {
  const int c = lane & 31, h = lane >> 5, p = c >> 2, j = c & 3;
  const long n = tile * 8 + p;
  const bool live = ACTIVE && n < a.N;
  const long nc = n < a.N ? n : a.N - 1;
  const float beta = a.net.beta;
  float* d1blk = a.d1q + tile * NT * 1024;
  float* d0blk = a.d0q + tile * NT * 1024;
  f16x8 ph[KS], pl[KS];
  float p_inv = 1.0f, m = 0.0f;
  if (ACTIVE) {
    // gradient of the sdf row's output: the value lane carries g_sdf, tangent lane k carries g_grad[k]
    float gq = 0.0f;
    if (live) gq = j == 0 ? (a.g_sdf ? a.g_sdf[nc] : 0.0f) : (a.g_grad ? a.g_grad[nc * 3 + (j - 1)] : 0.0f);
    const float* dap = a.da1v ? a.da1v + ((nc >> 5) * NT) * 1024 + ((int)(nc & 31) + 32 * h) * 4 : nullptr;
    float avn[16];
    float4 ddn[4];
    auto request = [&](int t) {  // tile t's saved outputs and (value lane) the colour path's gradient: one tile ahead of their use
      load_tile(a.a1q + (tile * NT + t) * 1024, lane, avn);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ddn[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dap && j == 0) ddn[g] = ldg4(dap + t * 1024 + g * 256);
      }
    };
    request(0);
#pragma unroll 2
    for (int t = 0; t < NT; ++t) {
      float av[16], dv[16], gin[16];
      float4 dd4[4];
#pragma unroll
      for (int r = 0; r < 16; ++r) av[r] = avn[r];
#pragma unroll
      for (int g = 0; g < 4; ++g) dd4[g] = ddn[g];
      if (t + 1 < NT) request(t + 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w4 = *reinterpret_cast<const float4*>(wl + 32 * t + 8 * g + 4 * h);
        const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
        const float dd[4] = {dd4[g].x, dd4[g].y, dd4[g].z, dd4[g].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) gin[4 * g + q] = fmaf(gq, ww[q], dd[q]);
      }
      softplus_rev_tile(av, gin, beta, j, dv);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(live ? dv[r] : 0.0f));
      store_tile(d1blk + t * 1024, lane, dv);
    }
    publish_max(a.gmax, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p_inv = planes_from_tiles<NT>(d1blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- layer 0: incoming gradient = W1^T D1
  m = 0.0f;
  for (int u = 0; u < NT; ++u) {
    f32x16 acc;
    f32x4 aq[4];
    if (ACTIVE) {  // the tile's saved outputs, requested ahead of its product with loads the compiler does not see (a visible load
      // would be waited for with vmcnt(0): the whole DMA ring); only the product's own DMA pieces are younger
#pragma unroll
      for (int g = 0; g < 4; ++g) aq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) hidden_load4(aq[g], a.a0q + (tile * NT + u) * 1024 + g * 256 + lane * 4);
    }
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE) {
      hidden_wait<((KS + GSLABS - 1) / GSLABS) * PW>(aq);
      const float inv = p_inv * sl[u];
      float av[16], dv[16], gin[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        av[r] = aq[r >> 2][r & 3];
        gin[r] = acc[r] * inv;
      }
      softplus_rev_tile(av, gin, beta, j, dv);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(live ? dv[r] : 0.0f));
      store_tile(d0blk + u * 1024, lane, dv);
    }
  }
  if (ACTIVE) {
    publish_max(a.gmax + 1, m, live, true, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's d0 stores have left before it reads them back
    p_inv = planes_from_tiles<NT>(d0blk, lane, m, ph, pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- d(encode rows) = W0^T D0, stacked layout
  for (int u = 0; u < ct; ++u) {
    f32x16 acc;
    prod<KS, ACTIVE>(ws, ph, pl, acc);
    if (ACTIVE && live && a.dET) {
      const float inv = p_inv * sl[NT + u];
      float* out = a.dET + ((long)j * a.N + n) * a.ldE;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * u + 8 * g + 4 * h;
        if (fo < a.ldE) {
          float4 o = make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
          if (a.dxpe && j == 0 && fo < 40) {
            const float4 e = ldg4(a.dxpe + n * 40 + fo);
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
          }
          stg4(out + fo, o);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

__global__ __launch_bounds__(THREADS, 2) void field_geo_bwd_kernel(const GeoBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING + (H + 32) * 4];
  float* wl = reinterpret_cast<float*>(smem + RING);  // w_sdf
  float* sl = wl + H;                                       // NT + ct tile scales
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = (a.net.in_dim + 31) / 32;
  for (int i = tid; i < H; i += THREADS) wl[i] = a.net.w_sdf[i];
  for (int i = tid; i < NT + ct; i += THREADS) sl[i] = a.scales[i];
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WStream ws;
  ws_setup(ws, a.stream, smem, wave, lane, a.total_groups);
  const long G = gridDim.x;
  for (long base = 0; base < a.n_tiles; base += WAVES * G) {
    const long tile = base + wave * G + blockIdx.x;
    if (tile < a.n_tiles) geo_bwd_tile<true>(a, ws, wl, sl, tile, lane, ct);
    else geo_bwd_tile<false>(a, ws, wl, sl, 0, lane, ct);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// out[o][f] += sum_rows w[row][o] X[row][f] over a tile-native X [rows, 32 nt] (o < n_out <= 4): the weight gradients of the
// field's narrow output layers (the sdf row: 1 output over the quad-native a1; the albedo rows: 3 outputs over c1).
// quad weights: w(row) = g_sdf[row >> 2] for row & 3 == 0, g_grad[row >> 2][(row & 3) - 1] otherwise (one output);
// bias[o] += sum_rows w[row][o] (quad form: value rows only).
struct ColsumArgs {
  const float* X; int nt; int rows;
  const float* w4; int n_out;                  // [rows, 4] row-major, or null for the quad form
  const float* g_sdf; const float* g_grad;     // quad form ([rows / 4], [rows / 4, 3]; either may be null)
  float* out; int ldo; float* bias;
  int blocks_per_wg;
};

__global__ __launch_bounds__(512) void native_weighted_colsum_kernel(const ColsumArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int row_blocks = (a.rows + 31) / 32;
  const int b0 = blockIdx.x * a.blocks_per_wg, b1 = min(row_blocks, b0 + a.blocks_per_wg);
  for (int t = wave; t < a.nt; t += 8) {
    float acc[4][16];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[o][r] = 0.0f;
    for (int bb = b0; bb < b1; bb += 4) {  // four row blocks in flight per wave
      float v[4][16], w[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = bb + u;
        const long row = (long)b * 32 + c;
        w[u][0] = w[u][1] = w[u][2] = w[u][3] = 0.0f;
        if (b < b1) {
          load_tile(a.X + ((long)b * a.nt + t) * 1024, lane, v[u]);
          if (row < a.rows) {
            if (a.w4) {
              const float4 q = ldg4(a.w4 + row * 4);
              w[u][0] = q.x; w[u][1] = q.y; w[u][2] = q.z; w[u][3] = q.w;
            } else {
              const int j = (int)(row & 3);
              const long n = row >> 2;
              w[u][0] = j == 0 ? (a.g_sdf ? a.g_sdf[n] : 0.0f) : (a.g_grad ? a.g_grad[n * 3 + (j - 1)] : 0.0f);
              if (j == 0) bs[0] += w[u][0];
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[u][r] = 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (a.w4) {
#pragma unroll
          for (int o = 0; o < 4; ++o) bs[o] += w[u][o];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < a.n_out)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = fmaf(w[u][o], v[u][r], acc[o][r]);
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < a.n_out) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[o][r];
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          if (c == 0 && v != 0.0f) atomicAdd(a.out + (long)o * a.ldo + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3), v);
        }
        if (a.bias && t == 0) {
          float v = h == 0 ? bs[o] : 0.0f;
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          if (lane == 0 && v != 0.0f) atomicAdd(a.bias + o, v);
        }
      }
  }
}

int check_field_net(const nsky_field_net* n, const char* who) {
  NSKY_CHECK_ARG(n, "%s: null network", who);
  NSKY_CHECK_ARG(n->in_dim >= 4 && n->in_dim <= 80 && n->in_dim % 4 == 0 && n->npe >= 0 && n->npe <= 39 && n->beta > 0.0f,
                 "%s: in_dim %d (4..80, multiple of 4), npe %d (<= 39), beta %g", who, n->in_dim, n->npe, n->beta);
  return NSKY_OK;
}

int persistent_grid(int n_tiles) {
  static int cus = [] {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }();
  const int wgs = (n_tiles + WAVES - 1) / WAVES;
  return wgs < cus * WGS_PER_CU ? wgs : cus * WGS_PER_CU;
}

}  // namespace

extern "C" int nsky_chain_stream_layout(const nsky_chain_layer* layers, int32_t n_layers, int64_t* stream_bytes, int32_t* n_tiles, int32_t* n_groups) {
  NSKY_CHECK_ARG(layers && n_layers >= 1 && n_layers <= NSKY_CHAIN_MAX_LAYERS, "nsky_chain_stream_layout: 1..%d layers", NSKY_CHAIN_MAX_LAYERS);
  for (int i = 0; i < n_layers; ++i)
    NSKY_CHECK_ARG(layers[i].rows >= 1 && layers[i].K >= 1 && layers[i].K <= PACK_KMAX, "nsky_chain_stream_layout: layer %d: rows %d, K %d (<= %d)", i,
                   layers[i].rows, layers[i].K, PACK_KMAX);
  long groups; int tiles;
  chain_layout(layers, n_layers, groups, tiles);
  if (stream_bytes) *stream_bytes = (groups + RING_GROUPS + 2) * (int64_t)GROUP;
  if (n_tiles) *n_tiles = tiles;
  if (n_groups) *n_groups = (int32_t)groups;
  return NSKY_OK;
}

extern "C" int nsky_chain_pack(const nsky_chain_layer* layers, int32_t n_layers, void* stream_buf, float* scales, nsky_stream_t stream) {
  if (int rc = nsky_chain_stream_layout(layers, n_layers, nullptr, nullptr, nullptr)) return rc;
  NSKY_CHECK_ARG(stream_buf && scales && ((uintptr_t)stream_buf % 16) == 0, "nsky_chain_pack: null / unaligned buffer");
  ChainLayers L;
  L.n = n_layers;
  for (int i = 0; i < n_layers; ++i) {
    L.l[i] = layers[i];
    NSKY_CHECK_ARG(L.l[i].W && L.l[i].ld >= (L.l[i].transposed ? L.l[i].rows : L.l[i].K), "nsky_chain_pack: layer %d: weights / ld", i);
  }
  long groups; int tiles;
  chain_layout(layers, n_layers, groups, tiles);
  hipLaunchKernelGGL(chain_pack_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, L, (unsigned char*)stream_buf, scales);
  NSKY_CHECK_LAUNCH("nsky_chain_pack");
  return NSKY_OK;
}

#define NSKY_AL16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int nsky_field_geo_fwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, const float* ET,
                                  int32_t ldE, int32_t N, float* a0q, float* a1q, float* Eq, float* a1max, float* sdf, float* grad, float* qmax,
                                  nsky_stream_t stream) {
  if (int rc = check_field_net(net, "nsky_field_geo_fwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && scales && ET && a0q && a1q && sdf && grad && N > 0 && net->b0 && net->b1 && net->w_sdf, "nsky_field_geo_fwd: null operand / empty batch");
  NSKY_CHECK_ARG(ldE % 4 == 0 && ldE >= net->in_dim && NSKY_AL16(ET) && NSKY_AL16(a0q) && NSKY_AL16(a1q) && NSKY_AL16(stream_buf) && (!Eq || NSKY_AL16(Eq)),
                 "nsky_field_geo_fwd: alignment / ldE");
  NSKY_CHECK_ARG(ksteps_of(net->in_dim) == 5, "nsky_field_geo_fwd: encode rows of 68..80 columns expected, got %d", net->in_dim);
  NSKY_CHECK_ARG(total_groups == NT * groups_of(net->in_dim) + NT * groups_of(H), "nsky_field_geo_fwd: stream of %d groups does not hold W0, W1", total_groups);
  GeoFwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.scales = scales; a.total_groups = total_groups; a.ET = ET; a.ldE = ldE; a.N = N;
  a.n_tiles = ceil_div(N, 8); a.a0q = a0q; a.a1q = a1q; a.Eq = Eq; a.a1max = a1max; a.sdf = sdf; a.grad = grad; a.qmax = qmax;
  const dim3 grid(persistent_grid(a.n_tiles));
  hipLaunchKernelGGL((field_geo_fwd_kernel<5>), grid, dim3(THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_field_geo_fwd");
  return NSKY_OK;
}

extern "C" int nsky_field_colour_fwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, const float* ET,
                                     int32_t ldE, int32_t N, const float* a1q, const float* a1max, float* a1v, float* feat, float* xpe, float* c0, float* c1,
                                     float* alb, nsky_stream_t stream) {
  if (int rc = check_field_net(net, "nsky_field_colour_fwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && scales && ET && a1q && a1max && feat && xpe && c0 && c1 && alb && N > 0 && net->b2f && net->bc0 && net->bc1 && net->wc2 && net->bc2,
                 "nsky_field_colour_fwd: null operand / empty batch");
  NSKY_CHECK_ARG(ldE % 4 == 0 && ldE >= net->npe + 1 && NSKY_AL16(ET) && NSKY_AL16(a1q) && NSKY_AL16(feat) && NSKY_AL16(xpe) && NSKY_AL16(c0) && NSKY_AL16(c1) && NSKY_AL16(alb) &&
                     (!a1v || NSKY_AL16(a1v)) && net->ldc2 >= H, "nsky_field_colour_fwd: alignment / leading dimensions");
  NSKY_CHECK_ARG(total_groups == NT * (2 * groups_of(H) + groups_of(300)), "nsky_field_colour_fwd: stream of %d groups does not hold W2f, Wc0, Wc1", total_groups);
  ColFwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.scales = scales; a.total_groups = total_groups; a.ET = ET; a.ldE = ldE; a.N = N;
  a.n_tiles = ceil_div(N, 32); a.a1q = a1q; a.a1max = a1max; a.a1v = a1v; a.feat = feat; a.xpe = xpe; a.c0 = c0; a.c1 = c1; a.alb = alb;
  hipLaunchKernelGGL(field_colour_fwd_kernel, dim3(persistent_grid(a.n_tiles)), dim3(THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_field_colour_fwd");
  return NSKY_OK;
}

extern "C" int nsky_field_colour_bwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, int32_t N,
                                     const float* g_alb, const float* alb, const float* c0, const float* c1, float* dpc2, float* dpc1, float* dpc0,
                                     float* dfeat, float* dxpe, float* da1v, float* gmax, nsky_stream_t stream) {
  if (int rc = check_field_net(net, "nsky_field_colour_bwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && scales && g_alb && alb && c0 && c1 && dpc2 && dpc1 && dpc0 && dfeat && da1v && gmax && N > 0 && net->wc2,
                 "nsky_field_colour_bwd: null operand / empty batch");
  NSKY_CHECK_ARG(NSKY_AL16(alb) && NSKY_AL16(c0) && NSKY_AL16(c1) && NSKY_AL16(dpc2) && NSKY_AL16(dpc1) && NSKY_AL16(dpc0) && NSKY_AL16(dfeat) && NSKY_AL16(da1v) &&
                     (!dxpe || NSKY_AL16(dxpe)) && net->ldc2 >= H, "nsky_field_colour_bwd: alignment");
  NSKY_CHECK_ARG(total_groups == (2 * NT + 10) * groups_of(H), "nsky_field_colour_bwd: stream of %d groups does not hold Wc1^T, Wc0^T, W2f^T", total_groups);
  ColBwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.scales = scales; a.total_groups = total_groups; a.N = N; a.n_tiles = ceil_div(N, 32);
  a.g_alb = g_alb; a.alb = alb; a.c0 = c0; a.c1 = c1; a.dpc2 = dpc2; a.dpc1 = dpc1; a.dpc0 = dpc0; a.dfeat = dfeat; a.dxpe = dxpe; a.da1v = da1v; a.gmax = gmax;
  hipLaunchKernelGGL(field_colour_bwd_kernel, dim3(persistent_grid(a.n_tiles)), dim3(THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_field_colour_bwd");
  return NSKY_OK;
}

extern "C" int nsky_field_geo_bwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, int32_t N,
                                  const float* g_sdf, const float* g_grad, const float* da1v, const float* dxpe, const float* a0q, const float* a1q,
                                  float* d1q, float* d0q, float* dET, int32_t ldE, float* gmax, nsky_stream_t stream) {
  if (int rc = check_field_net(net, "nsky_field_geo_bwd")) return rc;
  NSKY_CHECK_ARG(stream_buf && scales && a0q && a1q && d1q && d0q && gmax && N > 0 && net->w_sdf, "nsky_field_geo_bwd: null operand / empty batch");
  NSKY_CHECK_ARG(NSKY_AL16(a0q) && NSKY_AL16(a1q) && NSKY_AL16(d1q) && NSKY_AL16(d0q) && (!da1v || NSKY_AL16(da1v)) && (!dxpe || NSKY_AL16(dxpe)) &&
                     (!dET || (NSKY_AL16(dET) && ldE % 4 == 0 && ldE >= net->in_dim)), "nsky_field_geo_bwd: alignment / ldE");
  const int ct = (net->in_dim + 31) / 32;
  NSKY_CHECK_ARG(total_groups == (NT + ct) * groups_of(H), "nsky_field_geo_bwd: stream of %d groups does not hold W1^T, W0^T", total_groups);
  GeoBwdArgs a;
  a.net = *net; a.stream = (const unsigned char*)stream_buf; a.scales = scales; a.total_groups = total_groups; a.N = N; a.n_tiles = ceil_div(N, 8);
  a.g_sdf = g_sdf; a.g_grad = g_grad; a.da1v = da1v; a.dxpe = dxpe; a.a0q = a0q; a.a1q = a1q; a.d1q = d1q; a.d0q = d0q; a.dET = dET; a.ldE = ldE; a.gmax = gmax;
  hipLaunchKernelGGL(field_geo_bwd_kernel, dim3(persistent_grid(a.n_tiles)), dim3(THREADS), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_field_geo_bwd");
  return NSKY_OK;
}

extern "C" int nsky_native_weighted_colsum(const float* X, int32_t nt, int32_t rows, const float* w4, int32_t n_out, const float* g_sdf,
                                           const float* g_grad, float* out, int32_t ldo, float* bias, nsky_stream_t stream) {
  NSKY_CHECK_ARG(X && out && nt >= 1 && rows >= 0 && NSKY_AL16(X) && ldo >= 32 * nt, "nsky_native_weighted_colsum: bad argument");
  NSKY_CHECK_ARG(w4 ? (n_out >= 1 && n_out <= 4 && NSKY_AL16(w4)) : (n_out == 1 && rows % 4 == 0), "nsky_native_weighted_colsum: weights (n_out %d)", n_out);
  if (rows == 0) return NSKY_OK;
  ColsumArgs a;
  a.X = X; a.nt = nt; a.rows = rows; a.w4 = w4; a.n_out = n_out; a.g_sdf = g_sdf; a.g_grad = g_grad; a.out = out; a.ldo = ldo; a.bias = bias;
  const int row_blocks = ceil_div(rows, 32);
  a.blocks_per_wg = ceil_div(row_blocks, 256);  // one workgroup per CU: every one ends with 512 n_out atomics onto the same 256 n_out addresses
  hipLaunchKernelGGL(native_weighted_colsum_kernel, dim3(ceil_div(row_blocks, a.blocks_per_wg)), dim3(512), 0, (hipStream_t)stream, a);
  NSKY_CHECK_LAUNCH("nsky_native_weighted_colsum");
  return NSKY_OK;
}
