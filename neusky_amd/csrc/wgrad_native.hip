// Weight gradient of a dense layer straight over two TILE-NATIVE matrices (include/neusky_hip.h: 32 x 32 blocks in
// v_mfma_f32_32x32x16 accumulator order, i.e. batch row on the lane, features in the registers):
//
//     dW[n, k] += sum_rows dZ[row, n] * X[row, k]        db[n] += sum_rows dZ[row, n]
//
// the sum runs over the batch (~2.6e5 rows for the DDF's FiLM-SIREN chain), both operands are streamed ONCE, and the
// kernel is HBM bound: 2 x 4 B x (N + K) per row against 3 x 2 N K fp16 MFMA flops.
//
// Products are fp32-grade from three v_mfma_f32_32x32x16_f16 into ONE accumulator: every element is multiplied by a
// power of two (dZ: from the largest |dZ| the backward kernels publish; X: a constant) and split into fp16 hi + fp16
// residual; hi*hi + hi*lo + lo*hi is accumulated in fp32 and the scale is undone on the way out.
//
// The contraction index is the batch row = the LANE of the native layout, so each operand needs one transpose.  A block
// (32 rows x 32 features, 4 KB) is loaded as four lane-linear global_load_dwordx4 (one 16-byte unit = 4 features of one row
// per lane), split in registers and written as two fp16 images [row][32 features] (64-byte rows, 16-byte chunks XOR-swizzled
// by (row >> 1) & 3: ds_write_b64 and the transposed read are both conflict free); the MFMA fragments (8 consecutive rows
// of one feature) come back through ds_read_b64_tr_b16, gfx950's transposing LDS read.
//
// Workgroup = 8 waves (4 x 2, two per SIMD) with 2 x 4 accumulator tiles each: a 256 x 256 block of dW per workgroup, 128
// accumulator registers per lane, one workgroup per CU (128 KB of images, two stages); or 4 waves (2 x 2) with 2 x 2 tiles:
// 128 x 128, two workgroups per CU.  Wave w stages tile w of each operand.
// The batch is split over the grid (split-K) and reduced with contiguous float atomics into dW / db (the callers'
// accumulators: zero-filled slabs or the parameters' own .grad).  One barrier per 32 rows; the loads of the
// next two row blocks are in flight while a block is multiplied (see the pipeline comment in the kernel).
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WgradProblem {
  const float* A;   // dZ, tile native, nnt_a tiles per row block
  const float* B;   // X, tile native, nnt_b tiles per row block
  float* dW;        // [32 nnt_a][ldw]
  float* db;        // [32 nnt_a] or null
  const float* a_scale_max;  // device scalar: largest |dZ| (null: dZ is taken as it is)
  float b_scale;    // power of two applied to X
  const float* b_scale_max;  // device scalar: largest |X| (null: the constant b_scale) -- operands that carry input tangents have no bound
  int nnt_a, nnt_b, ldw;   // tiles of 32 features each operand is walked in (row-major: ceil(width / 32) rounded up to the block)
  int tile0;        // first output block of this problem in the launch's block list
  int lda, ldb;     // row-major operands: leading dimensions (floats)
  int wa, wb;       // features that exist (dW is [wa, wb]); native: 32 nnt
  int bias_rows;    // db sums the first bias_rows rows only
  int bias_mod;     // > 1: db sums the rows with row % bias_mod == 0 only (quad-native gradients: the value rows)
};
struct WgradArgs {
  WgradProblem p[NSKY_WGRAD_MAX_PROBLEMS];
  int n_problems, rows, row_blocks, blocks_per_split, n_tiles;  // n_tiles: output blocks of all problems together
};

__device__ __forceinline__ float pow2_scale(float m, float& inv) {  // m s < 2^15
  if (!(m > 0.0f) || !(m < 3.0e38f)) { inv = 1.0f; return 1.0f; }
  int e;
  (void)frexpf(m, &e);
  e = max(-100, min(100, e));
  inv = ldexpf(1.0f, e - 15);
  return ldexpf(1.0f, 15 - e);
}

// consecutive hardware workgroup ids go round-robin over the 8 XCDs: hand every XCD one contiguous run of logical ids, so
// the workgroups of one batch split (which read the same rows of X, and of dZ where the layer is wider than one block)
// share an L2
__device__ __forceinline__ int xcd_contiguous(int id, int total) {
  const int xcd = id & 7, idx = id >> 3, per = total >> 3, rem = total & 7;
  return xcd * per + min(xcd, rem) + idx;
}

template <bool BF>
__device__ __forceinline__ f32x16 mma(f16x8 x, f16x8 y, f32x16 acc) {
  typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
  if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, x), __builtin_bit_cast(b16x8, y), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
}

// RM: both operands are ROW-MAJOR [rows, ld] matrices (the SDF / colour field's stacked value + tangent rows) and the products are
// the 2-term bf16 split (hi hi + hi lo + lo hi, no pre-scaling: gradients keep bf16's exponent range; 2^-16 per product) instead of
// the pre-scaled fp16 one; a load instruction then covers 8 rows x 128 B, lane = (row 8 u + lane / 8, features 4 (lane % 8) ..).
template <int WM, int WN, int TM, int TN, bool RM>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 8 ? 1 : 2) void wgrad_native_kernel(WgradArgs a) {
  constexpr int NW = WM * WN;       // waves; also the 32-feature tiles per operand and workgroup (wave w stages tile w of each)
  constexpr int NT = NW;
  static_assert(WM * TM == NT && WN * TN == NT, "square output block");
  constexpr int OPB = NT * 4096;    // bytes of one operand's images in a stage (tile: hi 2 KB, lo 2 KB)
  constexpr int STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
  const int gtile = logical % a.n_tiles, split = logical / a.n_tiles;
  int pi = 0;
  while (pi + 1 < a.n_problems && gtile >= a.p[pi + 1].tile0) ++pi;
  const WgradProblem& P = a.p[pi];
  const int tile = gtile - P.tile0;
  const int kblocks = P.nnt_b / NT;
  const int nb = tile / kblocks, kb = tile % kblocks;
  const int r_beg = split * a.blocks_per_split;
  const int r_end = min(a.row_blocks, r_beg + a.blocks_per_split);
  if (r_beg >= r_end) return;

  float a_inv = 1.0f;
  const float a_scale = P.a_scale_max ? pow2_scale(*P.a_scale_max, a_inv) : 1.0f;
  float b_unused;
  const float b_scale = P.b_scale_max ? pow2_scale(*P.b_scale_max, b_unused) : P.b_scale;
  const bool clamp_b = P.b_scale_max == nullptr;

  // ---- staging: this wave's block of each operand.  Native: lane = (row c, half hh), unit u = features 8 u + 4 hh .. + 3 of row c.
  // Row-major: lane = (row 8 u + lane / 8 of the block, features 4 (lane % 8) .. + 3 of the tile)
  const int c = lane & 31, hh = lane >> 5;
  const int rq = lane >> 3, fq = lane & 7;
  const int fa = (nb * NT + wave) * 32 + 4 * fq, fb = (kb * NT + wave) * 32 + 4 * fq;  // RM: first feature of this lane's quad
  const bool fa_ok = fa < P.wa, fb_ok = fb < P.wb;
  const float* pa = RM ? P.A + (long)r_beg * 32 * P.lda + min(fa, max(P.wa - 4, 0))
                       : P.A + ((long)r_beg * P.nnt_a + nb * NT + wave) * 1024 + lane * 4;
  const float* pb = RM ? P.B + (long)r_beg * 32 * P.ldb + min(fb, max(P.wb - 4, 0))
                       : P.B + ((long)r_beg * P.nnt_b + kb * NT + wave) * 1024 + lane * 4;
  const long a_step = RM ? 32l * P.lda : (long)P.nnt_a * 1024, b_step = RM ? 32l * P.ldb : (long)P.nnt_b * 1024;
  int rm_row0 = 0;  // RM: first row of the block pa2 / pb2 point at (a unit's row is clamped to the last existing row; dropped when staged)
  // eight staging units per row block and wave: unit 2 u = 16 bytes of dZ (features 8 u + 4 hh .. + 3 of row c), 2 u + 1 = of X
  f32x4 raw[8];
  const float* pa2 = pa;  // block the next reload of a unit reads
  const float* pb2 = pb;
  auto reload = [&](int uu) {
    if (RM) {
      const int row = min(rm_row0 + 8 * (uu >> 1) + rq, a.rows - 1) - rm_row0;
      raw[uu] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(((uu & 1) ? pb2 + (long)row * P.ldb : pa2 + (long)row * P.lda)));
    } else {
      raw[uu] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(((uu & 1) ? pb2 : pa2) + (uu >> 1) * 256));
    }
  };
  float bias[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[u][i] = 0.0f;
  const bool want_bias = P.db != nullptr && kb == 0;
  // image byte offset of this lane's 8-byte slot of unit u: row c, chunk u ^ ((c >> 1) & 3), half hh
  // (RM: row 8 u + rq, chunk (fq >> 1) ^ ((row >> 1) & 3), half fq & 1)
  const int w_base = wave * 4096 + c * 64 + 8 * hh;
  const int w_swz = (c >> 1) & 3;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
  // live_row: the unit's row exists; bias_on: it also counts for the bias sum (native: one flag per block and lane)
  auto stage_unit = [&](int uu, int row0, bool row_block_live, unsigned char* stage) {  // split unit uu, write it to the images of `stage`
    const int op = uu & 1, u = uu >> 1;
    const float s = op == 0 ? a_scale : b_scale;
    const int row = row0 + (RM ? 8 * u + rq : c);
    const bool live = row_block_live && row < a.rows && (!RM || (op == 0 ? fa_ok : fb_ok));
    const bool bias_on = op == 0 && row < P.bias_rows && (P.bias_mod <= 1 || (row % P.bias_mod) == 0);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = live ? raw[uu][i] * s : 0.0f;
      // an X operand under a CONSTANT scale is clamped to the fp16 range: beyond the scale's reach it saturates instead of turning the
      // whole gradient into NaN (operands scaled by their own maximum stay below 2^15)
      if (op == 1 && clamp_b) v[i] = __builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f);
      if (op == 0) bias[u][i] += bias_on ? v[i] : 0.0f;
    }
    unsigned char* img;
    if (RM) {
      const int r = 8 * u + rq;
      img = stage + op * OPB + wave * 4096 + r * 64 + 16 * ((fq >> 1) ^ ((r >> 1) & 3)) + 8 * (fq & 1);
      b16x4 h4, l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) { h4[i] = (__bf16)v[i]; l4[i] = (__bf16)(v[i] - (float)h4[i]); }
      *reinterpret_cast<b16x4*>(img) = h4;
      *reinterpret_cast<b16x4*>(img + 2048) = l4;
    } else {
      img = stage + op * OPB + w_base + 16 * (u ^ w_swz);
      f16x4 h4, l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) { h4[i] = (_Float16)v[i]; l4[i] = (_Float16)(v[i] - (float)h4[i]); }
      *reinterpret_cast<f16x4*>(img) = h4;
      *reinterpret_cast<f16x4*>(img + 2048) = l4;
    }
  };

  // ---- fragments: lane (r = lane & 31, h = lane >> 5) of a 16-row k-step gets rows 8 h + 0..7 of feature r from two transposed
  // reads; it SUPPLIES the address of row 8 h + 4 rd + q, features 16 gsel + 4 p .. + 3 (q = i >> 2, p = i & 3, i = lane & 15)
  const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, gsel = (lane >> 4) & 1, h = lane >> 5;
  int t_off[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int row = 8 * h + 4 * rd + q;
    t_off[rd] = row * 64 + 16 * ((2 * gsel + (p >> 1)) ^ ((row >> 1) & 3)) + 8 * (p & 1);
  }
  auto frag = [&](const unsigned char* img, int ks) -> f16x8 {  // img: one plane of one tile
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + ks * 1024 + t_off[0]));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + ks * 1024 + t_off[1]));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(f16x8, v);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Software pipeline over 32-row blocks.  While block r is multiplied out of image stage `cur`, block r + 1 sits in the eight
  // staging units (its loads were issued one block earlier) and moves into the other stage ONE UNIT PER (k-step, X tile) POINT of
  // the MFMA stream; the unit's registers are refilled at once with block r + 2, so eight 1 KB loads per wave stay in flight
  // the whole time and the split's VALU work rides between the MFMAs instead of after them.
  rm_row0 = r_beg * 32;
#pragma unroll
  for (int uu = 0; uu < 8; ++uu) reload(uu);
  {
    if (r_beg + 1 < r_end) { pa2 += a_step; pb2 += b_step; rm_row0 += 32; }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      stage_unit(uu, r_beg * 32, true, smem);
      reload(uu);
    }
    if (r_beg + 2 < r_end) { pa2 += a_step; pb2 += b_step; rm_row0 += 32; }
  }
  static_assert(TN == 4 || TN == 2, "one staging unit per (k-step, X tile) point, or two");
  constexpr int UPP = 8 / (2 * TN);  // staging units per point
  int cur = 0;
  for (int r = r_beg; r < r_end; ++r) {
    __syncthreads();  // stage cur is complete; nobody still reads the other one
    // past the end of the split the units are staged all the same (as zeros, into the stage nobody reads): no branch in the
    // block, so the loads are waited for by count and not all at once
    const bool live = r + 1 < r_end;
    const unsigned char* st = smem + cur * STAGE;
    unsigned char* nxt = smem + (cur ^ 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // the wave's TM dZ tiles stay in registers for the k-step; the X tiles pass through one at a time, the next one's
      // fragments requested before the current one's 3 TM MFMAs (the scheduling fences keep the compiler from hoisting every
      // read of the k-step to its top)
      f16x8 ah[TM], al[TM], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag(st + (wm * TM + i) * 4096, ks);
        al[i] = frag(st + (wm * TM + i) * 4096 + 2048, ks);
      }
      bh[0] = frag(st + OPB + (wn * TN) * 4096, ks);
      bl[0] = frag(st + OPB + (wn * TN) * 4096 + 2048, ks);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (j + 1 < TN) {
          bh[(j + 1) & 1] = frag(st + OPB + (wn * TN + j + 1) * 4096, ks);
          bl[(j + 1) & 1] = frag(st + OPB + (wn * TN + j + 1) * 4096 + 2048, ks);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<RM>(al[i], bh[j & 1], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<RM>(ah[i], bl[j & 1], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mma<RM>(ah[i], bh[j & 1], acc[i][j]);
#pragma unroll
        for (int k = 0; k < UPP; ++k) {
          const int uu = (ks * TN + j) * UPP + k;
          stage_unit(uu, (r + 1) * 32, live, nxt);
          reload(uu);  // block r + 2 (or, past the end of the split, a block that is never staged)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (r + 3 < r_end) { pa2 += a_step; pb2 += b_step; rm_row0 += 32; }
    cur ^= 1;
  }

  // ---- out: accumulator register g of tile (i, j) is dW row 32 (wm TM + i) + 8 (g / 4) + 4 h + (g & 3), column 32 (wn TN + j) + lane & 31
  const float out_scale = a_inv / b_scale;
  const int n0 = (nb * NT + wm * TM) * 32, k0 = (kb * NT + wn * TN) * 32;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float* out = P.dW + (long)(n0 + 32 * i + 4 * h) * P.ldw + k0 + 32 * j + (lane & 31);
      const bool col_ok = k0 + 32 * j + (lane & 31) < P.wb;
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (col_ok && n0 + 32 * i + 4 * h + 8 * (g >> 2) + (g & 3) < P.wa)
          atomicAdd(out + (long)(8 * (g >> 2) + (g & 3)) * P.ldw, acc[i][j][g] * out_scale);
      __builtin_amdgcn_sched_barrier(0);
    }
  if (want_bias && RM) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = (bias[0][i] + bias[1][i]) + (bias[2][i] + bias[3][i]);  // the four units hold different rows of the same features
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);  // over the 8 row groups
      if (rq == 0 && fa + i < P.wa) atomicAdd(P.db + fa + i, v * a_inv);
    }
  } else if (want_bias) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = bias[u][i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);  // over the 32 rows of this lane half
        if (c == 0 && (nb * NT + wave) * 32 + 8 * u + 4 * hh + i < P.wa) atomicAdd(P.db + (nb * NT + wave) * 32 + 8 * u + 4 * hh + i, v * a_inv);
      }
  }
}

}  // namespace

extern "C" int nsky_wgrad_native_batch(const nsky_wgrad_problem* problems, int32_t n_problems, int32_t rows, nsky_stream_t stream) {
  NSKY_CHECK_ARG(problems && n_problems >= 1 && n_problems <= NSKY_WGRAD_MAX_PROBLEMS && rows >= 0,
                 "nsky_wgrad_native_batch: 1..%d problems expected, got %d", NSKY_WGRAD_MAX_PROBLEMS, n_problems);
  if (rows == 0) return NSKY_OK;
  WgradArgs a;
  const bool rm = problems[0].lda > 0;  // row-major operands (all problems of a launch alike)
  bool wide = true;
  for (int i = 0; i < n_problems; ++i) {
    const nsky_wgrad_problem& q = problems[i];
    NSKY_CHECK_ARG(q.dZ && q.X && q.dW, "nsky_wgrad_native_batch: problem %d: null argument", i);
    NSKY_CHECK_ARG((q.lda > 0) == rm && (q.ldb > 0) == rm, "nsky_wgrad_native_batch: problem %d: row-major and tile-native operands in one launch", i);
    if (rm) {
      NSKY_CHECK_ARG(q.width_a >= 4 && q.width_b >= 4 && q.width_a % 4 == 0 && q.width_b % 4 == 0 && q.lda >= q.width_a && q.ldb >= q.width_b &&
                         q.lda % 4 == 0 && q.ldb % 4 == 0, "nsky_wgrad_native_batch: problem %d: row-major widths / leading dimensions must be multiples of 4", i);
      NSKY_CHECK_ARG(q.ldw >= q.width_b && q.a_scale_max == nullptr, "nsky_wgrad_native_batch: problem %d: ldw %d < %d, or a scale with row-major operands", i,
                     q.ldw, q.width_b);
      wide = wide && q.width_a > 128 && q.width_b > 128;
    } else {
      NSKY_CHECK_ARG(q.nnt_a > 0 && q.nnt_b > 0 && q.nnt_a % 4 == 0 && q.nnt_b % 4 == 0,
                     "nsky_wgrad_native_batch: problem %d: both operands need a multiple of 128 features (got %d and %d tiles of 32)", i,
                     q.nnt_a, q.nnt_b);
      NSKY_CHECK_ARG(q.width_a >= 0 && q.width_a <= 32 * q.nnt_a && q.width_b >= 0 && q.width_b <= 32 * q.nnt_b,
                     "nsky_wgrad_native_batch: problem %d: widths %d / %d exceed the tiles", i, q.width_a, q.width_b);
      NSKY_CHECK_ARG(q.ldw >= (q.width_b > 0 ? q.width_b : 32 * q.nnt_b), "nsky_wgrad_native_batch: problem %d: ldw %d too small", i, q.ldw);
      int e = 0;
      NSKY_CHECK_ARG(q.b_scale > 0.0f && frexpf(q.b_scale, &e) == 0.5f, "nsky_wgrad_native_batch: problem %d: b_scale must be a power of two", i);
      wide = wide && q.nnt_a % 8 == 0 && q.nnt_b % 8 == 0;
    }
  }
  // 256 x 256 output blocks, 8 waves, one workgroup per CU -- or, if any problem is narrower, 128 x 128, 4 waves, two per CU
  const int NT = wide ? 8 : 4;
  int tiles = 0;
  for (int i = 0; i < n_problems; ++i) {
    const nsky_wgrad_problem& q = problems[i];
    WgradProblem& P = a.p[i];
    P.A = q.dZ; P.B = q.X; P.dW = q.dW; P.db = q.db; P.a_scale_max = q.a_scale_max; P.b_scale = rm ? 1.0f : q.b_scale;
    P.b_scale_max = rm ? nullptr : q.b_scale_max;
    P.nnt_a = rm ? ceil_div(q.width_a, 32 * NT) * NT : q.nnt_a;
    P.nnt_b = rm ? ceil_div(q.width_b, 32 * NT) * NT : q.nnt_b;
    P.ldw = q.ldw; P.tile0 = tiles;
    P.lda = q.lda; P.ldb = q.ldb;
    P.wa = (rm || q.width_a > 0) ? q.width_a : 32 * q.nnt_a;
    P.wb = (rm || q.width_b > 0) ? q.width_b : 32 * q.nnt_b;
    P.bias_rows = q.bias_rows > 0 ? q.bias_rows : rows;
    P.bias_mod = rm ? 0 : q.bias_row_mod;
    tiles += (P.nnt_a / NT) * (P.nnt_b / NT);
  }
  a.n_problems = n_problems; a.rows = rows; a.n_tiles = tiles;
  a.row_blocks = ceil_div(rows, 32);
  const int slots = wide ? 256 : 512;  // workgroups resident at once
  int splits = slots / a.n_tiles;
  if (splits < 1) splits = 1;
  if (splits > a.row_blocks) splits = a.row_blocks;
  a.blocks_per_split = ceil_div(a.row_blocks, splits);
  splits = ceil_div(a.row_blocks, a.blocks_per_split);
  const dim3 grid(a.n_tiles * splits);
  const size_t smem = 2 * 2 * NT * 4096;
  static bool attr_set = [] {  // not a stream operation: once per process, outside any capture
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_native_kernel<4, 2, 2, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_native_kernel<2, 2, 2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_native_kernel<4, 2, 2, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_native_kernel<2, 2, 2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const hipStream_t st = (hipStream_t)stream;
  if (wide && rm) hipLaunchKernelGGL((wgrad_native_kernel<4, 2, 2, 4, true>), grid, dim3(512), smem, st, a);
  else if (wide) hipLaunchKernelGGL((wgrad_native_kernel<4, 2, 2, 4, false>), grid, dim3(512), smem, st, a);
  else if (rm) hipLaunchKernelGGL((wgrad_native_kernel<2, 2, 2, 2, true>), grid, dim3(256), smem, st, a);
  else hipLaunchKernelGGL((wgrad_native_kernel<2, 2, 2, 2, false>), grid, dim3(256), smem, st, a);
  NSKY_CHECK_LAUNCH("nsky_wgrad_native_batch");
  return NSKY_OK;
}

extern "C" int nsky_wgrad_native(const float* dZ, int32_t nnt_a, const float* X, int32_t nnt_b, int32_t rows, float* dW, int32_t ldw,
                                 float* db, const float* a_scale_max, float b_scale, nsky_stream_t stream) {
  nsky_wgrad_problem q = {};
  q.dZ = dZ; q.nnt_a = nnt_a; q.X = X; q.nnt_b = nnt_b; q.dW = dW; q.ldw = ldw; q.db = db; q.a_scale_max = a_scale_max; q.b_scale = b_scale;
  return nsky_wgrad_native_batch(&q, 1, rows, stream);
}
