// fp32 dense layer on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   C[M,N] = epilogue( sum_k A(m,k) B(n,k) + bias[n] )
//
// Workgroup = 256 threads = 4 waves (one per SIMD).  Block tile BM x BN x 32, each wave owns a
// (BM/WAVES_M) x (BN/WAVES_N) sub-tile built from 32x32 MFMA accumulators.  Both operand tiles live
// in LDS K-MAJOR ([k][m], row stride BM+2 floats) whatever their layout in HBM, so the MFMA operand
// fetch is one conflict-free ds_read_b32 per 32x2 fragment (lanes 0-31 read row k, lanes 32-63 row
// k+1).  k-contiguous operands are transposed on the way into LDS (8 lanes read one 128-B row
// segment, the (BM+2) stride keeps the scattered ds_write_b32 at 2-way = free).  Global->LDS is
// register staged and software pipelined: tile t+1 is fetched to VGPRs before the 64 MFMAs of tile
// t and written to the other LDS buffer after them, one barrier per tile.
#include "common.h"
#include "../../include/neusky_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;

struct EpiCtx {
  const float* bias;
  int epi;
  float p0, p1;
  const float* aux0; int ldaux0;
  const float* aux1; int ldaux1;
  const float* aux2; int ldaux2;
  float* out1; int ldout1;
  float* out2; int ldout2;
  int row_mod;
  float beta;
};

__device__ __forceinline__ void epilogue_store(const EpiCtx& e, float* C, int ldc, int row, int col, float acc) {
  float v = acc + (e.bias ? e.bias[col] : 0.0f);
  float r;
  switch (e.epi) {
    default:
    case NSKY_EPI_NONE: r = v; break;
    case NSKY_EPI_RELU: r = fmaxf(v, 0.0f); break;
    case NSKY_EPI_LEAKY: r = v > 0.0f ? v : e.p0 * v; break;
    case NSKY_EPI_SIGMOID: r = e.p0 * sigmoidf_(v); break;
    case NSKY_EPI_SOFTPLUS: {
      float bv = e.p0 * v;
      r = bv > 20.0f ? v : log1pf(expf(bv)) / e.p0;
      if (e.out1) e.out1[(long)row * e.ldout1 + col] = sigmoidf_(bv);
    } break;
    case NSKY_EPI_FILM: {
      float f = e.p0 * e.aux0[(long)row * e.ldaux0 + col] + e.p1;
      float ph = e.aux1[(long)row * e.ldaux1 + col];
      r = sinf(f * v + ph);
      if (e.out1) e.out1[(long)row * e.ldout1 + col] = v;
    } break;
    case NSKY_EPI_MUL_AUX: r = v * e.aux0[(long)(row % e.row_mod) * e.ldaux0 + col]; break;
    case NSKY_EPI_BWD_RELU: r = e.aux0[(long)row * e.ldaux0 + col] > 0.0f ? v : 0.0f; break;
    case NSKY_EPI_BWD_LEAKY: r = e.aux0[(long)row * e.ldaux0 + col] > 0.0f ? v : e.p0 * v; break;
    case NSKY_EPI_BWD_FILM: {
      float z = e.aux0[(long)row * e.ldaux0 + col];
      float f = e.p0 * e.aux1[(long)row * e.ldaux1 + col] + e.p1;
      float ph = e.aux2[(long)row * e.ldaux2 + col];
      float gc = v * cosf(f * z + ph);
      r = gc * f;
      e.out1[(long)row * e.ldout1 + col] = gc * z * e.p0;
      e.out2[(long)row * e.ldout2 + col] = gc;
    } break;
    case NSKY_EPI_EXP: r = expf(fminf(v, e.p0)); break;
  }
  float* dst = C + (long)row * ldc + col;
  if (e.beta != 0.0f) r += e.beta * (*dst);
  *dst = r;
}

template <int BT, bool KCONTIG>
struct TileLoader {
  // BT x BK tile; F4 float4 per thread
  static constexpr int F4 = BT * BK / 4 / 256;
  float4 v[F4];

  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int t0, int T, int k0, int kend, int tid) {
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      int f = it * 256 + tid;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        int row = f >> 3, kq = f & 7;
        int t = t0 + row, k = k0 + kq * 4;
        if (t < T && k < kend) x = *reinterpret_cast<const float4*>(P + (long)t * ld + k);
      } else {
        int krow = f / (BT / 4), t4 = f % (BT / 4);
        int k = k0 + krow, t = t0 + t4 * 4;
        if (k < kend && t < T) x = *reinterpret_cast<const float4*>(P + (long)k * ld + t);
      }
      v[it] = x;
    }
  }

  __device__ __forceinline__ void store(float* S, int tid) const {
    constexpr int LD = BT + 2;
#pragma unroll
    for (int it = 0; it < F4; ++it) {
      int f = it * 256 + tid;
      if (KCONTIG) {
        int row = f >> 3, kq = f & 7;
        float* s = S + (kq * 4) * LD + row;
        s[0] = v[it].x; s[LD] = v[it].y; s[2 * LD] = v[it].z; s[3 * LD] = v[it].w;
      } else {
        int krow = f / (BT / 4), t4 = f % (BT / 4);
        float2* s = reinterpret_cast<float2*>(S + krow * LD + t4 * 4);
        s[0] = make_float2(v[it].x, v[it].y);
        s[1] = make_float2(v[it].z, v[it].w);
      }
    }
  }
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool AK, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                       int ldc, int k_split_len, EpiCtx e) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA_S = BM + 2, LDB_S = BN + 2;
  __shared__ float smem[2 * BK * (LDA_S + LDB_S)];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA_S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * k_split_len;
  const int kend = min(K, kbeg + k_split_len);
  const int ntiles = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  TileLoader<BM, AK> la;
  TileLoader<BN, BKC> lb;
  if (ntiles > 0) {
    la.load(A, lda, m0, M, kbeg, kend, tid);
    lb.load(B, ldb, n0, N, kbeg, kend, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1 < ntiles);
    if (more) {
      la.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend, tid);
      lb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend, tid);
    }
    const float* as = As + cur * BK * LDA_S + (lane >> 5) * LDA_S + wm * WM + (lane & 31);
    const float* bs = Bs + cur * BK * LDB_S + (lane >> 5) * LDB_S + wn * WN + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[kk * 2 * LDA_S + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[kk * 2 * LDB_S + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      la.store(As + (cur ^ 1) * BK * LDA_S, tid);
      lb.store(Bs + (cur ^ 1) * BK * LDB_S, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  // Epilogue: park the accumulators in LDS (the operand buffers are dead after the last barrier),
  // then sweep the tile row-major so C, the aux operands and the side outputs move as whole rows.
  float* Cs = smem;  // BM*BN floats <= 2*BK*(LDA_S+LDB_S)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int cl = wn * WN + j * 32 + (lane & 31);
        Cs[rl * BN + cl] = acc[i][j][r];
      }
  __syncthreads();
  const bool atomic = gridDim.z > 1;
  for (int idx = tid; idx < BM * BN; idx += 256) {
    const int rl = idx / BN, cl = idx % BN;
    const int row = m0 + rl, col = n0 + cl;
    if (row < M && col < N) {
      const float v = Cs[idx];
      if (atomic)
        atomicAdd(C + (long)row * ldc + col, v);
      else
        epilogue_store(e, C, ldc, row, col, v);
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch(const nsky_gemm_desc* d, const EpiCtx& e, int splits, int k_split_len, hipStream_t s) {
  dim3 grid(ceil_div(d->M, BM), ceil_div(d->N, BN), splits);
#define NSKY_GEMM_LAUNCH(AK, BKC)                                                                              \
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, AK, BKC>), grid, dim3(256), 0, s, d->A, d->B, \
                     d->C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, k_split_len, e)
  if (d->a_kcontig && d->b_kcontig) NSKY_GEMM_LAUNCH(true, true);
  else if (d->a_kcontig && !d->b_kcontig) NSKY_GEMM_LAUNCH(true, false);
  else if (!d->a_kcontig && d->b_kcontig) NSKY_GEMM_LAUNCH(false, true);
  else NSKY_GEMM_LAUNCH(false, false);
#undef NSKY_GEMM_LAUNCH
}

__global__ void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx, float* __restrict__ out, int rows_per_block) {
  // block: 256 threads = 64 columns x 4 row-phases
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int phase = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.0f;
  if (col < N)
    for (int r = r0 + phase; r < r1; r += 4) s += X[(long)r * ldx + col];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (phase == 0 && col < N) atomicAdd(out + col, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

}  // namespace

extern "C" int nsky_gemm_f32(const nsky_gemm_desc* d, nsky_stream_t stream) {
  NSKY_CHECK_ARG(d && d->A && d->B && d->C, "nsky_gemm_f32: null operand");
  NSKY_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "nsky_gemm_f32: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  NSKY_CHECK_ARG(d->lda % 4 == 0 && d->ldb % 4 == 0, "nsky_gemm_f32: lda/ldb must be multiples of 4 (got %d,%d)", d->lda, d->ldb);
  NSKY_CHECK_ARG(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "nsky_gemm_f32: A/B must be 16-byte aligned");
  if (d->a_kcontig || d->b_kcontig) NSKY_CHECK_ARG(d->K % 4 == 0, "nsky_gemm_f32: K=%d must be a multiple of 4 for k-contiguous operands", d->K);
  if (d->a_kcontig) NSKY_CHECK_ARG(d->lda >= d->K, "nsky_gemm_f32: lda < K"); else NSKY_CHECK_ARG(d->lda >= d->M, "nsky_gemm_f32: lda < M");
  if (d->b_kcontig) NSKY_CHECK_ARG(d->ldb >= d->K, "nsky_gemm_f32: ldb < K"); else NSKY_CHECK_ARG(d->ldb >= d->N, "nsky_gemm_f32: ldb < N");
  NSKY_CHECK_ARG(d->ldc >= d->N, "nsky_gemm_f32: ldc < N");
  int splits = d->k_splits > 1 ? d->k_splits : 1;
  int k_split_len = d->K;
  if (splits > 1) {
    NSKY_CHECK_ARG(d->epi == NSKY_EPI_NONE && !d->bias && d->beta == 0.0f, "nsky_gemm_f32: split-K needs a plain epilogue");
    k_split_len = ((d->K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (d->K + k_split_len - 1) / k_split_len;
  }
  switch (d->epi) {
    case NSKY_EPI_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1, "nsky_gemm_f32: FILM needs aux0 (freq) and aux1 (phase)"); break;
    case NSKY_EPI_MUL_AUX: case NSKY_EPI_BWD_RELU: case NSKY_EPI_BWD_LEAKY: NSKY_CHECK_ARG(d->aux0, "nsky_gemm_f32: epilogue needs aux0"); break;
    case NSKY_EPI_BWD_FILM: NSKY_CHECK_ARG(d->aux0 && d->aux1 && d->aux2 && d->out1 && d->out2, "nsky_gemm_f32: BWD_FILM needs aux0..2, out1, out2"); break;
    default: break;
  }
  EpiCtx e;
  e.bias = d->bias; e.epi = d->epi; e.p0 = d->p0; e.p1 = d->p1;
  e.aux0 = d->aux0; e.ldaux0 = d->ldaux0; e.aux1 = d->aux1; e.ldaux1 = d->ldaux1; e.aux2 = d->aux2; e.ldaux2 = d->ldaux2;
  e.out1 = d->out1; e.ldout1 = d->ldout1; e.out2 = d->out2; e.ldout2 = d->ldout2;
  e.row_mod = d->row_mod > 0 ? d->row_mod : d->M;
  e.beta = d->beta;
  hipStream_t s = (hipStream_t)stream;
  if (d->N <= 32)
    launch<128, 32, 4, 1>(d, e, splits, k_split_len, s);
  else if (d->N <= 64)
    launch<128, 64, 2, 2>(d, e, splits, k_split_len, s);
  else
    launch<128, 128, 2, 2>(d, e, splits, k_split_len, s);
  NSKY_CHECK_LAUNCH("nsky_gemm_f32");
  return NSKY_OK;
}

extern "C" int nsky_colsum_f32(const float* X, int32_t M, int32_t N, int32_t ldx, float* out, nsky_stream_t stream) {
  NSKY_CHECK_ARG(X && out && M > 0 && N > 0 && ldx >= N, "nsky_colsum_f32: bad arguments");
  const int rows_per_block = 512;
  dim3 grid(ceil_div(N, 64), ceil_div(M, rows_per_block));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, M, N, ldx, out, rows_per_block);
  NSKY_CHECK_LAUNCH("nsky_colsum_f32");
  return NSKY_OK;
}
