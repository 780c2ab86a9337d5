/* neusky_hip.h - C ABI of libneusky_hip.so (MI355X / gfx950 HIP kernels for the NeuSky hot path).
 *
 * The reference (JADGardner/neusky) is 100% Python and has NO native boundary of its own
 * (SURVEY.md F1): every native kernel it runs belongs to tiny-cuda-nn / torch / nerfacc.  This
 * header is therefore the drop-in seam SURVEY.md 8(b) defines: each entry point replaces the
 * external kernel (or the chain of torch ops) the reference reaches at the cited call site.
 *
 * Conventions (all entry points)
 *   - extern "C", plain device pointers + sizes + scalars + a hipStream_t; no torch types.
 *   - ownership: the caller allocates every buffer (device memory); the callee never allocates,
 *     frees or retains a pointer.  All matrices are row-major float32 unless stated; leading
 *     dimensions (ld*) are in elements.
 *   - errors: return 0 on success, <0 on error (never throws); nsky_last_error() returns the text
 *     of the calling thread's last error.
 *   - threading: re-entrant, no global mutable state except the thread-local error string;
 *     stream-ordered; no host synchronisation and no allocation inside (hipGraph-capturable).
 */
#ifndef NEUSKY_HIP_H
#define NEUSKY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nsky_stream_t; /* hipStream_t */

const char* nsky_last_error(void);
int nsky_abi_version(void); /* 16 (bumped when a struct layout or an entry point's signature changes) */

/* ------------------------------------------------------------------------------------------
 * Dense layer on the matrix cores: C[M,N] = epilogue( sum_k A(m,k) * B(n,k) + bias[n] )
 * exact fp32 (v_mfma_f32_32x32x2_f32).  Replaces torch nn.Linear / F.linear + activation at
 *   neusky/fields/sdf_albedo_field.py:172,199-207,233 (geo + colour nets),
 *   neusky/utils/siren.py:114-119,138-144,207 (mapping net, FiLM sine layers, head),
 * and the autograd-generated backward GEMMs of the same layers.
 *   a_kcontig: 1 -> A stored [M][lda] (k contiguous); 0 -> A stored [K][lda] (m contiguous, i.e. A^T view)
 *   b_kcontig: 1 -> B stored [N][ldb] (k contiguous, torch Linear weight); 0 -> B stored [K][ldb]
 *   k-contiguous operands need K % 4 == 0 (zero padded); every ld % 4 == 0, bases 16-byte aligned.
 *   k_splits > 1: split the K range over grid.z and atomically add into C (C pre-zeroed by the
 *                 caller, epilogue must be NSKY_EPI_NONE, bias NULL) - used for weight gradients.
 */
enum {
  NSKY_EPI_NONE = 0,
  NSKY_EPI_RELU = 1,
  NSKY_EPI_LEAKY = 2,      /* slope p0 */
  NSKY_EPI_SIGMOID = 3,    /* C = p0 * sigmoid(v) */
  NSKY_EPI_SOFTPLUS = 4,   /* beta p0 (threshold 20); out1 (optional) = sigmoid(beta v) = d softplus/dv */
  NSKY_EPI_FILM = 5,       /* C = sin((p0*aux0+p1) * v + aux1); out1 (optional) = v */
  NSKY_EPI_MUL_AUX = 6,    /* C = v * aux0[row % row_mod] */
  NSKY_EPI_BWD_RELU = 7,   /* C = v * (aux0 > 0) */
  NSKY_EPI_BWD_LEAKY = 8,  /* C = v * (aux0 > 0 ? 1 : p0) */
  NSKY_EPI_BWD_FILM = 9,   /* z=aux0, F=aux1, P=aux2: u=(p0 F+p1) z+P; C=v cos(u)(p0 F+p1); out1=v cos(u) z p0; out2=v cos(u) */
  NSKY_EPI_EXP = 10        /* C = exp(min(v, p0)) */
};

/* contraction arithmetic: exact fp32 MFMA (default); or operands split on the fly into 2 / 3 bf16 terms and
 * rebuilt from 3 / 6 bf16 MFMAs with fp32 accumulation (relative product error ~2^-16 / ~2^-22); or F16X2: split into
 * fp16 hi + fp16 residual scaled by 2^11, rebuilt from 3 fp16 MFMAs in two fp32 accumulators (~2^-21, fp32-grade) --
 * for operands within fp16's range only (|x| <= 65504, larger magnitudes saturate): forward layers, not gradients.
 * N <= 64 always runs the fp32 kernel. */
enum { NSKY_PREC_F32 = 0, NSKY_PREC_BF16X2 = 2, NSKY_PREC_BF16X3 = 3, NSKY_PREC_F16X2 = 4 };

typedef struct nsky_gemm_desc {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_kcontig, b_kcontig;
  const float* bias;
  int32_t epi;
  float p0, p1;
  const float* aux0; int32_t ldaux0;
  const float* aux1; int32_t ldaux1;
  const float* aux2; int32_t ldaux2;
  float* out1; int32_t ldout1;
  float* out2; int32_t ldout2;
  int32_t row_mod;   /* 0 = M */
  int32_t k_splits;  /* 0/1 = none */
  float beta;        /* C = result + beta * C_old (non-split only; 0 or 1) */
  float* a_rowsum;   /* optional [M]: ACCUMULATES sum_k A(m,k) (bias gradient when A = dZ^T); atomics */
  int32_t precision; /* NSKY_PREC_*: arithmetic of the contraction (operands and result stay fp32 in memory) */
  int32_t rowsum_k_limit; /* a_rowsum sums only k < rowsum_k_limit (multiple of 32; 0 = all of K): the value rows of a stacked
                             [value; tangent] gradient matrix carry the bias, the tangent rows do not */
  int32_t a_native_nt, b_native_nt; /* > 0: that operand (a_kcontig / b_kcontig must be 0) is a TILE-NATIVE matrix with this many
                             32-feature tiles per row, as the FiLM chain kernels store activations and gradients (below) */
  const float* a_scale_max; /* optional device scalar = largest |A| (NSKY_PREC_F16X2, N > 64): A is staged times the power of two
                             that brings it to ~2^14, so the fp16 split keeps fp32-grade products for gradients of any
                             magnitude; C and a_rowsum are scaled back */
} nsky_gemm_desc;

int nsky_gemm_f32(const nsky_gemm_desc* d, nsky_stream_t stream);

/* Same contraction against a PRE-SPLIT B operand (a weight matrix, re-read by every row tile of A): the two 16-bit
 * planes of B -- fp16 hi + 2^11-scaled fp16 residual (NSKY_PREC_F16X2) or bf16 hi + bf16 residual (NSKY_PREC_BF16X2) --
 * are produced once per optimisation step by nsky_split_planes and streamed into LDS by LDS-DMA, as is the raw fp32 A,
 * which is split in place there (no register staging; same MFMA sequence and results as nsky_gemm_f32 with that precision).
 *   nsky_split_planes: planes(n, k) = W[n][k] (transpose = 0: forward layers, W = torch Linear weight [out, in]) or
 *                      W[k][n] (transpose = 1: input gradients dX = dZ W); hi / lo are [rows_pad][ldp] uint16, zero padded,
 *                      rows_pad % 256 == 0, ldp % 32 == 0.
 *   nsky_gemm_f32_planes: d->B, ldb, b_kcontig are ignored; A must be k-contiguous; K % 4 == 0 (finite A); no split-K, no a_rowsum;
 *                      d->precision selects the plane format (must match the split); epilogues as nsky_gemm_f32. */
int nsky_split_planes(const float* W, int32_t n_rows, int32_t n_k, int32_t ldw, int32_t transpose, int32_t precision,
                      uint16_t* hi, uint16_t* lo, int32_t rows_pad, int32_t ldp, nsky_stream_t stream);
int nsky_gemm_f32_planes(const nsky_gemm_desc* d, const uint16_t* B_hi, const uint16_t* B_lo, int32_t ldp, nsky_stream_t stream);

/* column sums: out[n] (+)= sum_m X[m,n]  (bias gradients) */
int nsky_colsum_f32(const float* X, int32_t M, int32_t N, int32_t ldx, float* out, nsky_stream_t stream);
/* out[c] += sum_r w[r * w_stride] X[r][c]: the weight gradient of a ONE-output dense layer (the sdf head of the geo net,
 * sdf_albedo_field.py:169-174 and :233-238) -- a matrix-vector product streamed at HBM rate instead of a 1-row GEMM. */
int nsky_weighted_colsum_f32(const float* X, int32_t M, int32_t N, int32_t ldx, const float* w, int32_t w_stride, float* out,
                             nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FiLM-SIREN chain as ONE kernel per row tile (no per-layer launches, no [M, 2 n_film H] frequency / phase matrix, no
 * layer-to-layer activation round trip through HBM).  Replaces the whole module neusky/utils/siren.py:108-208
 * (CustomMappingNetwork :108-131, FiLMLayer :133-145, DDFFiLMSiren.forward :189-208; frequencies = raw * 15 + 30 :200) at
 * its call sites neusky/fields/directional_distance_field.py:233-243,275-276 (DDF) and the RENI-shaped illumination
 * decode driven from neusky/models/neusky_model.py:488-506,535-549.
 *   net: geometry + the fp32 parameter pointers in torch nn.Linear layout ([out, in], rows contiguous, leading dimensions
 *        in elements).  hidden = width of the FiLM layers AND of the mapping layers (128 or 256); cond_dim <= 320,
 *        x_dim <= 16, out_dim <= 4, at most NSKY_FILM_MAX_LAYERS layers each.
 *   nsky_film_stream_layout: bytes of the packed weight stream and number of 32-feature weight tiles.
 *   nsky_film_pack: once per optimisation step: every weight tile -> power-of-two scaled fp16 hi + fp16 residual planes in
 *        MFMA-fragment order (direction 0 = forward stream, 1 = backward stream), plus `table` (NSKY_FILM_TABLE_FLOATS
 *        floats, 16-byte aligned): every bias of the network followed by one reciprocal scale per tile.
 *   nsky_film_chain_fwd: cond [M, ldcond] (first cond_dim columns), x [M, ldx] -> res [M, ldres] (first out_dim columns,
 *        raw head output; columns out_dim..3 are written too).  Side outputs, each a TILE-NATIVE [ceil32(M), hidden] matrix
 *        (below): y_save[i] (FiLM layer outputs; REQUIRED: they are also the hand-off to the next layer), z_save[i] (FiLM
 *        pre-activations W y + b; optional), h_save[l] (mapping activations after LeakyReLU(0.2); optional).
 *        Products are fp32-grade (three fp16 MFMAs on power-of-two pre-scaled hi / residual planes, fp32 accumulate).
 *   Tile-native layout of a [rows, width] fp32 matrix (rows padded to a multiple of 32, width % 32 == 0): 32 x 32 blocks of
 *        4 KB, block (R, t) at float offset (R * (width / 32) + t) * 1024; inside a block element (row c, feature f) at
 *        (f / 8) * 256 + (c + 32 * ((f / 4) & 1)) * 4 + (f & 3) -- the accumulator layout of v_mfma_f32_32x32x16, so the
 *        chain kernels move a tile with four 1 KB-contiguous wave instructions.  nsky_gemm_f32 reads such operands with
 *        a_native_nt / b_native_nt = width / 32 (weight gradients).
 */
#define NSKY_FILM_MAX_LAYERS 12
#define NSKY_FILM_TABLE_FLOATS 6656
typedef struct nsky_film_net {
  int32_t hidden, n_map, n_film;
  int32_t cond_dim, x_dim, out_dim;
  const float* map_w[NSKY_FILM_MAX_LAYERS]; const float* map_b[NSKY_FILM_MAX_LAYERS]; int32_t map_ld[NSKY_FILM_MAX_LAYERS];
  const float* mo_w; const float* mo_b; int32_t mo_ld;     /* mapping head [2 n_film hidden, hidden]: frequencies, then phases */
  const float* film_w[NSKY_FILM_MAX_LAYERS]; const float* film_b[NSKY_FILM_MAX_LAYERS]; int32_t film_ld[NSKY_FILM_MAX_LAYERS];
  const float* out_w; const float* out_b; int32_t out_ld;  /* head [out_dim, hidden] */
} nsky_film_net;
int nsky_film_stream_layout(const nsky_film_net* net, int32_t direction, int64_t* stream_bytes, int32_t* n_tiles);
int nsky_film_pack(const nsky_film_net* net, int32_t direction, void* stream_buf, float* table, nsky_stream_t stream);
int nsky_film_chain_fwd(const nsky_film_net* net, const void* stream_buf, const float* table, const float* cond,
                        int32_t ldcond, const float* x, int32_t ldx, int32_t M, float* const* h_save, float* const* z_save,
                        float* const* y_save, float* res, int32_t ldres, nsky_stream_t stream);
/* Backward of the chain (replaces the autograd-generated backward of siren.py:108-208; parameters' gradients are then plain
 * weight-gradient GEMMs, nsky_gemm_f32 with a_native_nt / b_native_nt, over the tile-native matrices written here).
 *   nsky_film_chain_bwd_film (stream / table packed with direction 1): d_res [M, ldres] -> for every FiLM layer i
 *        dz_save[i] = dL/d(W y + b) (tile-native [ceil32(M), hidden]) and, in dfp (tile-native [ceil32(M), 2 n_film hidden]),
 *        dL/dF_i in columns i hidden.. and dL/dphase_i in columns (n_film + i) hidden..; dfp_rowmax [ceil32(M)] = max |dfp| of each
 *        batch row.  F and phase are re-formed from h_last (the last mapping activation), z_save[i] is read back.
 *   nsky_film_chain_bwd_map (direction 2): dfp, dfp_rowmax, h_save[l] -> dpre_save[l] = dL/d(pre-activation of mapping layer l)
 *        (tile-native) and d_cond [M, ldcond] (row-major, pad columns zeroed; optional).
 * gmax (caller zero-fills): largest magnitude of each gradient matrix, for nsky_gemm_f32's a_scale_max -- bwd_film writes
 *        [i] = max |dz_save[i]| (i < n_film) and [n_film] = max |dfp|; bwd_map writes [l] = max |dpre_save[l]| (l < n_map).
 * hidden must be a multiple of 128 for the backward streams.  d_x (optional, [M, ldx], ldx <= 16): gradient w.r.t. the FiLM input
 * rows (the DDF's multi-view rays differentiate through their direction rows, neusky/models/ddf_model.py:297-322). */
int nsky_film_chain_bwd_film(const nsky_film_net* net, const void* stream_buf, const float* table, int32_t M, const float* d_res,
                             int32_t ldres, const float* h_last, const float* const* z_save, float* const* dz_save, float* dfp,
                             float* dfp_rowmax, float* gmax, float* d_x, int32_t ldx, nsky_stream_t stream);
int nsky_film_chain_bwd_map(const nsky_film_net* net, const void* stream_buf, const float* table, int32_t M, const float* dfp,
                            const float* dfp_rowmax, const float* const* h_save, float* const* dpre_save, float* d_cond,
                            int32_t ldcond, float* gmax, nsky_stream_t stream);
/* SDF value chain: SDFAlbedoField.get_sdf_at_pos (neusky/fields/sdf_albedo_field.py:169-174; nerfstudio SDFField geometry network:
 * Linear + Softplus(beta) twice, then the sdf row of the last Linear) for M encode rows, forward and backward as one kernel each
 * (same weight-stream machinery as the FiLM-SIREN chain: nsky_sdf_stream_layout / nsky_sdf_pack once per optimizer step and
 * direction, 0 = forward, 1 = backward).  a0_save / a1_save: the two softplus outputs, tile-native [ceil32(M), hidden] (scratch in
 * inference).  _bwd: g_sdf [M] -> dz1 / dz0 (tile-native pre-activation gradients: the operands of the weight gradients,
 * nsky_wgrad_native), dE [M, ldE] (optional), dw2 [hidden] += sum_rows g a1 and db2 [1] += sum_rows g (optional), gmax [2] = max |dz1|, max |dz0| (caller
 * zero-fills).  hidden = 256, in_dim <= 80 (multiple of 4). */
typedef struct {
  int32_t in_dim, hidden;
  const float* w0; int32_t ld0; const float* b0;   /* [hidden, in_dim] */
  const float* w1; int32_t ld1; const float* b1;   /* [hidden, hidden] */
  const float* w2; const float* b2;                /* the sdf row [hidden] of the last layer and its bias (1 value; may be null) */
  float beta;
} nsky_sdf_net;
int nsky_sdf_stream_layout(const nsky_sdf_net* net, int32_t direction, int64_t* stream_bytes, int32_t* n_tiles);
int nsky_sdf_pack(const nsky_sdf_net* net, int32_t direction, void* stream_buf, float* table, nsky_stream_t stream);
int nsky_sdf_chain_fwd(const nsky_sdf_net* net, const void* stream_buf, const float* table, const float* E, int32_t ldE, int32_t M,
                       float* a0_save, float* a1_save, float* sdf, nsky_stream_t stream);
int nsky_sdf_chain_bwd(const nsky_sdf_net* net, const void* stream_buf, const float* table, int32_t M, const float* g_sdf,
                       const float* a0_save, const float* a1_save, float* dz1, float* dz0, float* dE, int32_t ldE, float* dw2,
                       float* db2, float* gmax, nsky_stream_t stream);

/* Weight and bias gradient of one dense layer of the chain straight over two tile-native matrices (what autograd's
 * AddmmBackward / the MmBackward pair of nn.Linear computes for siren.py:59-66, :167-172 and the mapping network's layers):
 *     dW[n, k] += sum_rows dZ[row, n] X[row, k]   (n < 32 nnt_a, k < 32 nnt_b; dW row-major, leading dimension ldw)
 *     db[n]    += sum_rows dZ[row, n]             (db optional)
 * dW / db are ACCUMULATED into (float atomics over a batch split): the caller zero-fills them or passes a gradient slab.
 * nnt_a and nnt_b must be multiples of 4 (128 features).  a_scale_max: device scalar holding max |dZ| (gmax above), or null;
 * b_scale: power of two applied to X before the fp16 hi + residual split (2^6 suits activations up to ~500).  Products are
 * fp32-grade (three fp16 MFMAs per product, fp32 accumulation). */
int nsky_wgrad_native(const float* dZ, int32_t nnt_a, const float* X, int32_t nnt_b, int32_t rows, float* dW, int32_t ldw,
                      float* db, const float* a_scale_max, float b_scale, nsky_stream_t stream);
/* The same for up to NSKY_WGRAD_MAX_PROBLEMS layers that share the batch (every layer of a chain's backward) in ONE launch:
 * the batch split is shared, so the reduction traffic and the launch cost are paid once instead of per layer. */
#define NSKY_WGRAD_MAX_PROBLEMS 16
typedef struct {
  const float* dZ; int32_t nnt_a;
  const float* X;  int32_t nnt_b;
  float* dW;       int32_t ldw;
  float* db;                 /* may be null */
  const float* a_scale_max;  /* may be null */
  float b_scale;
  /* ROW-MAJOR operands instead of tile-native ones (the SDF / colour field's weight gradients, autograd's MmBackward of
   * sdf_albedo_field.py:147-161,199-207): lda, ldb > 0 = leading dimensions of dZ [rows, width_a] and X [rows, width_b] (widths and
   * leading dimensions multiples of 4; nnt_*, a_scale_max, b_scale unused); the products are then the 2-term bf16 split (2^-16 per
   * product, no pre-scaling).  All problems of one launch are of the same kind.  bias_rows > 0: db sums only the first bias_rows rows
   * (stacked value + tangent rows: only the value rows carry a bias). */
  int32_t lda, ldb, width_a, width_b, bias_rows;
  /* tile-native operands only: width_a / width_b > 0 = features of dZ / X that exist (dW is [width_a, width_b], ldw >= width_b; the
   * tiles beyond are walked but their outputs dropped); bias_row_mod = 4: db sums the rows with row % 4 == 0 only (quad-native
   * matrices of the SDF field, nsky_field_geo_bwd: row 4 n + j is row-set j of point n and only the value rows carry a bias). */
  int32_t bias_row_mod;
  /* tile-native operands only: device scalar holding max |X| (may be null: the constant b_scale applies).  Operands that carry input
   * tangents -- the quad-native encode rows and hidden activations of the SDF field -- have no a-priori bound (a hash level of resolution
   * 2048 with table differences of order one has d feature / d x in the thousands): nsky_field_geo_fwd publishes their maxima. */
  const float* b_scale_max;
} nsky_wgrad_problem;
int nsky_wgrad_native_batch(const nsky_wgrad_problem* problems, int32_t n_problems, int32_t rows, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SDF / albedo field as chain kernels (csrc/field_chain.hip).  Replaces SDFAlbedoField.get_outputs / get_colors,
 * neusky/fields/sdf_albedo_field.py:185-269: the geometry network it inherits from nerfstudio's SDFField ([x | PE6 | hash] -> Linear +
 * Softplus(beta) -> Linear + Softplus(beta) -> Linear -> [sdf | feat]), torch.autograd.grad(sdf, x, create_graph=True) (:231-238; here the
 * encode row's three tangent rows ride through the layers in forward mode) and the colour network (:147-161,199-207), forward and
 * backward (the reference's double backward) as two kernels each on the weight-stream machinery of the FiLM-SIREN chain.
 *
 * Packed weight streams of arbitrary layer lists: a layer is a [rows, K] matrix (transposed = 0: W[r][k] = W[r * ld + k]; transposed = 1:
 * W[r][k] = W[k * ld + r], i.e. the torch [out, in] matrix walked by input feature), cut into ceil(rows / 32) tiles of 32 output rows,
 * K <= 320.  nsky_chain_pack writes the stream and one reciprocal power-of-two scale per tile (scales[n_tiles]).
 */
#define NSKY_CHAIN_MAX_LAYERS 8
typedef struct { const float* W; int32_t ld, rows, K, transposed; } nsky_chain_layer;
int nsky_chain_stream_layout(const nsky_chain_layer* layers, int32_t n_layers, int64_t* stream_bytes, int32_t* n_tiles, int32_t* n_groups);
int nsky_chain_pack(const nsky_chain_layer* layers, int32_t n_layers, void* stream_buf, float* scales, nsky_stream_t stream);
/* The small vectors of the field the kernels keep in LDS (hidden width 256, geometric feature width 256).  The weight matrices come as
 * packed streams (total_groups = their n_groups), in this order:
 *   geo_fwd    : W0 [256, in_dim], W1 [256, 256]
 *   colour_fwd : W2f [256, 256] (feature rows of the last geometry layer), Wc0 [256, 300] (columns [feat 256 | 0 0 0 0 | x PE | 0]), Wc1 [256, 256]
 *   colour_bwd : Wc1^T, Wc0^T (rows = its 300 input columns), W2f^T      (transposed = 1)
 *   geo_bwd    : W1^T, W0^T (rows = in_dim)                              (transposed = 1) */
typedef struct nsky_field_net {
  int32_t in_dim;                         /* width of an encode row (multiple of 4; 68..80): [x | PE | hash] */
  int32_t npe;                            /* its leading x / PE columns that also feed the colour net (<= 39) */
  float beta;                             /* Softplus beta (sdf_albedo_field.py:163) */
  const float* b0; const float* b1;       /* [256] biases of the two hidden geometry layers */
  const float* w_sdf; const float* b_sdf; /* the sdf row [256] of the last geometry layer and its bias [1] (may be null) */
  const float* b2f;                       /* [256] bias of its feature rows */
  const float* bc0; const float* bc1;     /* [256] */
  const float* wc2; int32_t ldc2; const float* bc2;  /* output layer of the colour net [3, 256] (row stride ldc2), [3] */
} nsky_field_net;
/* QUAD-NATIVE matrices: tile-native [ceil32(4 N), width] whose row 4 n + j is row-set j of point n (j = 0: value, j = 1..3: d/dx_k).
 * ET: the stacked encode matrix [4 N, ldE] of nsky_encode_fwd (row j N + n).  Outputs: sdf [N], grad [N, 3] (d sdf / dx), a0q / a1q
 * (the two hidden layers' outputs, quad-native, width 256), Eq (optional; quad-native copy of the encode rows, width 128, for the first
 * layer's weight gradient), a1max [N] (optional: largest |a1| of each value row, the colour path's operand scale), qmax [2] (optional; the
 * caller zero-fills it: max |Eq| and max |a0q| over value and tangent rows, the b_scale_max of the first two layers' weight gradients). */
int nsky_field_geo_fwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, const float* ET,
                       int32_t ldE, int32_t N, float* a0q, float* a1q, float* Eq, float* a1max, float* sdf, float* grad, float* qmax,
                       nsky_stream_t stream);
/* feat = W2f a1 + b2f; albedo = sigmoid(Wc2 relu(Wc1 relu(Wc0 [feat | x PE] + bc0) + bc1) + bc2) -> alb [N, 4] (3 used).  Saves (tile-
 * native, rows = points): a1v [ceil32(N), 256] (optional: value rows of a1), feat [ceil32(N), 256] and xpe [ceil32(N), 128] (columns 0..255
 * and 256..303 of the colour net's input; xpe's columns 48.. are not written), c0, c1 [ceil32(N), 256]. */
int nsky_field_colour_fwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, const float* ET,
                          int32_t ldE, int32_t N, const float* a1q, const float* a1max, float* a1v, float* feat, float* xpe, float* c0, float* c1,
                          float* alb, nsky_stream_t stream);
/* g_alb [N, 3] -> dpc2 [N, 4] (output layer's pre-activation gradient, row-major), dpc1 / dpc0 / dfeat (tile-native pre-activation
 * gradients: weight-gradient operands), dxpe [N, 40] (optional: gradient of the encode row's x / PE columns), da1v (tile-native: gradient
 * of a1's value rows from the feature rows), gmax [3] = max |dpc1|, |dpc0|, |dfeat| (caller zero-fills). */
int nsky_field_colour_bwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, int32_t N,
                          const float* g_alb, const float* alb, const float* c0, const float* c1, float* dpc2, float* dpc1, float* dpc0,
                          float* dfeat, float* dxpe, float* da1v, float* gmax, nsky_stream_t stream);
/* g_sdf [N], g_grad [N, 3], da1v, dxpe (each optional) -> d1q / d0q (quad-native pre-activation gradients of the two hidden layers),
 * dET [4 N, ldE] (stacked, optional), gmax [2] = max |d1q|, |d0q| (caller zero-fills). */
int nsky_field_geo_bwd(const nsky_field_net* net, const void* stream_buf, const float* scales, int32_t total_groups, int32_t N,
                       const float* g_sdf, const float* g_grad, const float* da1v, const float* dxpe, const float* a0q, const float* a1q,
                       float* d1q, float* d0q, float* dET, int32_t ldE, float* gmax, nsky_stream_t stream);
/* out[o][f] += sum_rows w[row][o] X[row][f] (o < n_out <= 4) over a tile-native X [rows, 32 nt]; bias[o] += sum_rows w[row][o] (optional):
 * the weight gradients of the field's narrow output layers.  w4 [rows, 4] row-major, or (w4 null, n_out 1) the quad form over a
 * quad-native X: w(row) = g_sdf[row / 4] on value rows (the only ones that count for the bias), g_grad[row / 4][row % 4 - 1] on tangent rows. */
int nsky_native_weighted_colsum(const float* X, int32_t nt, int32_t rows, const float* w4, int32_t n_out, const float* g_sdf,
                                const float* g_grad, float* out, int32_t ldo, float* bias, nsky_stream_t stream);

/* Per-ray reductions of the renderers, one pass each way: expected depth clipped to the global [min, max] of the sample mid
 * points and, if max_clamp > 0, to max_clamp (nerfstudio DepthRenderer('expected'); neusky_model.py:591, :1342-1353),
 * accumulation (:595), weighted normal (:812, :1357) and albedo on white (:813).  weights / starts / ends [R,S]; normals / albedo
 * [R,S,3] (optional, with their outputs); sums [R,8] and bounds [2] are scratch kept for the backward: the caller sets bounds to
 * (+inf, -inf) before the forward call. */
int nsky_ray_reduce_fwd(const float* weights, const float* starts, const float* ends, const float* normals, const float* albedo,
                        int32_t R, int32_t S, float max_clamp, float* sums, float* bounds, float* p2p, float* accumulation,
                        float* normal, float* albedo_acc, nsky_stream_t stream);
int nsky_ray_reduce_bwd(const float* weights, const float* starts, const float* ends, const float* normals, const float* albedo,
                        const float* sums, const float* bounds, int32_t R, int32_t S, float max_clamp, const float* d_p2p,
                        const float* d_accumulation, const float* d_normal, const float* d_albedo_acc, float* d_weights,
                        float* d_normals, float* d_albedo, nsky_stream_t stream);


/* NeuS alphas of P isolated samples for three interval lengths each (the hash-grid density probe, neusky_model.py:715-732:
 * nerfstudio SDFField.get_alpha with `deltas` = the three axis gaps broadcast against [P,1]): alphas [P,3]; backward to sdf [P],
 * gradients [P,3] and the variance parameter (d_variance += ; may be NULL).  gap3_host: three floats in HOST memory. */
int nsky_point_alphas_fwd(const float* sdf, const float* grad, const float* dirs, const float* gap3_host, const float* variance, float anneal,
                          int32_t P, float* alphas, nsky_stream_t stream);
int nsky_point_alphas_bwd(const float* sdf, const float* grad, const float* dirs, const float* gap3_host, const float* variance, float anneal,
                          int32_t P, const float* d_alphas, float* d_sdf, float* d_grad, float* d_variance, nsky_stream_t stream);

/* The DDF's output activation (neusky/fields/directional_distance_field.py:297-299): t [n] = scale sigmoid(raw[:, 0]) on the padded
 * [n, ld] head output of the chain; backward d_raw [n, ld] = (d_t scale s (1 - s) | 0 ...). */
int nsky_sigmoid_column_fwd(const float* raw, int32_t ld, int64_t n, float scale, float* t, nsky_stream_t stream);
int nsky_sigmoid_column_bwd(const float* raw, int32_t ld, int64_t n, float scale, const float* d_t, float* d_raw, nsky_stream_t stream);

/* The proposal networks' density MLP (nerfstudio HashMLPDensityField, hidden_dim 16: Linear(in_dim, 16) + ReLU -> Linear(16, 1); called
 * through ProposalNetworkSampler at neusky/models/neusky_model.py:561) on P hash-encoded rows feat [P, ldf]: raw [P] = the density
 * head's pre-activation, both layers in registers.  Backward: d_raw [P] -> d_feat [P, ldf] (optional; pad columns zeroed) and the
 * parameter gradients, ADDED into dw0 [16, ldw0] / db0 [16] / dw1 [16] / db1 [1] (caller zero-fills).  w0 [16, ldw0], w1 [16]: torch
 * nn.Linear layout; hidden must be 16, in_dim <= 12. */
int nsky_proposal_mlp_fwd(const float* feat, int32_t ldf, int64_t P, int32_t in_dim, int32_t hidden, const float* w0, int32_t ldw0,
                          const float* b0, const float* w1, const float* b1, float* raw, nsky_stream_t stream);
int nsky_proposal_mlp_bwd(const float* feat, int32_t ldf, int64_t P, int32_t in_dim, int32_t hidden, const float* w0, int32_t ldw0,
                          const float* b0, const float* w1, const float* b1, const float* d_raw, float* d_feat, float* dw0, float* db0,
                          float* dw1, float* db1, nsky_stream_t stream);
/* Points along rays: out[i] = origins[i] + sign t[i] dirs[i % n_dirs] (the DDF's predicted termination points: ddf_model.py:243,
 * neusky_model.py:1716-1724 with the R x Dv visibility rows sharing their Dv directions), and its backward
 * d_t[i] = sign <d_out[i], dirs[i % n_dirs]> (origins and directions carry no gradient on this path). */
int nsky_ray_points_fwd(const float* origins, const float* dirs, int32_t n_dirs, float sign, const float* t, int64_t n, float* out,
                        nsky_stream_t stream);
int nsky_ray_points_bwd(const float* dirs, int32_t n_dirs, float sign, const float* d_out, int64_t n, float* d_t, nsky_stream_t stream);
/* unit rows of g [P,3] (torch.nn.functional.normalize(p=2, eps=1e-12), sdf_albedo_field.py:256) and its backward */
int nsky_normalize3_fwd(const float* g, int64_t P, float* n, nsky_stream_t stream);
int nsky_normalize3_bwd(const float* g, const float* d_n, int64_t P, float* d_g, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sample generators (replace host-side torch RNG + upload in the reference).  Counter-based RNG (Philox4x32-10) keyed by
 * (seed, *counter); every call advances *counter (a device uint64 the caller owns) by one, so a replayed HIP graph draws
 * fresh numbers.  Distributions follow the reference, RNG streams do not.
 *
 * nsky_ddf_vmf_samples: VMFDDFSampler.generate_ddf_samples (neusky/model_components/ddf_sampler.py:249-286 with
 *   random_vmf :205-247): n_positions points on the unit sphere (z >= 0 when upper_hemisphere), each with n_directions
 *   directions from the vMF lobe (concentration kappa) about its inward normal.  origins [n_positions * n_directions, 3] =
 *   point * radius (repeated), directions likewise. */
int nsky_ddf_vmf_samples(int32_t n_positions, int32_t n_directions, float kappa, float radius, int32_t upper_hemisphere,
                         uint64_t seed, uint64_t* counter, float* origins, float* directions, nsky_stream_t stream);

/* nsky_ddf_fit_rows_fwd: the DDF evaluations of DDFModel.get_outputs (neusky/models/ddf_model.py:193-219 the fit rays,
 *   :279-321 one multi-view ray per fit ray, :324-360 the sky rays), as rows for the DDF network: E = N + (want_mv ? N : 0) + Ns
 *   rows of q_pos [E,3] (sphere positions) and xrow [E,ldx] = [d_loc | NeRF2(d_loc) | 0] (get_localised_transforms :158-181 +
 *   directional_distance_field.py:188-191,270-271).  mv_points_in null: the multi-view points are drawn here (seed, *counter,
 *   advanced by one); mv_points_out [N,3] receives the points used (z folded to >= 0).  sky_gt [Ns] = |o - sphere exit| (:343);
 *   distance_weight [N] (optional) = 1 - (|p| / radius)^weight_exp (:224-238).
 * nsky_ddf_fit_rows_bwd: d_term_dist [N] from d_xrow_mv = gradient of the N multi-view rows (the only differentiable input is the
 *   fit rays' termination distance, :287). */
int nsky_ddf_fit_rows_fwd(const float* positions, const float* directions, const float* term_dist, int32_t N,
                          const float* mv_points_in, uint64_t seed, uint64_t* counter, const float* sky_o, const float* sky_d,
                          int32_t Ns, float radius, int32_t want_mv, float weight_exp, int32_t weight_include_z, float* q_pos,
                          float* xrow, int32_t ldx, float* mv_points_out, float* sky_gt, float* distance_weight, nsky_stream_t stream);
int nsky_ddf_fit_rows_bwd(const float* positions, const float* directions, const float* term_dist, const float* mv_points,
                          int32_t N, const float* d_xrow_mv, int32_t ldx, float* d_term_dist, nsky_stream_t stream);


/* Probe points of the hash-grid density loss (neusky_model.py:704-724): positions [P,3] = lattice + (u gap - gap / 2), u ~ U(0,1)^3,
 * directions [P,3] uniform on the sphere; gap3_host: three floats in HOST memory (the cell size per axis); (seed, *counter): Philox,
 * the counter is advanced by one. */
int nsky_grid_probe_points(const float* lattice, const float* gap3_host, int32_t P, uint64_t seed, uint64_t* counter, float* positions,
                           float* directions, nsky_stream_t stream);

/* HDR output of the RENI++ decoder for the direction grid and the batch's own rays (neusky_model.py:488-549: exp output activation,
 * unnormalised by the per-image scale): raw [U D + R, ldr] (the chain's head output, 3 used columns) ->
 * grid [U D, 3] = exp(raw) scale[u], rays [R, 3] = exp(raw) scale[ray_latent[r]].  _bwd: d_raw [U D + R, ldr] (pad columns zeroed;
 * d_grid / d_rays may be NULL = zero), d_scale [U] += (caller zero-fills; NULL = not wanted). */
int nsky_reni_output_fwd(const float* raw, int32_t ldr, const float* scale, const int64_t* ray_latent, int32_t U, int32_t D, int32_t R,
                         float* grid, float* rays, nsky_stream_t stream);
int nsky_reni_output_bwd(const float* raw, int32_t ldr, const float* scale, const int64_t* ray_latent, int32_t U, int32_t D, int32_t R,
                         const float* d_grid, const float* d_rays, float* d_raw, float* d_scale, nsky_stream_t stream);

/* Illumination directions of one step: IcosahedronSampler with a random rotation (neusky/model_components/illumination_samplers.py:75-110,
 * drawn per step at neusky_model.py:456-458) and the upper-hemisphere subset of the rotated set (neusky_model.py:1650-1657).
 * base [D,3]; rotation_in: a 3x3 row-major matrix, or NULL = drawn from (seed, *counter) (Philox; the counter is advanced by one);
 * dirs [D,3] = base R^T; sel [D/2] = indices of the D/2 largest z, ascending (= the z > 0 subset of a centrally symmetric set);
 * rot_out [9] optional.  D even, <= 1024. */
int nsky_illumination_directions(const float* base, int32_t D, const float* rotation_in, uint64_t seed, uint64_t* counter, float* dirs,
                                 int32_t* sel, float* rot_out, nsky_stream_t stream);
/* RENI++ decoder inputs (the rotation-invariant representation of RENIField, as fed at neusky_model.py:1207-1252): for latent
 * codes Z [U,L,3], a direction set [D,3] and R further (direction, latent index) pairs -- the batch's own rays (:535-549) --
 * rows u D + d (every latent set against every direction), then U D + r, of cond [U D + R, ldcond] = [|Z_xy|, Z_z, Z_xy . d_xy] per
 * latent (3 L columns, pad columns zeroed) and of xrow [.., ldx] = [|d_xy|, d_z | NeRF2(2 freqs, 0..2) of those | 0].
 * _bwd: d_latents [U,L,3] (overwritten) from d_cond. */
int nsky_reni_grid_inputs_fwd(const float* latents, const float* directions, int32_t U, int32_t L, int32_t D, const float* ray_dirs,
                              const int64_t* ray_latent, int32_t R, float* cond, int32_t ldcond, float* xrow, int32_t ldx,
                              nsky_stream_t stream);
int nsky_reni_grid_inputs_bwd(const float* latents, const float* directions, int32_t U, int32_t L, int32_t D, const float* ray_dirs,
                              const int64_t* ray_latent, int32_t R, const float* d_cond, int32_t ldcond, float* d_latents,
                              nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multiresolution hash-grid encode (tiny-cuda-nn HashGrid semantics, fp32) fused with the rest of
 * the MLP input row.  Replaces tcnn.Encoding + NeRFEncoding + torch.cat at
 *   neusky/fields/sdf_albedo_field.py:119-130 (+ inherited forward_geonetwork, called :172,180,233)
 *   neusky/fields/directional_distance_field.py:146-156, :267-268
 * and, with `tangents`, the forward-mode input Jacobian that the reference obtains with
 * torch.autograd.grad(sdf, inputs, create_graph=True) (sdf_albedo_field.py:235-238).
 *
 * Row layout written to Y (ldy >= width, pad columns are zeroed):
 *   [ x (3, if include_x) | PE: sin(2 pi x_i 2^e_f) (3*pe_freqs), sin(.. + pi/2) (3*pe_freqs) | hash features (2*L) ]
 *   with e_f = f * pe_max_exp / (pe_freqs - 1)  (nerfstudio NeRFEncoding, min_freq_exp = 0)
 * mode: 0 = feed x raw to the grid (DDF); 1 = (contract_Linf(x)+2)/4; 2 = (contract_L2(x)+2)/4.
 * T (optional): three more row blocks [3][P][ldy], d(row)/d(x_k).
 */
typedef struct nsky_hashgrid_desc {
  const float* table;      /* [offset[L]][2] */
  int32_t n_levels;        /* <= 16, 2 features per level */
  int32_t smoothstep;      /* 1: Smoothstep interpolation, 0: Linear */
  float scale[16];
  int32_t resolution[16];
  uint32_t offset[17];     /* rows; level l owns rows offset[l]..offset[l+1] */
} nsky_hashgrid_desc;

int nsky_encode_fwd(const nsky_hashgrid_desc* g, const float* x, int32_t P, int32_t mode, int32_t include_x,
                    int32_t pe_freqs, float pe_max_exp, float* Y, int32_t ldy, float* T, nsky_stream_t stream);

/* Backward of nsky_encode_fwd.  dY [P,lddy] = gradient w.r.t. the rows; dT (optional) [3][P][lddy] =
 * gradient w.r.t. the tangent rows (second-order path: eikonal / normals).  Accumulates (float adds, order not
 * fixed) into dtable [offset[L]][2] (NULL: a frozen table, only dx is formed); dx (optional, [P,3], overwritten) = dY . d(row)/dx.
 * workspace (optional): nsky_encode_bwd_workspace_bytes(g, P, dT != NULL) bytes of scratch, 256-byte aligned; with it, and at
 * least NSKY_ENCODE_BWD_OWNER_MIN_POINTS points, the table gradient is accumulated by chunk-owning workgroups in LDS
 * instead of by global atomics (same result up to the order of the float adds). */
#define NSKY_ENCODE_BWD_OWNER_MIN_POINTS 32768
int64_t nsky_encode_bwd_workspace_bytes(const nsky_hashgrid_desc* g, int32_t P, int32_t tangents);
int nsky_encode_bwd(const nsky_hashgrid_desc* g, const float* x, int32_t P, int32_t mode, int32_t include_x,
                    int32_t pe_freqs, float pe_max_exp, const float* dY, int32_t lddy, const float* dT, float* dtable,
                    float* dx, void* workspace, nsky_stream_t stream);

/* corner rows (uint32 [P][L][8]) exactly as the encode kernels address the table - the integer
 * part of the hash encode, exposed for BIT-EXACT parity tests. */
int nsky_hash_indices(const nsky_hashgrid_desc* g, const float* x, int32_t P, int32_t mode, uint32_t* idx,
                      nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hemisphere integral + alpha composite + sRGB on COMPACT inputs.  Replaces
 * RGBLambertianRendererWithVisibility.render_and_combine_rgb, neusky/model_components/renderers.py:60-130
 * (and the [R*S,D,*] broadcasts built at neusky/models/neusky_model.py:512-525, 1755-1759).
 *   albedo, normals [R,S,3]; weights [R,S]; dirs [D,3]; cam_colours [U,D,3]; cam_of_ray [R] (row of cam_colours);
 *   vis [R,D] or NULL; bg [R,3]  ->  rgb [R,3] (sRGB, clamped to [0,1]); lin [R,3] (optional: linear composite,
 *   needed by the backward).  D <= 1024.
 */
int nsky_hemi_composite_fwd(const float* albedo, const float* normals, const float* weights, const float* dirs,
                            const float* cam_colours, const int32_t* cam_of_ray, const float* vis, const float* bg,
                            int32_t R, int32_t S, int32_t D, float* rgb, float* lin, nsky_stream_t stream);
/* d_cam_colours [U,D,3] is ACCUMULATED (atomic adds; zero it first); d_vis / d_cam_colours may be NULL. */
int nsky_hemi_composite_bwd(const float* albedo, const float* normals, const float* weights, const float* dirs,
                            const float* cam_colours, const int32_t* cam_of_ray, const float* vis, const float* bg,
                            const float* lin, const float* d_rgb, int32_t R, int32_t S, int32_t D, float* d_albedo,
                            float* d_normals, float* d_weights, float* d_cam_colours, float* d_vis, float* d_bg,
                            nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NeuS alpha -> transmittance -> weights (+ accumulation, expected depth).  Replaces nerfstudio
 * SDFField.get_alpha (called neusky/fields/sdf_albedo_field.py:266), RaySamples.get_weights_and_
 * transmittance_from_alphas (neusky/models/neusky_model.py:565-568) and the accumulation / expected-depth
 * renderers (:591-595).  sdf, starts, ends [R,S]; grad [R,S,3]; ray_dirs [R,3]; variance: device scalar
 * (LearnedVariance parameter; inv_s = clip(exp(10 v), 1e-6, 1e6)).  S <= 256.
 * depth is sum(w mid)/(sum(w)+1e-10) WITHOUT the batch-global clip to [min mid, max mid] (host applies it).
 */
int nsky_neus_weights_fwd(const float* sdf, const float* grad, const float* ray_dirs, const float* starts,
                          const float* ends, const float* variance, float cos_anneal, int32_t R, int32_t S,
                          float* alpha, float* weights, float* trans_bg, float* accumulation, float* depth,
                          nsky_stream_t stream);
/* d_variance (device scalar) is ACCUMULATED. */
int nsky_neus_weights_bwd(const float* sdf, const float* grad, const float* ray_dirs, const float* starts,
                          const float* ends, const float* variance, float cos_anneal, int32_t R, int32_t S,
                          const float* d_weights, const float* d_trans_bg, float* d_sdf, float* d_grad,
                          float* d_variance, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DDF sky-visibility ray set-up and sigmoid.  Replaces the geometry of NeuSkyFactoModel.compute_visibility,
 * neusky/models/neusky_model.py:1667-1709 (surface point incl. the outside-sphere fix-up :1674-1683,
 * ray_sphere_intersection :1590-1622) + DDFModel.get_localised_transforms / einsum, neusky/models/ddf_model.py:158-200
 * + the NeRF direction encoding, neusky/fields/directional_distance_field.py:188-191,270-271.
 *   origins, ray_dirs [R,3]; depth [R]; sel_dirs [Dv,3] (the upper-hemisphere subset, :1650-1657)
 *   -> sphere_pts [R*Dv,3]; xrow [R*Dv,ldx] = [d_loc | NeRF2(d_loc) | 0]; surf_dist [R*Dv] = min(|x-p|, 2r) (:1724-1727);
 *      term_dist [R*Dv] (optional, :1697).
 */
int nsky_visibility_rays(const float* origins, const float* ray_dirs, const float* depth, const float* sel_dirs,
                         int32_t R, int32_t Dv, float radius, float* sphere_pts, float* xrow, int32_t ldx,
                         float* surf_dist, float* term_dist, nsky_stream_t stream);
/* vis[r, sel_index[j]] = 1 - sigmoid(scale * (surf_dist - t_hat - threshold))  (:1730-1740); the caller pre-fills
 * vis [R,D] with the lower-hemisphere constant (:1745-1748).  threshold: device scalar. */
int nsky_visibility_finish_fwd(const float* t_hat, const float* surf_dist, const float* threshold, float scale,
                               const int32_t* sel_index, int32_t R, int32_t Dv, int32_t D, float* vis,
                               nsky_stream_t stream);
/* d_threshold (device scalar) is ACCUMULATED. */
int nsky_visibility_finish_bwd(const float* t_hat, const float* surf_dist, const float* threshold, float scale,
                               const int32_t* sel_index, int32_t R, int32_t Dv, int32_t D, const float* d_vis,
                               float* d_t_hat, float* d_threshold, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Interlevel (proposal) loss, nerfstudio interlevel_loss as neusky/models/neusky_model.py:987-988 calls it.
 *   c [R,S+1] final spacing bins, w [R,S] final weights (both constants), sb [R,n+1] / wp [R,n] one proposal level.
 *   per_ray[r] = sum_s max(w_s - w_outer_s, 0)^2 / (w_s + 1e-7)   (the loss term is mean over R*S = sum(per_ray)/(R S))
 * Backward: d_per_ray [R] -> d_wp [R,n] (every element written).  n <= 4096.
 */
int nsky_interlevel_fwd(const float* c, const float* w, const float* sb, const float* wp, int32_t R, int32_t S, int32_t n,
                        float* per_ray, nsky_stream_t stream);
int nsky_interlevel_bwd(const float* c, const float* w, const float* sb, const float* wp, const float* d_per_ray, int32_t R,
                        int32_t S, int32_t n, float* d_wp, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Ray set-up of the proposal sampler (nerfstudio SphereCollider + UniformSampler + the spacing->euclidean map and
 * bin mid-points that ProposalNetworkSampler applies between levels; driven from neusky/models/neusky_model.py:213,561).
 * No gradient flows through any of these.
 *   nsky_sphere_collider : origins, directions [R,3] -> nears, fars [R] against the sphere |x| = radius
 *                          (nears >= near_plane, fars >= nears + 1e-6; rays that miss get [near_plane, near_plane + 1e-6])
 *   nsky_uniform_bins    : sbins [R,n+1] = linspace(0,1,n+1) jittered inside its own bin by jitter[R] (NULL: plain
 *                          lattice), ebins = sbins * far + (1 - sbins) * near
 *   nsky_bins_to_samples : sbins [R,n+1] -> ebins [R,n+1] (optional) and positions [R,n,3] = o + d (e_i + e_{i+1}) / 2
 *                          (optional)
 */
int nsky_sphere_collider(const float* origins, const float* directions, int32_t R, float radius, float near_plane, float* nears,
                         float* fars, nsky_stream_t stream);
int nsky_uniform_bins(const float* nears, const float* fars, const float* jitter, int32_t R, int32_t n, float* sbins, float* ebins,
                      nsky_stream_t stream);
int nsky_bins_to_samples(const float* sbins, const float* nears, const float* fars, const float* origins, const float* directions,
                         int32_t R, int32_t n, float* ebins, float* positions, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Proposal-network weights: nerfstudio HashMLPDensityField's trunc_exp density + RaySamples.get_weights, as the proposal
 * sampler drives them (neusky/models/neusky_model.py:561).  raw [R*n] with element stride ld_raw (the padded output of
 * the 1-wide density head), ebins [R,n+1] euclidean bin edges -> weights [R,n] =
 * nan_to_num((1 - exp(-delta sigma)) exp(-cumsum_excl(delta sigma))), sigma = exp(raw).  n <= 256.
 * Backward writes d_raw with the same stride (the ld_raw-1 pad columns are zero-filled).
 */
int nsky_density_weights_fwd(const float* raw, int32_t ld_raw, const float* ebins, int32_t R, int32_t n, float* weights,
                             nsky_stream_t stream);
int nsky_density_weights_bwd(const float* raw, int32_t ld_raw, const float* ebins, const float* d_weights, int32_t R, int32_t n,
                             float* d_raw, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Reverse-over-forward step of one Softplus layer that carries three input tangents.  Together with
 * nsky_gemm_f32 (NSKY_EPI_MUL_AUX) this replaces the double backward that the reference gets from
 * torch.autograd.grad(..., create_graph=True) at neusky/fields/sdf_albedo_field.py:235-238 feeding the eikonal
 * loss (neusky/models/neusky_model.py:958-960) and the shading normals (:251).
 *   s = sigmoid(beta z) [N,ld]; ta = tangent activations [3][N,ld]; da (optional) [N,ld];
 *   dta [3][N,ld]  OR  the outer product ggrad[N,3] x wvec[C];   out: dz [N,ld], du [3][N,ld];
 *   wsum (optional, outer-product form only) [C] += sum_{n,k} ggrad[n,k] ta_k[n,:] = the gradient of wvec, from the same pass.
 */
int nsky_softplus_tangent_bwd(const float* da, const float* s, const float* ta, const float* dta, const float* ggrad,
                              const float* wvec, float beta, int32_t N, int32_t C, int32_t ld, float* dz, float* du,
                              float* wsum, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Proposal PDF re-sampling: nerfstudio PDFSampler.generate_ray_samples as driven by
 * ProposalNetworkSampler (called at neusky/models/neusky_model.py:561).  weights [R,n0] (already annealed),
 * bins [R,n0+1] (spacing domain), u_base [nb] = linspace(0, 1-1/nb, nb) (+ 1/(2nb) when not stratified),
 * jitter [R] in [0,1) or NULL  ->  new_bins [R,nb], inds [R,nb] (optional; searchsorted(cdf,u,right)).
 * The CDF is accumulated sequentially in fp32, so inds are bit-reproducible against the oracle.
 */
int nsky_pdf_sample(const float* weights, const float* bins, const float* u_base, const float* jitter, int32_t R,
                    int32_t n0, int32_t nb, float histogram_padding, float eps, float* new_bins, int32_t* inds,
                    nsky_stream_t stream);

/*
 * Weight-norm parameterisation of the SDF / colour layers (nn.utils.weight_norm as nerfstudio's SDFField applies it;
 * used by neusky/fields/sdf_albedo_field.py:147-161 for the colour net and the inherited geo net), fused with the row /
 * column re-ordering and zero padding the consuming GEMMs want:
 *   out[r][c] = g[sr] * v[sr][sc] / ||v[sr]||_2,  sr = row_map[r], sc = col_map[c]; -1 marks a structural zero.
 * v [out_features, in_features] (ldv), g [out_features], inv_norm [out_features] (written; saved for the backward).
 * Every source row must appear at most once in row_map, every source column at most once in col_map.
 * Backward: d_out [n_rows_out, ldo] -> dv [out_features, ldv-strided rows listed in row_map], dg [out_features];
 * inverse_col[sc] = output column reading source column sc, or -1.  Rows of v absent from row_map are not written.
 */
int nsky_weight_norm_fwd(const float* v, const float* g, int32_t n_rows_out, int32_t n_cols_out, int32_t in_features,
                         int32_t ldv, const int32_t* row_map, const int32_t* col_map, float* out, int32_t ldo,
                         float* inv_norm, nsky_stream_t stream);
int nsky_weight_norm_bwd(const float* d_out, int32_t ldo, const float* v, const float* g, const float* inv_norm,
                         int32_t n_rows_out, int32_t in_features, int32_t ldv, const int32_t* row_map,
                         const int32_t* inverse_col, float* dv, float* dg, nsky_stream_t stream);

/* Copies n_segments float runs (src -> dst, n floats each) in one launch: the parameter gradients autograd keeps in tensors of its
 * own (torch's AccumulateGrad with an undefined .grad) into their places in the optimizer's gradient slab, instead of one add
 * kernel per parameter (nerfstudio Optimizers.zero_grad_all / optimizer_scheduler_step_all around neusky_config.py:216-237). */
typedef struct { const float* src; float* dst; int64_t n; } nsky_segment;
int nsky_gather_segments(const nsky_segment* segments, int32_t n_segments, nsky_stream_t stream);
/* The same for byte runs of any element type (host arrays of device pointers and byte counts): the next step's input tensors into the
 * static buffers a captured graph reads (nerfstudio's datamanager.next_train hand-over), one launch instead of one copy per tensor. */
int nsky_copy_segments(const void* const* src, void* const* dst, const int64_t* nbytes, int32_t n_segments, nsky_stream_t stream);

/* Adam update (torch.optim.Adam semantics, no weight decay / amsgrad) over a flat slab of n floats;
 * the five optimizer groups of neusky/configs/neusky_config.py:216-237.  grad_scale multiplies g first.  The hyper-parameters are
 * DOUBLES (ABI 16): torch forms 1 - beta, lr / (1 - beta1^t) and sqrt(1 - beta2^t) in double from the Python scalars and rounds each to
 * float once; a float beta2 = 0.999 already puts 1 - beta2 off by 1.3e-5 relative. */
int nsky_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                   double eps, int32_t step, float grad_scale, nsky_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Closed-form loss terms, one launch each way per model (instead of ~300 small torch launches per step).
 * NeuSky model, train branch of get_loss_dict, neusky/models/neusky_model.py:933-1035 (+ RENISkyPixelLoss,
 * neusky/model_components/losses.py:44-58, linear_to_sRGB neusky/utils/utils.py:25-30, nerfstudio monosdf_normal_loss):
 *   terms[8] = { rgb_l1 (:947-950), eikonal (:958-960), fg_mask BCE (:963-967), hashgrid density L1 (:990-993),
 *                ground-plane normal (:995-1000), sky pixel (:1002-1009), visibility-threshold MSE (:1011-1030),
 *                sdf level set (:1032-1035) }, UNSCALED (the coefficients are applied by the caller); a term whose input
 *   pointer is NULL is left 0.  wsum [R] (side output) = per-ray weight sums, needed by the backward.
 *   Backward: d_terms[8] -> gradients (every element written) for the inputs whose output pointer is non-NULL.
 */
typedef struct nsky_main_losses_desc {
  int32_t R, S, P, M;     /* rays, samples per ray, hash-grid probe points, sdf-at-termination rows */
  const float* rgb;       /* [R,3] rendered radiance */
  const float* image;     /* [R,3] */
  const float* mask;      /* [R,4] float {static, fg, ground, sky} (neusky_dataset.py:290) */
  const float* eik;       /* [R*S,3] sdf gradients */
  const float* weights;   /* [R,S] */
  const float* normal;    /* [R,3] rendered normals */
  const float* hdr_bg;    /* [R,3] linear HDR background */
  const float* grid;      /* [P*3] hash-grid probe alphas */
  const float* sdf_term;  /* [M] */
  const float* vis_thr;   /* [1] learnable visibility threshold */
  float sky_alpha, vis_target;
} nsky_main_losses_desc;
int nsky_main_losses_fwd(const nsky_main_losses_desc* d, float* terms, float* wsum, nsky_stream_t stream);
int nsky_main_losses_bwd(const nsky_main_losses_desc* d, const float* wsum, const float* d_terms, float* d_rgb, float* d_eik,
                         float* d_weights, float* d_normal, float* d_hdr_bg, float* d_grid, float* d_sdf_term, float* d_vis_thr,
                         nsky_stream_t stream);


/* Scalar training metrics of a model in one launch: out[0] = 10 log10(peak_sq / mean(((pred - gt) mask)^2)) over n elements (mask
 * optional: PSNR of neusky_model.py:1066-1068 with peak 1; depth PSNR of ddf_model.py:381-405 with peak = the DDF radius and the batch
 * mask); with variance (the NeuS deviation parameter, 1 float): out[1] = clip(exp(10 v), 1e-6, 1e6), out[2] = 1 / out[1]
 * (neusky_model.py:1071-1072).  None of it is differentiated. */
int nsky_train_metrics(const float* pred, const float* gt, const float* mask, int64_t n, float peak_sq, const float* variance, float* out,
                       nsky_stream_t stream);
/* Attention core of the RENI++ transformer decoder -- the illumination model the reference configures (neusky/configs/neusky_config.py:78-95:
 * conditioning="Attention", VN invariance, SO2 about z, 8 heads x 6 layers, hidden 128) and decodes at neusky/models/neusky_model.py:488-506,
 * 535-549.  The `reni` package holding the decoder is absent from the reference tree: the arithmetic follows the published architecture as
 * restated in oracle/neusky_oracle.py:reni_attention_decode (PARITY UNPINNED).  For every camera u, direction d and head h (head width 16):
 *     q~ = scale * [d_x q | d_y q | q],   s_n = q~ . K~[u,h,n],   p = softmax_n(s),   o = d_x (p V~)[0:16] + d_y (p V~)[16:32] + (p V~)[32:48]
 * with K~, V~ [U, n_heads, L, 48] the per-camera key / value parts (the token is linear in (d_x, d_y): three 16-wide parts per token, L <= 128),
 * Q, O, dO, dQ [U, D, 16 n_heads] row-major, dirs [U, D, 3], row_max / row_sum [U, n_heads, D] (saved by the forward for the backward).
 * Replaces the per-head `softmax(Q K^T) V` of the decoder's cross-attention (two batched matrix products, a softmax and their autograd nodes per
 * layer).  Arithmetic: D >= 32 rows per camera: matrix cores, fp32-grade fp16 hi + residual products (the chain kernels' arithmetic); a camera's
 * short blocks (a ray's own row): exact fp32 on the vector units.  _bwd returns dQ and the per-camera dK~, dV~ (summed over the camera's D rows); directions carry no gradient. */
int nsky_attn_core_fwd(const float* Q, const float* dirs, const float* Kt, const float* Vt, int32_t U, int32_t D, int32_t L, int32_t n_heads,
                       float scale, float* O, float* row_max, float* row_sum, nsky_stream_t stream);
int nsky_attn_core_bwd(const float* Q, const float* dirs, const float* Kt, const float* Vt, const float* O, const float* row_max,
                       const float* row_sum, const float* dO, int32_t U, int32_t D, int32_t L, int32_t n_heads, float scale, float* dQ,
                       float* dKt, float* dVt, float* drow_scratch /* U n_heads (D + 4) floats: D = dO . O per row and head, then the operand maxima of every (camera, head) */,
                       nsky_stream_t stream);
/* The rays' own rows of the same decoder (the background radiance along a ray's own direction, neusky_model.py:535-549): a ray attends to
 * the keys / values of ITS camera.  Q, O, dO, dQ [R, 16 n_heads], dirs [R, 3], row_max / row_sum [R, n_heads]; the rays sorted by camera:
 * perm[i] = the ray at sorted position i, seg[u] .. seg[u + 1] the positions of camera u (seg [U + 1]).  _bwd ADDS the rays' part to
 * dKt / dVt (which hold the grid rows' sums from nsky_attn_core_bwd, launched earlier on the same stream).  Exact fp32 on the vector units.
 * Replaces decoding the rays as R one-direction cameras (which projects R x 3 L token rows per layer). */
int nsky_attn_core_rays_fwd(const float* Q, const float* dirs, const int32_t* perm, const int32_t* seg, const float* Kt, const float* Vt, int32_t U,
                            int32_t R, int32_t L, int32_t n_heads, float scale, float* O, float* row_max, float* row_sum, nsky_stream_t stream);
int nsky_attn_core_rays_bwd(const float* Q, const float* dirs, const int32_t* perm, const int32_t* seg, const float* Kt, const float* Vt, const float* O,
                            const float* row_max, const float* row_sum, const float* dO, int32_t U, int32_t R, int32_t L, int32_t n_heads, float scale,
                            float* dQ, float* dKt, float* dVt, nsky_stream_t stream);
/* Residual add + layer norm of the decoder's row stream ([M, W] row-major, W = 64, 128, 256 or 512): s = x + r (r NULL: s = x, not stored),
 * y = LayerNorm(s) gamma + beta (biased variance, eps inside the root: torch.nn.LayerNorm), stats [M, 2] = (mean, rstd) of every row.
 * _bwd, for frozen gamma / beta (the RENI++ decoder is fixed, neusky_config.py:87): ds = d LayerNorm / d s (dy) + ds_in (ds_in NULL: none) --
 * the gradient of x and of r alike.  Replace torch's add + native_layer_norm (+ their backward nodes) of every decoder block. */
int nsky_add_layer_norm_fwd(const float* x, const float* r, const float* gamma, const float* beta, int64_t M, int32_t W, float eps, float* s, float* y,
                            float* stats, nsky_stream_t stream);
int nsky_add_layer_norm_bwd(const float* s, const float* stats, const float* gamma, const float* dy, const float* ds_in, int64_t M, int32_t W, float* ds,
                            nsky_stream_t stream);
/* The step's objective: total = sum over segments of scale_s * sum_i coef_s[i] x_s[i] (coef NULL: 1).  Replaces the dozen scalar
 * multiplies, sums and adds that scale and merge the loss dictionaries (nerfstudio scale_dict + functools.reduce(torch.add, ...),
 * neusky_pipeline.py:283-289; interlevel_loss' mean, neusky_model.py:987-988) by one launch each way.  One workgroup; bwd writes
 * grad_s[i] = g[0] * scale_s * coef_s[i] for every segment whose grad pointer is set. */
#define NSKY_TOTAL_MAX_SEGMENTS 8
typedef struct nsky_total_segment {
  const float* x;      /* [n] */
  const float* coef;   /* [n] or NULL */
  float* grad;         /* [n] or NULL (bwd) */
  int32_t n;
  float scale;
} nsky_total_segment;
int nsky_weighted_total_fwd(const nsky_total_segment* segments, int32_t n_segments, float* total, nsky_stream_t stream);
int nsky_weighted_total_bwd(const nsky_total_segment* segments, int32_t n_segments, const float* g, nsky_stream_t stream);
/* DDF model, get_loss_dict, neusky/models/ddf_model.py:407-493:
 *   terms[5] = { depth L1 x scene-centre weight (:427-433), sdf L2, sdf L1, multi-view hinge^2 with the reference's
 *                [M] - [M,1] -> [M,M] broadcast (:475-483), sky-ray L1 (:485-490) }, unscaled. */
typedef struct nsky_ddf_losses_desc {
  int32_t Mr, Mm, Ms;          /* fit rays, multi-view rays, sky rays */
  const float* expected;       /* [Mr] expected termination distance */
  const float* term;           /* [Mr] target termination distance */
  const float* mask;           /* [Mr] */
  const float* dist_weight;    /* [Mr] or NULL */
  const float* sdf;            /* [Mr] sdf at termination or NULL */
  const float* mv_expected; const float* mv_term;    /* [Mm] */
  const float* sky_expected; const float* sky_term;  /* [Ms] */
  int32_t want_depth, want_sdf_l2, want_sdf_l1, mask_to_circumference, inverse_depth_weight;
  float radius;
} nsky_ddf_losses_desc;
int nsky_ddf_losses_fwd(const nsky_ddf_losses_desc* d, float* terms, nsky_stream_t stream);
/* d_term / d_mv_term: gradients w.r.t. the target distances (the field's own rendering of the fit rays; NULL when the pipeline
 * detached them, neusky_pipeline.py stop_sdf_gradients) */
int nsky_ddf_losses_bwd(const nsky_ddf_losses_desc* d, const float* d_terms, float* d_expected, float* d_sdf, float* d_mv_expected,
                        float* d_sky_expected, float* d_term, float* d_mv_term, nsky_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUSKY_HIP_H */
