// Shared machinery of the chain kernels (film_chain.hip: FiLM-SIREN and the sdf value chain; field_chain.hip: the SDF / albedo
// field): packed weight stream (one tile = 32 output features x K as fp16 hi + residual planes in MFMA-fragment order),
// the LDS ring it is streamed through by global_load_lds_dwordx4, the register-resident product, the fp16 split helpers and
// the tile-native activation layout.  Everything here is device-inline code in an anonymous namespace: each translation unit
// gets its own copy.
#pragma once
#include "common.h"
#include "../../include/neusky_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SLAB = 2048;
constexpr int GSLABS = 8;
constexpr int GROUP = SLAB * GSLABS;  // 16 KB
constexpr int RING_GROUPS = 8;
constexpr int RING_BYTES = GROUP * RING_GROUPS;  // 128 KB
constexpr int BIAS_FLOATS = 6144;                // 24 KB: every bias of the network
constexpr int SCALE_FLOATS = 512;                // reciprocal tile scales, stream order
constexpr int MAXL = NSKY_FILM_MAX_LAYERS;
constexpr float Y_SCALE = 16384.0f;              // sine outputs live in [-1, 1]: fixed power-of-two scale
constexpr int PACK_KMAX = 320;

__host__ __device__ inline int ksteps_of(int K) { return (K + 15) / 16; }
__host__ __device__ inline int groups_of(int K) { return (ksteps_of(K) + GSLABS - 1) / GSLABS; }

struct TileDesc {
  const float* W;
  int ld, row0, nrows, K, transposed, k0;
  long group;
  int slab0 = 0;  // first slab inside the group (tiles of K <= 64 may share a group: product_pair)
};

// one block per tile: absmax -> power-of-two scale -> fp16 hi / residual planes in fragment order (tile = blockIdx.x)
__device__ __forceinline__ void pack_tile(const TileDesc& d, unsigned char* __restrict__ stream, float* __restrict__ scales,
                                          float (*w)[PACK_KMAX + 1], float* red) {
  const int tid = threadIdx.x;
  const int Kp = ksteps_of(d.K) * 16;
  float m = 0.0f;
  for (int idx = tid; idx < 32 * Kp; idx += 256) {
    const int r = idx / Kp, k = idx % Kp;
    float v = 0.0f;
    if (r < d.nrows && k < d.K) v = d.transposed ? d.W[(long)(d.k0 + k) * d.ld + d.row0 + r] : d.W[(long)(d.row0 + r) * d.ld + k];
    w[r][k] = v;
    m = fmaxf(m, fabsf(v));
  }
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  m = red[0];
  float sc = 1.0f;
  if (m > 0.0f && m < 3.0e38f) {
    int e;
    (void)frexpf(m, &e);  // m < 2^e
    e = max(-100, min(100, e));
    sc = ldexpf(1.0f, 15 - e);  // m sc < 2^15
  }
  if (tid == 0) scales[blockIdx.x] = 1.0f / sc;
  unsigned char* base = stream + d.group * GROUP + (long)d.slab0 * SLAB;
  const int KS = ksteps_of(d.K);
  for (int idx = tid; idx < KS * 64; idx += 256) {
    const int ks = idx >> 6, lane = idx & 63;
    const int r = lane & 31, h = lane >> 5;
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = w[r][16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)] * sc;
      const _Float16 xh = (_Float16)x;
      hi[j] = xh;
      lo[j] = (_Float16)(x - (float)xh);
    }
    *reinterpret_cast<f16x8*>(base + (long)ks * SLAB + lane * 16) = hi;
    *reinterpret_cast<f16x8*>(base + (long)ks * SLAB + 1024 + lane * 16) = lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  // LDS-DMA hidden from hipcc's waitcnt bookkeeping; M0 saved and restored inside the statement; completion counted by hand
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Weight stream consumer.  Invariant while slabs of group g are being consumed: every piece of groups <= g + 1 has landed and
// all four waves know it (barrier).  Fragment reads run TWO k-steps ahead of the MFMAs (three register buffers, rotated at
// compile time inside a product) and one k-step ahead across a tile boundary, so an LDS round trip (~150-200 cycles under
// load) hides behind 6 MFMAs; the waits are counted (lgkmcnt(2): only the youngest pair may still be in flight; LDS returns
// in order, so a scalar load the compiler slips in can only make the wait stricter).
// Transition g -> g + 1 (after this wave's last fragment of group g has arrived): wait for the wave's own pieces of group g + 2
// (at most RING_GROUPS - 3 younger groups x 4 pieces outstanding; every other vector-memory operation of the wave only makes the
// wait longer), barrier (all pieces of g + 2 landed; nobody reads group g any more), refill the slot of g with group g + 8.
// LDS reads in flight across the barrier belong to group g + 1: never the slot being refilled.
struct WStream {
  const unsigned char* src;  // this lane's source address inside group 0 (stream + wave * 4096 + lane * 16)
  uint32_t dst;              // this wave's destination inside ring slot 0 (lds0 + wave * 4096)
  uint32_t lds_lane;         // lds0 + lane * 16
  int g;                     // group of the slab whose fragments are carried in (ch, cl): slab 0 of the next tile
  f16x8 ch, cl;              // fragments requested ahead of the next product (landed: every product settles them at its end)
  int total, sg;             // WRAP streams (persistent workgroups walk the stream again for every row tile): groups in the
                             // stream, and the source group the next DMA issue reads (wraps to 0 after total - 1)
};

// PW = 1 KB pieces of a 16 KB group this wave moves: 4 with four waves per workgroup, 2 with eight
// RG: groups in the LDS ring (a power of two; 8 = 128 KB with one workgroup per CU, 4 = 64 KB with two)
template <int PW = 4, int RG = RING_GROUPS>
__device__ __forceinline__ void ws_issue(const WStream& w, int group) {
  const unsigned char* s = w.src + (long)group * GROUP;
  const uint32_t d = w.dst + (uint32_t)(group & (RG - 1)) * GROUP;
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    glds16(s + p * 1024, d + p * 1024);
  }
}

// the same for a stream that is walked cyclically: the ring slot follows the running group count, the source wraps
template <int PW = 4, int RG = RING_GROUPS>
__device__ __forceinline__ void ws_issue_wrap(WStream& w, int group) {
  const unsigned char* s = w.src + (long)w.sg * GROUP;
  const uint32_t d = w.dst + (uint32_t)(group & (RG - 1)) * GROUP;
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    glds16(s + p * 1024, d + p * 1024);
  }
  w.sg = w.sg + 1 == w.total ? 0 : w.sg + 1;
}

// The two fragment registers are read-write operands of BOTH the request and the wait: the compiler sees one value that is
// modified in place, so it neither renames it nor copies it while the LDS read is still in flight (a copy of a register whose
// load has not landed would capture stale data; cdna_hip_programming.md section 5.7 item 1).
__device__ __forceinline__ void frag_read(f16x8& h, f16x8& l, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "+v"(h), "+v"(l) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_wait(f16x8& h, f16x8& l) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(h), "+v"(l) : "n"(N) : "memory");
}
// The LAST wait of a product and the hand-over of the landed fragments to the registers that carry them into the next product, as
// ONE statement.  A tied ("+v") operand lets the register allocator satisfy the tie with a copy IN FRONT of the statement when the
// value has to change registers anyway (here it does: three buffers rotate over KS k-steps, the carried pair must end where the
// next product expects buffer 0), and a copy in front of the wait captures whatever the destination of the in-flight ds_read held:
// the field kernels of round 3 were built that way (v_mov_b64 x 4, then s_waitcnt lgkmcnt(0)) and gave one wave a stale first
// k-step of its next tile in about one launch of six (tools/isa_lint.py checks every kernel for the pattern).
// Here the moves are the statement's own, behind the wait; sources and destinations are separate (early-clobber) operands.
__device__ __forceinline__ void frag_settle(f16x8& ch, f16x8& cl, const f16x8& h, const f16x8& l) {
  typedef unsigned long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 sh = __builtin_bit_cast(u64x2, h), sl = __builtin_bit_cast(u64x2, l);
  unsigned long h0, h1, l0, l1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b64 %0, %4\n\tv_mov_b64 %1, %5\n\tv_mov_b64 %2, %6\n\tv_mov_b64 %3, %7"
               : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1) : "v"(sh[0]), "v"(sh[1]), "v"(sl[0]), "v"(sl[1]) : "memory");
  ch = __builtin_bit_cast(f16x8, u64x2{h0, h1});
  cl = __builtin_bit_cast(f16x8, u64x2{l0, l1});
}

template <int RG = RING_GROUPS>
__device__ __forceinline__ uint32_t ws_addr(const WStream& w, int group, int slab) {
  return w.lds_lane + (uint32_t)(group & (RG - 1)) * GROUP + slab * SLAB;
}

template <int PW = 4>
__device__ __forceinline__ void ws_begin(WStream& w) {
#pragma unroll
  for (int g = 0; g < RING_GROUPS; ++g) ws_issue<PW>(w, g);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PW * (RING_GROUPS - 2)) : "memory");  // groups 0 and 1 landed
  w.g = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { w.ch[j] = (_Float16)0.0f; w.cl[j] = (_Float16)0.0f; }
  frag_read(w.ch, w.cl, ws_addr(w, 0, 0));
  frag_wait<0>(w.ch, w.cl);
}

template <int PW = 4, int RG = RING_GROUPS>
__device__ __forceinline__ void ws_begin_wrap(WStream& w, int total_groups) {
  w.total = total_groups;
  w.sg = 0;
#pragma unroll
  for (int g = 0; g < RG; ++g) ws_issue_wrap<PW, RG>(w, g);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PW * (RG - 2)) : "memory");  // groups 0 and 1 landed
  w.g = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { w.ch[j] = (_Float16)0.0f; w.cl[j] = (_Float16)0.0f; }
  frag_read(w.ch, w.cl, ws_addr<RG>(w, 0, 0));
  frag_wait<0>(w.ch, w.cl);
}

template <int PW = 4, bool WRAP = false, int RG = RING_GROUPS>
__device__ __forceinline__ void ws_transition(WStream& w, int from_group) {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PW * (RG - 3)) : "memory");
  if (WRAP) ws_issue_wrap<PW, RG>(w, from_group + RG);
  else ws_issue<PW, RG>(w, from_group + RG);
}

// what a wave WITHOUT a row tile does in place of a product (the last round of a persistent workgroup): its share of the group
// hand-shakes (counted wait, barrier, DMA pieces of the refill) and nothing else -- the same number of barriers as product<KS>
template <int KS, int PW, bool WRAP, int RG = RING_GROUPS>
__device__ __forceinline__ void product_skip(WStream& w) {
  constexpr int NG = (KS + GSLABS - 1) / GSLABS;
  const int g0 = w.g;
#pragma unroll
  for (int i = 0; i < NG; ++i) ws_transition<PW, WRAP, RG>(w, g0 + i);
  w.g = g0 + NG;
}

// acc += W_tile X over KS k-steps, B planes in registers.  One wave per SIMD issues in order, so everything that is not an
// MFMA is placed in the shadow of one: the LDS requests of k-step ks + NB - 1 (and a group transition: barrier + 4 DMA pieces) right
// behind the first MFMA of k-step ks, the counted wait for the fragments of ks + 1 behind the third.
// NB fragment buffers rotate: reads run NB - 1 k-steps ahead of the MFMAs (NB = 4 where a wave has the registers: the last wait of a
// product, for slab 0 of the next tile, then has had 9 MFMAs to be satisfied instead of 6).
template <int KS, int PW = 4, bool WRAP = false, int RG = RING_GROUPS, int NB = 3>
__device__ __forceinline__ void product(WStream& w, const f16x8 (&bh)[KS], const f16x8 (&bl)[KS], f32x16& acc) {
  constexpr int NG = (KS + GSLABS - 1) / GSLABS;  // groups of this tile; slab index KS stands for slab 0 of the next tile
  constexpr int AHEAD = NB - 1;
  f16x8 fh[NB], fl[NB];
  fh[0] = w.ch;
  fl[0] = w.cl;
#pragma unroll
  for (int b = 1; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) { fh[b][j] = (_Float16)0.0f; fl[b][j] = (_Float16)0.0f; }
  const int g0 = w.g;
  auto request = [&](int s) {  // s static
    if (s < KS) frag_read(fh[s % NB], fl[s % NB], ws_addr<RG>(w, g0 + s / GSLABS, s % GSLABS));
    else frag_read(fh[s % NB], fl[s % NB], ws_addr<RG>(w, g0 + NG, 0));
  };
#pragma unroll
  for (int s = 1; s < AHEAD; ++s)
    if (s <= KS) request(s);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % NB], bl[ks], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + AHEAD <= KS) request(ks + AHEAD);  // into the buffer of k-step ks - 1, whose MFMAs have been issued
    if ((ks & (GSLABS - 1)) == GSLABS - 1 || ks == KS - 1) ws_transition<PW, WRAP, RG>(w, g0 + ks / GSLABS);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks % NB], bh[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % NB], bh[ks], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // fragments of k-step ks + 1 (slab KS = first slab of the next tile): the younger pairs (slabs ks + 2 .. min(ks + AHEAD, KS)) may
    // still be in flight
    static_assert(AHEAD == 2 || AHEAD == 3, "two or three k-steps of read-ahead");
    if (ks < KS - 1) {  // (ks is a constant after unrolling: one of the two waits survives)
      if (AHEAD == 3 && ks + 3 <= KS) frag_wait<4>(fh[(ks + 1) % NB], fl[(ks + 1) % NB]);
      else frag_wait<2>(fh[(ks + 1) % NB], fl[(ks + 1) % NB]);
    }
  }
  // the last wait (lgkmcnt(0)) and the carry in one statement: nothing is in flight behind it, no load crosses a loop back-edge or
  // a branch join.  (Even when the carried slab sits in buffer 0 -- NB = 4, KS a multiple of 4 -- the register allocator does not
  // keep buffer 0 in the carry's registers and would copy in front of a tied wait: tools/isa_lint.py caught exactly that.)
  frag_settle(w.ch, w.cl, fh[KS % NB], fl[KS % NB]);
  w.g = g0 + NG;
}

// Two tiles of K = 64 (four k-steps each) sharing one group and one set of B planes: slabs 0..3 accumulate into a0, slabs 4..7 into
// a1.  The accumulators are NOT zeroed: the mapping backward's head product adds k-block after k-block into them.
template <int PW = 4, int RG = RING_GROUPS>
__device__ __forceinline__ void product_pair(WStream& w, const f16x8 (&bh)[4], const f16x8 (&bl)[4], f32x16& a0, f32x16& a1) {
  constexpr int KS = 8;
  f16x8 fh[3], fl[3];
  fh[0] = w.ch;
  fl[0] = w.cl;
#pragma unroll
  for (int j = 0; j < 8; ++j) { fh[1][j] = fh[2][j] = fl[1][j] = fl[2][j] = (_Float16)0.0f; }
  const int g0 = w.g;
  auto request = [&](int s) {  // s static
    if (s < KS) frag_read(fh[s % 3], fl[s % 3], ws_addr<RG>(w, g0, s));
    else frag_read(fh[s % 3], fl[s % 3], ws_addr<RG>(w, g0 + 1, 0));
  };
  auto step = [&](int ks, f32x16& acc) {  // ks static
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % 3], bl[ks & 3], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 <= KS) request(ks + 2);
    if (ks == KS - 1) ws_transition<PW, false, RG>(w, g0);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks % 3], bh[ks & 3], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % 3], bh[ks & 3], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 <= KS) frag_wait<2>(fh[(ks + 1) % 3], fl[(ks + 1) % 3]);
  };
  request(1);
  step(0, a0); step(1, a0); step(2, a0); step(3, a0);
  step(4, a1); step(5, a1); step(6, a1); step(7, a1);
  frag_settle(w.ch, w.cl, fh[KS % 3], fl[KS % 3]);
  w.g = g0 + 1;
}

// the same over the first ksn (wave-uniform, run time; >= 1) of KS k-steps: the mapping network's first layer.  No read-ahead
// inside the tile (every step sits in its own branch); 8 short tiles per row tile.
template <int KS>
__device__ __forceinline__ void product_dyn(WStream& w, int ksn, const f16x8 (&bh)[KS], const f16x8 (&bl)[KS], f32x16& acc) {
  const int g0 = w.g;
  const int ng = (ksn + GSLABS - 1) / GSLABS;
  f16x8 ah = w.ch, al = w.cl;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks < ksn) {
      const bool last = ks == ksn - 1;
      if ((ks & (GSLABS - 1)) == GSLABS - 1 || last) ws_transition(w, g0 + ks / GSLABS);
      f16x8 nh = ah, nl = al;
      frag_read(nh, nl, last ? ws_addr(w, g0 + ng, 0) : ws_addr(w, g0 + (ks + 1) / GSLABS, (ks + 1) % GSLABS));
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      frag_settle(ah, al, nh, nl);
    }
  }
  w.ch = ah;
  w.cl = al;
  w.g = g0 + ng;
}

__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 xh = (_Float16)x[j];
    hi[j] = xh;
    lo[j] = (_Float16)(x[j] - (float)xh);
  }
}

// power-of-two scale s with m s < 2^15 (m = largest magnitude of the row); returns s, inv = 1 / s
__device__ __forceinline__ float row_scale(float m, float& inv) {
  m = fmaxf(m, __shfl_xor(m, 32, 64));  // lanes l and l ^ 32 hold the two halves of one batch row
  if (!(m > 0.0f) || !(m < 3.0e38f)) {
    inv = 1.0f;
    return 1.0f;
  }
  int e;
  (void)frexpf(m, &e);
  e = max(-100, min(100, e));
  inv = ldexpf(1.0f, e - 15);
  return ldexpf(1.0f, 15 - e);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4_nt(const float* p) {  // bypasses this CU's vector L1 (served by the XCD's L2)
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// B planes of KS k-steps from an fp32 row [dim] (dim % 4 == 0, columns >= dim read as 0): returns 1 / scale
template <int KS>
__device__ __forceinline__ float load_planes(const float* __restrict__ rowp, int dim, int ksn, int h, f16x8 (&ph)[KS], f16x8 (&pl)[KS]) {
  float v[KS][8];
  float m = 0.0f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int feat = 16 * ks + 8 * u + 4 * h;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ks < ksn && feat < dim) q = ldg4(rowp + feat);
      v[ks][4 * u] = q.x; v[ks][4 * u + 1] = q.y; v[ks][4 * u + 2] = q.z; v[ks][4 * u + 3] = q.w;
      m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
  float inv;
  const float s = row_scale(m, inv);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[ks][j] * s;
    split8(x, ph[ks], pl[ks]);
  }
  return inv;
}

// sin with Cody-Waite reduction to [-pi/4, pi/4] + minimax polynomials (|err| < 2e-7 for |x| < 1e4), cos alongside
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
  const float k = rintf(x * 0.6366197723675814f);
  float r = fmaf(-k, 1.5707962513e+00f, x);
  r = fmaf(-k, 7.5497894159e-08f, r);
  r = fmaf(-k, 5.3903029534e-15f, r);
  const float r2 = r * r;
  float sp = fmaf(r2, 2.7183114939e-06f, -1.9839334836e-04f);
  sp = fmaf(sp, r2, 8.3333293855e-03f);
  sp = fmaf(sp, r2, -1.6666666567e-01f);
  sp = fmaf(sp * r2, r, r);
  float cp = fmaf(r2, 2.4433157117e-05f, -1.3887316255e-03f);
  cp = fmaf(cp, r2, 4.1666645683e-02f);
  cp = fmaf(cp, r2, -0.5f);
  cp = fmaf(cp, r2, 1.0f);
  const int q = (int)k;
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// One of the two alone (the chain forward needs the sine, the FiLM backward the cosine): reduction by multiples of pi to
// [-pi/2, pi/2] (two-term Cody-Waite: the fused multiply-adds keep the products exact, the third term of pi is k 1e-15), ONE
// polynomial, the sign from the parity of k.  14 instructions instead of 23; |err| < 1.4e-7 on the reduced range (fitted and
// checked in float32 arithmetic), the reduction adds |k| 1e-15.
__device__ __forceinline__ float sin_cw(float x) {
  const float k = rintf(x * 0.31830988618379067f);
  float r = fmaf(-k, 3.14159274101257324f, x);
  r = fmaf(-k, -8.74227766e-08f, r);
  const float r2 = r * r;
  float p = fmaf(r2, 2.6348141091e-06f, -1.9822760078e-04f);
  p = fmaf(p, r2, 8.3332424983e-03f);
  p = fmaf(p, r2, -1.6666665673e-01f);
  const float s = fmaf(p * r2, r, r);
  return __int_as_float(__float_as_int(s) ^ ((int)k << 31));
}
__device__ __forceinline__ float cos_cw(float x) {
  const float k = rintf(x * 0.31830988618379067f);
  float r = fmaf(-k, 3.14159274101257324f, x);
  r = fmaf(-k, -8.74227766e-08f, r);
  const float r2 = r * r;
  float p = fmaf(r2, -2.6297973932e-07f, 2.4774602934e-05f);
  p = fmaf(p, r2, -1.3888651738e-03f);
  p = fmaf(p, r2, 4.1666660458e-02f);
  p = fmaf(p, r2, -0.5f);
  const float c = fmaf(p, r2, 1.0f);
  return __int_as_float(__float_as_int(c) ^ ((int)k << 31));
}

// Tile-native activation layout ("native"): the [rows, width] matrix is cut into 32-row x 32-feature blocks of 4 KB, block
// (R, t) at float offset (R * (width / 32) + t) * 1024, and inside a block element (row c, feature f) sits at
// (f / 8) * 256 + (c + 32 * ((f / 4) & 1)) * 4 + (f & 3): exactly the accumulator layout of v_mfma_f32_32x32x16 (register
// 4 g + q of lane (c, h) = feature 8 g + 4 h + q of batch row c), so a wave stores / loads a tile with four 1 KB-contiguous
// float4 instructions and the lane that stored a piece is the lane that reads it back.  Rows are padded to a multiple of 32.
__device__ __forceinline__ void store_tile(float* blk, int lane, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) stg4(blk + g * 256 + lane * 4, make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]));
}
// a tile nothing reads again before the backward (saved h / z): non-temporal, it need not displace the weight stream in L2
__device__ __forceinline__ void store_tile_nt(float* blk, int lane, const float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    __builtin_nontemporal_store(f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}, reinterpret_cast<f32x4*>(blk + g * 256 + lane * 4));
}
__device__ __forceinline__ void load_tile(const float* blk, int lane, float (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = ldg4(blk + g * 256 + lane * 4);
    v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
  }
}

// largest magnitude of a gradient matrix -> device scalar (the weight-gradient GEMM pre-scales its fp16 split with it)
__device__ __forceinline__ void publish_max(float* slot, float v, bool live_row, bool wave_live, int lane) {
  v = live_row ? v : 0.0f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  if (wave_live && lane == 0 && v > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}

__device__ __forceinline__ void hidden_load4(f32x4& q, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(q) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void hidden_wait(f32x4 (&q)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void hidden_wait8(f32x4 (&p)[4], f32x4 (&q)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(N) : "memory");
}
// a load of data THIS lane stored earlier in the launch: bypass the CU's vector L1 (nt), served by the XCD's L2
__device__ __forceinline__ void hidden_load4_nt(f32x4& q, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(q) : "v"(p) : "memory");
}

// planes of KS k-steps from NT tiles this wave stored (scaled by the row maximum m): the lane that stored a piece loads it
template <int NT>
__device__ __forceinline__ float planes_from_tiles(const float* blk, int lane, float m, f16x8 (&ph)[2 * NT], f16x8 (&pl)[2 * NT]) {
  float inv;
  const float s = row_scale(m, inv);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
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
  return inv;
}

__device__ __forceinline__ float softplus_b(float v, float beta, float inv_beta) {  // torch.nn.functional.softplus(beta, threshold 20)
  const float bv = beta * v;
  const float t = __expf(-fabsf(bv));
  const float u = 1.0f + t, um1 = u - 1.0f;
  const float l = um1 == 0.0f ? t : __logf(u) * (t * __builtin_amdgcn_rcpf(um1));
  return bv > 20.0f ? v : (fmaxf(bv, 0.0f) + l) * inv_beta;
}

}  // namespace
