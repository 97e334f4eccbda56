// zk_attn_dev.h -- device side of the MFMA attention kernels (d = 64): argument block, relative-position helpers,
// LDS staging, and the tile functions `attn_fwd_tile` / `attn_bwd_fused64_tile` over a caller-provided LDS buffer.
// Included by zk_attn.hip (one tile per workgroup launch) and zk_layer.hip (the per-XCD layer program calls the same
// functions phase after phase: bit-identical results).  See zk_attn.hip for the design notes.
#pragma once
#include "zk_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  int ldq, ldk, ldv;
  long bsq, bsk, bsv;      // batch strides (elements) of q / k / v
  int kv_group;            // key/value (and kmask) batch index = b / kv_group (beam-tiled queries)
  int B, nh, Lq, Lk, d;
  const float* kmask;      // [B, Lk] 1 = valid key, 0 = pad; may be null
  int causal; int q_pos0;  // absolute position of query row 0 (decode: the time step)
  float scale; float mask_inf;
  const bf16_t* rpr_k; const bf16_t* rpr_v; int max_rel;   // [2*max_rel+1, d] tables or null
  uint32_t thr; float inv_keep; const uint64_t* seed; uint32_t sid;
  int ldmask;              // row stride of kmask (= the host-side Lk)
  // decode step replayed from a hipGraph: the time step lives in device memory.  pos_flags bit 0:
  // q_pos0 = *pos_dev; bit 1: only keys 0 .. *pos_dev are valid (self-attention over the cache)
  const int* pos_dev; int pos_flags;
  // Relative-position terms on the MFMA path (modules/rpr.py:10-75), decomposed so that the table
  // products are plain GEMMs done by the caller.  All four are [B*Lq, ldg] (gq, gd fp32; pb, dsb bf16) with the entry of
  // (token t, head h, relative index r) at t*ldg + h*nrp + r:
  //   gq  in : Q_h . Rk^T   -> gathered into the scores      gd  in : dO_h . Rv^T -> gathered into dP
  //   pb  out: sum over keys with relative index r of P (after dropout)   -> O += pb . Rv, dRv = pb^T dO
  //   dsb out: the same bucket sums of dS                                  -> dQ += dsb . Rk, dRk = dsb^T Q
  const float* gq; const float* gd; bf16_t* pb; bf16_t* dsb; int ldg; int nrp;
};

__device__ __forceinline__ void attn_apply_pos(AttnArgs& a) {
  if (a.pos_dev != nullptr) {
    const int t = *a.pos_dev;
    if (a.pos_flags & 1) a.q_pos0 = t;
    if (a.pos_flags & 2) a.Lk = min(a.Lk, t + 1);
  }
}

__device__ __forceinline__ int rel_index(int i_abs, int j, int max_rel) {
  int dlt = i_abs - j;                         // modules/rpr.py:66-75
  dlt = dlt < -max_rel ? -max_rel : (dlt > max_rel ? max_rel : dlt);
  return dlt + max_rel;
}

// additive mask of key j for query i (absolute position), reference semantics
__device__ __forceinline__ float mask_bias(const AttnArgs& a, int b, int i_abs, int j) {
  float bias = 0.f;
  if (a.kmask != nullptr && a.kmask[(size_t)(b / a.kv_group) * a.ldmask + j] == 0.f) bias -= a.mask_inf;
  if (a.causal && j > i_abs) bias -= a.mask_inf;
  return bias;
}

// Bucket sums over the relative index for the 16 query rows a wave owns.  `val(row, j)` reads entry
// (local row, key j) of a [64 x Lk] LDS tile; keys j in [0, Lk).  Interior indices 0 < r < 2m pick the
// single key j = i_abs - (r - m); r = 0 collects j >= i_abs + m, r = 2m collects j <= i_abs - m
// (the clipped tails of modules/rpr.py:66-75); r > 2m is padding (0).
// One wave, 16 query rows: `emit(row, r, value)` receives every bucket r in [0, nr) of the wave's local rows
// (row in [w*16, w*16+16)).  The interior indices cost one LDS read each; the two clipped tails -- sums over up to Lk
// keys -- are split over four lanes per row (16 keys each per 64-key tile) and combined with two shuffles instead of
// one lane walking the whole row (that loop made the relative-position attention kernels 3.5x slower than the plain
// ones).  nr > 2m+1: padding buckets, emitted as 0.  All 64 lanes must call it.
template <typename F, typename E>
__device__ __forceinline__ void rpr_bucket_wave(const AttnArgs& a, int nr, int i0, int w, int lane, F val, E emit) {
  const int m = a.max_rel;
  // tails
  {
    const int row = w * 16 + (lane >> 2), q = lane & 3;
    const int i = i0 + row, ia = a.q_pos0 + i;
    float t0 = 0.f, t1 = 0.f;
    if (i < a.Lq) {
      for (int jb = q * 16; jb < a.Lk; jb += 64) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
          const int j = jb + u;
          if (j < a.Lk) {
            const float p = val(row, j);
            if (j >= ia + m) t0 += p;
            if (j <= ia - m) t1 += p;
          }
        }
      }
    }
    t0 = quad_sum(t0);
    t1 = quad_sum(t1);
    if (q == 0 && i < a.Lq) { emit(row, 0, t0); if (m > 0) emit(row, 2 * m, t1); }
  }
  // interior indices and padding.  e / nr without the ~40-instruction integer division of a runtime divisor (nine
  // iterations x two calls per kernel): (e * ceil(2^16 / nr)) >> 16 is exact for nr <= 64 and e < 1024
  const int inv16 = (65536 + nr - 1) / nr;
  for (int e = lane; e < 16 * nr; e += 64) {
    const int qrow = (e * inv16) >> 16;
    const int row = w * 16 + qrow, r = e - qrow * nr;
    const int i = i0 + row;
    if (i >= a.Lq || r == 0 || r == 2 * m) continue;
    float acc = 0.f;
    if (r < 2 * m) {
      const int j = a.q_pos0 + i - (r - m);
      if (j >= 0 && j < a.Lk) acc = val(row, j);
    }
    emit(row, r, acc);
  }
}
// Bucket sums over the relative index for the 16 query rows a wave owns, written to the global [.., ldg] layout.
// `val(row, j)` reads entry (local row, key j) of a [64 x Lk] LDS tile; keys j in [0, Lk).
template <typename F>
__device__ __forceinline__ void rpr_bucket_rows(const AttnArgs& a, bf16_t* __restrict__ dst, int b, int h, int i0,
                                                int w, int lane, F val) {
  rpr_bucket_wave(a, a.nrp, i0, w, lane, val, [&](int row, int r, float v) {
    dst[((size_t)b * a.Lq + i0 + row) * a.ldg + h * a.nrp + r] = f2bf(v);
  });
}
// Multi-tile form: entry u of this lane (e = lane + 64 u) accumulates the keys [j0, j0 + 64) of the tile
// currently in LDS; `val(row, jl)` reads local key jl.  MAXU * 64 >= 16 * nrp entries per wave.
#define RPR_MAXU 16
template <typename F>
__device__ __forceinline__ void rpr_bucket_accum(const AttnArgs& a, float (&acc)[RPR_MAXU], int i0, int j0, int w,
                                                 int lane, F val) {
  const int m = a.max_rel;
#pragma unroll
  for (int u = 0; u < RPR_MAXU; ++u) {
    const int e = lane + 64 * u;
    if (e >= 16 * a.nrp) break;
    const int row = w * 16 + e / a.nrp, r = e % a.nrp;
    const int i = i0 + row;
    if (i >= a.Lq) continue;
    const int ia = a.q_pos0 + i;
    const int jend = min(j0 + 64, a.Lk);
    if (r > 0 && r < 2 * m) {
      const int j = ia - (r - m);
      if (j >= j0 && j < jend) acc[u] += val(row, j - j0);
    } else if (r == 0) {
      for (int j = max(ia + m, j0); j < jend; ++j) acc[u] += val(row, j - j0);
    } else if (r == 2 * m) {
      for (int j = min(ia - m, jend - 1); j >= j0; --j) acc[u] += val(row, j - j0);
    }
  }
}
__device__ __forceinline__ void rpr_bucket_store(const AttnArgs& a, const float (&acc)[RPR_MAXU], bf16_t* __restrict__ dst,
                                                 int b, int h, int i0, int w, int lane) {
#pragma unroll
  for (int u = 0; u < RPR_MAXU; ++u) {
    const int e = lane + 64 * u;
    if (e >= 16 * a.nrp) break;
    const int i = i0 + w * 16 + e / a.nrp;
    if (i < a.Lq) dst[((size_t)b * a.Lq + i) * a.ldg + h * a.nrp + e % a.nrp] = f2bf(acc[u]);
  }
}
__device__ __forceinline__ float rpr_gather(const AttnArgs& a, const float* __restrict__ tab, int b, int h, int i,
                                            int j) {
  return tab[((size_t)b * a.Lq + i) * a.ldg + h * a.nrp + rel_index(a.q_pos0 + i, j, a.max_rel)];
}

// =====================================================================================
// MFMA kernels (d = 64)
// =====================================================================================
// make ATTNTRACE=1: workgroup (0, 0, 0) of the single-tile backward kernels stamps the constant 100 MHz clock at its phase
// boundaries (scripts/attn_bwd_trace.py); nothing in the default build
#ifdef ZK_ATTN_TRACE
extern __device__ unsigned long long zk_attn_trace_buf[16];
#define ZK_AT(i)                                                                                                     \
  do {                                                                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)                                   \
      zk_attn_trace_buf[i] = __builtin_amdgcn_s_memrealtime();                                                       \
  } while (0)
#else
#define ZK_AT(i)
#endif
#define AD 64            // head dim
#define ALD 72           // LDS row stride (bf16): 144 B rows -> conflict-free b128 fragment reads
#define TQ 64            // query / key tile

__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// fragment of a [rows][ALD] LDS tile: row r0+(lane&15), 8 consecutive k at kk*32+8*(lane>>4)
__device__ __forceinline__ uint4 frag(const bf16_t* t, int r0, int kk, int lane) {
  return *reinterpret_cast<const uint4*>(t + (r0 + (lane & 15)) * ALD + kk * 32 + (lane >> 4) * 8);
}
// stage rows [row0, row0+64) x 64 channels (channel-contiguous in HBM) -> dst[64][ALD]; rows >= nrows -> 0
template <bool FRESH = false>
__device__ __forceinline__ void stage_direct(bf16_t* dst, const bf16_t* src, int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    const int r = t >> 3, c = (t & 7) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row0 + r < nrows) v = zk_ld16<FRESH>(src + (size_t)(row0 + r) * ld + c);
    *reinterpret_cast<uint4*>(dst + r * ALD + c) = v;
  }
}
__device__ __forceinline__ uint32_t half_of4(const uint4& v, int i) {
  const uint32_t w = (i >> 1) == 0 ? v.x : ((i >> 1) == 1 ? v.y : ((i >> 1) == 2 ? v.z : v.w));
  return (i & 1) ? (w >> 16) : (w & 0xffffu);
}
// transposed staging: dst[phys(c)][r] = src[row0+r][c]; channel rows stored in permuted order
// phys(c) = (c%8)*8 + c/8 so that the 8-byte LDS writes of neighbouring lanes hit different banks.
template <bool FRESH = false>
__device__ __forceinline__ void stage_trans(bf16_t* dst, const bf16_t* src, int ld, int row0, int nrows, int tid) {
  if (tid < 128) {
    const int dc = tid & 7, rq = tid >> 3;   // channel chunk (8 channels), row quad (4 rows)
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = row0 + rq * 4 + u;
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      if (r < nrows) v[u] = zk_ld16<FRESH>(src + (size_t)r * ld + dc * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint2 o;
      o.x = half_of4(v[0], i) | (half_of4(v[1], i) << 16);
      o.y = half_of4(v[2], i) | (half_of4(v[3], i) << 16);
      *reinterpret_cast<uint2*>(dst + (i * 8 + dc) * ALD + rq * 4) = o;
    }
  }
}
__device__ __forceinline__ int chan_of_phys(int q) { return (q & 7) * 8 + (q >> 3); }
// The same staging in two halves -- all global loads of a prologue are issued before the first LDS store, so
// their latencies overlap instead of adding up (one stage_* call after another measured ~1500 cycles each).
// Branch-free: rows past the end are clamped for the address and zeroed by a select.
struct DirectRegs { uint4 v[2]; };
template <bool FRESH = false>
__device__ __forceinline__ void load_direct(DirectRegs& R, const bf16_t* src, int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    const int r = row0 + (t >> 3), c = (t & 7) * 8;
    const uint4 v = zk_ld16<FRESH>(src + (size_t)min(r, nrows - 1) * ld + c);
    R.v[it] = r < nrows ? v : make_uint4(0u, 0u, 0u, 0u);
  }
}
__device__ __forceinline__ void store_direct(bf16_t* dst, const DirectRegs& R, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    *reinterpret_cast<uint4*>(dst + (t >> 3) * ALD + (t & 7) * 8) = R.v[it];
  }
}
struct TransRegs { uint4 v[4]; };
template <bool FRESH = false>
__device__ __forceinline__ void load_trans(TransRegs& R, const bf16_t* src, int ld, int row0, int nrows, int t128) {
  const int dc = t128 & 7, rq = t128 >> 3;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = row0 + rq * 4 + u;
    const uint4 v = zk_ld16<FRESH>(src + (size_t)min(r, nrows - 1) * ld + dc * 8);
    R.v[u] = r < nrows ? v : make_uint4(0u, 0u, 0u, 0u);
  }
}
__device__ __forceinline__ void store_trans(bf16_t* dst, const TransRegs& R, int t128) {
  const int dc = t128 & 7, rq = t128 >> 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint2 o;
    o.x = half_of4(R.v[0], i) | (half_of4(R.v[1], i) << 16);
    o.y = half_of4(R.v[2], i) | (half_of4(R.v[3], i) << 16);
    *reinterpret_cast<uint2*>(dst + (i * 8 + dc) * ALD + rq * 4) = o;
  }
}
// [64 rows][64 channels] bf16 tile in LDS (row stride ALD) -> global rows [0, nrows), 16 bytes per thread and store
__device__ __forceinline__ void store_tile_rows(const bf16_t* tile, bf16_t* dst, size_t ld, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    const int r = t >> 3, c = (t & 7) * 8;
    if (r < nrows) *reinterpret_cast<uint4*>(dst + (size_t)r * ld + c) = *reinterpret_cast<const uint4*>(tile + r * ALD + c);
  }
}

__device__ __forceinline__ float quad16_max(float v) {
  v = row16_max(v);
  return v;
}
__device__ __forceinline__ float quad16_sum(float v) {
  v = row16_sum(v);
  return v;
}

// ---- forward: grid (ceil(Lq/64), nh, B); NKT = ceil(Lk/64) <= 4
// LDS bytes of one forward tile: sQ, sK (K tile, later V^T tile), sP
// (+ with relative positions folded in: the key table [r][channel], the value table transposed [channel][r], the
//  bucket sums of P [query][r] -- bf16 tiles of 64 x ALD -- and G = Q.Rk^T as fp32 [query][GLD])
#define GLD 68
template <int NKT, bool RPR = false>
struct AttnFwdLds {
  static constexpr int BASE = (2 * TQ * ALD + TQ * (NKT * 64 + 8)) * 2;
  static constexpr int BYTES = BASE + (RPR ? 3 * TQ * ALD * 2 + TQ * GLD * 4 : 0);
};
// one (64-query tile qt, head h, sentence b) problem by a 256-thread workgroup (the caller has applied attn_apply_pos); smem: AttnFwdLds<NKT>::BYTES, 16-byte
// aligned, the workgroup's ONLY LDS object (shared with whatever ran before: the function starts with a barrier-free
// overwrite, callers separate it from earlier readers of smem with a workgroup barrier)
// RPR (modules/rpr.py:10-75 folded into the tile): with only 2*max_rel+1 <= 64 distinct table rows, q.R_k[idx(i,j)] is
// a gather from G = Q_h.Rk^T (64 x 64, one MFMA pass over the LDS-resident table) and sum_j P[i,j] R_v[idx(i,j)] is
// PB.Rv with PB[i,r] = the sum of P over the keys of relative index r -- no products through HBM, no extra launches.
// PRE (zk_proj_attn_out_ln: the projection ran in this workgroup): 1 = the Q tile, 3 = the Q and K tiles and V^T
// (`vt_pre`, the layout store_trans leaves: [physical channel][key], rows >= Lk zero) are already in the LDS, written behind a
// workgroup barrier by the caller -- nothing of them is loaded here.  PRE = 3 needs NKT = 1 (self-attention of one tile).
template <int NKT, bool FRESH = false, bool RPR = false, int PRE = 0>
__device__ __forceinline__ void attn_fwd_tile(unsigned char* smem, const AttnArgs& a, bf16_t* __restrict__ out, int ldo,
                                              float* __restrict__ lse, int qt, int h, int b,
                                              const bf16_t* vt_pre = nullptr) {
  static_assert(PRE == 0 || PRE == 1 || (PRE == 3 && NKT == 1), "PRE = 3 is the single-tile self-attention");
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sK = sQ + TQ * ALD;     // K tile, later V^T tile
  bf16_t* sP = sK + TQ * ALD;
  constexpr int PLD = NKT * 64 + 8;
  [[maybe_unused]] bf16_t* sRk = reinterpret_cast<bf16_t*>(smem + AttnFwdLds<NKT, false>::BASE);
  [[maybe_unused]] bf16_t* sRvT = sRk + TQ * ALD;
  [[maybe_unused]] bf16_t* sPB = sRvT + TQ * ALD;
  [[maybe_unused]] float* sG = reinterpret_cast<float*>(sPB + TQ * ALD);
  [[maybe_unused]] const int nrel = 2 * a.max_rel + 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i0 = qt * TQ;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)(b / a.kv_group) * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)(b / a.kv_group) * a.bsv + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  // the loads that do not depend on anything computed here are issued together: Q, the first K tile, the first
  // V tile (kept in registers until P is ready) and the key mask -- one memory round trip instead of four
  DirectRegs rQ, rK;
  [[maybe_unused]] DirectRegs rRk;
  TransRegs rVt;
  if (PRE == 0) load_direct<FRESH>(rQ, qb, a.ldq, i0, a.Lq, tid);
  if (PRE != 3) {
    load_direct<FRESH>(rK, kb, a.ldk, 0, a.Lk, tid);
    if (tid < 128) load_trans<FRESH>(rVt, vb, a.ldv, 0, a.Lk, tid);
  }
  if (RPR) {
    // the two tables ride in the same round trip (staged one after the other behind the Q / K stores they cost the
    // forward two more L2 latencies: 17.6 us per launch against 10.4 without relative positions)
    load_direct(rRk, a.rpr_k, AD, 0, nrel, tid);
    if (tid >= 128) load_trans(rVt, a.rpr_v, AD, 0, nrel, tid - 128);
  }
  float kbias[NKT * 4];
#pragma unroll
  for (int t = 0; t < NKT * 4; ++t) {
    const int j = t * 16 + (lane & 15);
    kbias[t] = (a.kmask != nullptr && a.kmask[(size_t)(b / a.kv_group) * a.ldmask + min(j, a.Lk - 1)] == 0.f) ? -a.mask_inf : 0.f;
  }
  if (PRE == 0) store_direct(sQ, rQ, tid);
  if (PRE != 3) store_direct(sK, rK, tid);
  if (RPR) {
    store_direct(sRk, rRk, tid);                          // [r][channel], rows >= nrel zero
    if (tid >= 128) store_trans(sRvT, rVt, tid - 128);    // [physical channel][r]
  }
  f32x4_t S[NKT * 4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt > 0) {
      __syncthreads();   // previous tile's readers are done with sK
      stage_direct<FRESH>(sK, kb, a.ldk, kt * 64, a.Lk, tid);
    }
    __syncthreads();
    const uint4 q0 = frag(sQ, w * 16, 0, lane), q1 = frag(sQ, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16(q0, frag(sK, nt * 16, 0, lane), acc);
      acc = mfma16(q1, frag(sK, nt * 16, 1, lane), acc);
      S[kt * 4 + nt] = acc;
    }
  }
  if (RPR) {
    // G rows of this wave's 16 queries (only this wave reads them back)
    const uint4 q0 = frag(sQ, w * 16, 0, lane), q1 = frag(sQ, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16(q0, frag(sRk, nt * 16, 0, lane), acc);
      acc = mfma16(q1, frag(sRk, nt * 16, 1, lane), acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sG[(w * 16 + (lane >> 4) * 4 + r) * GLD + nt * 16 + (lane & 15)] = acc[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  // scale + mask; C layout: col (key) = lane&15, row (query) = (lane>>4)*4 + reg
  const int rbase = i0 + w * 16 + (lane >> 4) * 4;
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int t = 0; t < NKT * 4; ++t) {
    const int j = t * 16 + (lane & 15);
    const bool kvalid = j < a.Lk;
    const float kb_ = kvalid ? kbias[t] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float raw = S[t][r];
      if (a.gq != nullptr && kvalid && rbase + r < a.Lq) raw += rpr_gather(a, a.gq, b, h, rbase + r, j);
      if (RPR && kvalid) raw += sG[(w * 16 + (lane >> 4) * 4 + r) * GLD + rel_index(a.q_pos0 + rbase + r, j, a.max_rel)];
      float s = raw * a.scale + kb_;
      if (a.causal && j > a.q_pos0 + rbase + r) s -= a.mask_inf;
      s = kvalid ? s : -INFINITY;
      S[t][r] = s;
      mx[r] = fmaxf(mx[r], s);
    }
  }
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) mx[r] = quad16_max(mx[r]);
#pragma unroll
  for (int t = 0; t < NKT * 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(S[t][r] - mx[r]);
      S[t][r] = e;
      sum[r] += e;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sum[r] = quad16_sum(sum[r]);
    const int i = rbase + r;
    if ((lane & 15) == 0 && i < a.Lq && lse != nullptr)
      lse[((size_t)b * a.nh + h) * a.Lq + i] = mx[r] + __logf(sum[r]);
    sum[r] = 1.f / sum[r];
  }
  // P (bf16, after dropout) -> sP rows owned by this wave
  // RPR: PB[i][r] = the sum of P over the keys of relative index r, for this wave's rows, straight from these registers:
  // an interior bucket 0 < r < 2m has exactly one key (the entry IS the bucket: a scattered 2-byte store), the two clipped
  // tails are sums over the row (partials per lane, a 16-lane reduction, one store).  Rows beyond Lq and buckets beyond 2m
  // stay zero: the wave clears its 16 x 64 slab first (LDS operations of one wave execute in order).  [A second walk over
  // sP -- rpr_bucket_wave -- cost the backward 6 of its 18 us, profiles/r03_attn_bwd_phases_v1.txt.]
  [[maybe_unused]] float tlo[4] = {0.f, 0.f, 0.f, 0.f}, thi[4] = {0.f, 0.f, 0.f, 0.f};
  if (RPR) {
    for (int e = lane; e < 16 * 8; e += 64)
      *reinterpret_cast<uint4*>(sPB + (w * 16 + (e >> 3)) * ALD + (e & 7) * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int t = 0; t < NKT * 4; ++t) {
    const int j = t * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = S[t][r] * sum[r];
      if (a.thr) {
        const uint64_t idx = (((uint64_t)b * a.nh + h) * a.Lq + (rbase + r)) * a.Lk + j;
        p *= zk_drop_scale(seed, a.sid, idx, a.thr, a.inv_keep);
      }
      const bf16_t pb16 = f2bf(p);
      const int il = w * 16 + (lane >> 4) * 4 + r;
      sP[il * PLD + j] = pb16;
      if (RPR) {
        const float pr = (rbase + r < a.Lq && j < a.Lk) ? bf2f(pb16) : 0.f;
        const int ri = rel_index(a.q_pos0 + rbase + r, j, a.max_rel);
        if (ri == 0) tlo[r] += pr;
        else if (ri == 2 * a.max_rel) thi[r] += pr;
        else sPB[il * ALD + ri] = f2bf(pr);
      }
    }
  }
  if (RPR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s0 = row16_sum(tlo[r]), s1 = row16_sum(thi[r]);
      if ((lane & 15) == 0) {
        const int il = w * 16 + (lane >> 4) * 4 + r;
        sPB[il * ALD] = f2bf(s0);
        if (a.max_rel > 0) sPB[il * ALD + 2 * a.max_rel] = f2bf(s1);
      }
    }
  }
  if (a.pb != nullptr) {   // relative-position value term: bucket sums of this wave's own rows of P
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    rpr_bucket_rows(a, a.pb, b, h, i0, w, lane, [&](int row, int j) { return bf2f(sP[row * PLD + j]); });
  }
  // O = P V, V^T staged per key tile into sK
  f32x4_t O[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) O[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    __syncthreads();   // sK readers done / sP complete
    if (kt == 0) { if (PRE != 3 && tid < 128) store_trans(sK, rVt, tid); }
    else stage_trans<FRESH>(sK, vb, a.ldv, kt * 64, a.Lk, tid);
    __syncthreads();
    const bf16_t* sVt = PRE == 3 ? vt_pre : sK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 pa = *reinterpret_cast<const uint4*>(sP + (w * 16 + (lane & 15)) * PLD + kt * 64 + kk * 32 +
                                                       (lane >> 4) * 8);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) O[nb] = mfma16(pa, frag(sVt, nb * 16, kk, lane), O[nb]);
    }
  }
  if (RPR) {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();             // this wave's sPB rows are complete (sRvT since the first barrier)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 pa = frag(sPB, w * 16, kk, lane);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) O[nb] = mfma16(pa, frag(sRvT, nb * 16, kk, lane), O[nb]);
    }
  }
  // O through LDS (sQ is dead) so that every thread stores 16 bytes of a row
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[(w * 16 + (lane >> 4) * 4 + r) * ALD + c] = f2bf(O[nb][r]);
  }
  __syncthreads();
  store_tile_rows(sQ, out + ((size_t)b * a.Lq + i0) * ldo + h * AD, ldo, a.Lq - i0, tid);
}

// ---- backward, single tile (Lq <= 64 and Lk <= 64): grid (1, nh, B) -> dQ, dK, dV in ONE pass.
// The two-kernel form above reads Q, K, V, dO twice and recomputes S and dP; at the training
// shapes of the north star (L = 64) a whole (batch, head) problem is one 64x64 tile, so one
// workgroup computes P and dS once (wave w owns query rows 16w..16w+15) and all three gradients
// follow from LDS: dQ = dS K, dK = dS^T Q, dV = P^T dO.  The operand tiles of the first phase are
// dead by then and their LDS is reused for dS, P^T and dS^T (7 tiles = 64.5 KB -> 2 workgroups/CU).
#define ATTN_BWD64_LDS_BYTES (7 * TQ * ALD * 2 + TQ * 4)
// RPR: the relative-position terms of modules/rpr.py:10-75 inside the tile (tables in LDS): G = Q.Rk^T and Gd = dO.Rv^T
// by MFMA, gathered into the scores / dP; the bucket sums of dS and P stay in LDS; dQ += dsb.Rk; and the table
// gradients dRk = dsb^T Q, dRv = pb^T dO of this (sentence, head) go to `rpr_part` (fp32 [B*nh][2][64][64], summed by
// the caller) -- instead of four grouped-GEMM launches and two reductions around the kernel.  Six more bf16 tiles and
// two fp32 tiles: one workgroup per CU.
#define ATTN_BWD64_RPR_LDS_BYTES (ATTN_BWD64_LDS_BYTES + 6 * TQ * ALD * 2 + 2 * TQ * GLD * 4)
// OPROJ: the gradient of the attention output is not read but COMPUTED here from the gradient of the output projection's
// result: dO_h = dY . W_o[h*64 .. h*64+63, :]^T  (dY [rows, n], W_o [H, n] row-major, n = op.n a multiple of 128) -- the
// 64 x 64 x n product of this (sentence, head) and nothing more, so the dgrad GEMM launch of o_map (and the dO matrix in HBM)
// disappears.  W_o's 64 rows go through LDS in slabs of 128 columns (two of the tiles that are idle until the prologue ends,
// double-buffered); the dY fragments of a wave's 16 rows come straight from global memory (no other wave needs them).
struct AttnOProj { const bf16_t* dy; int lddy; const bf16_t* w; int ldw; int n; };
template <bool RPR = false, bool OPROJ = false>
__device__ __forceinline__ void attn_bwd_fused64_tile(unsigned char* smem, const AttnArgs& a, const bf16_t* __restrict__ o, int ldo,
                                                      const bf16_t* __restrict__ dout, int lddo,
                                                      const float* __restrict__ lse,
                                                      bf16_t* __restrict__ dq, int lddq,
                                                      bf16_t* __restrict__ dk, int lddk,
                                                      bf16_t* __restrict__ dv, int lddv, int h, int b,
                                                      float* __restrict__ rpr_part = nullptr,
                                                      const AttnOProj op = AttnOProj()) {
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);      // phase 2: dS   [query][key]
  bf16_t* sK = sQ + TQ * ALD;                          // phase 2: P^T  [key][query]
  bf16_t* sV = sK + TQ * ALD;                          // phase 2: dS^T [key][query]
  bf16_t* sdO = sV + TQ * ALD;
  bf16_t* sKt = sdO + TQ * ALD;
  bf16_t* sQt = sKt + TQ * ALD;
  bf16_t* sdOt = sQt + TQ * ALD;
  float* sL = reinterpret_cast<float*>(sdOt + TQ * ALD);
  [[maybe_unused]] bf16_t* sRk = reinterpret_cast<bf16_t*>(smem + ATTN_BWD64_LDS_BYTES);   // key table   [r][channel]
  [[maybe_unused]] bf16_t* sRv = sRk + TQ * ALD;                                            // value table [r][channel]
  [[maybe_unused]] bf16_t* sRkT = sRv + TQ * ALD;                                           // key table   [phys channel][r]
  [[maybe_unused]] bf16_t* sSB = sRkT + TQ * ALD;                                           // bucket sums of dS  [query][r]
  [[maybe_unused]] bf16_t* sSBt = sSB + TQ * ALD;                                           //                    [r][query]
  [[maybe_unused]] bf16_t* sPBt = sSBt + TQ * ALD;                                          // bucket sums of P   [r][query]
  [[maybe_unused]] float* sG = reinterpret_cast<float*>(sPBt + TQ * ALD);                   // Q.Rk^T  [query][GLD]
  [[maybe_unused]] float* sGd = sG + TQ * GLD;                                              // dO.Rv^T [query][GLD]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)b * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)b * a.bsv + h * AD;
  (void)o; (void)ldo;
  const bf16_t* dob = dout + (size_t)b * a.Lq * lddo + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  // every global load of the prologue first (tiles, the O / dO rows of D_i, lse, the key mask), then the LDS stores
  ZK_AT(0);
  DirectRegs rQ, rdO, rK, rV;
  TransRegs t0, t1;
  load_direct(rQ, qb, a.ldq, 0, a.Lq, tid);
  if (!OPROJ) load_direct(rdO, dob, lddo, 0, a.Lq, tid);
  load_direct(rK, kb, a.ldk, 0, a.Lk, tid);
  load_direct(rV, vb, a.ldv, 0, a.Lk, tid);
  if (tid < 128) {
    load_trans(t0, kb, a.ldk, 0, a.Lk, tid);
  } else {
    load_trans(t0, qb, a.ldq, 0, a.Lq, tid - 128);
    if (!OPROJ) load_trans(t1, dob, lddo, 0, a.Lq, tid - 128);
  }
  // OPROJ: dO rows 16w .. 16w+15 of this head = dY rows x W_o[h*64 .., :]^T in slabs of 128 columns.  The loads of the first
  // 512-column chunk are issued HERE, behind the tile loads and in front of everything that waits for a loaded value (the
  // key-mask compare below waits for the whole queue: with the chunk issued behind it the product started one memory
  // round trip late, ATTNTRACE).  Q, K, V go to their tiles before the first slab (their registers are needed); the four
  // tiles of the second phase (dO, K^T, Q^T, dO^T) are idle until the product is done: slab s uses tiles 3 + 2(s&1),
  // 4 + 2(s&1), one barrier per slab.
  [[maybe_unused]] f32x4_t dOa[4];
  [[maybe_unused]] DirectRegs wr[8];
  [[maybe_unused]] uint4 av[16];
  [[maybe_unused]] const bf16_t* wrow = OPROJ ? op.w + (size_t)h * AD * op.ldw : nullptr;
  [[maybe_unused]] const bf16_t* ap =
      OPROJ ? op.dy + ((size_t)b * a.Lq + min(w * 16 + (lane & 15), a.Lq - 1)) * op.lddy + (lane >> 4) * 8 : nullptr;
  auto oproj_issue = [&](int c0) {
    const int nsl = min(4, (op.n - c0) >> 7);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nsl) {
        load_direct(wr[2 * sl], wrow + c0 + sl * 128, op.ldw, 0, AD, tid);
        load_direct(wr[2 * sl + 1], wrow + c0 + sl * 128 + 64, op.ldw, 0, AD, tid);
#pragma unroll
        for (int u = 0; u < 4; ++u) av[4 * sl + u] = zk_ld16<false>(ap + c0 + sl * 128 + u * 32);
      }
    }
  };
  if (OPROJ) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) dOa[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    oproj_issue(0);
  }
  // D_i = sum_j P_ij dP_ij is taken from the P and dP this workgroup computes anyway (whole rows live in one tile),
  // not from rowsum(dO o O) over the stored bf16 O: no O / dO row loads, and sum_j dS_ij = 0 holds to fp32 rounding
  // (with the bf16 O the rows of dS kept a common offset ~2^-9 |dO.O| that leaked mean(K) into dQ -- 30 % of the
  // tiny q_map / k_map gradients of the 12-layer-encoder configuration, tests/test_gpu_fullsize.py)
  const int dr = tid >> 2, dpart = tid & 3;
  const int drc = min(dr, a.Lq - 1);
  const float lse_r = lse[((size_t)b * a.nh + h) * a.Lq + drc];
  float kbias4[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int j = nt * 16 + (lane & 15);
    kbias4[nt] = (a.kmask != nullptr && a.kmask[(size_t)b * a.ldmask + min(j, a.Lk - 1)] == 0.f) ? -a.mask_inf : 0.f;
  }
  if (OPROJ) {
    ZK_AT(1);
    // chunks of 512 columns with EVERY load of the chunk in flight before the first slab is multiplied (one slab of
    // prefetch distance measured +7.4 us per launch: each slab then waits out most of a cold L2 / HBM latency)
    for (int c0 = 0; c0 < op.n; c0 += 512) {
      const int nsl = min(4, (op.n - c0) >> 7);
      if (c0 > 0) oproj_issue(c0);
      else {
        store_direct(sQ, rQ, tid);
        store_direct(sK, rK, tid);
        store_direct(sV, rV, tid);
      }
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        if (sl < nsl) {
          bf16_t* w0 = sdO + (sl & 1) * 2 * TQ * ALD;
          bf16_t* w1 = w0 + TQ * ALD;
          store_direct(w0, wr[2 * sl], tid);
          store_direct(w1, wr[2 * sl + 1], tid);
          __syncthreads();
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            dOa[nb] = mfma16(av[4 * sl + 0], frag(w0, nb * 16, 0, lane), dOa[nb]);
            dOa[nb] = mfma16(av[4 * sl + 1], frag(w0, nb * 16, 1, lane), dOa[nb]);
            dOa[nb] = mfma16(av[4 * sl + 2], frag(w1, nb * 16, 0, lane), dOa[nb]);
            dOa[nb] = mfma16(av[4 * sl + 3], frag(w1, nb * 16, 1, lane), dOa[nb]);
          }
        }
      }
    }
    ZK_AT(2);
    __syncthreads();                     // the slabs are dead: the transposed tiles and dO may land
  } else {
    store_direct(sQ, rQ, tid);
    store_direct(sdO, rdO, tid);
    store_direct(sK, rK, tid);
    store_direct(sV, rV, tid);
  }
  if (tid < 128) {
    store_trans(sKt, t0, tid);
  } else {
    store_trans(sQt, t0, tid - 128);
    if (!OPROJ) store_trans(sdOt, t1, tid - 128);
  }
  if (OPROJ) {
    // dO (rounded to bf16 like the GEMM's output was) as [query][channel] and [phys channel][query]; rows >= Lq are 0
    const int r0 = w * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int c = nb * 16 + (lane & 15);
      uint32_t hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hv[r] = (r0 + r < a.Lq) ? (uint32_t)f2bf(dOa[nb][r]) : 0u;
        sdO[(r0 + r) * ALD + c] = (bf16_t)hv[r];
      }
      *reinterpret_cast<uint2*>(sdOt + ((c & 7) * 8 + (c >> 3)) * ALD + r0) = make_uint2(hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16));
    }
  }
  if (dpart == 0) sL[dr] = (dr < a.Lq) ? lse_r : 0.f;
  ZK_AT(3);
  if (RPR) {
    const int nrel = 2 * a.max_rel + 1;
    stage_direct(sRk, a.rpr_k, AD, 0, nrel, tid);
    stage_direct(sRv, a.rpr_v, AD, 0, nrel, tid);
    stage_trans(sRkT, a.rpr_k, AD, 0, nrel, tid);
    for (int e = tid; e < 3 * TQ * ALD / 8; e += 256)          // bucket tiles: rows / buckets that are never emitted
      reinterpret_cast<uint4*>(sSB)[e] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  if (RPR) {
    // G and Gd rows of this wave's 16 queries (read back by this wave only)
    const uint4 q0 = frag(sQ, w * 16, 0, lane), q1 = frag(sQ, w * 16, 1, lane);
    const uint4 g0 = frag(sdO, w * 16, 0, lane), g1 = frag(sdO, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
      ga = mfma16(q0, frag(sRk, nt * 16, 0, lane), ga);
      ga = mfma16(q1, frag(sRk, nt * 16, 1, lane), ga);
      gb = mfma16(g0, frag(sRv, nt * 16, 0, lane), gb);
      gb = mfma16(g1, frag(sRv, nt * 16, 1, lane), gb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sG[(w * 16 + (lane >> 4) * 4 + r) * GLD + nt * 16 + (lane & 15)] = ga[r];
        sGd[(w * 16 + (lane >> 4) * 4 + r) * GLD + nt * 16 + (lane & 15)] = gb[r];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  // ---- phase 1: P and dS of query rows 16w .. 16w+15 against all 64 keys, kept in registers
  ZK_AT(4);
  const int rloc = w * 16 + (lane >> 4) * 4;
  float pv[4][4], dsv[4][4];
  {
    float pu[4][4], dpv[4][4], Di[4] = {0.f, 0.f, 0.f, 0.f};
    const uint4 q0 = frag(sQ, w * 16, 0, lane), q1 = frag(sQ, w * 16, 1, lane);
    const uint4 g0 = frag(sdO, w * 16, 0, lane), g1 = frag(sdO, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t sc4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      sc4 = mfma16(q0, frag(sK, nt * 16, 0, lane), sc4);
      sc4 = mfma16(q1, frag(sK, nt * 16, 1, lane), sc4);
      dp = mfma16(g0, frag(sV, nt * 16, 0, lane), dp);
      dp = mfma16(g1, frag(sV, nt * 16, 1, lane), dp);
      const int j = nt * 16 + (lane & 15);
      const bool kvalid = j < a.Lk;
      const float kbias = kvalid ? kbias4[nt] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rloc + r;
        const bool live = kvalid && i < a.Lq;
        float raw = sc4[r], dpr = dp[r];
        if (a.gq != nullptr && live) { raw += rpr_gather(a, a.gq, b, h, i, j); dpr += rpr_gather(a, a.gd, b, h, i, j); }
        if (RPR && live) {
          const int ri = rel_index(a.q_pos0 + i, j, a.max_rel);
          raw += sG[i * GLD + ri];
          dpr += sGd[i * GLD + ri];
        }
        float sc = raw * a.scale + kbias;
        if (a.causal && j > a.q_pos0 + i) sc -= a.mask_inf;
        const float p = live ? __expf(sc - sL[i]) : 0.f;
        float ms = 1.f;
        if (a.thr) {
          const uint64_t idx = (((uint64_t)b * a.nh + h) * a.Lq + i) * a.Lk + j;
          ms = zk_drop_scale(seed, a.sid, idx, a.thr, a.inv_keep);
        }
        pv[nt][r] = p * ms;
        pu[nt][r] = p;
        dpv[nt][r] = dpr * ms;
        Di[r] += p * dpr * ms;
      }
    }
    // a query row's 64 keys sit in the 16 lanes of its group x 4 key tiles
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Di[r] = row16_sum(Di[r]);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dsv[nt][r] = pu[nt][r] * (dpv[nt][r] - Di[r]) * a.scale;
  }
  ZK_AT(5);
  __syncthreads();                     // every wave is done reading sQ / sK / sV / sdO
  ZK_AT(6);
  bf16_t* sdS = sQ;
  bf16_t* sPt = sK;
  bf16_t* sdSt = sV;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int j = nt * 16 + (lane & 15);
    uint2 pp, dd;
    pp.x = (uint32_t)f2bf(pv[nt][0]) | ((uint32_t)f2bf(pv[nt][1]) << 16);
    pp.y = (uint32_t)f2bf(pv[nt][2]) | ((uint32_t)f2bf(pv[nt][3]) << 16);
    dd.x = (uint32_t)f2bf(dsv[nt][0]) | ((uint32_t)f2bf(dsv[nt][1]) << 16);
    dd.y = (uint32_t)f2bf(dsv[nt][2]) | ((uint32_t)f2bf(dsv[nt][3]) << 16);
    *reinterpret_cast<uint2*>(sPt + j * ALD + rloc) = pp;       // 4 consecutive queries of key row j
    *reinterpret_cast<uint2*>(sdSt + j * ALD + rloc) = dd;
#pragma unroll
    for (int r = 0; r < 4; ++r) sdS[(rloc + r) * ALD + j] = f2bf(dsv[nt][r]);
  }
  __syncthreads();
  if (RPR) {
    rpr_bucket_wave(a, 2 * a.max_rel + 1, 0, w, lane, [&](int row, int j) { return bf2f(sdS[row * ALD + j]); },
                    [&](int row, int r, float v) { const bf16_t x = f2bf(v); sSB[row * ALD + r] = x; sSBt[r * ALD + row] = x; });
    rpr_bucket_wave(a, 2 * a.max_rel + 1, 0, w, lane, [&](int row, int j) { return bf2f(sPt[j * ALD + row]); },
                    [&](int row, int r, float v) { sPBt[r * ALD + row] = f2bf(v); });
    __syncthreads();                     // the transposed bucket tiles are read across waves
  }
  if (a.dsb != nullptr) {   // bucket sums for the relative-position table products (see AttnArgs)
    rpr_bucket_rows(a, a.dsb, b, h, 0, w, lane, [&](int row, int j) { return bf2f(sdS[row * ALD + j]); });
    rpr_bucket_rows(a, a.pb, b, h, 0, w, lane, [&](int row, int j) { return bf2f(sPt[j * ALD + row]); });
  }
  // ---- phase 2: wave w -> dQ rows 16w.. (queries) and dK / dV rows 16w.. (keys)
  ZK_AT(7);
  f32x4_t dQ[4], dK[4], dV[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    dQ[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dK[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dV[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint4 da = frag(sdS, w * 16, kk, lane);
    const uint4 pa = frag(sPt, w * 16, kk, lane);
    const uint4 dt = frag(sdSt, w * 16, kk, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      dQ[nb] = mfma16(da, frag(sKt, nb * 16, kk, lane), dQ[nb]);
      dV[nb] = mfma16(pa, frag(sdOt, nb * 16, kk, lane), dV[nb]);
      dK[nb] = mfma16(dt, frag(sQt, nb * 16, kk, lane), dK[nb]);
    }
  }
  ZK_AT(8);
  if (RPR) {
    f32x4_t tk[4], tv[4];                // rows r = 16w .. of dRk / dRv of this (sentence, head)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) { tk[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; tv[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 sb = frag(sSB, w * 16, kk, lane);       // dsb rows (queries) x r
      const uint4 st = frag(sSBt, w * 16, kk, lane);      // dsb^T rows (r) x queries
      const uint4 pt = frag(sPBt, w * 16, kk, lane);      // pb^T  rows (r) x queries
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        dQ[nb] = mfma16(sb, frag(sRkT, nb * 16, kk, lane), dQ[nb]);
        tk[nb] = mfma16(st, frag(sQt, nb * 16, kk, lane), tk[nb]);
        tv[nb] = mfma16(pt, frag(sdOt, nb * 16, kk, lane), tv[nb]);
      }
    }
    // this wave's 16 rows of the two [r][channel] partials through LDS (sG / sGd are dead since phase 1; a wave only
    // touches its own rows of them) so that they leave as 16-byte row pieces; rows >= 2*max_rel+1 are never read
    float* pk = rpr_part + ((size_t)b * a.nh + h) * 2 * TQ * AD;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sG[(rloc + r) * GLD + c] = tk[nb][r];
        sGd[(rloc + r) * GLD + c] = tv[nb][r];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int nrel = 2 * a.max_rel + 1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = it * 64 + lane, rr = w * 16 + (e >> 4), c4 = (e & 15) * 4;
      if (rr < nrel) {
        *reinterpret_cast<float4*>(pk + rr * AD + c4) = *reinterpret_cast<const float4*>(sG + rr * GLD + c4);
        *reinterpret_cast<float4*>(pk + TQ * AD + rr * AD + c4) = *reinterpret_cast<const float4*>(sGd + rr * GLD + c4);
      }
    }
  }
  // results through LDS ([row][channel] tiles over the transposed operands, which are dead now) so that every
  // thread stores 16 bytes instead of 48 scattered 2-byte elements
  __syncthreads();
  bf16_t* oQ = sKt; bf16_t* oK = sQt; bf16_t* oV = sdOt;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = rloc + r;     // query row for dQ, key row for dK / dV
      oQ[i * ALD + c] = f2bf(dQ[nb][r]);
      oK[i * ALD + c] = f2bf(dK[nb][r]);
      oV[i * ALD + c] = f2bf(dV[nb][r]);
    }
  }
  __syncthreads();
  ZK_AT(9);
  store_tile_rows(oQ, dq + (size_t)b * a.Lq * lddq + h * AD, lddq, a.Lq, tid);
  store_tile_rows(oK, dk + (size_t)b * a.Lk * lddk + h * AD, lddk, a.Lk, tid);
  store_tile_rows(oV, dv + (size_t)b * a.Lk * lddv + h * AD, lddv, a.Lk, tid);
  ZK_AT(10);
}

// ---- backward with relative positions, single tile, TWO workgroups per CU.
// attn_bwd_fused64_tile<true> keeps every tile of both phases resident (151 KB: one four-wave workgroup per CU, 45 us per
// launch against 17 for the plain kernel).  Here the tiles take turns in eight slots (+ the lse row: 72.3 KB):
//   phase 1   T0 Q  T1 K  T2 V  T3 dO  T4 Rk  T5 Rv  T6-7 G (fp32 [64][GLD]: first Q.Rk^T for the scores, then -- same
//             buffer, the rows are private to their wave -- dO.Rv^T for dP)
//   phase 2a  T0 dS  T1 P^T  T2 dS^T  T3 K^T  T6 Q^T  T7 dO^T  T4 Rk^T   (the transposed operands wait in registers
//             since the prologue)                                        -> dQ, dK, dV
//   phase 2b  T2 dsb  T3 dsb^T  T5 pb^T (every wave writes -- values or zeros -- all rows / columns of its 16 queries)
//                                                                        -> dQ += dsb.Rk, dRk = dsb^T Q, dRv = pb^T dO
//   output    T0-3 the two fp32 table partials, T4-6 dQ, dK, dV as [row][channel] for 16-byte row stores
// Same arithmetic as the resident form (same MFMA sequences per output, same bucket sums); the only difference is the
// dropout multiplier of dP, applied from a kept-mask instead of being regenerated.
#define ATTN_BWD64_RPR2_LDS_BYTES (8 * TQ * ALD * 2 + TQ * 4)
__device__ __forceinline__ void attn_bwd_rpr64_tile(unsigned char* smem, const AttnArgs& a,
                                                    const bf16_t* __restrict__ dout, int lddo,
                                                    const float* __restrict__ lse,
                                                    bf16_t* __restrict__ dq, int lddq,
                                                    bf16_t* __restrict__ dk, int lddk,
                                                    bf16_t* __restrict__ dv, int lddv, int h, int b,
                                                    float* __restrict__ rpr_part) {
  constexpr int TB = TQ * ALD;
  bf16_t* T0 = reinterpret_cast<bf16_t*>(smem);
  bf16_t* T1 = T0 + TB; bf16_t* T2 = T1 + TB; bf16_t* T3 = T2 + TB;
  bf16_t* T4 = T3 + TB; bf16_t* T5 = T4 + TB; bf16_t* T6 = T5 + TB; bf16_t* T7 = T6 + TB;
  float* sG = reinterpret_cast<float*>(T6);            // [64][GLD] fp32 = 17408 B <= two tiles
  float* sL = reinterpret_cast<float*>(T7 + TB);
  static_assert(TQ * GLD * 4 <= 2 * TQ * ALD * 2, "G must fit two tiles");
  static_assert(2 * TQ * GLD * 4 <= 4 * TQ * ALD * 2, "the two table partials must fit four tiles");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)b * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)b * a.bsv + h * AD;
  const bf16_t* dob = dout + (size_t)b * a.Lq * lddo + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;
  const int nrel = 2 * a.max_rel + 1;

  // ---- prologue: every global load first
  ZK_AT(0);
  DirectRegs rQ, rdO, rK, rV, rRk, rRv;
  TransRegs t0, t1;
  load_direct(rQ, qb, a.ldq, 0, a.Lq, tid);
  load_direct(rdO, dob, lddo, 0, a.Lq, tid);
  load_direct(rK, kb, a.ldk, 0, a.Lk, tid);
  load_direct(rV, vb, a.ldv, 0, a.Lk, tid);
  load_direct(rRk, a.rpr_k, AD, 0, nrel, tid);
  load_direct(rRv, a.rpr_v, AD, 0, nrel, tid);
  if (tid < 128) {
    load_trans(t0, kb, a.ldk, 0, a.Lk, tid);
    load_trans(t1, a.rpr_k, AD, 0, nrel, tid);
  } else {
    load_trans(t0, qb, a.ldq, 0, a.Lq, tid - 128);
    load_trans(t1, dob, lddo, 0, a.Lq, tid - 128);
  }
  const int dr = tid >> 2, dpart = tid & 3;
  const float lse_r = lse[((size_t)b * a.nh + h) * a.Lq + min(dr, a.Lq - 1)];
  float kbias4[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int j = nt * 16 + (lane & 15);
    kbias4[nt] = (a.kmask != nullptr && a.kmask[(size_t)b * a.ldmask + min(j, a.Lk - 1)] == 0.f) ? -a.mask_inf : 0.f;
  }
  store_direct(T0, rQ, tid);
  store_direct(T3, rdO, tid);
  store_direct(T1, rK, tid);
  store_direct(T2, rV, tid);
  store_direct(T4, rRk, tid);
  store_direct(T5, rRv, tid);
  if (dpart == 0) sL[dr] = (dr < a.Lq) ? lse_r : 0.f;
  ZK_AT(3);
  __syncthreads();
  ZK_AT(4);

  // ---- phase 1: P and dS of query rows 16w .. 16w+15 against all 64 keys, kept in registers
  const int rloc = w * 16 + (lane >> 4) * 4;
  float pv[4][4], dsv[4][4];
  {
    float pu[4][4], Di[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t kept = 0;                                     // bit nt*4 + r: the element survived dropout
    const uint4 q0 = frag(T0, w * 16, 0, lane), q1 = frag(T0, w * 16, 1, lane);
    const uint4 g0 = frag(T3, w * 16, 0, lane), g1 = frag(T3, w * 16, 1, lane);
    // G = Q.Rk^T rows of this wave's queries (read back by this wave only)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t ga = {0.f, 0.f, 0.f, 0.f};
      ga = mfma16(q0, frag(T4, nt * 16, 0, lane), ga);
      ga = mfma16(q1, frag(T4, nt * 16, 1, lane), ga);
#pragma unroll
      for (int r = 0; r < 4; ++r) sG[(rloc + r) * GLD + nt * 16 + (lane & 15)] = ga[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t sc4 = {0.f, 0.f, 0.f, 0.f};
      sc4 = mfma16(q0, frag(T1, nt * 16, 0, lane), sc4);
      sc4 = mfma16(q1, frag(T1, nt * 16, 1, lane), sc4);
      const int j = nt * 16 + (lane & 15);
      const bool kvalid = j < a.Lk;
      const float kbias = kvalid ? kbias4[nt] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rloc + r;
        const bool live = kvalid && i < a.Lq;
        float raw = sc4[r];
        if (live) raw += sG[i * GLD + rel_index(a.q_pos0 + i, j, a.max_rel)];
        float sc = raw * a.scale + kbias;
        if (a.causal && j > a.q_pos0 + i) sc -= a.mask_inf;
        const float p = live ? __expf(sc - sL[i]) : 0.f;
        float ms = 1.f;
        if (a.thr) {
          const uint64_t idx = (((uint64_t)b * a.nh + h) * a.Lq + i) * a.Lk + j;
          ms = zk_drop_scale(seed, a.sid, idx, a.thr, a.inv_keep);
        }
        if (ms != 0.f) kept |= 1u << (nt * 4 + r);
        pv[nt][r] = p * ms;
        pu[nt][r] = p;
      }
    }
    // Gd = dO.Rv^T into the same rows of the same buffer
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t gb = {0.f, 0.f, 0.f, 0.f};
      gb = mfma16(g0, frag(T5, nt * 16, 0, lane), gb);
      gb = mfma16(g1, frag(T5, nt * 16, 1, lane), gb);
#pragma unroll
      for (int r = 0; r < 4; ++r) sG[(rloc + r) * GLD + nt * 16 + (lane & 15)] = gb[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float dpv[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t dp = {0.f, 0.f, 0.f, 0.f};
      dp = mfma16(g0, frag(T2, nt * 16, 0, lane), dp);
      dp = mfma16(g1, frag(T2, nt * 16, 1, lane), dp);
      const int j = nt * 16 + (lane & 15);
      const bool kvalid = j < a.Lk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rloc + r;
        float dpr = dp[r];
        if (kvalid && i < a.Lq) dpr += sG[i * GLD + rel_index(a.q_pos0 + i, j, a.max_rel)];
        const float ms = ((kept >> (nt * 4 + r)) & 1u) ? (a.thr ? a.inv_keep : 1.f) : 0.f;
        dpv[nt][r] = dpr * ms;
        Di[r] += pu[nt][r] * dpr * ms;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Di[r] = row16_sum(Di[r]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dsv[nt][r] = pu[nt][r] * (dpv[nt][r] - Di[r]) * a.scale;
  }
  ZK_AT(5);
  __syncthreads();                     // every wave is done with the operand tiles, the tables and G
  ZK_AT(6);
  bf16_t* sdS = T0; bf16_t* sPt = T1; bf16_t* sdSt = T2;
  bf16_t* sKt = T3; bf16_t* sRkT = T4; bf16_t* sQt = T6; bf16_t* sdOt = T7;
  uint2 ppk[4], ddk[4];                 // this lane's 4 x 4 entries of P and dS as bf16 pairs: also the source of the bucket sums
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int j = nt * 16 + (lane & 15);
    uint2 pp, dd;
    pp.x = (uint32_t)f2bf(pv[nt][0]) | ((uint32_t)f2bf(pv[nt][1]) << 16);
    pp.y = (uint32_t)f2bf(pv[nt][2]) | ((uint32_t)f2bf(pv[nt][3]) << 16);
    dd.x = (uint32_t)f2bf(dsv[nt][0]) | ((uint32_t)f2bf(dsv[nt][1]) << 16);
    dd.y = (uint32_t)f2bf(dsv[nt][2]) | ((uint32_t)f2bf(dsv[nt][3]) << 16);
    ppk[nt] = pp; ddk[nt] = dd;
    *reinterpret_cast<uint2*>(sPt + j * ALD + rloc) = pp;
    *reinterpret_cast<uint2*>(sdSt + j * ALD + rloc) = dd;
#pragma unroll
    for (int r = 0; r < 4; ++r) sdS[(rloc + r) * ALD + j] = (bf16_t)((r & 1) ? ((r & 2 ? dd.y : dd.x) >> 16) : ((r & 2 ? dd.y : dd.x) & 0xffffu));
  }
  if (tid < 128) {
    store_trans(sKt, t0, tid);
    store_trans(sRkT, t1, tid);
  } else {
    store_trans(sQt, t0, tid - 128);
    store_trans(sdOt, t1, tid - 128);
  }
  __syncthreads();
  ZK_AT(7);
  // ---- phase 2a: wave w -> dQ rows 16w.. (queries) and dK / dV rows 16w.. (keys)
  f32x4_t dQ[4], dK[4], dV[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    dQ[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dK[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dV[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint4 da = frag(sdS, w * 16, kk, lane);
    const uint4 pa = frag(sPt, w * 16, kk, lane);
    const uint4 dt = frag(sdSt, w * 16, kk, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      dQ[nb] = mfma16(da, frag(sKt, nb * 16, kk, lane), dQ[nb]);
      dV[nb] = mfma16(pa, frag(sdOt, nb * 16, kk, lane), dV[nb]);
      dK[nb] = mfma16(dt, frag(sQt, nb * 16, kk, lane), dK[nb]);
    }
  }
  ZK_AT(8);
  __syncthreads();                     // dS^T (T2) and K^T (T3) are dead
  // ---- phase 2b: bucket sums of dS and P over the relative index, then the table terms
  bf16_t* sSB = T2; bf16_t* sSBt = T3; bf16_t* sPBt = T5;
  {
    // this wave owns rows 16w.. of sSB and columns 16w.. of the two transposed tiles: zeros first, then the emitted buckets
    // (LDS operations of one wave execute in order)
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = it * 64 + lane;                         // 128 pieces of 16 bytes
      *reinterpret_cast<uint4*>(sSB + (w * 16 + (e >> 3)) * ALD + (e & 7) * 8) = z;
      *reinterpret_cast<uint4*>(sSBt + (e >> 1) * ALD + w * 16 + (e & 1) * 8) = z;
      *reinterpret_cast<uint4*>(sPBt + (e >> 1) * ALD + w * 16 + (e & 1) * 8) = z;
    }
  }
  // Bucket sums over the relative index straight from the registers of phase 1 (the resident form walks the LDS tiles
  // again: 6.2 of its 18 us, profiles/r03_attn_bwd_phases_v1.txt).  Entry (i, j) of this lane belongs to bucket
  // r = clip(i - j, -m, m) + m: an interior bucket 0 < r < 2m has exactly ONE key, so the entry IS the bucket -- a scattered
  // 2-byte store; the two clipped tails r = 0 (j >= i + m) and r = 2m (j <= i - m) are sums over the row: four partials per
  // lane, a 16-lane reduction, one store.  Values are the bf16-rounded P / dS the products above used.
  {
    const int m = a.max_rel;
    float tl_ds[4] = {0.f, 0.f, 0.f, 0.f}, th_ds[4] = {0.f, 0.f, 0.f, 0.f};
    float tl_p[4] = {0.f, 0.f, 0.f, 0.f}, th_p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int j = nt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rloc + r;
        const uint32_t dw = (r & 2) ? ddk[nt].y : ddk[nt].x, pw = (r & 2) ? ppk[nt].y : ppk[nt].x;
        const bf16_t db = (bf16_t)((r & 1) ? (dw >> 16) : (dw & 0xffffu)), pb_ = (bf16_t)((r & 1) ? (pw >> 16) : (pw & 0xffffu));
        const int ri = rel_index(a.q_pos0 + i, j, m);
        if (ri == 0) { tl_ds[r] += bf2f(db); tl_p[r] += bf2f(pb_); }
        else if (ri == 2 * m) { th_ds[r] += bf2f(db); th_p[r] += bf2f(pb_); }
        else { sSB[i * ALD + ri] = db; sSBt[ri * ALD + i] = db; sPBt[ri * ALD + i] = pb_; }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a0 = row16_sum(tl_ds[r]), a1 = row16_sum(th_ds[r]), b0 = row16_sum(tl_p[r]), b1 = row16_sum(th_p[r]);
      if ((lane & 15) == 0) {
        const int i = rloc + r;
        const bf16_t x0 = f2bf(a0), y0 = f2bf(b0);
        sSB[i * ALD] = x0; sSBt[i] = x0; sPBt[i] = y0;
        if (m > 0) {
          const bf16_t x1 = f2bf(a1), y1 = f2bf(b1);
          sSB[i * ALD + 2 * m] = x1; sSBt[2 * m * ALD + i] = x1; sPBt[2 * m * ALD + i] = y1;
        }
      }
    }
  }
  ZK_AT(11);
  __syncthreads();                     // the transposed bucket tiles are read across waves
  f32x4_t tk[4], tv[4];                // rows r = 16w .. of dRk / dRv of this (sentence, head)
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) { tk[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; tv[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint4 sb = frag(sSB, w * 16, kk, lane);
    const uint4 st = frag(sSBt, w * 16, kk, lane);
    const uint4 pt = frag(sPBt, w * 16, kk, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      dQ[nb] = mfma16(sb, frag(sRkT, nb * 16, kk, lane), dQ[nb]);
      tk[nb] = mfma16(st, frag(sQt, nb * 16, kk, lane), tk[nb]);
      tv[nb] = mfma16(pt, frag(sdOt, nb * 16, kk, lane), tv[nb]);
    }
  }
  ZK_AT(12);
  __syncthreads();                     // every tile is dead: outputs through LDS for 16-byte row stores
  float* oTk = reinterpret_cast<float*>(T0);
  float* oTv = oTk + TQ * GLD;
  bf16_t* oQ = T4; bf16_t* oK = T5; bf16_t* oV = T6;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = rloc + r;     // table row r for tk / tv, query row for dQ, key row for dK / dV
      oTk[i * GLD + c] = tk[nb][r];
      oTv[i * GLD + c] = tv[nb][r];
      oQ[i * ALD + c] = f2bf(dQ[nb][r]);
      oK[i * ALD + c] = f2bf(dK[nb][r]);
      oV[i * ALD + c] = f2bf(dV[nb][r]);
    }
  }
  __syncthreads();
  ZK_AT(9);
  float* pk = rpr_part + ((size_t)b * a.nh + h) * 2 * TQ * AD;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = it * 256 + tid, rr = e >> 4, c4 = (e & 15) * 4;
    if (rr < nrel) {
      *reinterpret_cast<float4*>(pk + rr * AD + c4) = *reinterpret_cast<const float4*>(oTk + rr * GLD + c4);
      *reinterpret_cast<float4*>(pk + TQ * AD + rr * AD + c4) = *reinterpret_cast<const float4*>(oTv + rr * GLD + c4);
    }
  }
  store_tile_rows(oQ, dq + (size_t)b * a.Lq * lddq + h * AD, lddq, a.Lq, tid);
  store_tile_rows(oK, dk + (size_t)b * a.Lk * lddk + h * AD, lddk, a.Lk, tid);
  store_tile_rows(oV, dv + (size_t)b * a.Lk * lddv + h * AD, lddv, a.Lk, tid);
  ZK_AT(10);
}
