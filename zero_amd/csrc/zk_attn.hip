// zk_attn.hip -- fused scaled-dot-product attention, forward and backward (gfx950).
//
// Reference: func.py:218-256 `dot_attention` core -- q *= d^-0.5 (func.py:222), logits = q k^T
// (+ relative-position term, modules/rpr.py:10-41), logits += bias where bias is
// (1-mask)*-1e8 on padded keys (func.py:384-387) or -1e8 above the diagonal (func.py:376-383;
// FINITE on purpose: a fully masked row softmaxes to uniform), softmax, dropout on the
// weights, weights @ v (+ relative-position value term).  split_heads / combine_heads
// (func.py:68-104) are folded into the addressing: q/k/v/out are [B*L, ld] matrices whose
// head h lives at columns [h*d, (h+1)*d).
//
// Two implementations, both HIP:
//   * reference ("naive") kernels: one wave per query (or key) row, any d<=256, Lk<=512,
//     relative positions supported.  Used for parity checks, odd shapes, RPR and decode.
//   * MFMA kernels (d == 64, no RPR): 64-query tile per workgroup, 4 waves x 16 rows,
//     v_mfma_f32_16x16x32_bf16, K / V^T / P staged in LDS, softmax in registers with
//     16-lane shuffle reductions.  Backward = flash-style recompute from the saved
//     log-sum-exp: kernel A (per query tile) -> dQ and D=rowsum(dO*O); kernel B (per key
//     tile) -> dK, dV.
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
template <typename F>
__device__ __forceinline__ void rpr_bucket_rows(const AttnArgs& a, bf16_t* __restrict__ dst, int b, int h, int i0,
                                                int w, int lane, F val) {
  const int m = a.max_rel;
  for (int e = lane; e < 16 * a.nrp; e += 64) {
    const int row = w * 16 + e / a.nrp, r = e % a.nrp;
    const int i = i0 + row;
    if (i >= a.Lq) continue;
    const int ia = a.q_pos0 + i;
    float acc = 0.f;
    if (r > 0 && r < 2 * m) {
      const int j = ia - (r - m);
      if (j >= 0 && j < a.Lk) acc = val(row, j);
    } else if (r == 0) {
      for (int j = max(ia + m, 0); j < a.Lk; ++j) acc += val(row, j);
    } else if (r == 2 * m) {
      for (int j = min(ia - m, a.Lk - 1); j >= 0; --j) acc += val(row, j);
    }
    dst[((size_t)b * a.Lq + i) * a.ldg + h * a.nrp + r] = f2bf(acc);
  }
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
// reference kernels
// =====================================================================================
#define NAIVE_MAXK 8   // keys per lane -> Lk <= 512

// score of (query i, key j) for head h -- full dot product by one lane
__device__ __forceinline__ float naive_score(const AttnArgs& a, int b, int h, int i, int j) {
  const bf16_t* qp = a.q + (size_t)b * a.bsq + (size_t)i * a.ldq + h * a.d;
  const bf16_t* kp = a.k + (size_t)(b / a.kv_group) * a.bsk + (size_t)j * a.ldk + h * a.d;
  float s = 0.f;
  if (a.rpr_k != nullptr) {
    const bf16_t* rp = a.rpr_k + (size_t)rel_index(a.q_pos0 + i, j, a.max_rel) * a.d;
    for (int c = 0; c < a.d; ++c) s += bf2f(qp[c]) * (bf2f(kp[c]) + bf2f(rp[c]));
  } else {
    for (int c = 0; c < a.d; ++c) s += bf2f(qp[c]) * bf2f(kp[c]);
  }
  return s * a.scale + mask_bias(a, b, a.q_pos0 + i, j);
}

__global__ void __launch_bounds__(256) k_attn_fwd_naive(AttnArgs a, bf16_t* __restrict__ out, int ldo,
                                                        float* __restrict__ lse) {
  __shared__ float sp[4][NAIVE_MAXK * 64];
  attn_apply_pos(a);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + w;
  const long nrows = (long)a.B * a.nh * a.Lq;
  if (row >= nrows) return;   // whole wave exits together
  const int i = (int)(row % a.Lq);
  const int h = (int)((row / a.Lq) % a.nh);
  const int b = (int)(row / ((long)a.Lq * a.nh));
  const uint64_t seed = a.thr ? *a.seed : 0;
  float s[NAIVE_MAXK];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NAIVE_MAXK; ++t) {
    const int j = t * 64 + lane;
    s[t] = (j < a.Lk) ? naive_score(a, b, h, i, j) : -INFINITY;
    m = fmaxf(m, s[t]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NAIVE_MAXK; ++t) {
    const int j = t * 64 + lane;
    s[t] = (j < a.Lk) ? __expf(s[t] - m) : 0.f;
    sum += s[t];
  }
  sum = wave_sum(sum);
  if (lane == 0 && lse != nullptr) lse[row] = m + __logf(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int t = 0; t < NAIVE_MAXK; ++t) {
    const int j = t * 64 + lane;
    float p = s[t] * inv;
    if (a.thr && j < a.Lk) p *= zk_drop_scale(seed, a.sid, (uint64_t)row * a.Lk + j, a.thr, a.inv_keep);
    sp[w][t * 64 + lane] = p;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave have landed
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < a.d; c += 64) {
    float o = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
      float vv = bf2f(a.v[(size_t)(b / a.kv_group) * a.bsv + (size_t)j * a.ldv + h * a.d + c]);
      if (a.rpr_v != nullptr)
        vv += bf2f(a.rpr_v[(size_t)rel_index(a.q_pos0 + i, j, a.max_rel) * a.d + c]);
      o += sp[w][j] * vv;
    }
    out[((size_t)b * a.Lq + i) * ldo + h * a.d + c] = f2bf(o);
  }
}

// backward A: per query row -> dq, D (= sum_j p_ij dp_ij), relative-position table grads
__global__ void __launch_bounds__(256) k_attn_bwd_dq_naive(AttnArgs a, const bf16_t* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dq, int lddq,
                                                           float* __restrict__ Dbuf,
                                                           float* __restrict__ drpr_k,
                                                           float* __restrict__ drpr_v) {
  __shared__ float sds[4][NAIVE_MAXK * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + w;
  const long nrows = (long)a.B * a.nh * a.Lq;
  if (row >= nrows) return;
  const int i = (int)(row % a.Lq);
  const int h = (int)((row / a.Lq) % a.nh);
  const int b = (int)(row / ((long)a.Lq * a.nh));
  const uint64_t seed = a.thr ? *a.seed : 0;
  const float l = lse[row];
  const bf16_t* dop = dout + ((size_t)b * a.Lq + i) * lddo + h * a.d;
  const bf16_t* qp = a.q + (size_t)b * a.bsq + (size_t)i * a.ldq + h * a.d;
  float p[NAIVE_MAXK], dp[NAIVE_MAXK], ms[NAIVE_MAXK];
  float D = 0.f;
#pragma unroll
  for (int t = 0; t < NAIVE_MAXK; ++t) {
    const int j = t * 64 + lane;
    p[t] = 0.f; dp[t] = 0.f; ms[t] = 1.f;
    if (j < a.Lk) {
      p[t] = __expf(naive_score(a, b, h, i, j) - l);
      const bf16_t* vp = a.v + (size_t)(b / a.kv_group) * a.bsv + (size_t)j * a.ldv + h * a.d;
      float acc = 0.f;
      if (a.rpr_v != nullptr) {
        const bf16_t* rp = a.rpr_v + (size_t)rel_index(a.q_pos0 + i, j, a.max_rel) * a.d;
        for (int c = 0; c < a.d; ++c) acc += bf2f(dop[c]) * (bf2f(vp[c]) + bf2f(rp[c]));
      } else {
        for (int c = 0; c < a.d; ++c) acc += bf2f(dop[c]) * bf2f(vp[c]);
      }
      if (a.thr) ms[t] = zk_drop_scale(seed, a.sid, (uint64_t)row * a.Lk + j, a.thr, a.inv_keep);
      dp[t] = acc * ms[t];
      D += p[t] * dp[t];
    }
  }
  D = wave_sum(D);
  if (lane == 0 && Dbuf != nullptr) Dbuf[row] = D;
#pragma unroll
  for (int t = 0; t < NAIVE_MAXK; ++t) {
    const int j = t * 64 + lane;
    const float ds = p[t] * (dp[t] - D);
    sds[w][t * 64 + lane] = ds;
    if (j < a.Lk && drpr_k != nullptr) {
      const int ri = rel_index(a.q_pos0 + i, j, a.max_rel);
      const float pd = p[t] * ms[t];
      for (int c = 0; c < a.d; ++c) {
        unsafeAtomicAdd(drpr_k + (size_t)ri * a.d + c, ds * a.scale * bf2f(qp[c]));
        unsafeAtomicAdd(drpr_v + (size_t)ri * a.d + c, pd * bf2f(dop[c]));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < a.d; c += 64) {
    float o = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
      float kk = bf2f(a.k[(size_t)(b / a.kv_group) * a.bsk + (size_t)j * a.ldk + h * a.d + c]);
      if (a.rpr_k != nullptr)
        kk += bf2f(a.rpr_k[(size_t)rel_index(a.q_pos0 + i, j, a.max_rel) * a.d + c]);
      o += sds[w][j] * kk;
    }
    dq[((size_t)b * a.Lq + i) * lddq + h * a.d + c] = f2bf(o * a.scale);
  }
}

// backward B: per key row -> dk, dv.  lanes over channels, loop over queries.
__global__ void __launch_bounds__(256) k_attn_bwd_dkv_naive(AttnArgs a, const bf16_t* __restrict__ dout, int lddo,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ Dbuf,
                                                            bf16_t* __restrict__ dk, int lddk,
                                                            bf16_t* __restrict__ dv, int lddv) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + w;
  const long nrows = (long)a.B * a.nh * a.Lk;
  if (row >= nrows) return;
  const int j = (int)(row % a.Lk);
  const int h = (int)((row / a.Lk) % a.nh);
  const int b = (int)(row / ((long)a.Lk * a.nh));
  const uint64_t seed = a.thr ? *a.seed : 0;
  const bf16_t* kp = a.k + (size_t)(b / a.kv_group) * a.bsk + (size_t)j * a.ldk + h * a.d;
  const bf16_t* vp = a.v + (size_t)(b / a.kv_group) * a.bsv + (size_t)j * a.ldv + h * a.d;
  // up to 4 channels per lane (d <= 256)
  float kc[4], vc[4], dkc[4] = {0, 0, 0, 0}, dvc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = u * 64 + lane;
    kc[u] = (c < a.d) ? bf2f(kp[c]) : 0.f;
    vc[u] = (c < a.d) ? bf2f(vp[c]) : 0.f;
  }
  for (int i = 0; i < a.Lq; ++i) {
    const long qrow = ((long)b * a.nh + h) * a.Lq + i;
    const bf16_t* qp = a.q + (size_t)b * a.bsq + (size_t)i * a.ldq + h * a.d;
    const bf16_t* dop = dout + ((size_t)b * a.Lq + i) * lddo + h * a.d;
    const int ri = rel_index(a.q_pos0 + i, j, a.max_rel);
    float qv[4], dov[4];
    float s = 0.f, dpv = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = u * 64 + lane;
      qv[u] = 0.f; dov[u] = 0.f;
      if (c < a.d) {
        qv[u] = bf2f(qp[c]);
        dov[u] = bf2f(dop[c]);
        float ke = kc[u], ve = vc[u];
        if (a.rpr_k != nullptr) {
          ke += bf2f(a.rpr_k[(size_t)ri * a.d + c]);
          ve += bf2f(a.rpr_v[(size_t)ri * a.d + c]);
        }
        s += qv[u] * ke;
        dpv += dov[u] * ve;
      }
    }
    s = wave_sum(s);
    dpv = wave_sum(dpv);
    s = s * a.scale + mask_bias(a, b, a.q_pos0 + i, j);
    const float p = __expf(s - lse[qrow]);
    const float ms = a.thr ? zk_drop_scale(seed, a.sid, (uint64_t)qrow * a.Lk + j, a.thr, a.inv_keep) : 1.f;
    const float ds = p * (dpv * ms - Dbuf[qrow]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dvc[u] += p * ms * dov[u];
      dkc[u] += ds * a.scale * qv[u];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = u * 64 + lane;
    if (c < a.d) {
      dk[((size_t)b * a.Lk + j) * lddk + h * a.d + c] = f2bf(dkc[u]);
      dv[((size_t)b * a.Lk + j) * lddv + h * a.d + c] = f2bf(dvc[u]);
    }
  }
}

// =====================================================================================
// MFMA kernels (d = 64)
// =====================================================================================
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
__device__ __forceinline__ void stage_direct(bf16_t* dst, const bf16_t* src, int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    const int r = t >> 3, c = (t & 7) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row0 + r < nrows) v = *reinterpret_cast<const uint4*>(src + (size_t)(row0 + r) * ld + c);
    *reinterpret_cast<uint4*>(dst + r * ALD + c) = v;
  }
}
__device__ __forceinline__ uint32_t half_of4(const uint4& v, int i) {
  const uint32_t w = (i >> 1) == 0 ? v.x : ((i >> 1) == 1 ? v.y : ((i >> 1) == 2 ? v.z : v.w));
  return (i & 1) ? (w >> 16) : (w & 0xffffu);
}
// transposed staging: dst[phys(c)][r] = src[row0+r][c]; channel rows stored in permuted order
// phys(c) = (c%8)*8 + c/8 so that the 8-byte LDS writes of neighbouring lanes hit different banks.
__device__ __forceinline__ void stage_trans(bf16_t* dst, const bf16_t* src, int ld, int row0, int nrows, int tid) {
  if (tid < 128) {
    const int dc = tid & 7, rq = tid >> 3;   // channel chunk (8 channels), row quad (4 rows)
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = row0 + rq * 4 + u;
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      if (r < nrows) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + dc * 8);
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
__device__ __forceinline__ void load_direct(DirectRegs& R, const bf16_t* src, int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = tid + it * 256;
    const int r = row0 + (t >> 3), c = (t & 7) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)min(r, nrows - 1) * ld + c);
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
__device__ __forceinline__ void load_trans(TransRegs& R, const bf16_t* src, int ld, int row0, int nrows, int t128) {
  const int dc = t128 & 7, rq = t128 >> 3;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = row0 + rq * 4 + u;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)min(r, nrows - 1) * ld + dc * 8);
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
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float quad16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// ---- forward: grid (ceil(Lq/64), nh, B); NKT = ceil(Lk/64) <= 4
template <int NKT>
__global__ void __launch_bounds__(256) k_attn_fwd_mfma(AttnArgs a, bf16_t* __restrict__ out, int ldo,
                                                       float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) bf16_t sQ[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sK[TQ * ALD];    // K tile, later V^T tile
  __shared__ __attribute__((aligned(16))) bf16_t sP[TQ * (NKT * 64 + 8)];
  constexpr int PLD = NKT * 64 + 8;
  attn_apply_pos(a);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)(b / a.kv_group) * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)(b / a.kv_group) * a.bsv + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  // the loads that do not depend on anything computed here are issued together: Q, the first K tile, the first
  // V tile (kept in registers until P is ready) and the key mask -- one memory round trip instead of four
  DirectRegs rQ, rK;
  TransRegs rVt;
  load_direct(rQ, qb, a.ldq, i0, a.Lq, tid);
  load_direct(rK, kb, a.ldk, 0, a.Lk, tid);
  if (tid < 128) load_trans(rVt, vb, a.ldv, 0, a.Lk, tid);
  float kbias[NKT * 4];
#pragma unroll
  for (int t = 0; t < NKT * 4; ++t) {
    const int j = t * 16 + (lane & 15);
    kbias[t] = (a.kmask != nullptr && a.kmask[(size_t)(b / a.kv_group) * a.ldmask + min(j, a.Lk - 1)] == 0.f) ? -a.mask_inf : 0.f;
  }
  store_direct(sQ, rQ, tid);
  store_direct(sK, rK, tid);
  f32x4_t S[NKT * 4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt > 0) {
      __syncthreads();   // previous tile's readers are done with sK
      stage_direct(sK, kb, a.ldk, kt * 64, a.Lk, tid);
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
      sP[(w * 16 + (lane >> 4) * 4 + r) * PLD + j] = f2bf(p);
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
    if (kt == 0) { if (tid < 128) store_trans(sK, rVt, tid); }
    else stage_trans(sK, vb, a.ldv, kt * 64, a.Lk, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 pa = *reinterpret_cast<const uint4*>(sP + (w * 16 + (lane & 15)) * PLD + kt * 64 + kk * 32 +
                                                       (lane >> 4) * 8);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) O[nb] = mfma16(pa, frag(sK, nb * 16, kk, lane), O[nb]);
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

// ---- backward A: grid (ceil(Lq/64), nh, B) -> dQ, Dbuf
__global__ void __launch_bounds__(256) k_attn_bwd_dq_mfma(AttnArgs a, const bf16_t* __restrict__ o, int ldo,
                                                          const bf16_t* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dq, int lddq,
                                                          float* __restrict__ Dbuf) {
  __shared__ __attribute__((aligned(16))) bf16_t sQ[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdO[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sK[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sV[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sKt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdS[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sPd[TQ * ALD];   // dropped P of the tile (relative-position sums only)
  __shared__ float sD[TQ];
  __shared__ float sL[TQ];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
  float acc_dsb[RPR_MAXU], acc_pb[RPR_MAXU];
#pragma unroll
  for (int u = 0; u < RPR_MAXU; ++u) { acc_dsb[u] = 0.f; acc_pb[u] = 0.f; }
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)(b / a.kv_group) * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)(b / a.kv_group) * a.bsv + h * AD;
  const bf16_t* ob = o + (size_t)b * a.Lq * ldo + h * AD;
  const bf16_t* dob = dout + (size_t)b * a.Lq * lddo + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  stage_direct(sQ, qb, a.ldq, i0, a.Lq, tid);
  stage_direct(sdO, dob, lddo, i0, a.Lq, tid);
  {  // D_i = sum_c dO[i][c] * O[i][c]; 4 threads per row
    const int r = tid >> 2, part = tid & 3;
    float acc = 0.f;
    if (i0 + r < a.Lq) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float x[8], y[8];
        unpack8(*reinterpret_cast<const uint4*>(dob + (size_t)(i0 + r) * lddo + part * 16 + u * 8), x);
        unpack8(*reinterpret_cast<const uint4*>(ob + (size_t)(i0 + r) * ldo + part * 16 + u * 8), y);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += x[c] * y[c];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) {
      sD[r] = acc;
      const int i = i0 + r;
      sL[r] = (i < a.Lq) ? lse[((size_t)b * a.nh + h) * a.Lq + i] : 0.f;
      if (i < a.Lq && Dbuf != nullptr) Dbuf[((size_t)b * a.nh + h) * a.Lq + i] = acc;
    }
  }
  f32x4_t dQ[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) dQ[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int rloc = w * 16 + (lane >> 4) * 4;
  const int nkt = (a.Lk + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_direct(sK, kb, a.ldk, kt * 64, a.Lk, tid);
    stage_direct(sV, vb, a.ldv, kt * 64, a.Lk, tid);
    stage_trans(sKt, kb, a.ldk, kt * 64, a.Lk, tid);
    __syncthreads();
    const uint4 q0 = frag(sQ, w * 16, 0, lane), q1 = frag(sQ, w * 16, 1, lane);
    const uint4 g0 = frag(sdO, w * 16, 0, lane), g1 = frag(sdO, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = mfma16(q0, frag(sK, nt * 16, 0, lane), s);
      s = mfma16(q1, frag(sK, nt * 16, 1, lane), s);
      dp = mfma16(g0, frag(sV, nt * 16, 0, lane), dp);
      dp = mfma16(g1, frag(sV, nt * 16, 1, lane), dp);
      const int j = kt * 64 + nt * 16 + (lane & 15);
      const bool kvalid = j < a.Lk;
      float kbias = 0.f;
      if (kvalid && a.kmask != nullptr && a.kmask[(size_t)(b / a.kv_group) * a.ldmask + j] == 0.f) kbias = -a.mask_inf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + rloc + r;
        float raw = s[r], dpv = dp[r];
        if (a.gq != nullptr && kvalid && i < a.Lq) { raw += rpr_gather(a, a.gq, b, h, i, j); dpv += rpr_gather(a, a.gd, b, h, i, j); }
        float sc = raw * a.scale + kbias;
        if (a.causal && j > a.q_pos0 + i) sc -= a.mask_inf;
        const float p = kvalid ? __expf(sc - sL[rloc + r]) : 0.f;
        float ms = 1.f;
        if (a.thr) {
          const uint64_t idx = (((uint64_t)b * a.nh + h) * a.Lq + i) * a.Lk + j;
          ms = zk_drop_scale(seed, a.sid, idx, a.thr, a.inv_keep);
        }
        dpv *= ms;
        const float ds = p * (dpv - sD[rloc + r]) * a.scale;
        sdS[(rloc + r) * ALD + nt * 16 + (lane & 15)] = f2bf(ds);
        if (a.dsb != nullptr) sPd[(rloc + r) * ALD + nt * 16 + (lane & 15)] = f2bf(i < a.Lq ? p * ms : 0.f);
      }
    }
    __syncthreads();
    if (a.dsb != nullptr) {   // per-index sums of dS and P over this key tile (rows of this wave)
      rpr_bucket_accum(a, acc_dsb, i0, kt * 64, w, lane, [&](int row, int jl) { return bf2f(sdS[row * ALD + jl]); });
      rpr_bucket_accum(a, acc_pb, i0, kt * 64, w, lane, [&](int row, int jl) { return bf2f(sPd[row * ALD + jl]); });
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 da = frag(sdS, w * 16, kk, lane);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) dQ[nb] = mfma16(da, frag(sKt, nb * 16, kk, lane), dQ[nb]);
    }
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + rloc + r;
      if (i < a.Lq) dq[((size_t)b * a.Lq + i) * lddq + h * AD + c] = f2bf(dQ[nb][r]);
    }
  }
  if (a.dsb != nullptr) {
    rpr_bucket_store(a, acc_dsb, a.dsb, b, h, i0, w, lane);
    rpr_bucket_store(a, acc_pb, a.pb, b, h, i0, w, lane);
  }
}

// ---- backward B: grid (ceil(Lk/64), nh, B) -> dK, dV
__global__ void __launch_bounds__(256) k_attn_bwd_dkv_mfma(AttnArgs a, const bf16_t* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ Dbuf,
                                                           bf16_t* __restrict__ dk, int lddk,
                                                           bf16_t* __restrict__ dv, int lddv) {
  __shared__ __attribute__((aligned(16))) bf16_t sK[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sV[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sQ[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdO[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sQt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdOt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sPt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdSt[TQ * ALD];
  __shared__ float sD[TQ];
  __shared__ float sL[TQ];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)(b / a.kv_group) * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)(b / a.kv_group) * a.bsv + h * AD;
  const bf16_t* dob = dout + (size_t)b * a.Lq * lddo + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  stage_direct(sK, kb, a.ldk, j0, a.Lk, tid);
  stage_direct(sV, vb, a.ldv, j0, a.Lk, tid);
  f32x4_t dK[4], dV[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) { dK[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dV[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  const int jloc = w * 16 + (lane >> 4) * 4;   // key rows of this lane in the C layout
  float kbias[4];
  bool kval[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = j0 + jloc + r;
    kval[r] = j < a.Lk;
    kbias[r] = (kval[r] && a.kmask != nullptr && a.kmask[(size_t)(b / a.kv_group) * a.ldmask + j] == 0.f) ? -a.mask_inf : 0.f;
  }
  const int nqt = (a.Lq + 63) / 64;
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    stage_direct(sQ, qb, a.ldq, qt * 64, a.Lq, tid);
    stage_direct(sdO, dob, lddo, qt * 64, a.Lq, tid);
    stage_trans(sQt, qb, a.ldq, qt * 64, a.Lq, tid);
    stage_trans(sdOt, dob, lddo, qt * 64, a.Lq, tid);
    if (tid < 64) {
      const int i = qt * 64 + tid;
      sL[tid] = (i < a.Lq) ? lse[((size_t)b * a.nh + h) * a.Lq + i] : 0.f;
      sD[tid] = (i < a.Lq) ? Dbuf[((size_t)b * a.nh + h) * a.Lq + i] : 0.f;
    }
    __syncthreads();
    const uint4 k0 = frag(sK, w * 16, 0, lane), k1 = frag(sK, w * 16, 1, lane);
    const uint4 v0 = frag(sV, w * 16, 0, lane), v1 = frag(sV, w * 16, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {   // 16-query blocks; C layout: col = query, row = key
      f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
      st = mfma16(k0, frag(sQ, nt * 16, 0, lane), st);
      st = mfma16(k1, frag(sQ, nt * 16, 1, lane), st);
      dpt = mfma16(v0, frag(sdO, nt * 16, 0, lane), dpt);
      dpt = mfma16(v1, frag(sdO, nt * 16, 1, lane), dpt);
      const int il = nt * 16 + (lane & 15);
      const int i = qt * 64 + il;
      const bool qvalid = i < a.Lq;
      const float li = sL[il], Di = sD[il];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + jloc + r;
        float raw = st[r], dpr = dpt[r];
        if (a.gq != nullptr && qvalid && kval[r]) { raw += rpr_gather(a, a.gq, b, h, i, j); dpr += rpr_gather(a, a.gd, b, h, i, j); }
        float sc = raw * a.scale + kbias[r];
        if (a.causal && j > a.q_pos0 + i) sc -= a.mask_inf;
        const float p = (qvalid && kval[r]) ? __expf(sc - li) : 0.f;
        float ms = 1.f;
        if (a.thr) {
          const uint64_t idx = (((uint64_t)b * a.nh + h) * a.Lq + i) * a.Lk + j;
          ms = zk_drop_scale(seed, a.sid, idx, a.thr, a.inv_keep);
        }
        const float ds = p * (dpr * ms - Di) * a.scale;
        sPt[(jloc + r) * ALD + il] = f2bf(p * ms);
        sdSt[(jloc + r) * ALD + il] = f2bf(ds);
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 pa = frag(sPt, w * 16, kk, lane);
      const uint4 da = frag(sdSt, w * 16, kk, lane);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        dV[nb] = mfma16(pa, frag(sdOt, nb * 16, kk, lane), dV[nb]);
        dK[nb] = mfma16(da, frag(sQt, nb * 16, kk, lane), dK[nb]);
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = chan_of_phys(nb * 16 + (lane & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + jloc + r;
      if (j < a.Lk) {
        dk[((size_t)b * a.Lk + j) * lddk + h * AD + c] = f2bf(dK[nb][r]);
        dv[((size_t)b * a.Lk + j) * lddv + h * AD + c] = f2bf(dV[nb][r]);
      }
    }
  }
}

// ---- backward, single tile (Lq <= 64 and Lk <= 64): grid (1, nh, B) -> dQ, dK, dV in ONE pass.
// The two-kernel form above reads Q, K, V, dO twice and recomputes S and dP; at the training
// shapes of the north star (L = 64) a whole (batch, head) problem is one 64x64 tile, so one
// workgroup computes P and dS once (wave w owns query rows 16w..16w+15) and all three gradients
// follow from LDS: dQ = dS K, dK = dS^T Q, dV = P^T dO.  The operand tiles of the first phase are
// dead by then and their LDS is reused for dS, P^T and dS^T (7 tiles = 64.5 KB -> 2 workgroups/CU).
__global__ void __launch_bounds__(256) k_attn_bwd_fused64(AttnArgs a, const bf16_t* __restrict__ o, int ldo,
                                                          const bf16_t* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dq, int lddq,
                                                          bf16_t* __restrict__ dk, int lddk,
                                                          bf16_t* __restrict__ dv, int lddv) {
  __shared__ __attribute__((aligned(16))) bf16_t sQ[TQ * ALD];    // phase 2: dS   [query][key]
  __shared__ __attribute__((aligned(16))) bf16_t sK[TQ * ALD];    // phase 2: P^T  [key][query]
  __shared__ __attribute__((aligned(16))) bf16_t sV[TQ * ALD];    // phase 2: dS^T [key][query]
  __shared__ __attribute__((aligned(16))) bf16_t sdO[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sKt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sQt[TQ * ALD];
  __shared__ __attribute__((aligned(16))) bf16_t sdOt[TQ * ALD];
  __shared__ float sL[TQ];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* qb = a.q + (size_t)b * a.bsq + h * AD;
  const bf16_t* kb = a.k + (size_t)b * a.bsk + h * AD;
  const bf16_t* vb = a.v + (size_t)b * a.bsv + h * AD;
  (void)o; (void)ldo;
  const bf16_t* dob = dout + (size_t)b * a.Lq * lddo + h * AD;
  const uint64_t seed = a.thr ? *a.seed : 0;

  // every global load of the prologue first (tiles, the O / dO rows of D_i, lse, the key mask), then the LDS stores
  DirectRegs rQ, rdO, rK, rV;
  TransRegs t0, t1;
  load_direct(rQ, qb, a.ldq, 0, a.Lq, tid);
  load_direct(rdO, dob, lddo, 0, a.Lq, tid);
  load_direct(rK, kb, a.ldk, 0, a.Lk, tid);
  load_direct(rV, vb, a.ldv, 0, a.Lk, tid);
  if (tid < 128) {
    load_trans(t0, kb, a.ldk, 0, a.Lk, tid);
  } else {
    load_trans(t0, qb, a.ldq, 0, a.Lq, tid - 128);
    load_trans(t1, dob, lddo, 0, a.Lq, tid - 128);
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
  store_direct(sQ, rQ, tid);
  store_direct(sdO, rdO, tid);
  store_direct(sK, rK, tid);
  store_direct(sV, rV, tid);
  if (tid < 128) {
    store_trans(sKt, t0, tid);
  } else {
    store_trans(sQt, t0, tid - 128);
    store_trans(sdOt, t1, tid - 128);
  }
  if (dpart == 0) sL[dr] = (dr < a.Lq) ? lse_r : 0.f;
  __syncthreads();
  // ---- phase 1: P and dS of query rows 16w .. 16w+15 against all 64 keys, kept in registers
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
      Di[r] += __shfl_xor(Di[r], 1, 64);
      Di[r] += __shfl_xor(Di[r], 2, 64);
      Di[r] += __shfl_xor(Di[r], 4, 64);
      Di[r] += __shfl_xor(Di[r], 8, 64);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dsv[nt][r] = pu[nt][r] * (dpv[nt][r] - Di[r]) * a.scale;
  }
  __syncthreads();                     // every wave is done reading sQ / sK / sV / sdO
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
  if (a.dsb != nullptr) {   // bucket sums for the relative-position table products (see AttnArgs)
    rpr_bucket_rows(a, a.dsb, b, h, 0, w, lane, [&](int row, int j) { return bf2f(sdS[row * ALD + j]); });
    rpr_bucket_rows(a, a.pb, b, h, 0, w, lane, [&](int row, int j) { return bf2f(sPt[j * ALD + row]); });
  }
  // ---- phase 2: wave w -> dQ rows 16w.. (queries) and dK / dV rows 16w.. (keys)
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
  store_tile_rows(oQ, dq + (size_t)b * a.Lq * lddq + h * AD, lddq, a.Lq, tid);
  store_tile_rows(oK, dk + (size_t)b * a.Lk * lddk + h * AD, lddk, a.Lk, tid);
  store_tile_rows(oV, dv + (size_t)b * a.Lk * lddv + h * AD, lddv, a.Lk, tid);
}

// =====================================================================================
// C-ABI
// =====================================================================================
static int fill_args(AttnArgs* a, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, int B,
                     int nh, int Lq, int Lk, int d, const float* kmask, int causal, int q_pos0, float scale,
                     float mask_inf, const void* rpr_k, const void* rpr_v, int max_rel, float drop_p,
                     const uint64_t* seed, uint32_t sid) {
  a->q = (const bf16_t*)q; a->k = (const bf16_t*)k; a->v = (const bf16_t*)v;
  a->ldq = ldq; a->ldk = ldk; a->ldv = ldv;
  a->bsq = (long)Lq * ldq; a->bsk = (long)Lk * ldk; a->bsv = (long)Lk * ldv; a->kv_group = 1;
  a->B = B; a->nh = nh; a->Lq = Lq; a->Lk = Lk; a->d = d;
  a->kmask = kmask; a->causal = causal; a->q_pos0 = q_pos0; a->scale = scale; a->mask_inf = mask_inf;
  a->rpr_k = (const bf16_t*)rpr_k; a->rpr_v = (const bf16_t*)rpr_v; a->max_rel = max_rel;
  a->thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  a->inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  a->seed = seed; a->sid = sid;
  a->ldmask = Lk; a->pos_dev = nullptr; a->pos_flags = 0;
  a->gq = nullptr; a->gd = nullptr; a->pb = nullptr; a->dsb = nullptr; a->ldg = 0; a->nrp = 0;
  return 0;
}

static bool attn_mfma_ok(const AttnArgs& a, int extra_ld_or) {
  // relative positions run on the MFMA kernels only in the decomposed form (gather tables supplied)
  if (a.d != AD || ((a.rpr_k != nullptr || a.rpr_v != nullptr) && a.gq == nullptr)) return false;
  if ((a.ldq | a.ldk | a.ldv | extra_ld_or) % 8) return false;
  if ((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) != 0) return false;
  return true;
}

extern "C" {

int zk_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int nh, int Lq, int Lk,
                int d, int ldq, int ldk, int ldv, int ldo, const float* kmask, int causal, int q_pos0,
                float scale, float mask_inf, const void* rpr_k, const void* rpr_v, int max_rel, float drop_p,
                const uint64_t* seed, uint32_t sid, long bsq, long bsk, long bsv, int kv_group, int impl,
                const int* pos_dev, int pos_flags, const void* rpr_gq, void* rpr_pb, int rpr_ldg, int rpr_nrp,
                hipStream_t stream) {
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr), "zk_attn_fwd: rpr_k and rpr_v go together");
  ZK_CHECK_ARG((rpr_gq == nullptr) == (rpr_pb == nullptr), "zk_attn_fwd: rpr_gq and rpr_pb go together");
  ZK_CHECK_ARG(rpr_gq == nullptr || (rpr_nrp >= 2 * max_rel + 1 && rpr_nrp <= 64 && rpr_ldg >= nh * rpr_nrp),
               "zk_attn_fwd: rpr tables need 2*max_rel+1 <= nrp <= 64 and ldg >= nh*nrp");
  ZK_CHECK_ARG(kv_group >= 1, "zk_attn_fwd: kv_group must be >= 1");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_attn_fwd: dropout needs a seed pointer");
  ZK_CHECK_ARG(Lk >= 1, "zk_attn_fwd: Lk must be >= 1");
  if (B == 0 || Lq == 0) return 0;
  AttnArgs a;
  fill_args(&a, q, k, v, ldq, ldk, ldv, B, nh, Lq, Lk, d, kmask, causal, q_pos0, scale, mask_inf, rpr_k, rpr_v,
            max_rel, drop_p, seed, sid);
  if (bsq > 0) a.bsq = bsq;
  if (bsk > 0) a.bsk = bsk;
  if (bsv > 0) a.bsv = bsv;
  a.kv_group = kv_group;
  a.pos_dev = pos_dev; a.pos_flags = pos_dev ? pos_flags : 0;
  a.gq = (const float*)rpr_gq; a.pb = (bf16_t*)rpr_pb; a.ldg = rpr_ldg; a.nrp = rpr_nrp;
  const bool ok = attn_mfma_ok(a, ldo) && Lk <= 256 && (a.bsq % 8 == 0) && (a.bsk % 8 == 0) && (a.bsv % 8 == 0) && (((uintptr_t)out & 15) == 0);
  ZK_CHECK_ARG(impl != 2 || ok, "zk_attn_fwd: MFMA kernel needs d=64, no rpr, Lk<=256, ld%%8==0");
  if (impl == 2 || (impl == 0 && ok)) {
    dim3 grid((Lq + TQ - 1) / TQ, nh, B);
    const int nkt = (Lk + 63) / 64;
    if (nkt == 1) hipLaunchKernelGGL(k_attn_fwd_mfma<1>, grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
    else if (nkt == 2) hipLaunchKernelGGL(k_attn_fwd_mfma<2>, grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
    else if (nkt == 3) hipLaunchKernelGGL(k_attn_fwd_mfma<3>, grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
    else hipLaunchKernelGGL(k_attn_fwd_mfma<4>, grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  ZK_CHECK_ARG(Lk <= NAIVE_MAXK * 64, "zk_attn_fwd: Lk=%d exceeds the reference kernel limit %d", Lk,
               NAIVE_MAXK * 64);
  ZK_CHECK_ARG(d <= 256, "zk_attn_fwd: d=%d > 256", d);
  const long rows = (long)B * nh * Lq;
  hipLaunchKernelGGL(k_attn_fwd_naive, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, (bf16_t*)out, ldo,
                     lse);
  ZK_LAUNCH_CHECK();
  return 0;
}

size_t zk_attn_bwd_workspace(int B, int nh, int Lq) { return (size_t)B * nh * Lq * sizeof(float); }

// dq/dk/dv are [B*L, ld] matrices like q/k/v.  drpr_k/drpr_v (fp32 [2*max_rel+1, d]) are
// ACCUMULATED into (caller zeroes them).
int zk_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                void* dq, void* dk, void* dv, float* drpr_k, float* drpr_v, int B, int nh, int Lq, int Lk, int d,
                int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, const float* kmask,
                int causal, int q_pos0, float scale, float mask_inf, const void* rpr_k, const void* rpr_v,
                int max_rel, float drop_p, const uint64_t* seed, uint32_t sid, int impl, void* workspace,
                size_t ws_bytes, const void* rpr_gq, const void* rpr_gd, void* rpr_pb, void* rpr_dsb, int rpr_ldg,
                int rpr_nrp, hipStream_t stream) {
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr), "zk_attn_bwd: rpr_k and rpr_v go together");
  ZK_CHECK_ARG(rpr_k == nullptr || rpr_gq != nullptr || (drpr_k != nullptr && drpr_v != nullptr),
               "zk_attn_bwd: rpr needs grad outputs");
  ZK_CHECK_ARG(rpr_gq == nullptr || (rpr_gd != nullptr && rpr_pb != nullptr && rpr_dsb != nullptr &&
                                     rpr_nrp >= 2 * max_rel + 1 && rpr_nrp <= 64 && rpr_ldg >= nh * rpr_nrp),
               "zk_attn_bwd: decomposed rpr needs gq, gd, pb, dsb and 2*max_rel+1 <= nrp <= 64");
  ZK_CHECK_ARG(ws_bytes >= zk_attn_bwd_workspace(B, nh, Lq), "zk_attn_bwd: workspace too small");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_attn_bwd: dropout needs a seed pointer");
  if (B == 0 || Lq == 0 || Lk == 0) return 0;
  AttnArgs a;
  fill_args(&a, q, k, v, ldq, ldk, ldv, B, nh, Lq, Lk, d, kmask, causal, q_pos0, scale, mask_inf, rpr_k, rpr_v,
            max_rel, drop_p, seed, sid);
  float* Dbuf = (float*)workspace;
  a.gq = (const float*)rpr_gq; a.gd = (const float*)rpr_gd; a.pb = (bf16_t*)rpr_pb; a.dsb = (bf16_t*)rpr_dsb;
  a.ldg = rpr_ldg; a.nrp = rpr_nrp;
  const bool ok = attn_mfma_ok(a, ldo | lddo | lddq | lddk | lddv) &&
                  ((((uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0);   // 16-byte row accesses
  ZK_CHECK_ARG(rpr_gq == nullptr || (ok && impl != 1), "zk_attn_bwd: decomposed rpr runs on the MFMA kernels only");
  ZK_CHECK_ARG(impl != 2 || ok, "zk_attn_bwd: MFMA kernel needs d=64, no rpr, ld%%8==0");
  ZK_CHECK_ARG(impl != 3 || ok, "zk_attn_bwd: MFMA kernels need d=64, no rpr, ld%%8==0");
  if ((impl == 0 || impl == 2) && ok && Lq <= TQ && Lk <= TQ) {      // impl 3 forces the two-kernel form
    hipLaunchKernelGGL(k_attn_bwd_fused64, dim3(1, nh, B), dim3(256), 0, stream, a, (const bf16_t*)out, ldo,
                       (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  if (impl == 2 || impl == 3 || (impl == 0 && ok)) {
    hipLaunchKernelGGL(k_attn_bwd_dq_mfma, dim3((Lq + TQ - 1) / TQ, nh, B), dim3(256), 0, stream, a,
                       (const bf16_t*)out, ldo, (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, Dbuf);
    ZK_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_attn_bwd_dkv_mfma, dim3((Lk + TQ - 1) / TQ, nh, B), dim3(256), 0, stream, a,
                       (const bf16_t*)dout, lddo, lse, Dbuf, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  ZK_CHECK_ARG(Lk <= NAIVE_MAXK * 64, "zk_attn_bwd: Lk=%d exceeds the reference kernel limit %d", Lk,
               NAIVE_MAXK * 64);
  ZK_CHECK_ARG(d <= 256, "zk_attn_bwd: d=%d > 256", d);
  const long qrows = (long)B * nh * Lq, krows = (long)B * nh * Lk;
  hipLaunchKernelGGL(k_attn_bwd_dq_naive, dim3((unsigned)((qrows + 3) / 4)), dim3(256), 0, stream, a,
                     (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, Dbuf, drpr_k, drpr_v);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attn_bwd_dkv_naive, dim3((unsigned)((krows + 3) / 4)), dim3(256), 0, stream, a,
                     (const bf16_t*)dout, lddo, lse, Dbuf, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
  ZK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
