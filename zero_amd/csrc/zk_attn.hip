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
#include "zk_attn_dev.h"
#include "zk_prog.h"
#include "zk_gemm2_dev.h"      // gemm_tile: the output projection + residual + LayerNorm of k_attn_out_ln

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
// MFMA kernels (d = 64): tile functions in zk_attn_dev.h
// =====================================================================================
// ---- forward: grid (ceil(Lq/64), nh, B); NKT = ceil(Lk/64) <= 4
template <int NKT, bool RPR = false>
__global__ void __launch_bounds__(256) k_attn_fwd_mfma(AttnArgs a, bf16_t* __restrict__ out, int ldo,
                                                       float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[AttnFwdLds<NKT, RPR>::BYTES];
  attn_apply_pos(a);
  attn_fwd_tile<NKT, false, RPR>(smem, a, out, ldo, lse, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- attention + output projection + residual + LayerNorm of a sub-layer in ONE launch (zk_attn_out_ln; func.py:218-256
// dot_attention core, func.py:206-216 o_map, func.py:321-324 + 289-303 in the order of transformer.py:57-58).
// Workgroup (sentence b, head h) -- 512 threads, the grid of the 64x64 o_map tiles with SENTENCE-ALIGNED row tiles (rows
// b Lq .. b Lq + Lq - 1, Lq <= 64) and nh = N/64 column tiles -- first computes the attention output of its (b, h) into
// `att` exactly as k_attn_fwd_mfma does (waves 0-3; the producer waves of the GEMM only keep the barrier count), makes
// it visible and raises flag (b, h); once the nh heads of the sentence are there it runs its o_map tile over
// A = att[rows of b, :] with the in-launch LayerNorm epilogue (gemm_tile<.., LN = 3>, the row block's workgroups ARE the
// sentence's heads).  The attention launch, its drain and the cold start of the projection disappear; att still goes to
// memory once (the backward reads it).  Visibility: with the sentence's workgroups on one XCD (local) the L2 is the
// point of coherence -- stores acknowledged (vmcnt) before the flag, the LDS-DMA of the K loop misses the L1 (nothing of
// `att` was read by this CU before) -- otherwise agent-scope release / acquire around the flag.
#define ZK_ATTN_FWD_BARRIERS(NKT) (4 * (NKT))     // workgroup barriers inside attn_fwd_tile<NKT> (no relative positions)
// PRO > 0 (zk_proj_attn_out_ln): the projection in front of the attention (func.py:206-216: the merged qkv_map of a
// self-attention, PRO = 3, or the q_map of a cross-attention, PRO = 1) runs HERE as well: workgroup (b, h) needs nothing but
// its own head's 64 columns of q (k, v) over its sentence's rows -- x[rows of b, :] W[:, p H + h*64 ..] + bias, PRO tiles of
// the 64x64 GEMM with K = H -- so the projection launch, its drain and the attention's cold start disappear without any
// exchange between workgroups.  The tiles are written to the q / k / v matrices as before (the backward reads them) and read
// back by this workgroup's attention tile past the L1 (FRESH loads: the stores were acknowledged by the L2).  Same tile
// function, same K order as the projection launch: bit-identical.
#ifdef ZK_ATTN_TRACE
__device__ unsigned long long zk_attn_wg_trace[4096];
#define ZK_WG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) zk_attn_wg_trace[2 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ZK_WG_STAMP(k)
#endif
struct AttnPro { const bf16_t* x; int ldx; const bf16_t* w; int ldw; const float* bias; int K; int N; };

// The projection tiles of one (sentence, head) as ONE K loop (PRO = 3: q, k, v; PRO = 1: q): out[m0.., p nslab + n0 + 0..63] =
// x[m0.., :] W[:, p nslab + n0 + 0..63] + bias for p < PRO.  Three calls of gemm_tile cost three ring fills, three epilogues
// through the LDS and fetched the x rows three times (measured: as long as the projection launch they replaced).  Here the
// ring runs through: K tile kt is a half stage {x tile, W_q tile} followed (PRO = 3) by a half stage {W_k tile, W_v tile};
// the x fragments of a K tile stay in registers for the second half, so x goes through the address path once; four
// producer waves issue every LDS-DMA, three half stages in flight; the accumulators leave straight from the registers
// (64-byte row pieces) -- no LDS round trip, no barrier.  Per element the same MFMAs in the same K order as gemm_tile and
// the same rounding (acc + bias -> bf16): bit-identical to the projection launch.  K % 64 == 0; rows >= M are not stored.
template <int PRO>
__device__ __forceinline__ void proj_heads_tile(unsigned char* smem, const bf16_t* __restrict__ X, int ldx,
                                                const bf16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                int K, int m0, int M, int n0, int nslab,
                                                bf16_t* tQ, bf16_t* tK, bf16_t* tVt, uint4 (&pk)[PRO]) {
  constexpr int NS = 4, HALF = 128 * 64;            // bf16 elements per half stage (two 64 x 64 operand tiles)
  constexpr int HPK = PRO == 3 ? 2 : 1;             // half stages per K tile
  constexpr int PER = 4;                            // DMA instructions per producer wave per half stage
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int dwave = wave - 4, wm = wave >> 1, wn = wave & 1;
  const int nk = K >> 6, nh = nk * HPK;
  const uint32_t ring_addr = lds_addr(ring);
  f32x16_t acc[PRO];
#pragma unroll
  for (int p = 0; p < PRO; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  // the epilogue's bias values (PRO x 64 floats) are brought into the LDS by the producer waves before the K loop (whose
  // barriers order them): no global load stands between the last MFMA and the attention, and no registers are held
  // across the K loop [24 + 8 prefetched registers per thread cost the kernel its second workgroup per CU: 49 us a launch]
  __shared__ __attribute__((aligned(16))) float s_pbias[PRO * 64];
  const int row = tid >> 3, cc = (tid & 7) * 8;
  if (producer && tid - 256 < PRO * 64) {
    const int i = tid - 256;
    s_pbias[i] = bias != nullptr ? bias[(i >> 6) * nslab + n0 + (i & 63)] : 0.f;
  }
  if (producer) {
    DmaPlan<64, 4> planA, planB;
    dma_plan<64, false, 4>(planA, ldx, m0, M, dwave, lane);
    dma_plan<64, true, 4>(planB, ldw, 0, 64, dwave, lane);
    const bf16_t* baseA = X;
    const bf16_t* baseB = W + n0;
    int hs = 0;                                     // next half stage to issue
    auto issue = [&]() {
      const uint32_t st = ring_addr + (uint32_t)((hs % NS) * HALF * 2);
      if (PRO == 1 || (hs & 1) == 0) {
        dma_tile<64, false, false, 4>(planA, baseA, 0, K, st, dwave);
        dma_tile<64, true, false, 4>(planB, baseB, 0, K, st + 64 * 128, dwave);
        baseA += 64;
        if (PRO == 1) baseB += (size_t)64 * ldw;
      } else {
        dma_tile<64, true, false, 4>(planB, baseB + nslab, 0, K, st, dwave);
        dma_tile<64, true, false, 4>(planB, baseB + 2 * nslab, 0, K, st + 64 * 128, dwave);
        baseB += (size_t)64 * ldw;
      }
      ++hs;
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) if (i < nh) issue();
    for (int s = 0; s < nh; ++s) {
      // half stage s has landed once at most min(NS - 2, nh - 1 - s) later ones are still in flight
      const int later = nh - 1 - s;
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (hs < nh) issue();                         // refills the half stage everybody finished reading at this barrier
    }
  } else {
    bf16x8_t af[4];
    for (int s = 0; s < nh; ++s) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s == 0) ZK_AT(1);
      if (s == nh - 1) ZK_AT(7);
      const bf16_t* h0 = ring + (s % NS) * HALF;
      const bf16_t* h1 = h0 + 64 * 64;
      if (PRO == 1 || (s & 1) == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) af[kk] = load_frag<64, false>(h0, wm * 32, kk, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk], load_frag<64, true>(h1, wn * 32, kk, lane), acc[0], 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          acc[PRO > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk], load_frag<64, true>(h0, wn * 32, kk, lane),
                                                                         acc[PRO > 1 ? 1 : 0], 0, 0, 0);
          acc[PRO > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk], load_frag<64, true>(h1, wn * 32, kk, lane),
                                                                         acc[PRO > 2 ? 2 : 0], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue through the LDS (the ring is dead): PRO fp32 tiles, then every thread stores 16-byte row pieces
  // [straight from the registers -- 48 two-byte stores per lane -- the launch was 5 us LONGER than with three gemm_tile calls]
  // and leaves the SAME bf16 values where the attention tile of this workgroup expects its operands (tQ / tK: [row][ALD];
  // tVt: [physical channel][key] as store_trans writes it; rows >= M zero as load_direct / load_trans return them): the
  // attention starts from the LDS -- no store acknowledgement, no load round trip between the projection and the attention
  // (2.3 us of the launch, profiles/r06_attn_out_ln_timeline_*).  The global copies (the backward reads them) are stored
  // by the CALLER from pk[] (this thread's row m0 + tid / 8, columns p nslab + n0 + (tid % 8) 8 .. +7; rows >= M: none)
  // behind the attention tile, next to the attention output's stores: a workgroup barrier in between would wait for them.
  constexpr int CLD = 64 + 4;
  float* sC = reinterpret_cast<float*>(smem);
  ZK_AT(2);
  __syncthreads();                                  // every compute wave is done reading the last half stages
  ZK_AT(8);
  if (!producer) {
    // C layout of the 32x32 MFMA: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = wn * 32 + (lane & 31);
#pragma unroll
    for (int p = 0; p < PRO; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) sC[p * 64 * CLD + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CLD + col] = acc[p][r];
  }
  // (LDS-only barriers from here on: __syncthreads() would also wait for global memory)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  ZK_AT(9);
  lds_barrier();
  ZK_AT(10);
  auto tile8 = [&](int p, int r, int c0, const float* bv) -> uint4 {          // bf16(acc + bias) of 8 consecutive columns
    float v[8];
    const float4 a = *reinterpret_cast<const float4*>(sC + p * 64 * CLD + r * CLD + c0);
    const float4 b = *reinterpret_cast<const float4*>(sC + p * 64 * CLD + r * CLD + c0 + 4);
    const float4 ba = *reinterpret_cast<const float4*>(bv), bb = *reinterpret_cast<const float4*>(bv + 4);
    v[0] = a.x * 1.f + ba.x; v[1] = a.y * 1.f + ba.y; v[2] = a.z * 1.f + ba.z; v[3] = a.w * 1.f + ba.w;
    v[4] = b.x * 1.f + bb.x; v[5] = b.y * 1.f + bb.y; v[6] = b.z * 1.f + bb.z; v[7] = b.w * 1.f + bb.w;
    return pack8(v);
  };
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int p = 0; p < PRO; ++p) pk[p] = tile8(p, row, cc, s_pbias + p * 64 + cc);
  ZK_AT(11);
  lds_barrier();                                    // every read of the fp32 tiles is done: the operand tiles may land on them
  ZK_AT(12);
  const bool live = m0 + row < M;
  *reinterpret_cast<uint4*>(tQ + row * ALD + cc) = live ? pk[0] : zero4;
  if constexpr (PRO == 3) {
    *reinterpret_cast<uint4*>(tK + row * ALD + cc) = live ? pk[1] : zero4;
    // V^T as store_trans leaves it: element (key = row, channel c = cc + j) at [phys(c)][row], phys(c) = (c % 8) 8 + c / 8 =
    // 8 j + cc / 8 -- eight two-byte writes per thread from the packed row piece it already holds [a second pass of 128
    // threads over the fp32 tile (four more reads + conversions each) made the read-back phase 2.4 us]
    const uint4 vv = live ? pk[2] : zero4;
    const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      tVt[(8 * j + (cc >> 3)) * ALD + row] = (bf16_t)((j & 1) ? (w4[j >> 1] >> 16) : (w4[j >> 1] & 0xffffu));
  }
  ZK_AT(13);
  lds_barrier();
}

template <int NKT, int PRO = 0>
__global__ void __launch_bounds__(512) k_attn_out_ln(AttnArgs a, bf16_t* __restrict__ att, int ldatt, float* __restrict__ lse,
                                                     const bf16_t* __restrict__ Wo, int ldw, int M, int N, TileSched ts,
                                                     GemmEpi e, unsigned long long* __restrict__ flags, AttnPro pro) {
  constexpr int GEMM_LDS = DldsCfg<64, 64, 4>::LDS_BYTES, ATT_LDS = AttnFwdLds<NKT, false>::BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > ATT_LDS ? GEMM_LDS : ATT_LDS];
  int tm, tn, z;
  tile_of_block(ts, tm, tn, z);                   // tm: sentence, tn: head
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ZK_AT(0);
  ZK_WG_STAMP(0);
  [[maybe_unused]] uint4 ppk[PRO > 0 ? PRO : 1];
  if constexpr (PRO > 0) {
    const int M_pro = min(M, (tm + 1) * a.Lq);
    bf16_t* tQ = reinterpret_cast<bf16_t*>(smem);                                 // attn_fwd_tile's sQ, sK; V^T behind its tiles
    proj_heads_tile<PRO>(smem, pro.x, pro.ldx, pro.w, pro.ldw, pro.bias, pro.K, tm * a.Lq, M_pro, tn * 64, N, tQ, tQ + TQ * ALD,
                         reinterpret_cast<bf16_t*>(smem + AttnFwdLds<NKT, false>::BASE), ppk);
  }
  ZK_AT(3);
  if (wave < 4) {
    attn_fwd_tile<NKT, false, false, PRO>(smem, a, att, ldatt, lse, 0, tn, tm,
                                                                      reinterpret_cast<const bf16_t*>(smem + AttnFwdLds<NKT, false>::BASE));
  } else {
#pragma unroll 1
    for (int i = 0; i < ZK_ATTN_FWD_BARRIERS(NKT); ++i) __syncthreads();
  }
  if constexpr (PRO > 0) {                          // the projected tiles' global copies (see proj_heads_tile)
    const int gm = tm * a.Lq + (tid >> 3);
    if (gm < min(M, (tm + 1) * a.Lq)) {
      bf16_t* dst = const_cast<bf16_t*>(a.q) + (size_t)gm * a.ldq + tn * 64 + (tid & 7) * 8;
#pragma unroll
      for (int p = 0; p < PRO; ++p) *reinterpret_cast<uint4*>(dst + p * N) = ppk[p];
    }
  }
  ZK_AT(4);
  const bool local = e.sy_local != 0;
  const uint32_t tag = (*e.sy_epoch << 8) | e.sy_site;
  unsigned long long* fl = flags + (size_t)tm * a.nh;
  __builtin_amdgcn_s_waitcnt(0);                  // this thread's rows of att have reached the L2
  __syncthreads();
  if (tid == 0) {
    if (local) __hip_atomic_store(fl + tn, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else __hip_atomic_store(fl + tn, (unsigned long long)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < a.nh && tid != tn) {
    int spins = 0;
    for (;;) {
      unsigned long long v;
      if (local) {
        const unsigned long long zero = 0;      // an atomic OR of zero executes in the L2 whatever the L1 holds
        asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(fl + tid), "v"(zero) : "memory");
      } else {
        v = __hip_atomic_load(fl + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((uint32_t)v == tag) break;
      if (++spins > (1 << 15)) {
        if (e.sy_err != nullptr) __hip_atomic_store(e.sy_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  ZK_AT(5);
  if (!local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // The A operand is clamped to the SENTENCE's own rows (M_own): with Lq < 64 the 64-row tile would otherwise fetch rows
  // of sentence tm + 1 -- masked later by sy_rows, but the fetch could leave lines of `att` in this CU's L1 BEFORE that
  // sentence's heads have written them, and a workgroup of sentence tm + 1 placed on this CU in a later resident round
  // would then read them stale on the sy_local path (no L1 invalidate there).  Clamped rows only feed rows >= sy_rows.
  const int M_own = min(M, (tm + 1) * a.Lq);
  gemm_tile<64, 64, 4, false, false, 4, 4, false, false, 3>(smem, att, Wo, M_own, N, ldatt, ldw, 0, N, tm * a.Lq, tn * 64,
                                                            nullptr, e, 1);
  ZK_AT(6);
  ZK_WG_STAMP(1);
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
    acc = quad_sum(acc);
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

// ---- backward, single tile (Lq <= 64 and Lk <= 64): grid (1, nh, B) -> dQ, dK, dV in ONE pass (attn_bwd_fused64_tile)
// (256, 2): two waves per SIMD = two workgroups per CU for the variants whose LDS allows it -- the OPROJ prologue keeps a
// 512-column chunk of its operands in flight and would otherwise take 286 registers and halve the occupancy
template <bool RPR = false, bool OPROJ = false>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_fused64(AttnArgs a, const bf16_t* __restrict__ o, int ldo,
                                                             const bf16_t* __restrict__ dout, int lddo,
                                                             const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dq, int lddq,
                                                             bf16_t* __restrict__ dk, int lddk,
                                                             bf16_t* __restrict__ dv, int lddv, float* __restrict__ rpr_part,
                                                             AttnOProj op) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[RPR ? ATTN_BWD64_RPR_LDS_BYTES : ATTN_BWD64_LDS_BYTES];
  int h = blockIdx.y, b = blockIdx.z;
  if (gridDim.y == 1) {
    // 1-D grid (bwd64_grid): workgroups go round-robin over the 8 XCDs, so id = xcd + 8 * slot puts the nh heads of a
    // sentence on ONE XCD -- they read the same 64 rows of dY (OPROJ) and the same rows of the qkv buffer, which then come
    // from HBM once instead of once per head (head-major ids sent every head of a sentence to a different XCD: 47 MB
    // fetched per launch against 17 MB algorithmic, profiles/r05_pmc_traffic.json)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    h = slot % a.nh;
    b = (slot / a.nh) * 8 + xcd;
    if (b >= a.B) return;
  }
  attn_bwd_fused64_tile<RPR, OPROJ>(smem, a, o, ldo, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, h, b, rpr_part, op);
}

#ifdef ZK_EXPERIMENTS   // measured, no gain over the two launches (profiles/r04_negative_results.txt item 8)
// ---- attention backward + the dgrad that consumes its dq / dk / dv + the LayerNorm backward below, ONE launch
// (zk_attn_bwd_ln; the mirror of k_attn_out_ln): workgroup (sentence b, head h) -- 256 threads, the grid of the 64x64 tiles
// of the dgrad with sentence-aligned row tiles -- runs the single-tile attention backward of its (b, h) with the o_map dgrad
// folded in (attn_bwd_fused64_tile<false, true>, unchanged), which writes its 64 columns of dQ / dK / dV; once the nh heads
// of the sentence have raised their flags it runs its tile of   dx = [dQ dK dV](b) W_qkv^T (+ ds)   (or dQ W_q^T for the
// cross-attention) with the LayerNorm-backward epilogue of zk_gemm_ln_bwd (gemm_tile<.., LN = 4>, four waves issuing
// their own LDS-DMA).  Visibility as in k_attn_out_ln.
__global__ void __launch_bounds__(256, 2) k_attn_bwd_ln(AttnArgs a, const bf16_t* __restrict__ o, int ldo,
                                                        const float* __restrict__ lse, bf16_t* __restrict__ dq, int lddq,
                                                        bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv,
                                                        AttnOProj op, const bf16_t* __restrict__ dA, int lda,
                                                        const bf16_t* __restrict__ W, int ldw, int K, int M, int N,
                                                        TileSched ts, GemmEpi e, unsigned long long* __restrict__ flags) {
  constexpr int GEMM_LDS = DldsCfg<64, 64, 4>::LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[GEMM_LDS > ATTN_BWD64_LDS_BYTES ? GEMM_LDS : ATTN_BWD64_LDS_BYTES];
  int tm, tn, z;
  tile_of_block(ts, tm, tn, z);                   // tm: sentence, tn: head
  const int tid = threadIdx.x;
  attn_bwd_fused64_tile<false, true>(smem, a, o, ldo, nullptr, 0, lse, dq, lddq, dk, lddk, dv, lddv, tn, tm, nullptr, op);
  const bool local = e.sy_local != 0;
  const uint32_t tag = (*e.sy_epoch << 8) | e.sy_site;
  unsigned long long* fl = flags + (size_t)tm * a.nh;
  __builtin_amdgcn_s_waitcnt(0);                  // this thread's rows of dQ / dK / dV have reached the L2
  __syncthreads();
  if (tid == 0) {
    if (local) __hip_atomic_store(fl + tn, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else __hip_atomic_store(fl + tn, (unsigned long long)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < a.nh && tid != tn) {
    int spins = 0;
    for (;;) {
      unsigned long long v;
      if (local) {
        const unsigned long long zero = 0;
        asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(fl + tid), "v"(zero) : "memory");
      } else {
        v = __hip_atomic_load(fl + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((uint32_t)v == tag) break;
      if (++spins > (1 << 15)) {
        if (e.sy_err != nullptr) __hip_atomic_store(e.sy_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (!local) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int M_own = min(M, (tm + 1) * a.Lq);      // (as in k_attn_out_ln: never fetch another sentence's rows of dA)
  gemm_tile<64, 64, 4, false, true, 4, 0, false, false, 4>(smem, dA, W, M_own, N, lda, ldw, 0, K, tm * a.Lq, tn * 64, nullptr, e, 1);
}

#endif  // ZK_EXPERIMENTS

// relative positions folded in, tiles taking turns in 72 KB of LDS: two workgroups per CU (attn_bwd_rpr64_tile)
__global__ void __launch_bounds__(256, 2) k_attn_bwd_rpr64(AttnArgs a, const bf16_t* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse, bf16_t* __restrict__ dq, int lddq,
                                                           bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv,
                                                           int lddv, float* __restrict__ rpr_part) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ATTN_BWD64_RPR2_LDS_BYTES];
  attn_bwd_rpr64_tile(smem, a, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, blockIdx.y, blockIdx.z, rpr_part);
}

// sum of the per-(sentence, head) table-gradient partials of the folded relative-position backward:
// part fp32 [nslices][2][64][64] -> dk / dv fp32 [n] (n = (2*max_rel+1)*64 leading elements of each table slab).
// Block = 16 columns x 16 slice groups: every thread adds nslices/16 values with all its loads in flight.
__global__ void __launch_bounds__(256) k_rpr_part_reduce(const float* __restrict__ part, int nslices, int n,
                                                         float* __restrict__ dk, float* __restrict__ dv) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const float* src = part + (size_t)blockIdx.y * TQ * AD;
  float t = 0.f;
  if (c < n)
    for (int s = g; s < nslices; s += 16) t += src[(size_t)s * 2 * TQ * AD + c];
  red[g][cl] = t;
  __syncthreads();
  if (g == 0 && c < n) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][cl];
    (blockIdx.y == 0 ? dk : dv)[c] = v;
  }
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

extern "C" int zk_zero(void* p, size_t bytes, hipStream_t stream);   // zk_elem.hip
// grid of k_attn_bwd_fused64: one id per (sentence, head), sentences dealt to the XCDs (see the kernel); tuning key 15 bit 2
// restores the head-major 3-D grid for A/B runs
static dim3 bwd64_grid(int nh, int B) { return (g_tune[15] & 4) ? dim3(1, nh, B) : dim3(nh * ((B + 7) / 8) * 8); }
#ifdef ZK_ATTN_TRACE
__device__ unsigned long long zk_attn_trace_buf[16];
extern "C" int zk_attn_wg_trace_read(unsigned long long* out, int n) {      // {start, end} of every workgroup of the last k_attn_out_ln launch
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_attn_wg_trace), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int zk_attn_trace_read(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zk_attn_trace_buf), sizeof(unsigned long long) * 16);
}
#endif
extern "C" {
size_t zk_attn_bwd_rpr_workspace(int B, int nh, int Lq) {
  return (size_t)B * nh * Lq * sizeof(float) + (size_t)B * nh * 2 * TQ * AD * sizeof(float);
}

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
  // relative positions folded into the MFMA tile (impl bit 8 = 256, set by the caller): tables in LDS, no products in HBM
  const bool fold = (impl & 256) && rpr_k != nullptr && rpr_gq == nullptr && d == AD && 2 * max_rel + 1 <= 64 &&
                    ((((uintptr_t)rpr_k | (uintptr_t)rpr_v) & 15) == 0);
  impl &= 255;
  if (fold) { a.rpr_k = nullptr; a.rpr_v = nullptr; }          // attn_mfma_ok() refuses undecomposed tables
  bool ok = attn_mfma_ok(a, ldo) && Lk <= 256 && (a.bsq % 8 == 0) && (a.bsk % 8 == 0) && (a.bsv % 8 == 0) && (((uintptr_t)out & 15) == 0);
  if (fold) { a.rpr_k = (const bf16_t*)rpr_k; a.rpr_v = (const bf16_t*)rpr_v; }
  const bool folded = fold && ok && impl != 1;
  if (fold && !folded) ok = false;
  ZK_CHECK_ARG(impl != 2 || ok, "zk_attn_fwd: MFMA kernel needs d=64, no rpr, Lk<=256, ld%%8==0");
  if (impl == 2 || (impl == 0 && ok)) {
    dim3 grid((Lq + TQ - 1) / TQ, nh, B);
    const int nkt = (Lk + 63) / 64;
    if (folded) {
      if (zk_prog_active()) return zk_prog_reject("attention with relative positions");
      if (nkt == 1) hipLaunchKernelGGL((k_attn_fwd_mfma<1, true>), grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
      else if (nkt == 2) hipLaunchKernelGGL((k_attn_fwd_mfma<2, true>), grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
      else if (nkt == 3) hipLaunchKernelGGL((k_attn_fwd_mfma<3, true>), grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
      else hipLaunchKernelGGL((k_attn_fwd_mfma<4, true>), grid, dim3(256), 0, stream, a, (bf16_t*)out, ldo, lse);
      ZK_LAUNCH_CHECK();
      return 0;
    }
    if (zk_prog_active()) return zk_prog_record_attn_fwd(a, (bf16_t*)out, ldo, lse, nkt);
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
  if (zk_prog_active()) return zk_prog_reject("attention forward on the reference kernel");
  const long rows = (long)B * nh * Lq;
  hipLaunchKernelGGL(k_attn_fwd_naive, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, (bf16_t*)out, ldo,
                     lse);
  ZK_LAUNCH_CHECK();
  return 0;
}

// Attention forward + output projection + residual + LayerNorm in one launch (see k_attn_out_ln): the arguments of
// zk_attn_fwd (MFMA path only: d = 64, Lq <= 64, Lk <= 256, no relative positions; `att` [B*Lq, nh*64] and lse are still
// written, the backward reads them) and of zk_gemm_add_ln (W_o [nh*64, N = nh*64]; slots / epoch / site / err shared with
// it; `flags`: zk_attn_out_ln_flags(B, nh) bytes, zero-filled once).  Returns 2 without launching when the shape is
// not covered (the caller issues zk_attn_fwd and zk_gemm_add_ln).
size_t zk_attn_out_ln_flags(int B, int nh) { return (size_t)B * nh * sizeof(unsigned long long); }

static int attn_out_ln_impl(const void* q, const void* k, const void* v, void* att, float* lse, int B, int nh, int Lq, int Lk, int d,
                   int ldq, int ldk, int ldv, int ldatt, const float* kmask, int causal, float scale, float mask_inf,
                   float attn_drop_p, const uint64_t* seed, uint32_t attn_sid, int kv_group, const void* Wo, int ldw,
                   const float* bias, const void* residual, int ldr, float drop_p, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, void* flags, size_t flags_bytes, const uint32_t* epoch, uint32_t site, int* err,
                   hipStream_t stream, int npro, const AttnPro& pro) {
  const int N = nh * AD, M = B * Lq;
  if (d != AD || Lq < 1 || Lq > 64 || Lk < 1 || Lk > 256 || N > 1024 || kv_group < 1) return 2;
  ZK_CHECK_ARG(residual != nullptr && ldr % 8 == 0 && gamma != nullptr && beta != nullptr && y != nullptr && att != nullptr,
               "zk_attn_out_ln: att, residual (row stride a multiple of 8), gamma, beta and y are required");
  ZK_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "zk_attn_out_ln: mean / rstd must both be given");
  ZK_CHECK_ARG((attn_drop_p == 0.f && drop_p == 0.f) || seed != nullptr, "zk_attn_out_ln: dropout needs a seed pointer");
  ZK_CHECK_ARG(slots != nullptr && flags != nullptr && epoch != nullptr && site >= 1 && site <= 255,
               "zk_attn_out_ln: slots, flags, epoch and a site in 1..255 are required");
  ZK_CHECK_ARG(slots_bytes >= (size_t)((M + 127) / 128 * 128) * (size_t)(N / 64) * 16 && flags_bytes >= zk_attn_out_ln_flags(B, nh),
               "zk_attn_out_ln: slots / flags too small");
  if (B == 0) return 0;
  AttnArgs a;
  fill_args(&a, q, k, v, ldq, ldk, ldv, B, nh, Lq, Lk, d, kmask, causal, 0, scale, mask_inf, nullptr, nullptr, 0, attn_drop_p,
            seed, attn_sid);
  a.kv_group = kv_group;
  const uintptr_t al = (uintptr_t)att | (uintptr_t)Wo | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)gamma |
                       (uintptr_t)beta | (uintptr_t)s_out | (uintptr_t)y | (uintptr_t)slots | (uintptr_t)flags;
  if (!attn_mfma_ok(a, ldatt | ldw) || (al & 15) != 0 || (a.bsq % 8) || (a.bsk % 8) || (a.bsv % 8)) return 2;
  if (zk_prog_active()) return 2;
  GemmEpi e;
  e.C = s_out; e.ldc = N; e.out_f32 = 0; e.alpha = 1.f; e.bias = bias;
  e.res = (const bf16_t*)residual; e.ldr = ldr; e.act = 0; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f;
  e.thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  e.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  e.seed = seed; e.sid = sid;
  e.ln_eps = eps; e.ln_invh = 1.f / (float)N;
  e.sy_slots = (unsigned long long*)slots; e.sy_epoch = epoch; e.sy_site = site;
  e.sy_gamma = gamma; e.sy_beta = beta; e.sy_y = (bf16_t*)y; e.sy_ldy = N;
  e.sy_mean = mean; e.sy_rstd = rstd; e.sy_err = err; e.sy_rows = Lq;
  TileSched ts;
  ts.tiles_m = B; ts.tiles_n = nh; ts.n_major = 0; ts.xcd_remap = 1;
  const long nwg = (long)B * nh;
  e.sy_local = (nwg % (8 * nh) == 0 && !(g_tune[15] & 1)) ? 1 : 0;
  const int nkt = (Lk + 63) / 64;
  const dim3 grid((unsigned)nwg), blk(512);
#define ZK_AOL(NKT_, PRO_) hipLaunchKernelGGL((k_attn_out_ln<NKT_, PRO_>), grid, blk, 0, stream, a, (bf16_t*)att, ldatt, lse, \
                                               (const bf16_t*)Wo, ldw, M, N, ts, e, (unsigned long long*)flags, pro)
#define ZK_AOL_N(PRO_) do { if (nkt == 1) ZK_AOL(1, PRO_); else if (nkt == 2) ZK_AOL(2, PRO_); else if (nkt == 3) ZK_AOL(3, PRO_); \
                            else ZK_AOL(4, PRO_); } while (0)
  if (npro == 0) ZK_AOL_N(0);
  else if (npro == 1) ZK_AOL_N(1);
  else ZK_AOL(1, 3);                               // (merged qkv_map: Lk = Lq <= 64, one key tile)
#undef ZK_AOL_N
#undef ZK_AOL
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_attn_out_ln(const void* q, const void* k, const void* v, void* att, float* lse, int B, int nh, int Lq, int Lk, int d,
                   int ldq, int ldk, int ldv, int ldatt, const float* kmask, int causal, float scale, float mask_inf,
                   float attn_drop_p, const uint64_t* seed, uint32_t attn_sid, int kv_group, const void* Wo, int ldw,
                   const float* bias, const void* residual, int ldr, float drop_p, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, void* flags, size_t flags_bytes, const uint32_t* epoch, uint32_t site, int* err,
                   hipStream_t stream) {
  AttnPro pro = {nullptr, 0, nullptr, 0, nullptr, 0, 0};
  return attn_out_ln_impl(q, k, v, att, lse, B, nh, Lq, Lk, d, ldq, ldk, ldv, ldatt, kmask, causal, scale, mask_inf, attn_drop_p,
                          seed, attn_sid, kv_group, Wo, ldw, bias, residual, ldr, drop_p, sid, gamma, beta, eps, s_out, y, mean,
                          rstd, slots, slots_bytes, flags, flags_bytes, epoch, site, err, stream, 0, pro);
}

// zk_attn_out_ln with the projection in front of the attention inside the launch as well (k_attn_out_ln<.., PRO>):
// q (k, v) = x Wp + bp is computed by the workgroup that consumes it.  pro = 3: Wp [Kp, 3 nh 64] is the merged qkv_map of a
// self-attention (q, k = q + nh*64, v = q + 2 nh*64 columns of ONE [B*Lq, ldq] matrix, Lk = Lq, kv_group = 1); pro = 1: Wp
// [Kp, nh 64] is the q_map of a cross-attention (k, v as for zk_attn_out_ln).  x [B*Lq, ldx]; the projected tiles are still
// written (the backward reads them).  Returns 2 (nothing launched) when the shape is not covered.
int zk_proj_attn_out_ln(const void* x, int ldx, const void* Wp, int ldwp, const float* bp, int Kp, int pro,
                   const void* q, const void* k, const void* v, void* att, float* lse, int B, int nh, int Lq, int Lk, int d,
                   int ldq, int ldk, int ldv, int ldatt, const float* kmask, int causal, float scale, float mask_inf,
                   float attn_drop_p, const uint64_t* seed, uint32_t attn_sid, int kv_group, const void* Wo, int ldw,
                   const float* bias, const void* residual, int ldr, float drop_p, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, void* flags, size_t flags_bytes, const uint32_t* epoch, uint32_t site, int* err,
                   hipStream_t stream) {
  ZK_CHECK_ARG(pro == 1 || pro == 3, "zk_proj_attn_out_ln: pro must be 1 (q_map) or 3 (merged qkv_map)");
  ZK_CHECK_ARG(x != nullptr && Wp != nullptr && q != nullptr, "zk_proj_attn_out_ln: x, Wp and q are required");
  const int N = nh * AD;
  if (Kp < 64 || Kp % 64 != 0 || ldx % 8 != 0 || ldwp % 8 != 0 || ldq % 8 != 0) return 2;
  if ((((uintptr_t)x | (uintptr_t)Wp | (uintptr_t)bp | (uintptr_t)q) & 15) != 0) return 2;
  if (pro == 3) {
    const bf16_t* qq = (const bf16_t*)q;
    if ((const bf16_t*)k != qq + N || (const bf16_t*)v != qq + 2 * N || ldk != ldq || ldv != ldq || Lk != Lq || kv_group != 1)
      return 2;
  }
  AttnPro ap = {(const bf16_t*)x, ldx, (const bf16_t*)Wp, ldwp, bp, Kp, pro * N};
  return attn_out_ln_impl(q, k, v, att, lse, B, nh, Lq, Lk, d, ldq, ldk, ldv, ldatt, kmask, causal, scale, mask_inf, attn_drop_p,
                          seed, attn_sid, kv_group, Wo, ldw, bias, residual, ldr, drop_p, sid, gamma, beta, eps, s_out, y, mean,
                          rstd, slots, slots_bytes, flags, flags_bytes, epoch, site, err, stream, pro, ap);
}

#ifdef ZK_EXPERIMENTS   // measured, no gain over the two launches (profiles/r04_negative_results.txt item 8)
// Attention backward + the dgrad of the projection(s) in front of it + the LayerNorm backward of the sub-layer below, one
// launch (see k_attn_bwd_ln).  The attention arguments of zk_attn_bwd on its single-tile path with the o_map dgrad folded in
// (d = 64, Lq, Lk <= 64, no relative positions; (oproj_dy, oproj_w): dO = dY W_o[h*64.., :]^T); dq / dk / dv are written as
// before (the weight gradients read them).  Then dx = dA W^T + residual with dA [B*Lq, K] = the matrix dq (dk, dv) are
// columns of (K = 3 nh 64 for a merged q/k/v projection, nh 64 for a query projection), W [N = nh*64, ldw] K-contiguous,
// and the arguments of zk_gemm_ln_bwd for the LayerNorm below; partials: [B][3][N] (one row per SENTENCE: nblk = B for the
// reduction).  flags: zk_attn_out_ln_flags(B, nh) bytes, zero-filled once (may be the forward's).  Returns 2 without
// launching when the shape is not covered.
int zk_attn_bwd_ln(const void* q, const void* k, const void* v, const void* out, const float* lse, void* dq, void* dk, void* dv,
                   int B, int nh, int Lq, int Lk, int d, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                   const float* kmask, int causal, float scale, float mask_inf, float attn_drop_p, const uint64_t* seed,
                   uint32_t attn_sid, const void* oproj_dy, int oproj_lddy, const void* oproj_w, int oproj_ldw, int oproj_n,
                   const void* dA, int lda, const void* W, int ldw, int K, const void* residual, int ldr, const void* s,
                   const float* mean, const float* rstd, const float* gamma, float drop_p, uint32_t sid, void* dsum,
                   void* dy_out, float* partials, void* slots, size_t slots_bytes, void* flags, size_t flags_bytes,
                   const uint32_t* epoch, uint32_t site, int* err, hipStream_t stream) {
  const int N = nh * AD, M = B * Lq;
  if (d != AD || Lq < 1 || Lq > TQ || Lk < 1 || Lk > TQ || N > 1024 || K < 64 || K % 64 != 0) return 2;
  ZK_CHECK_ARG(oproj_dy != nullptr && oproj_w != nullptr, "zk_attn_bwd_ln: the (dY, W_o) pair is required");
  ZK_CHECK_ARG(s != nullptr && mean != nullptr && rstd != nullptr && gamma != nullptr && dsum != nullptr && partials != nullptr,
               "zk_attn_bwd_ln: s, mean, rstd, gamma, dsum and partials are required");
  ZK_CHECK_ARG(residual == nullptr || ldr % 8 == 0, "zk_attn_bwd_ln: residual row stride must be a multiple of 8");
  ZK_CHECK_ARG((attn_drop_p == 0.f && drop_p == 0.f) || seed != nullptr, "zk_attn_bwd_ln: dropout needs a seed pointer");
  ZK_CHECK_ARG(drop_p == 0.f || dy_out != nullptr, "zk_attn_bwd_ln: dropout needs dy_out");
  ZK_CHECK_ARG(slots != nullptr && flags != nullptr && epoch != nullptr && site >= 1 && site <= 255,
               "zk_attn_bwd_ln: slots, flags, epoch and a site in 1..255 are required");
  ZK_CHECK_ARG(slots_bytes >= (size_t)((M + 127) / 128 * 128) * (size_t)(N / 64) * 16 && flags_bytes >= zk_attn_out_ln_flags(B, nh),
               "zk_attn_bwd_ln: slots / flags too small");
  if (B == 0) return 0;
  AttnArgs a;
  fill_args(&a, q, k, v, ldq, ldk, ldv, B, nh, Lq, Lk, d, kmask, causal, 0, scale, mask_inf, nullptr, nullptr, 0, attn_drop_p,
            seed, attn_sid);
  const bool shape = oproj_n >= 128 && oproj_n % 128 == 0 && oproj_lddy % 8 == 0 && oproj_ldw % 8 == 0 &&
                     oproj_lddy >= oproj_n && oproj_ldw >= oproj_n && ((((uintptr_t)oproj_dy | (uintptr_t)oproj_w) & 15) == 0);
  const uintptr_t al = (uintptr_t)out | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)dA | (uintptr_t)W |
                       (uintptr_t)residual | (uintptr_t)s | (uintptr_t)gamma | (uintptr_t)dsum | (uintptr_t)dy_out |
                       (uintptr_t)partials | (uintptr_t)slots | (uintptr_t)flags;
  if (!attn_mfma_ok(a, ldo | lddq | lddk | lddv | lda | ldw) || !shape || (al & 15) != 0 || zk_prog_active()) return 2;
  AttnOProj op = {(const bf16_t*)oproj_dy, oproj_lddy, (const bf16_t*)oproj_w, oproj_ldw, oproj_n};
  GemmEpi e;
  e.C = nullptr; e.ldc = N; e.out_f32 = 0; e.alpha = 1.f; e.bias = nullptr;
  e.res = (const bf16_t*)residual; e.ldr = ldr; e.act = 0; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f;
  e.thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  e.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  e.seed = seed; e.sid = sid;
  e.ln_invh = 1.f / (float)N;
  e.sy_slots = (unsigned long long*)slots; e.sy_epoch = epoch; e.sy_site = site;
  e.sy_gamma = gamma; e.sy_y = (bf16_t*)dsum; e.sy_ldy = N; e.sy_err = err;
  e.sy_s = (const bf16_t*)s; e.sy_lds = N; e.sy_mean_in = mean; e.sy_rstd_in = rstd;
  e.sy_dy = (bf16_t*)dy_out; e.sy_part = partials; e.sy_rows = Lq;
  TileSched ts;
  ts.tiles_m = B; ts.tiles_n = nh; ts.n_major = 0; ts.xcd_remap = 1;
  const long nwg = (long)B * nh;
  e.sy_local = (nwg % (8 * nh) == 0 && !(g_tune[15] & 1)) ? 1 : 0;
  hipLaunchKernelGGL(k_attn_bwd_ln, dim3((unsigned)nwg), dim3(256), 0, stream, a, (const bf16_t*)out, ldo, lse, (bf16_t*)dq,
                     lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, op, (const bf16_t*)dA, lda, (const bf16_t*)W, ldw, K, M, N, ts, e,
                     (unsigned long long*)flags);
  ZK_LAUNCH_CHECK();
  return 0;
}

#endif  // ZK_EXPERIMENTS

size_t zk_attn_bwd_workspace(int B, int nh, int Lq) { return (size_t)B * nh * Lq * sizeof(float); }

// dq/dk/dv are [B*L, ld] matrices like q/k/v.  drpr_k/drpr_v (fp32 [2*max_rel+1, d]) are
// ACCUMULATED into (caller zeroes them).
int zk_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                void* dq, void* dk, void* dv, float* drpr_k, float* drpr_v, int B, int nh, int Lq, int Lk, int d,
                int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, const float* kmask,
                int causal, int q_pos0, float scale, float mask_inf, const void* rpr_k, const void* rpr_v,
                int max_rel, float drop_p, const uint64_t* seed, uint32_t sid, int impl, void* workspace,
                size_t ws_bytes, const void* rpr_gq, const void* rpr_gd, void* rpr_pb, void* rpr_dsb, int rpr_ldg,
                int rpr_nrp, const void* oproj_dy, int oproj_lddy, const void* oproj_w, int oproj_ldw, int oproj_n,
                hipStream_t stream) {
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr), "zk_attn_bwd: rpr_k and rpr_v go together");
  ZK_CHECK_ARG((oproj_dy == nullptr) == (oproj_w == nullptr), "zk_attn_bwd: oproj_dy and oproj_w go together");
  ZK_CHECK_ARG(dout != nullptr || oproj_dy != nullptr, "zk_attn_bwd: dout or the (dY, W_o) pair it is the product of");
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
  // impl | 256 with tables but no decomposed products: relative positions folded into the single-tile kernel
  const bool fold = (impl & 256) && rpr_k != nullptr && rpr_gq == nullptr && d == AD && 2 * max_rel + 1 <= 64 &&
                    drpr_k != nullptr && drpr_v != nullptr && ((((uintptr_t)rpr_k | (uintptr_t)rpr_v) & 15) == 0) &&
                    ws_bytes >= zk_attn_bwd_workspace(B, nh, Lq) + (size_t)B * nh * 2 * TQ * AD * sizeof(float);
  // impl | 256 is also a CONTRACT on the table gradients: the call overwrites drpr_k / drpr_v whichever kernels end up
  // running (the folded kernel writes them; the reference kernels below accumulate with atomics, so they are cleared
  // first).  Without the bit the table gradients are accumulated into and the caller clears them.
  const bool overwrite_tables = (impl & 256) && drpr_k != nullptr && drpr_v != nullptr;
  // impl | 512 (with | 256): the folded kernel's per-(sentence, head) table-gradient partials stay in the workspace
  // (fp32 [B*nh][2][64][64] behind the B*nh*Lq floats of D) and the CALLER sums them -- e.g. all attention layers of a
  // step in one grouped reduction launch instead of one launch per layer.  Only honoured when the folded kernel runs:
  // the return value is 1 then (0: the tables were written here, as without the bit).
  const bool defer_tables = (impl & 512) != 0;
  // impl | 1024 (with | 256): the folded kernel in its first form (every tile resident, 151 KB of LDS, one workgroup per
  // CU) instead of the 72-KB form with two workgroups per CU -- kept for A/B runs and as the reference of the kernel tests
  const bool resident_tiles = (impl & 1024) != 0;
  impl &= 255;
  if (fold) { a.rpr_k = nullptr; a.rpr_v = nullptr; }
  bool ok = attn_mfma_ok(a, ldo | lddo | lddq | lddk | lddv) &&
            ((((uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0);   // 16-byte row accesses
  if (fold) { a.rpr_k = (const bf16_t*)rpr_k; a.rpr_v = (const bf16_t*)rpr_v; }
  const bool folded = fold && ok && (impl == 0 || impl == 2) && Lq <= TQ && Lk <= TQ;
  if (fold && !folded) ok = false;
  // (oproj_dy, oproj_w) given: dout = dY . W_o[h*64 .., :]^T is computed inside the single-tile kernel instead of read
  // (the o_map dgrad GEMM of the caller disappears).  Only that kernel does it: when the call would run anything else,
  // NOTHING is launched and the return value is 2 -- the caller then forms dout itself and calls again without the pair.
  AttnOProj op = {(const bf16_t*)oproj_dy, oproj_lddy, (const bf16_t*)oproj_w, oproj_ldw, oproj_n};
  if (oproj_dy != nullptr) {
    const bool single = (impl == 0 || impl == 2) && ok && Lq <= TQ && Lk <= TQ && rpr_gq == nullptr && !zk_prog_active() &&
                        (rpr_k == nullptr || folded);
    const bool shape = oproj_n >= 128 && oproj_n % 128 == 0 && oproj_lddy % 8 == 0 && oproj_ldw % 8 == 0 &&
                       oproj_lddy >= oproj_n && oproj_ldw >= oproj_n &&
                       ((((uintptr_t)oproj_dy | (uintptr_t)oproj_w) & 15) == 0);
    if (!single || !shape) return 2;
  }
  ZK_CHECK_ARG(rpr_gq == nullptr || (ok && impl != 1), "zk_attn_bwd: decomposed rpr runs on the MFMA kernels only");
  ZK_CHECK_ARG(impl != 2 || ok, "zk_attn_bwd: MFMA kernel needs d=64, no rpr, ld%%8==0");
  ZK_CHECK_ARG(impl != 3 || ok, "zk_attn_bwd: MFMA kernels need d=64, no rpr, ld%%8==0");
  if (folded && Lq <= TQ && Lk <= TQ) {
    if (zk_prog_active()) return zk_prog_reject("attention backward with relative positions");
    // table-gradient partials behind Dbuf in the workspace, summed over the B*nh (sentence, head) tiles afterwards
    float* part = (float*)workspace + (size_t)B * nh * Lq;
    if (oproj_dy != nullptr)
      hipLaunchKernelGGL((k_attn_bwd_fused64<true, true>), bwd64_grid(nh, B), dim3(256), 0, stream, a, (const bf16_t*)out, ldo,
                         (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, part, op);
    else if (!resident_tiles)
      hipLaunchKernelGGL(k_attn_bwd_rpr64, dim3(1, nh, B), dim3(256), 0, stream, a, (const bf16_t*)dout, lddo, lse,
                         (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, part);
    else
      hipLaunchKernelGGL((k_attn_bwd_fused64<true, false>), bwd64_grid(nh, B), dim3(256), 0, stream, a, (const bf16_t*)out, ldo,
                         (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, part, op);
    ZK_LAUNCH_CHECK();
    if (defer_tables) return 1;
    const int n = (2 * max_rel + 1) * AD;
    hipLaunchKernelGGL(k_rpr_part_reduce, dim3((n + 15) / 16, 2), dim3(256), 0, stream, (const float*)part, B * nh, n,
                       drpr_k, drpr_v);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  if (zk_prog_active()) {
    if (!((impl == 0 || impl == 2) && ok && Lq <= TQ && Lk <= TQ && rpr_gq == nullptr))
      return zk_prog_reject("attention backward that is not one 64x64 MFMA tile per (sentence, head)");
    return zk_prog_record_attn_bwd64(a, (const bf16_t*)out, ldo, (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq,
                                     (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
  }
  if ((impl == 0 || impl == 2) && ok && Lq <= TQ && Lk <= TQ) {      // impl 3 forces the two-kernel form
    if (oproj_dy != nullptr)
      hipLaunchKernelGGL((k_attn_bwd_fused64<false, true>), bwd64_grid(nh, B), dim3(256), 0, stream, a, (const bf16_t*)out, ldo,
                         (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv,
                         (float*)nullptr, op);
    else
      hipLaunchKernelGGL((k_attn_bwd_fused64<false, false>), bwd64_grid(nh, B), dim3(256), 0, stream, a, (const bf16_t*)out, ldo,
                         (const bf16_t*)dout, lddo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv,
                         (float*)nullptr, op);
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
  if (zk_prog_active()) return zk_prog_reject("attention backward on the reference kernels");
  if (overwrite_tables && rpr_k != nullptr) {
    const size_t nb = (size_t)(2 * max_rel + 1) * d * sizeof(float);
    if (zk_zero(drpr_k, nb, stream) != 0 || zk_zero(drpr_v, nb, stream) != 0) return -1;
  }
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
