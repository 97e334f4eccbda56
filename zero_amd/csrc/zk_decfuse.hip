// zk_decfuse.hip -- one attention sub-layer of a cached decode step in ONE launch plus the LayerNorm that follows it
// (transformer.py:120-175 decoder body at Lq = 1, func.py:124-287 dot_attention with a cache, search.py:115-236).
//
// A decode step runs on B*K rows (128 at the benchmark shape): every op is a ~5-7 us latency chain whatever it
// computes, so a step costs (number of launches) x ~7 us.  An attention sub-layer was four launches -- q (or qkv)
// projection, attention, output projection, residual + LayerNorm -- each a GEMM on 128 rows that kept 16 CUs busy.
// Here one workgroup owns (16 consecutive beam rows = 4 sentences at beam 4, head h) -- the 16 rows of one MFMA tile;
// with one sentence per workgroup every workgroup re-read the head's 128 KB of weights and the launch was bound by the
// L2 -> CU path (32 MB for 32 sentences; profiles/r02_dec_attn_phases.txt):
//   prologue  (optional) the PREVIOUS sub-layer's residual + LayerNorm for the 16 rows, recomputed by each of the nh
//             workgroups of the row group (16 x H elements: cheaper than a launch; zk_lndec_dev.h);
//   q_h     = bf16(x . Wq[:, h] + bq[h])          16 x 64 on the matrix cores (16x16x32 bf16), the 64 KB slice of the
//             TRANSPOSED weight streamed once, every fragment prefetched into registers at entry
//             (self-attention: k_h, v_h too; written to the per-beam caches at slot `time`)
//   P       = softmax(scale * q_h K_h^T + mask)   keys of the sentence (cross) or of the beam's cache (self)
//   ctx_h   = bf16(P V_h)
//   part[h] = ctx_h . Wo[h rows, :]               16 x H fp32, the head's share of the output projection
// and the NEXT launch (zk_ln_decode with parts, or the prologue of the next fused sub-layer) adds the nh partial
// products in a fixed order, the bias and the residual and normalises.  Nothing is exchanged between workgroups inside the
// launch.  Values are rounded to bf16 where the launch-per-op path stores bf16 (q, k, v, P, ctx, y), the fp32 sums run in
// a different order: results agree to fp32 rounding, not bit for bit.
// Everything that does not depend on values computed in the launch (weight fragments, the first keys / values of every
// thread, the prologue's inputs) is requested before the first use: the launch is ONE memory round trip plus the math.
#include "zk_common.h"
#include "zk_lndec_dev.h"

typedef __bf16 dbf16x8_t __attribute__((ext_vector_type(8)));
typedef float df32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ df32x4_t dmfma16(const uint4& a, const uint4& b, df32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dbf16x8_t, a), __builtin_bit_cast(dbf16x8_t, b), c, 0,
                                                 0, 0);
}

// make DECTRACE=1: workgroup 0 stamps the constant 100 MHz clock at the phase boundaries (scripts/dec_attn_trace.py)
#ifdef ZK_DEC_TRACE
__device__ unsigned long long zk_dec_trace_buf[16];
__device__ int zk_dec_trace_mode;     // bit 0 / 1 / 2: skip the q-weight / o-weight / key-value prefetch (timing only)
#define ZK_DM(bit) (zk_dec_trace_mode & (1 << (bit)))
#define ZK_DT(i)                                                                                        \
  do {                                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0) zk_dec_trace_buf[i] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#else
#define ZK_DT(i)
#define ZK_DM(bit) 0
#endif

#define ZK_DEC_GROUP_DEFAULT 4
struct DecAttnArgs {
  LnDecArgs pro;                 // pro.gamma == NULL: no prologue, pro.x is the block input
  // TRANSPOSED projection weights (row = output channel, K-contiguous: the MFMA B fragment is one 16-byte load)
  const bf16_t* wqt; int ldw; const float* bq;   // cross: q_map^T [H, H]; self: qkv_map^T [3H, H] (q | k | v rows), bias [3H]
  const bf16_t* k; const bf16_t* v; int ldk, ldv; long bsk, bsv;
  // cross: key j of sentence b at k + b*bsk + j*ldk.  self: caches, key j of ROW r at k + r*bsk + j*ldk (written here
  // at j = time)
  const float* kmask; int ldmask;                // cross: [B, ldmask] 1 = valid key
  const bf16_t* wot; int ldwo;                   // o_map^T [H, H]
  float* part;                                   // [nh][B*R][H]
  int B, R, nh, Lk;                              // Lk: cross = source length; self = cache capacity
  float scale, mask_inf;
  const int* time_dev; int time;                 // self: this step's slot (the device value wins); cross with relative
                                                 // positions: the query's position
  int gr;                                        // beam rows per workgroup (<= 16)
  // relative positions (modules/rpr.py:10-75 at Lq = 1; round 4: folded into this launch, transformer_rpr's decode step
  // no longer takes the launch-per-op path): tables bf16 [2 max_rel + 1][64] (shared by the heads) or NULL.  Key j of a
  // query at position t reads row clip(t - j, -max_rel, max_rel) + max_rel: the scores get q . Rk[row], the context
  // sum_j P_j Rv[row].
  const bf16_t* rpr_k; const bf16_t* rpr_v; int max_rel;
};

// 512 threads = 8 waves; one workgroup = (16 consecutive beam rows, head h).  MAXC = ceil(H / 512).
template <bool SELF, int MAXC>
__global__ void __launch_bounds__(512) k_dec_attn(DecAttnArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lk8 = (lane >> 4) * 8;
  const int h = blockIdx.x % a.nh, row0 = (blockIdx.x / a.nh) * a.gr;
  const int H = a.pro.H, R = a.R, rows = a.B * a.R;
  const int NR = min(a.gr, rows - row0);                 // valid rows of this group
  const bool rpr = a.rpr_k != nullptr;
  const int t = (SELF || rpr) ? (a.time_dev != nullptr ? *a.time_dev : a.time) : 0;
  const int Lk = SELF ? min(a.Lk, t + 1) : a.Lk;
  const int LkPad = (a.Lk + 3) & ~3;
  const int XLD = H + 8;
  constexpr int NP = SELF ? 3 : 1;                      // projections of this head: q (k, v)
  constexpr int KS = 8 * MAXC;                          // 32-wide k steps per half of K (H / 64 of them are real)
  constexpr int NOT = 4 * MAXC;                         // output tiles of 16 columns per wave (H / 128 real)
  bf16_t* sXb = reinterpret_cast<bf16_t*>(smem);        // [16][XLD]; rows >= NR are never read back
  float* sPart = reinterpret_cast<float*>(sXb + 16 * XLD);   // [3][8 waves][16 rows][16]: halves of the projection tiles
  float* sQ = sPart + 3 * 8 * 256;                      // [16][64] each
  float* sKc = sQ + 1024;
  float* sVc = sKc + 1024;
  float* sS = sVc + 1024;                               // [16][LkPad]
  bf16_t* sCb = reinterpret_cast<bf16_t*>(sS + 16 * LkPad);   // [16][72]
  float* sRk = reinterpret_cast<float*>(sCb + 16 * 72);       // relative positions: [2 max_rel + 1][64] fp32, keys ...
  float* sRv = sRk + (2 * a.max_rel + 1) * 64;                // ... and values (only with tables: dec_attn_lds)
  auto rel_row = [&](int j) {                                 // modules/rpr.py:66-75: clip(i - j) + max_rel
    int dlt = t - j;
    dlt = dlt < -a.max_rel ? -a.max_rel : (dlt > a.max_rel ? a.max_rel : dlt);
    return (dlt + a.max_rel) * 64;
  };
  // keys / values of row r (cross: of its sentence)
  auto kbase = [&](int r) { return a.k + (size_t)(SELF ? row0 + r : (row0 + r) / R) * a.bsk + h * 64; };
  auto vbase = [&](int r) { return a.v + (size_t)(SELF ? row0 + r : (row0 + r) / R) * a.bsv + h * 64; };

  ZK_DT(0);
  // ---- everything that does not depend on values computed here is requested first: one memory round trip
  // (H > 512: only the q weights are requested up front, k / v when their turn comes -- register budget)
  const int ptile = w & 3, khalf = w >> 2, nks = H / 64;
  constexpr bool EARLY = MAXC == 1;
  constexpr int NPR = EARLY ? NP : 1;
  uint4 wq_r[NPR][KS];
  auto load_wq = [&](int p, uint4 (&dst)[KS]) {
    const bf16_t* wp = a.wqt + (size_t)(p * H + h * 64 + ptile * 16 + l15) * a.ldw + khalf * (H / 2) + lk8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (ks < nks) dst[ks] = *reinterpret_cast<const uint4*>(wp + ks * 32);
  };
  uint4 wo_r[NOT][2];
  auto load_wo = [&]() {
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
      const int tile = w + 8 * i;
      if (tile * 16 < H) {
        const bf16_t* wp = a.wot + (size_t)(tile * 16 + l15) * a.ldwo + h * 64 + lk8;
        wo_r[i][0] = *reinterpret_cast<const uint4*>(wp);
        wo_r[i][1] = *reinterpret_cast<const uint4*>(wp + 32);
      }
    }
  };
  // the block input first (it heads the longest dependent chain): the previous sub-layer's residual + LayerNorm in
  // registers (wave w: rows w and w + 8), or the given rows
  uint4 xr[2][MAXC];
  if (a.pro.gamma == nullptr) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < H && w + 8 * n < NR) xr[n][i] = *reinterpret_cast<const uint4*>(a.pro.x + (size_t)(row0 + w + 8 * n) * H + c);
      }
  }
  float bq_r[NP][2];
  // scores: one (row, key) pair per thread (the whole 128-byte key row of the head); context: thread (row, 8 columns,
  // keys cjp, cjp + 4, ..)
  const int cr = tid >> 5, ccg = (tid >> 2) & 7, cjp = tid & 3;
  const int npair = NR * Lk;
  uint4 k_r[8], v_r[8];
  float km_r = 1.f;
  auto pair_key = [&](int e) {       // key row of pair e (clamped: always valid memory)
    const int ec = min(e, npair - 1), r = ec / Lk;
    int j = ec - r * Lk;
    if (SELF) j = min(j, max(t - 1, 0));
    return kbase(r) + (size_t)j * a.ldk;
  };
  auto ctx_val = [&](int j) {
    const int vr = min(cr, NR - 1);
    const int jc = SELF ? min(j, max(t - 1, 0)) : min(j, Lk - 1);
    return vbase(vr) + (size_t)jc * a.ldv + ccg * 8;
  };
  auto prefetch = [&]() {
    ZK_DT(8);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      bq_r[p][0] = a.bq[p * H + h * 64 + (tid & 63)];
      bq_r[p][1] = bq_r[p][0];
    }
    {
      const bf16_t* kp = pair_key(tid);
#pragma unroll
      for (int u = 0; u < 8; ++u) k_r[u] = *reinterpret_cast<const uint4*>(kp + u * 8);
      if (!SELF && a.kmask != nullptr) {
        const int ec = min(tid, npair - 1), r = ec / Lk;
        km_r = a.kmask[(size_t)((row0 + r) / R) * a.ldmask + (ec - r * Lk)];
      }
    }
    if (!SELF) {                       // (self: register budget)
#pragma unroll
      for (int q = 0; q < 8; ++q) v_r[q] = *reinterpret_cast<const uint4*>(ctx_val(cjp + 4 * q));
    }
    if (EARLY) {
#pragma unroll
      for (int p = 0; p < NPR; ++p)
        if (!ZK_DM(0)) load_wq(p, wq_r[p]);
    }
    ZK_DT(9);
  };

  ZK_DT(1);
  // the prologue's own loads are requested first, the prefetches behind them (a wave's loads return in order: the
  // LayerNorm must not wait for the weights), then the LayerNorm is computed while the weights arrive
  if (a.pro.gamma != nullptr) ln_decode_rows<MAXC, 2>(a.pro, row0 + w, 8, row0 + NR, lane, h == 0, xr, prefetch);
  else prefetch();
  if (!EARLY) load_wq(0, wq_r[0]);          // H > 512: after the LayerNorm (register budget)
  ZK_DT(10);
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H && w + 8 * n < NR) *reinterpret_cast<uint4*>(sXb + (w + 8 * n) * XLD + c) = xr[n][i];
    }
  if (rpr) {
    const int n = (2 * a.max_rel + 1) * 64;
    for (int e = tid; e < n; e += 512) { sRk[e] = bf2f(a.rpr_k[e]); sRv[e] = bf2f(a.rpr_v[e]); }
  }
  ZK_DT(11);
  __syncthreads();

  ZK_DT(2);
  // ---- projections of this head on the matrix cores: wave (tile, half of K) -> partial 16 x 16 tiles
  {
    df32x4_t acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = df32x4_t{0.f, 0.f, 0.f, 0.f};
    if (EARLY) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < nks) {
          const uint4 af = *reinterpret_cast<const uint4*>(sXb + l15 * XLD + khalf * (H / 2) + ks * 32 + lk8);
#pragma unroll
          for (int p = 0; p < NP; ++p) acc[p] = dmfma16(af, wq_r[p % NPR][ks], acc[p]);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p > 0) load_wq(p, wq_r[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks < nks) {
            const uint4 af = *reinterpret_cast<const uint4*>(sXb + l15 * XLD + khalf * (H / 2) + ks * 32 + lk8);
            acc[p] = dmfma16(af, wq_r[0][ks], acc[p]);
          }
        }
      }
    }
    // the output-projection fragments are needed last: requested now (the q / k / v fragments are dead), they arrive
    // behind the scores, the softmax and the context
    if (!ZK_DM(1)) load_wo();
    // C rows (lane >> 4) * 4 + i
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) sPart[((p * 8 + w) * 16 + (lane >> 4) * 4 + i) * 16 + l15] = acc[p][i];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int o = tid + 512 * n, r = o >> 6, c = o & 63, tile = c >> 4, cc = c & 15;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float s = sPart[((p * 8 + tile) * 16 + r) * 16 + cc] + sPart[((p * 8 + 4 + tile) * 16 + r) * 16 + cc] +
                      bq_r[p][n];
      const bf16_t sb = f2bf(s);
      if (p == 0) sQ[o] = bf2f(sb);
      else {
        (p == 1 ? sKc : sVc)[o] = bf2f(sb);
        if (r < NR && t < a.Lk) {
          bf16_t* dst = const_cast<bf16_t*>(p == 1 ? a.k : a.v);
          dst[(size_t)(row0 + r) * (p == 1 ? a.bsk : a.bsv) + (size_t)t * (p == 1 ? a.ldk : a.ldv) + h * 64 + c] = sb;
        }
      }
    }
  }
  __syncthreads();

  ZK_DT(3);
  // ---- scores: one (row, key) pair per thread
  for (int e = tid; e < npair; e += 512) {
    const int r = e / Lk, j = e - r * Lk;
    float dot = 0.f;
    if (!SELF || j < t) {
      if (e != tid) {
        const bf16_t* kp = pair_key(e);
#pragma unroll
        for (int u = 0; u < 8; ++u) k_r[u] = *reinterpret_cast<const uint4*>(kp + u * 8);
        if (!SELF && a.kmask != nullptr) km_r = a.kmask[(size_t)((row0 + r) / R) * a.ldmask + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float kf[8];
        unpack8(k_r[u], kf);
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += sQ[r * 64 + u * 8 + i] * kf[i];
      }
    } else {
      for (int i = 0; i < 64; ++i) dot += sQ[r * 64 + i] * sKc[r * 64 + i];
    }
    if (rpr) {
      const float* rk = sRk + rel_row(j);
      for (int i = 0; i < 64; ++i) dot += sQ[r * 64 + i] * rk[i];
    }
    sS[r * LkPad + j] = dot * a.scale + ((!SELF && km_r == 0.f) ? -a.mask_inf : 0.f);
  }
  __syncthreads();

  ZK_DT(4);
  // ---- softmax, wave w: rows w and w + 8; P rounded to bf16 as the tile kernels store it
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r = w + 8 * n;
    if (r < NR) {
      float* s = sS + r * LkPad;
      float mx = -INFINITY;
      for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, s[j]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < Lk; j += 64) { const float e = __expf(s[j] - mx); s[j] = e; sum += e; }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      for (int j = lane; j < Lk; j += 64) s[j] = bf2f(f2bf(s[j] * inv));
    }
  }
  __syncthreads();

  ZK_DT(5);
  // ---- context of this head: thread (row, 8 columns, every 4th key), the four key phases combined inside the quad
  {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int j0 = 0; j0 < Lk; j0 += 32) {
      if (j0 > 0 || SELF) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v_r[q] = *reinterpret_cast<const uint4*>(ctx_val(j0 + cjp + 4 * q));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + cjp + 4 * q;
        if (j < Lk && cr < NR) {
          float vf[8];
          if (!SELF || j < t) unpack8(v_r[q], vf);
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[i] = sVc[cr * 64 + ccg * 8 + i];
          }
          const float pr = sS[cr * LkPad + j];
          if (rpr) {
            const float* rv = sRv + rel_row(j) + ccg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) vf[i] += rv[i];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += pr * vf[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = quad_sum(acc[i]);
    if (cjp == 0) *reinterpret_cast<uint4*>(sCb + cr * 72 + ccg * 8) = pack8(acc);
  }
  __syncthreads();

  ZK_DT(6);
  // ---- the head's share of the output projection: wave w owns the 16-column tiles w, w + 8, ...
  {
    const uint4 a0 = *reinterpret_cast<const uint4*>(sCb + l15 * 72 + lk8);
    const uint4 a1 = *reinterpret_cast<const uint4*>(sCb + l15 * 72 + 32 + lk8);
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
      const int tile = w + 8 * i;
      if (tile * 16 < H) {
        df32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = dmfma16(a0, wo_r[i][0], acc);
        acc = dmfma16(a1, wo_r[i][1], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = (lane >> 4) * 4 + q;
          if (r < NR) a.part[((size_t)h * rows + row0 + r) * H + tile * 16 + l15] = acc[q];
        }
      }
    }
  }
  ZK_DT(7);
}

static size_t dec_attn_lds(int H, int Lk, int max_rel = -1) {
  return 2 * 16 * (size_t)(H + 8) + sizeof(float) * (3 * 8 * 256 + 3 * 1024 + 16 * (size_t)((Lk + 3) & ~3)) + 2 * 16 * 72 +
         (max_rel >= 0 ? sizeof(float) * 2 * (size_t)(2 * max_rel + 1) * 64 : 0);
}

// LDS bytes one workgroup of zk_dec_cross / zk_dec_self needs for (H, Lk): the host asks BEFORE it commits a decode batch
// to the fused path (a shape over the 160 KiB of a CU takes the launch-per-op path instead of failing mid-decode)
// max_rel < 0: no relative-position tables
extern "C" size_t zk_dec_attn_lds(int H, int Lk, int max_rel) { return dec_attn_lds(H, Lk, max_rel); }

template <bool SELF, int MAXC>
static int launch_dec_attn(const DecAttnArgs& a, hipStream_t stream) {
  const size_t lds = dec_attn_lds(a.pro.H, a.Lk, a.rpr_k != nullptr ? a.max_rel : -1);
  ZK_CHECK_ARG(lds <= 160 * 1024, "zk_dec_attn: %zu bytes of LDS needed (H=%d, Lk=%d)", lds, a.pro.H, a.Lk);
  auto kern = k_dec_attn<SELF, MAXC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return zk_set_error((int)e, "zk_dec_attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int groups = (a.B * a.R + a.gr - 1) / a.gr;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * a.nh)), dim3(512), lds, stream, a);
  ZK_LAUNCH_CHECK();
  return 0;
}

// Rows per workgroup.  More rows = fewer re-reads of the head's 128 KB of weights through the L2 -> CU path (32 MB per
// launch at 4 rows, 32 sentences x 8 heads), fewer rows = more CUs sharing the row-wise work (LayerNorm prologue,
// softmax).  zk_dec_group(n) overrides (n = 0: default) -- scripts/dec_attn_trace.py sweeps it.
static int g_dec_group = 0;
extern "C" int zk_dec_group(int n) {
  const int old = g_dec_group;
  if (n >= 0 && n <= 16) g_dec_group = n;
  return old;
}
static int dec_group_rows(int R) {
  if (g_dec_group > 0) return g_dec_group;
  int g = R;                         // one sentence ...
  while (g * 2 <= ZK_DEC_GROUP_DEFAULT) g *= 2;   // ... or as many whole sentences as fit the default
  return g > 16 ? 16 : g;
}

template <bool SELF>
static int dispatch_dec_attn(const DecAttnArgs& a, hipStream_t stream) {
  const int H = a.pro.H;
  return H <= 512 ? launch_dec_attn<SELF, 1>(a, stream)
                  : H == 1024 ? launch_dec_attn<SELF, 2>(a, stream) : launch_dec_attn<SELF, 4>(a, stream);
}

static int check_common(const char* who, const LnDecArgs& p, int B, int R, int nh, int Lk, int ldw, int ldwo, const void* wq,
                        const void* wo, const void* part) {
  ZK_CHECK_ARG(B >= 0 && R >= 1 && nh >= 1 && Lk >= 1, "%s: bad sizes B=%d R=%d nh=%d Lk=%d", who, B, R, nh, Lk);
  ZK_CHECK_ARG(p.H == nh * 64 && p.H >= 128 && p.H <= 2048 && (p.H & (p.H - 1)) == 0,
               "%s: H=%d must be nh * 64 and a power of two in 128 .. 2048", who, p.H);
  ZK_CHECK_ARG(ldw % 8 == 0 && ldwo % 8 == 0 && ldwo >= p.H, "%s: weight strides must be multiples of 8", who);
  ZK_CHECK_ARG(p.x && wq && wo && part, "%s: x, weights and the partial-product buffer are required", who);
  ZK_CHECK_ARG((((uintptr_t)p.x | (uintptr_t)wq | (uintptr_t)wo | (uintptr_t)part) & 15) == 0, "%s: 16-byte alignment", who);
  if (p.gamma != nullptr) {
    ZK_CHECK_ARG(p.beta && p.out && (p.ybuf || p.z || p.parts), "%s: the LayerNorm prologue needs beta, xout and y", who);
    ZK_CHECK_ARG((p.cat_in != nullptr || p.z == nullptr) && (p.cat_in == nullptr || p.z != nullptr || p.parts != nullptr) &&
                 (p.cache == nullptr) == (p.cat_out == nullptr),
                 "%s: the gate needs cat_in and z (or its partial sums); cache / cat_out go together", who);
    ZK_CHECK_ARG(p.parts == nullptr ||
                 (p.z == nullptr && p.nparts >= 1 && p.part_stride >= (long)B * R * p.H * (p.cat_in ? 2 : 1)),
                 "%s: partial sums exclude z and need nparts >= 1, part_stride >= rows * H (2H for the gate)", who);
  }
  return 0;
}

extern "C" {

#ifdef ZK_DEC_TRACE
int zk_dec_trace_set_mode(int mode) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(zk_dec_trace_mode), &mode, sizeof(int));
}
int zk_dec_trace_read(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zk_dec_trace_buf), sizeof(unsigned long long) * 16);
}
#endif

// Encoder-decoder attention sub-layer of a decode step (func.py:206-216 cached mk / mv, :217-287): grid B * nh.
// Prologue arguments as zk_ln_decode (gamma == NULL: none, x is the block input; otherwise xout receives
// LayerNorm(x + y) -- the residual input of the LayerNorm that follows this sub-layer).  k / v: keys and values of the
// B sentences (row j of sentence b at + b*bs + j*ld), kmask [B, ldmask] or NULL.  out_parts fp32 [nh][B*R][H]: feed
// it to zk_ln_decode(parts = out_parts, nparts = nh, part_stride = B*R*H, bias = o_map bias).
int zk_dec_cross(const void* x, void* ybuf, const float* gamma, const float* beta, void* xout, int H, float eps,
                 const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                 float* cache, void* cat_out, float inv_count, const int* time_dev, const void* wqt, int ldwq,
                 const float* bq, const void* k, const void* v, int ldk, int ldv, long bsk, long bsv, const float* kmask,
                 int ldmask, const void* wot, int ldwo, float* out_parts, int B, int R, int nh, int Lk, float scale,
                 float mask_inf, const void* rpr_k, const void* rpr_v, int max_rel, int pos, const int* pos_dev,
                 hipStream_t stream) {
  DecAttnArgs a{};
  a.pro = LnDecArgs{(const bf16_t*)x, (bf16_t*)ybuf, gamma, beta, (bf16_t*)xout, B * R, H, eps, (const bf16_t*)z,
                    (const bf16_t*)cat_in, parts, nparts, part_stride, bias, cache, (bf16_t*)cat_out, inv_count, time_dev};
  if (int rc = check_common("zk_dec_cross", a.pro, B, R, nh, Lk, ldwq, ldwo, wqt, wot, out_parts)) return rc;
  ZK_CHECK_ARG(k && v && bq && ldk % 8 == 0 && ldv % 8 == 0 && bsk % 8 == 0 && bsv % 8 == 0 &&
               (((uintptr_t)k | (uintptr_t)v) & 15) == 0, "zk_dec_cross: keys / values must be 16-byte aligned rows");
  if (B == 0) return 0;
  a.wqt = (const bf16_t*)wqt; a.ldw = ldwq; a.bq = bq;
  a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldk = ldk; a.ldv = ldv; a.bsk = bsk; a.bsv = bsv;
  a.kmask = kmask; a.ldmask = ldmask;
  a.wot = (const bf16_t*)wot; a.ldwo = ldwo; a.part = out_parts;
  a.B = B; a.R = R; a.nh = nh; a.Lk = Lk; a.scale = scale; a.mask_inf = mask_inf;
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr) && (rpr_k == nullptr || (max_rel >= 0 && max_rel <= 31)),
               "zk_dec_cross: relative positions need both tables and 0 <= max_rel <= 31");
  a.rpr_k = (const bf16_t*)rpr_k; a.rpr_v = (const bf16_t*)rpr_v; a.max_rel = rpr_k ? max_rel : 0;
  a.time = pos; a.time_dev = rpr_k ? pos_dev : nullptr;
  a.gr = dec_group_rows(R);
  return dispatch_dec_attn<false>(a, stream);
}

// Self-attention sub-layer of a decode step over per-beam caches (func.py:199-205): wqkv [H, 3H] (q | k | v column
// blocks) with bias [3H]; kcache / vcache bf16 [B*R, Tmax, H]: this step's key / value are written at slot
// time (*time_dev when given) and keys 0 .. time are attended.  Same prologue / out_parts contract as zk_dec_cross.
int zk_dec_self(const void* x, void* ybuf, const float* gamma, const float* beta, void* xout, int H, float eps,
                const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                float* cache, void* cat_out, float inv_count, const int* ln_time_dev, const void* wqkvt, int ldw,
                const float* bqkv, void* kcache, void* vcache, int Tmax, int time, const int* time_dev, const void* wot,
                int ldwo, float* out_parts, int B, int R, int nh, float scale, const void* rpr_k, const void* rpr_v,
                int max_rel, hipStream_t stream) {
  DecAttnArgs a{};
  a.pro = LnDecArgs{(const bf16_t*)x, (bf16_t*)ybuf, gamma, beta, (bf16_t*)xout, B * R, H, eps, (const bf16_t*)z,
                    (const bf16_t*)cat_in, parts, nparts, part_stride, bias, cache, (bf16_t*)cat_out, inv_count,
                    ln_time_dev};
  if (int rc = check_common("zk_dec_self", a.pro, B, R, nh, Tmax, ldw, ldwo, wqkvt, wot, out_parts)) return rc;
  ZK_CHECK_ARG(kcache && vcache && bqkv && (((uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0,
               "zk_dec_self: caches must be 16-byte aligned");
  ZK_CHECK_ARG(time_dev != nullptr || (time >= 0 && time < Tmax), "zk_dec_self: time=%d outside the cache (Tmax=%d)", time,
               Tmax);
  if (B == 0) return 0;
  a.wqt = (const bf16_t*)wqkvt; a.ldw = ldw; a.bq = bqkv;
  a.k = (const bf16_t*)kcache; a.v = (const bf16_t*)vcache; a.ldk = H; a.ldv = H;
  a.bsk = (long)Tmax * H; a.bsv = (long)Tmax * H;
  a.wot = (const bf16_t*)wot; a.ldwo = ldwo; a.part = out_parts;
  a.B = B; a.R = R; a.nh = nh; a.Lk = Tmax; a.scale = scale; a.mask_inf = 0.f;
  a.time_dev = time_dev; a.time = time;
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr) && (rpr_k == nullptr || (max_rel >= 0 && max_rel <= 31)),
               "zk_dec_self: relative positions need both tables and 0 <= max_rel <= 31");
  a.rpr_k = (const bf16_t*)rpr_k; a.rpr_v = (const bf16_t*)rpr_v; a.max_rel = rpr_k ? max_rel : 0;
  a.gr = dec_group_rows(R);
  return dispatch_dec_attn<true>(a, stream);
}

}  // extern "C"
