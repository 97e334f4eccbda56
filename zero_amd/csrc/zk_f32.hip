// zk_f32.hip -- the fp32 decode path (round 5; hp.decode_dtype = "float32").
//
// The reference computes in float32 by default (utils/dtype.py:12-15, `default_dtype`, run.py); the build's product
// path stores activations and GEMM weights as bf16.  Two correct implementations that round at different points cannot
// be token-identical on a beam search over 32000 candidates (measured in round 4: the fp32 and the bf16-storage ORACLES
// part on 3 % of the sentences), so the north star's "token-id exact greedy decode" needs a decode mode that rounds where
// the reference rounds: fp32 master weights, fp32 activations, fp32 accumulation.  These kernels are that mode -- the
// encoder pass and the cached decoder step of models/transformer.py:15-218 / transformer_aan.py:92-260 on plain fp32
// row-major matrices.  They are written for fidelity first (fmaf chains in a fixed order, IEEE division / sqrt, libm
// expf) and for a decode step of 128 rows second; the bf16 kernels stay the throughput path.
//
//   zk_f32_gemm       func.py:14-65 linear (+ bias, + ReLU) and the logits product (transformer.py:182-196): C = A B
//                     or A B^T on v_mfma_f32_32x32x2_f32 (exact fp32: an fmaf chain per output, MI355X_MICROARCH.md),
//                     one wave per 32 x 32 tile straight from global memory, the K range split over the four waves of a
//                     workgroup when the grid would not fill the chip (decode: 128 rows)
//   zk_f32_embed      transformer.py:16-33 / 88-119 embedding x sqrt(H) + bias + timing signal (func.py:341-369)
//   zk_f32_add_ln     func.py:321-324 + 289-303: LN(x + y), biased variance, eps inside the square root
//   zk_f32_attn       func.py:218-256: q pre-scaled, + (1 - mask) x (-inf value), softmax, x V; one wave per (row, head)
//   zk_f32_aan_step   transformer_aan.py:110-112: y = (x + cache) / (t + 1), cache += x
//   zk_f32_gate       transformer_aan.py:186-189: sigmoid(i) x + sigmoid(f) y
//   (the self-attention cache append / reorder of func.py:199-205, search.py:206-209 are byte moves: zk_cache_rows, zk_gather_rows)
#include "zk_common.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ GEMM
// One wave = one 32 x 32 output tile.  MFMA 32x32x2: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31].
// A chunk is 16 consecutive k: lane (i, h) holds A[i][k0 + 8 h .. + 7] (two float4 loads) and the eight B values of the
// same k; instruction j of a chunk multiplies k = k0 + j (h = 0) and k = k0 + 8 + j (h = 1).  Every output is therefore
// ONE fmaf chain over k in the fixed order k0, k0 + 8, k0 + 1, k0 + 9, ... (then, with KSPLIT > 1, the four waves'
// partial sums are added in wave order).
struct F32Chunk {
  float a[8];
  float b[8];
};

template <bool TB>
__device__ __forceinline__ void f32_load_chunk(F32Chunk& c, const float* __restrict__ A, const float* __restrict__ B,
                                               int lda, int ldb, int arow, int bcol, int k0, int kend, int lane) {
  const int h = lane >> 5;
  const int k = k0 + 8 * h;
  const float* ap = A + (size_t)arow * lda + k;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k + 4 * u < kend) v = *reinterpret_cast<const float4*>(ap + 4 * u);      // (K is a multiple of 4)
    c.a[4 * u] = v.x; c.a[4 * u + 1] = v.y; c.a[4 * u + 2] = v.z; c.a[4 * u + 3] = v.w;
  }
  if (TB) {      // B [N, K]: row = output column
    const float* bp = B + (size_t)bcol * ldb + k;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k + 4 * u < kend) v = *reinterpret_cast<const float4*>(bp + 4 * u);
      c.b[4 * u] = v.x; c.b[4 * u + 1] = v.y; c.b[4 * u + 2] = v.z; c.b[4 * u + 3] = v.w;
    }
  } else {       // B [K, N]
#pragma unroll
    for (int j = 0; j < 8; ++j) c.b[j] = (k + j < kend) ? B[(size_t)(k + j) * ldb + bcol] : 0.f;
  }
}

template <bool TB, int KSPLIT>
__global__ void __launch_bounds__(256) k_f32_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                  float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                  const float* __restrict__ bias, int act, int tiles_n) {
  __shared__ float red[KSPLIT > 1 ? 3 * 1024 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tile, kbeg, kend;
  if (KSPLIT > 1) {
    tile = blockIdx.x;
    const int per = ((K + KSPLIT * 16 - 1) / (KSPLIT * 16)) * 16;     // k range of a wave: a multiple of the chunk
    kbeg = wave * per;
    kend = min(K, kbeg + per);
  } else {
    tile = blockIdx.x * 4 + wave;
    kbeg = 0;
    kend = K;
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const bool live = tm * 32 < M;                 // (KSPLIT == 1: the last workgroup may hold tiles past the end)
  const int arow = min(tm * 32 + (lane & 31), M - 1);
  const int bcol = min(tn * 32 + (lane & 31), N - 1);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (live && kbeg < kend) {
    // PF chunks of 16 k in flight per wave (a chunk's loads take ~1-2 us from L2 / HBM, its eight MFMAs 0.2 us: with one
    // chunk of prefetch a decode-step product was a chain of exposed round trips).  Chunks past kend load zeros and add
    // exact zeros; the k order of every output's fmaf chain is unchanged.
    constexpr int PF = 4;
    F32Chunk ch[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) f32_load_chunk<TB>(ch[p], A, B, lda, ldb, arow, bcol, kbeg + 16 * p, kend, lane);
    for (int k0 = kbeg; k0 < kend; k0 += 16 * PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[p].a[j], ch[p].b[j], acc, 0, 0, 0);
        f32_load_chunk<TB>(ch[p], A, B, lda, ldb, arow, bcol, k0 + 16 * (p + PF), kend, lane);
      }
    }
  }
  if (KSPLIT > 1) {
    // partial tiles of waves 1 .. 3 through LDS, added by wave 0 in wave order
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave - 1) * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < KSPLIT - 1; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[w * 1024 + r * 64 + lane];
  }
  if (!live) return;
  const int col = tn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M) {
      float v = acc[r] + bv;
      if (act == 1) v = fmaxf(v, 0.f);
      C[(size_t)row * ldc + col] = v;
    }
  }
}

extern "C" int zk_f32_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int tb,
                           const float* bias, int act, hipStream_t stream) {
  ZK_CHECK_ARG(A != nullptr && B != nullptr && C != nullptr && M >= 0 && N >= 1 && K >= 4 && K % 4 == 0,
               "zk_f32_gemm: M=%d N=%d K=%d (K must be a positive multiple of 4)", M, N, K);
  ZK_CHECK_ARG(lda % 4 == 0 && lda >= K && ldc >= N && (tb ? (ldb % 4 == 0 && ldb >= K) : ldb >= N),
               "zk_f32_gemm: leading dimensions lda=%d ldb=%d ldc=%d", lda, ldb, ldc);
  ZK_CHECK_ARG((((uintptr_t)A | (uintptr_t)(tb ? B : A)) & 15) == 0, "zk_f32_gemm: operands must be 16-byte aligned");
  ZK_CHECK_ARG(act == 0 || act == 1, "zk_f32_gemm: act must be 0 (none) or 1 (ReLU)");
  if (M == 0) return 0;
  const int tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32;
  const long tiles = (long)tiles_m * tiles_n;
  // fewer than two waves per SIMD chip-wide and a K loop worth splitting: the four waves of a workgroup share a tile
  const bool split = tiles < 2048 && K >= 256;
#define ZK_F32_GEMM(TB_, KS_, GRID_)                                                                                      \
  hipLaunchKernelGGL((k_f32_gemm<TB_, KS_>), dim3((unsigned)(GRID_)), dim3(256), 0, stream, A, B, C, M, N, K, lda, ldb, \
                     ldc, bias, act, tiles_n)
  if (split) {
    if (tb) ZK_F32_GEMM(true, 4, tiles); else ZK_F32_GEMM(false, 4, tiles);
  } else {
    if (tb) ZK_F32_GEMM(true, 1, (tiles + 3) / 4); else ZK_F32_GEMM(false, 1, (tiles + 3) / 4);
  }
#undef ZK_F32_GEMM
  ZK_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ embedding
// out[r] = (table[ids[r]] * scale + bias) + timing[pos(r)],  pos(r) = (pos_dev ? *pos_dev : pos0) + r % L;
// all_pad (may be NULL): a device flag (zk_all_equal) -- non-zero: every id of the batch is the pad id and the step's
// input is exact zeros before the timing signal is added (transformer.py:113-115).  One wave per row.
__global__ void __launch_bounds__(256) k_f32_embed(const int* __restrict__ ids, int rows, int L, const float* __restrict__ table,
                                                   const float* __restrict__ bias, const float* __restrict__ timing,
                                                   int timing_rows, float* __restrict__ out, int H, float scale, int pos0,
                                                   const int* __restrict__ pos_dev, const int* __restrict__ all_pad) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int id = ids[r];
  const int pos = min((pos_dev != nullptr ? *pos_dev : pos0) + r % L, timing_rows - 1);
  const bool zero = all_pad != nullptr && *all_pad != 0;
  const float* e = table + (size_t)id * H;
  const float* t = timing + (size_t)pos * H;
  for (int c = lane; c < H; c += 64) {
    float v = 0.f;
    if (!zero) { v = e[c] * scale; v = v + bias[c]; }
    out[(size_t)r * H + c] = v + t[c];
  }
}

extern "C" int zk_f32_embed(const int* ids, int rows, int L, const float* table, const float* bias, const float* timing,
                            int timing_rows, float* out, int H, float scale, int pos0, const int* pos_dev,
                            const int* all_pad, hipStream_t stream) {
  ZK_CHECK_ARG(ids != nullptr && table != nullptr && bias != nullptr && timing != nullptr && out != nullptr && L >= 1 &&
               H >= 1 && timing_rows >= 1, "zk_f32_embed: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_f32_embed, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, rows, L, table, bias, timing, timing_rows,
                     out, H, scale, pos0, pos_dev, all_pad);
  ZK_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ residual + LayerNorm
// out = gamma (s - mean) / sqrt(var + eps) + beta,  s = x + y (y may be NULL), var = mean((s - mean)^2): the two passes of
// func.py:289-303 over a row held in registers.  One wave per row, H <= 64 * ZK_F32_LN_MAXU.
#define ZK_F32_LN_MAXU 32
__global__ void __launch_bounds__(256) k_f32_add_ln(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ out, int rows, int H, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float v[ZK_F32_LN_MAXU];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < ZK_F32_LN_MAXU; ++u) {
    const int c = u * 64 + lane;
    v[u] = 0.f;
    if (c < H) {
      v[u] = x[(size_t)r * H + c];
      if (y != nullptr) v[u] = v[u] + y[(size_t)r * H + c];
      s += v[u];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < ZK_F32_LN_MAXU; ++u) {
    const int c = u * 64 + lane;
    if (c < H) { const float d = v[u] - mean; q += d * d; }
  }
  const float var = wave_sum(q) / (float)H;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int u = 0; u < ZK_F32_LN_MAXU; ++u) {
    const int c = u * 64 + lane;
    if (c < H) out[(size_t)r * H + c] = gamma[c] * (v[u] - mean) * rs + beta[c];
  }
}

extern "C" int zk_f32_add_ln(const float* x, const float* y, const float* gamma, const float* beta, float* out, int rows, int H,
                             float eps, hipStream_t stream) {
  ZK_CHECK_ARG(x != nullptr && gamma != nullptr && beta != nullptr && out != nullptr && H >= 1 && H <= 64 * ZK_F32_LN_MAXU,
               "zk_f32_add_ln: H=%d (at most %d)", H, 64 * ZK_F32_LN_MAXU);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_f32_add_ln, dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, gamma, beta, out, rows, H, eps);
  ZK_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ attention
// One wave per (query row i of sentence b, head h): scores over the keys in lanes (an fmaf chain over the head's d
// channels each, q pre-multiplied by `scale` as func.py:222 does), + (1 - kmask) * (-mask_inf) (func.py:372-387: finite,
// a fully masked row softmaxes to uniform), softmax = exp(s - max) / sum, then lane c accumulates sum_j p_j V[j][c] over
// the keys in order.  Sentence b reads the keys / values / mask of sentence b / kv_group (beam rows share their
// sentence's memory).  nkeys_dev (may be NULL): the number of valid keys is *nkeys_dev + 1 (a decode step's cache holds
// positions 0 .. time), otherwise Lk.  LDS: Lk floats per wave.
__global__ void __launch_bounds__(256) k_f32_attn(const float* __restrict__ q, const float* __restrict__ k,
                                                  const float* __restrict__ v, float* __restrict__ out, int B, int nh, int Lq,
                                                  int Lk, int d, int ldq, int ldk, int ldv, int ldo, long bsq, long bsk,
                                                  long bsv, long bso, const float* __restrict__ kmask, int ldmask,
                                                  int kv_group, float scale, float mask_inf, const int* __restrict__ nkeys_dev,
                                                  const float* __restrict__ rpr_k, const float* __restrict__ rpr_v, int max_rel,
                                                  int q_pos0, const int* __restrict__ q_pos_dev) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * 4 + wave;
  const bool active = idx < (long)B * nh * Lq;   // (every wave takes part in the workgroup barriers)
  const int i = (int)(idx % Lq), h = (int)((idx / Lq) % nh), b = active ? (int)(idx / ((long)Lq * nh)) : 0;
  const int bk = b / kv_group;
  const int nk = nkeys_dev != nullptr ? min(*nkeys_dev + 1, Lk) : Lk;
  const int qpos = (q_pos_dev != nullptr ? *q_pos_dev : q_pos0) + i;      // absolute position of the query (relative positions)
  float* sp = sm + (size_t)wave * (Lk + d);      // [Lk] scores -> probabilities, then [d] the scaled query
  float* sq = sp + Lk;
  if (active) {
    const float* qp = q + (size_t)b * bsq + (size_t)i * ldq + h * d;
    for (int c = lane; c < d; c += 64) sq[c] = qp[c] * scale;
  }
  __syncthreads();
  float mx = -3.0e38f;
  if (active) {
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int j = j0 + lane;
      float s = -3.0e38f;
      if (j < nk) {
        const float* kp = k + (size_t)bk * bsk + (size_t)j * ldk + h * d;
        s = 0.f;
        for (int c = 0; c < d; c += 4) {
          const float4 kv = *reinterpret_cast<const float4*>(kp + c);
          s = fmaf(sq[c], kv.x, s); s = fmaf(sq[c + 1], kv.y, s); s = fmaf(sq[c + 2], kv.z, s); s = fmaf(sq[c + 3], kv.w, s);
        }
        if (rpr_k != nullptr) {       // modules/rpr.py:10-41: logits = q k^T + q r^T, r = table[clip(i - j, -m, m) + m]
          const float* rp = rpr_k + (size_t)(min(max(qpos - j, -max_rel), max_rel) + max_rel) * d;
          float s2 = 0.f;
          for (int c = 0; c < d; c += 4) {
            const float4 rv4 = *reinterpret_cast<const float4*>(rp + c);
            s2 = fmaf(sq[c], rv4.x, s2); s2 = fmaf(sq[c + 1], rv4.y, s2); s2 = fmaf(sq[c + 2], rv4.z, s2); s2 = fmaf(sq[c + 3], rv4.w, s2);
          }
          s = s + s2;
        }
        if (kmask != nullptr) s = s + (1.0f - kmask[(size_t)bk * ldmask + j]) * (-mask_inf);
        sp[j] = s;
      }
      mx = fmaxf(mx, s);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
  if (active) {
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int j = j0 + lane;
      if (j < nk) {                               // (a lane reads back the scores it wrote itself)
        const float p = expf(sp[j] - mx);
        sp[j] = p;
        sum += p;
      }
    }
  }
  sum = wave_sum(sum);
  if (active) {
    for (int j0 = 0; j0 < nk; j0 += 64) {
      const int j = j0 + lane;
      if (j < nk) sp[j] = sp[j] / sum;            // torch.softmax: exp(x - max) / sum, THEN the product with V
    }
  }
  __syncthreads();
  if (!active) return;
  float* op = out + (size_t)b * bso + (size_t)i * ldo + h * d;
  for (int c = lane; c < d; c += 64) {
    const float* vp = v + (size_t)bk * bsv + h * d + c;
    float acc = 0.f;
    for (int j = 0; j < nk; ++j) acc = fmaf(sp[j], vp[(size_t)j * ldv], acc);
    if (rpr_v != nullptr) {         // o = P V + sum_j P_j r_v[clip(i - j) + m]
      float acc2 = 0.f;
      for (int j = 0; j < nk; ++j)
        acc2 = fmaf(sp[j], rpr_v[(size_t)(min(max(qpos - j, -max_rel), max_rel) + max_rel) * d + c], acc2);
      acc = acc + acc2;
    }
    op[c] = acc;
  }
}

extern "C" int zk_f32_attn(const float* q, const float* k, const float* v, float* out, int B, int nh, int Lq, int Lk, int d,
                           int ldq, int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, const float* kmask,
                           int ldmask, int kv_group, float scale, float mask_inf, const int* nkeys_dev, const float* rpr_k,
                           const float* rpr_v, int max_rel, int q_pos0, const int* q_pos_dev, hipStream_t stream) {
  ZK_CHECK_ARG(q != nullptr && k != nullptr && v != nullptr && out != nullptr && nh >= 1 && Lq >= 1 && Lk >= 1 && d >= 4 &&
               d % 4 == 0 && kv_group >= 1, "zk_f32_attn: bad shape (nh=%d Lq=%d Lk=%d d=%d)", nh, Lq, Lk, d);
  ZK_CHECK_ARG(ldk % 4 == 0 && bsk % 4 == 0 && (((uintptr_t)k) & 15) == 0, "zk_f32_attn: keys must be 16-byte aligned rows");
  ZK_CHECK_ARG((rpr_k == nullptr) == (rpr_v == nullptr) && (rpr_k == nullptr || (max_rel >= 0 && ((((uintptr_t)rpr_k) & 15) == 0))),
               "zk_f32_attn: relative positions need both tables (16-byte aligned) and max_rel >= 0");
  const size_t lds = (size_t)4 * (Lk + d) * sizeof(float);
  ZK_CHECK_ARG(lds <= 64 * 1024, "zk_f32_attn: %d keys need %zu bytes of LDS (at most 64 KiB)", Lk, lds);
  if (B <= 0) return 0;
  const long waves = (long)B * nh * Lq;
  hipLaunchKernelGGL(k_f32_attn, dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, stream, q, k, v, out, B, nh, Lq, Lk, d, ldq,
                     ldk, ldv, ldo, bsq, bsk, bsv, bso, kmask, ldmask, kv_group, scale, mask_inf, nkeys_dev, rpr_k, rpr_v, max_rel,
                     q_pos0, q_pos_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ average attention (decode)
// transformer_aan.py:110-112 at decode position t:  y = (x + cache) / (t + 1);  cache = x + cache.
// cat [rows, 2H] = [x | y] (the input of z_project, transformer_aan.py:186).  t = time_dev ? *time_dev : time.
__global__ void __launch_bounds__(256) k_f32_aan_step(const float* __restrict__ x, float* __restrict__ cache,
                                                      float* __restrict__ cat, int rows, int H, int time,
                                                      const int* __restrict__ time_dev) {
  const size_t n = (size_t)rows * H;
  const float div = (float)((time_dev != nullptr ? *time_dev : time) + 1);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / H, c = i % H;
    const float xv = x[i];
    const float s = xv + cache[i];
    cache[i] = s;
    cat[r * 2 * H + c] = xv;
    cat[r * 2 * H + H + c] = s / div;
  }
}

extern "C" int zk_f32_aan_step(const float* x, float* cache, float* cat, int rows, int H, int time, const int* time_dev,
                               hipStream_t stream) {
  ZK_CHECK_ARG(x != nullptr && cache != nullptr && cat != nullptr && H >= 1, "zk_f32_aan_step: bad arguments");
  if (rows <= 0) return 0;
  const size_t n = (size_t)rows * H;
  hipLaunchKernelGGL(k_f32_aan_step, dim3((unsigned)min((size_t)2048, (n + 255) / 256)), dim3(256), 0, stream, x, cache, cat,
                     rows, H, time, time_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}

// transformer_aan.py:186-189: z [rows, 2H] = [i | f];  g = sigmoid(i) x + sigmoid(f) y  with cat = [x | y]
__global__ void __launch_bounds__(256) k_f32_gate(const float* __restrict__ z, const float* __restrict__ cat,
                                                  float* __restrict__ g, int rows, int H) {
  const size_t n = (size_t)rows * H;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / H, c = i % H;
    const float zi = z[r * 2 * H + c], zf = z[r * 2 * H + H + c];
    const float si = 1.0f / (1.0f + expf(-zi)), sf = 1.0f / (1.0f + expf(-zf));
    g[i] = si * cat[r * 2 * H + c] + sf * cat[r * 2 * H + H + c];
  }
}

extern "C" int zk_f32_gate(const float* z, const float* cat, float* g, int rows, int H, hipStream_t stream) {
  ZK_CHECK_ARG(z != nullptr && cat != nullptr && g != nullptr && H >= 1, "zk_f32_gate: bad arguments");
  if (rows <= 0) return 0;
  const size_t n = (size_t)rows * H;
  hipLaunchKernelGGL(k_f32_gate, dim3((unsigned)min((size_t)2048, (n + 255) / 256)), dim3(256), 0, stream, z, cat, g, rows, H);
  ZK_LAUNCH_CHECK();
  return 0;
}

// out[r][0 .. cols) = a[r] + b[r] (row strides in elements): the merged attention's  o + aan_o  (func.py:258-275)
__global__ void __launch_bounds__(256) k_f32_add_rows(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      float* __restrict__ out, int ldo, int rows, int cols) {
  const size_t n = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols, c = i % cols;
    out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
  }
}

extern "C" int zk_f32_add_rows(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int cols,
                               hipStream_t stream) {
  ZK_CHECK_ARG(a != nullptr && b != nullptr && out != nullptr && cols >= 1, "zk_f32_add_rows: bad arguments");
  if (rows <= 0) return 0;
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(k_f32_add_rows, dim3((unsigned)min((size_t)2048, (n + 255) / 256)), dim3(256), 0, stream, a, lda, b, ldb, out,
                     ldo, rows, cols);
  ZK_LAUNCH_CHECK();
  return 0;
}
