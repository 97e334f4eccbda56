// zk_prep.hip -- everything a training step derives from the token ids ALONE, in one launch (round 4).
//
// The reference's step takes its ids through feed_dict and TensorFlow derives the rest on the device: the padding masks
// (func.py:372-387 "masking"), the per-sentence loss weights (transformer.py:198-211) and -- inside the gradient of
// tf.nn.embedding_lookup -- the grouping of the token rows by id (tf.IndexedSlices + unsorted_segment_sum,
// main.py:28).  Rounds 1-3 grouped the rows on the host (numpy argsort / unique in TransformerCore.upload), which is
// invisible when one static batch is replayed and a per-batch cost of two sorts + ~10 small blocking uploads in the
// loop a user runs.  zk_batch_prep does it on the device, as the first node of the captured step:
//
//   block 0          source side: token rows sorted by (id, row)  -> rows / seg / uid / n   (k_embed_bwd_sorted's input)
//   block 1          target side: row (b, t) carries id[b, t-1]; rows with t == 0 have no embedding (transformer.py:99-113)
//   blocks 2 .. 2+B  sentence b: source mask row; target mask row + loss weights  w = loss_scale * m / (len_b * B)
//
// The sort is a bitonic network over 64-bit keys (id << 32 | row): a total order, so the result is THE stable
// grouping numpy's argsort(kind="stable") + unique gives (rows ascending inside an id) -- bit-identical sums in
// k_embed_bwd_sorted.  One workgroup per side; every thread keeps 16 CONSECUTIVE keys in registers, so of the 78
// compare-exchange stages of 4096 keys 42 are register-local, 33 exchange with a lane of the same wave (__shfl_xor, no
// barrier) and only 3 cross waves through LDS.  (The first version ran every stage through LDS behind a 16-wave
// barrier: 42.6 us per batch, 0.9 % of the step; profiles/r04_rocprof_kernel_stats_v0.txt.)  Up to 4096 rows per side:
// 256 threads; up to 16384: 1024 threads; beyond: the plain LDS-free network on a caller-provided global scratch.
#include "zk_common.h"
#include <cstring>
#include <vector>
#include <mutex>
#include <unordered_map>


struct PrepSide {
  const int* ids;      // [B, L]
  int* rows;           // [T]     out: token rows grouped by id
  int* seg;            // [T + 1] out: group boundaries
  int* uid;            // [T]     out: the id of each group
  int* n;              // [1]     out: number of groups
  unsigned long long* scratch;   // [npad] global keys when npad > the LDS capacity of the instantiation
  int L, shift, npad;  // npad: power of two >= B * L
  int rb;              // key = id << rb | row.  rb < 32: the keys fit 32 bits (ids below 2^(31 - rb), the caller's promise:
};                     // max_id of zk_batch_prep) and the network runs on 32-bit keys (min / max in one instruction each,
                       // one ds_bpermute per exchange); rb = 32: 64-bit keys

struct PrepArgs {
  PrepSide side[2];
  float* smask;        // [B, Ls] or null
  float* tmask;        // [B, Lt] or null
  float* tw;           // [B, Lt] or null
  int B;
  float loss_scale;
};

// ---- the network in a caller-provided global scratch (more than 16384 rows per side: rare, slow, correct)
__device__ __forceinline__ void prep_sort_global(const PrepSide& s, int B, int* scan) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int T = B * s.L, N = s.npad;
  unsigned long long* key = s.scratch;
  for (int i = tid; i < N; i += NT) {
    unsigned long long k = ~0ull;
    if (i < T) {
      if (!s.shift) k = ((unsigned long long)(unsigned)s.ids[i] << 32) | (unsigned)i;
      else if (i % s.L != 0) k = ((unsigned long long)(unsigned)s.ids[i - 1] << 32) | (unsigned)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (N >> 1); i += NT) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1)), b = a + j;
        const unsigned long long x = key[a], y = key[b];
        const bool up = (a & k) == 0;
        if ((x > y) == up) { key[a] = y; key[b] = x; }
      }
      __syncthreads();
    }
  }
  // group heads -> uid / seg; every thread owns a contiguous chunk of the sorted order
  const int per = (N + NT - 1) / NT;
  const int p0 = tid * per, p1 = min(p0 + per, N);
  int heads = 0;
  for (int p = p0; p < p1; ++p) {
    const unsigned long long k = key[p];
    if (k == ~0ull) break;
    if (p == 0 || (unsigned)(k >> 32) != (unsigned)(key[p - 1] >> 32)) ++heads;
  }
  scan[tid] = heads;
  __syncthreads();
  for (int off = 1; off < NT; off <<= 1) {       // inclusive Hillis-Steele scan of the per-thread counts
    const int v = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int g = scan[tid] - heads;                              // groups that start before this thread's chunk
  for (int p = p0; p < p1; ++p) {
    const unsigned long long k = key[p];
    if (k == ~0ull) break;
    s.rows[p] = (int)(unsigned)(k & 0xffffffffull);
    if (p == 0 || (unsigned)(k >> 32) != (unsigned)(key[p - 1] >> 32)) {
      s.uid[g] = (int)(unsigned)(k >> 32);
      s.seg[g] = p;
      ++g;
    }
  }
  if (tid == NT - 1) {
    const int n = scan[NT - 1];
    s.seg[n] = s.shift ? B * max(s.L - 1, 0) : T;       // rows that carry an embedding
    s.n[0] = n;
  }
}

// ---- the network on registers: thread t holds keys t*E .. t*E+E-1 of the N = NT*E.  KT = uint32_t or unsigned long long.
template <typename KT> __device__ __forceinline__ KT prep_bperm(int paddr, KT v);
template <> __device__ __forceinline__ uint32_t prep_bperm<uint32_t>(int paddr, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(paddr, (int)v);
}
template <> __device__ __forceinline__ unsigned long long prep_bperm<unsigned long long>(int paddr, unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)(v & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

template <int J, int E, typename KT>
__device__ __forceinline__ void prep_stage_reg(KT (&r)[E], int base, int k) {
#pragma unroll
  for (int a = 0; a < E; ++a) {
    if (a & J) continue;
    const bool up = ((base + a) & k) == 0;
    const KT x = r[a], y = r[a + J];
    const KT lo = x < y ? x : y, hi = x < y ? y : x;
    r[a] = up ? lo : hi;
    r[a + J] = up ? hi : lo;
  }
}

template <int NT, int E, typename KT>
__device__ __forceinline__ void prep_sort_regs(const PrepSide& s, int B, KT* lds) {
  static_assert(E == 16, "the register stages below are written for 16 keys per thread");
  constexpr int N = NT * E;
  constexpr KT NONE = (KT)~(KT)0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = B * s.L, base = tid * E, rb = s.rb;
  KT r[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = base + e;
    KT k = NONE;
    if (i < T) {
      if (!s.shift) k = ((KT)(unsigned)s.ids[i] << rb) | (KT)(unsigned)i;
      else if (i % s.L != 0) k = ((KT)(unsigned)s.ids[i - 1] << rb) | (KT)(unsigned)i;
    }
    r[e] = k;
  }
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
        if (j == 8) prep_stage_reg<8, E, KT>(r, base, k);
        else if (j == 4) prep_stage_reg<4, E, KT>(r, base, k);
        else if (j == 2) prep_stage_reg<2, E, KT>(r, base, k);
        else prep_stage_reg<1, E, KT>(r, base, k);
        continue;
      }
      // the partner thread t ^ d holds the keys E*d away; k > j >= E: all E keys of a thread share their direction
      const int d = j / E;
      const bool keep_min = ((tid & d) == 0) == ((base & k) == 0);
      if (d < 64) {
        const int paddr = (lane ^ d) << 2;       // ds_bpermute byte address of the partner lane
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const KT p = prep_bperm<KT>(paddr, r[e]);
          const KT lo = p < r[e] ? p : r[e], hi = p < r[e] ? r[e] : p;
          r[e] = keep_min ? lo : hi;
        }
      } else {
        __syncthreads();                         // the readers of the previous exchange are done
#pragma unroll
        for (int e = 0; e < E; ++e) lds[e * NT + tid] = r[e];
        __syncthreads();
        const int pt = tid ^ d;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const KT p = lds[e * NT + pt];
          const KT lo = p < r[e] ? p : r[e], hi = p < r[e] ? r[e] : p;
          r[e] = keep_min ? lo : hi;
        }
      }
    }
  }
  // group heads: key p is a head when it is valid and its id differs from key p-1's
  __syncthreads();
  lds[tid] = r[E - 1];
  __syncthreads();
  const KT before = tid > 0 ? lds[tid - 1] : (KT)0;
  const KT rowmask = rb >= 32 ? (KT)0xffffffffu : (KT)(((KT)1 << rb) - 1);
  bool head[E];
  int heads = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const KT prev = e ? r[e - 1] : before;
    head[e] = r[e] != NONE && ((base + e) == 0 || (r[e] >> rb) != (prev >> rb));
    heads += head[e] ? 1 : 0;
  }
  int v = heads;                                  // inclusive scan inside the wave, then across the waves
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int n = __shfl_up(v, off, 64);
    if (lane >= off) v += n;
  }
  int* wsum = reinterpret_cast<int*>(lds + NT);
  __syncthreads();
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  int g = v - heads, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int c = wsum[w];
    if (w < wave) g += c;
    total += c;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (r[e] == NONE) continue;
    s.rows[base + e] = (int)(unsigned)(r[e] & rowmask);
    if (head[e]) {
      s.uid[g] = (int)(unsigned)(r[e] >> rb);
      s.seg[g] = base + e;
      ++g;
    }
  }
  if (tid == 0) {
    s.seg[total] = s.shift ? B * max(s.L - 1, 0) : T;     // rows that carry an embedding
    s.n[0] = total;
  }
}

template <int NT, int E>
__global__ void __launch_bounds__(NT) k_batch_prep(PrepArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NT * E];
  const int blk = blockIdx.x;
  if (blk < 2) {
    const PrepSide& s = a.side[blk];
    if (s.ids == nullptr || s.rows == nullptr) return;
    if (s.npad > NT * E) prep_sort_global(s, a.B, reinterpret_cast<int*>(keys));
    else if (s.rb < 32) prep_sort_regs<NT, E, uint32_t>(s, a.B, reinterpret_cast<uint32_t*>(keys));
    else prep_sort_regs<NT, E, unsigned long long>(s, a.B, keys);
    return;
  }
  // masks and loss weights of sentence b (func.py:372-387; transformer.py:198-211: per-sentence mean, then batch mean)
  const int b = blk - 2, tid = threadIdx.x;
  const int Ls = a.side[0].L, Lt = a.side[1].L;
  if (a.smask != nullptr)
    for (int t = tid; t < Ls; t += NT) a.smask[b * Ls + t] = (a.side[0].ids[b * Ls + t] != 0) ? 1.f : 0.f;
  if (a.side[1].ids == nullptr || (a.tmask == nullptr && a.tw == nullptr)) return;
  float* sm = reinterpret_cast<float*>(keys);
  float cnt = 0.f;
  for (int t = tid; t < Lt; t += NT) cnt += (a.side[1].ids[b * Lt + t] != 0) ? 1.f : 0.f;
  const float len = block_sum<NT / 64>(cnt, sm);
  for (int t = tid; t < Lt; t += NT) {
    const float mk = (a.side[1].ids[b * Lt + t] != 0) ? 1.f : 0.f;
    if (a.tmask != nullptr) a.tmask[b * Lt + t] = mk;
    if (a.tw != nullptr) a.tw[b * Lt + t] = a.loss_scale * mk / (len * (float)a.B);
  }
}

#ifdef ZK_EXPERIMENTS
// =====================================================================================
// zk_ln_fold -- what the step derives from the PARAMETERS alone for the LayerNorm-free forward (GemmEpi, zk_gemm_ln):
// for every linear layer whose input is LN(s) (qkv_map, q_map, ffn enlarge; func.py:289-303 feeding func.py:14-65)
//     LN(s) W + b = rstd (s (gamma o W) - mu colsum(gamma o W)) + (beta W + b)
// so the consumer GEMM reads the un-normalised sum s against  Wf = bf16(gamma_k W_kn)  and finishes in its epilogue
// with  c_n = sum_k Wf_kn  (of the ROUNDED values: it must cancel the mean of exactly what the MFMAs multiplied) and
// d_n = sum_k beta_k W_kn + b_n.  W is the fp32 master.  One launch for all layers at the head of the step; a block =
// 64 columns of one weight, 32 row lanes x 8 column groups, K / 32 rows per lane.  HBM: 6 B per weight element.
// =====================================================================================
struct FoldDesc {
  const float* W; const float* gamma; const float* beta; const float* b;
  bf16_t* Wf; float* c; float* d;
  int K, N, block_start, pad;
};

__global__ void __launch_bounds__(256) k_ln_fold(const FoldDesc* __restrict__ descs, int nprob) {
  __shared__ float red[2][32][64 + 1];
  const int bid = blockIdx.x;
  int p = 0;
  {
    int hi = nprob - 1;
    while (p < hi) {
      const int mid = (p + hi + 1) >> 1;
      if (descs[mid].block_start <= bid) p = mid; else hi = mid - 1;
    }
  }
  const FoldDesc d = descs[p];
  const int n0 = (bid - d.block_start) * 64;
  const int tid = threadIdx.x, cg = tid & 7, rl = tid >> 3;
  const int col = n0 + cg * 8;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ds[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = rl; k < d.K; k += 32) {
    const float4 a = *reinterpret_cast<const float4*>(d.W + (size_t)k * d.N + col);
    const float4 b = *reinterpret_cast<const float4*>(d.W + (size_t)k * d.N + col + 4);
    const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const float g = d.gamma[k], bt = d.beta[k];
    float wf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = g * w[j];
    const uint4 pk = pack8(wf);
    *reinterpret_cast<uint4*>(d.Wf + (size_t)k * d.N + col) = pk;
    unpack8(pk, wf);
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] += wf[j]; ds[j] += bt * w[j]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][rl][cg * 8 + j] = cs[j]; red[1][rl][cg * 8 + j] = ds[j]; }
  __syncthreads();
  if (tid < 128) {
    const int q = tid >> 6, n = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[q][i][n];
    if (q == 0) d.c[n0 + n] = t;
    else d.d[n0 + n] = t + (d.b != nullptr ? d.b[n0 + n] : 0.f);
  }
}

#endif  // ZK_EXPERIMENTS

// =====================================================================================
// zk_copy_many -- up to 16 small device-to-device copies in ONE launch: the id-dependent arrays of the NEXT batch, which
// a side stream uploaded and prepared (zk_batch_prep) into staging buffers while the previous step was still running,
// move into the static buffers the captured step reads (zero_amd/main.py Trainer.step).  4-byte granularity.
// =====================================================================================
#define ZK_COPY_MAX 16
struct CopyMany {
  uint32_t* dst[ZK_COPY_MAX];
  const uint32_t* src[ZK_COPY_MAX];
  unsigned words[ZK_COPY_MAX];
};
__global__ void __launch_bounds__(256) k_copy_many(CopyMany c) {
  const int p = blockIdx.y;
  const unsigned n = c.words[p];
  uint32_t* __restrict__ d = c.dst[p];
  const uint32_t* __restrict__ s = c.src[p];
  const unsigned n4 = ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) ? (n >> 2) : 0;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
    reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  for (unsigned i = n4 * 4 + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

static int prep_npad(long T) {
  long n = 2;
  while (n < T) n <<= 1;
  return (int)n;
}

extern "C" {
// bytes of global scratch zk_batch_prep needs for a side of `rows` token rows (0 while the keys fit in LDS)
size_t zk_batch_prep_workspace(int rows) {
  const int n = prep_npad(rows);
  return n > 16384 ? (size_t)n * 8 : 0;
}

// ids int32 [B, Ls] / [B, Lt] (target side optional: tgt_ids NULL -> only the source mask is made).
// *_rows [T], *_seg [T + 1], *_uid [T], *_n [1] int32 outputs (NULL rows pointer: that side is not sorted);
// smask [B, Ls], tmask / tw [B, Lt] fp32 outputs, each optional.  scratch: >= zk_batch_prep_workspace(B*Ls) +
// zk_batch_prep_workspace(B*Lt) bytes (may be NULL when both are 0).
int zk_batch_prep(const int* src_ids, const int* tgt_ids, int B, int Ls, int Lt, int* src_rows, int* src_seg,
                  int* src_uid, int* src_n, int* tgt_rows, int* tgt_seg, int* tgt_uid, int* tgt_n, float* smask,
                  float* tmask, float* tw, float loss_scale, int max_id, void* scratch, size_t scratch_bytes,
                  hipStream_t stream) {
  ZK_CHECK_ARG(B >= 0 && Ls >= 0 && Lt >= 0, "zk_batch_prep: bad dims B=%d Ls=%d Lt=%d", B, Ls, Lt);
  ZK_CHECK_ARG((long)B * Ls < (1l << 30) && (long)B * Lt < (1l << 30), "zk_batch_prep: too many token rows");
  if (B == 0) return 0;
  ZK_CHECK_ARG(src_ids != nullptr, "zk_batch_prep: source ids are required");
  PrepArgs a;
  a.B = B; a.loss_scale = loss_scale; a.smask = smask; a.tmask = tmask; a.tw = tw;
  const size_t need_s = src_rows ? zk_batch_prep_workspace(B * Ls) : 0;
  const size_t need_t = (tgt_ids && tgt_rows) ? zk_batch_prep_workspace(B * Lt) : 0;
  ZK_CHECK_ARG(need_s + need_t == 0 || (scratch != nullptr && scratch_bytes >= need_s + need_t),
               "zk_batch_prep: scratch too small (%zu < %zu)", scratch_bytes, need_s + need_t);
  // 32-bit keys when every id (< max_id, the caller's promise; 0 = unknown) and every row index fit 31 bits together
  auto row_bits = [&](long T) {
    int rb = 1;
    while ((1l << rb) < T) ++rb;
    int tb = 1;
    while (max_id > 0 && (1l << tb) < (long)max_id) ++tb;
    return (max_id > 0 && rb + tb <= 31) ? rb : 32;
  };
  a.side[0] = PrepSide{src_ids, src_rows, src_seg, src_uid, src_n, (unsigned long long*)scratch, Ls, 0,
                       prep_npad((long)B * Ls), row_bits((long)B * Ls)};
  a.side[1] = PrepSide{tgt_ids, tgt_ids ? tgt_rows : nullptr, tgt_seg, tgt_uid, tgt_n,
                       (unsigned long long*)((char*)scratch + need_s), Lt, 1, prep_npad((long)B * Lt),
                       row_bits((long)B * Lt)};
  ZK_CHECK_ARG(src_rows == nullptr || (src_seg && src_uid && src_n), "zk_batch_prep: source outputs incomplete");
  ZK_CHECK_ARG(a.side[1].rows == nullptr || (tgt_seg && tgt_uid && tgt_n), "zk_batch_prep: target outputs incomplete");
  int big = 0;
  if (src_rows) big = a.side[0].npad;
  if (a.side[1].rows && a.side[1].npad > big) big = a.side[1].npad;
  const dim3 grid(2 + B);
  if (big <= 4096) hipLaunchKernelGGL((k_batch_prep<256, 16>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((k_batch_prep<1024, 16>), grid, dim3(1024), 0, stream, a);
  ZK_LAUNCH_CHECK();
  return 0;
}

// dsts / srcs / nbytes: HOST arrays of n (<= 16) device pointers and byte counts (multiples of 4, 4-byte aligned)
static int fill_copy_many(CopyMany* c, unsigned* gx_out, void* const* dsts, const void* const* srcs, const size_t* nbytes, int n) {
  size_t big = 0;
  for (int i = 0; i < ZK_COPY_MAX; ++i) {
    const bool on = i < n;
    ZK_CHECK_ARG(!on || (nbytes[i] % 4 == 0 && nbytes[i] < (1ull << 33) && (nbytes[i] == 0 || (dsts[i] != nullptr && srcs[i] != nullptr)) &&
                         ((((uintptr_t)dsts[i]) | ((uintptr_t)srcs[i])) & 3) == 0),
                 "zk_copy_many: copy %d must be 4-byte aligned, a multiple of 4 bytes and non-null", i);
    c->dst[i] = on ? (uint32_t*)dsts[i] : nullptr;
    c->src[i] = on ? (const uint32_t*)srcs[i] : nullptr;
    c->words[i] = on ? (unsigned)(nbytes[i] / 4) : 0;
    if (on && nbytes[i] > big) big = nbytes[i];
  }
  unsigned gx = (unsigned)((big / 16 + 255) / 256);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  *gx_out = gx;
  return 0;
}

int zk_copy_many(void* const* dsts, const void* const* srcs, const size_t* nbytes, int n, hipStream_t stream) {
  ZK_CHECK_ARG(n >= 0 && n <= ZK_COPY_MAX, "zk_copy_many: n=%d out of range (<= %d)", n, ZK_COPY_MAX);
  if (n == 0) return 0;
  CopyMany c;
  unsigned gx = 1;
  if (int rc = fill_copy_many(&c, &gx, dsts, srcs, nbytes, n)) return rc;
  hipLaunchKernelGGL(k_copy_many, dim3(gx, n), dim3(256), 0, stream, c);
  ZK_LAUNCH_CHECK();
  return 0;
}

// The captured training step starts with its zk_copy_many launch (the prepared batch moves from a staging set into the
// buffers the step reads).  Which staging set that is changes from step to step: instead of a launch of its own in front of
// every replay (a second submission per step: ~20 us of idle time at the step boundary), the parameters of that
// ONE node are rewritten in the instantiated graph.  exec: a handle of zk_graph_end whose capture holds exactly one
// zk_copy_many launch; the arguments as for zk_copy_many (n >= 1).
}  // extern "C"
static std::mutex g_copy_node_mu;
static std::unordered_map<void*, hipGraphNode_t> g_copy_node_of;      // executable -> its one zk_copy_many node
void zk_graph_forget_nodes(void* exec) {                               // (zk_common.h; called by zk_graph_destroy)
  std::lock_guard<std::mutex> lk(g_copy_node_mu);
  g_copy_node_of.erase(exec);
}
// how many zk_copy_many launches a capture of n copies makes (1 while n <= ZK_COPY_MAX): the host asks BEFORE it chooses the
// in-graph commit, whose graph must hold exactly one such node
extern "C" int zk_copy_many_max(void) { return ZK_COPY_MAX; }
extern "C" {
int zk_graph_set_copy_many(void* exec, void* const* dsts, const void* const* srcs, const size_t* nbytes, int n) {
  ZK_CHECK_ARG(exec != nullptr && n >= 1 && n <= ZK_COPY_MAX, "zk_graph_set_copy_many: null graph or n=%d out of range", n);
  // the node is looked up ONCE per executable (this call is on the per-step path: walking the several hundred nodes of the
  // step's graph every step cost host time of the order of what the in-graph commit saves)
  hipGraphNode_t node = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_copy_node_mu);
    auto it = g_copy_node_of.find(exec);
    if (it != g_copy_node_of.end()) node = it->second;
  }
  if (node == nullptr) {
    hipGraph_t graph = zk_graph_template_of(exec);
    ZK_CHECK_ARG(graph != nullptr, "zk_graph_set_copy_many: not a handle of zk_graph_end");
    size_t nn = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &nn);
    if (e != hipSuccess) return zk_set_error((int)e, "hipGraphGetNodes: %s", hipGetErrorString(e));
    std::vector<hipGraphNode_t> nodes(nn);
    e = hipGraphGetNodes(graph, nodes.data(), &nn);
    if (e != hipSuccess) return zk_set_error((int)e, "hipGraphGetNodes: %s", hipGetErrorString(e));
    int found = 0;
    for (size_t i = 0; i < nn; ++i) {
      hipGraphNodeType t;
      if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess || t != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams p;
      if (hipGraphKernelNodeGetParams(nodes[i], &p) != hipSuccess) continue;
      if (p.func == (void*)k_copy_many) { node = nodes[i]; ++found; }
    }
    ZK_CHECK_ARG(found == 1, "zk_graph_set_copy_many: the graph holds %d zk_copy_many launches (need exactly one)", found);
    std::lock_guard<std::mutex> lk(g_copy_node_mu);
    g_copy_node_of[exec] = node;
  }
  hipError_t e;
  CopyMany c;
  unsigned gx = 1;
  if (int rc = fill_copy_many(&c, &gx, dsts, srcs, nbytes, n)) return rc;
  void* args[1] = {&c};
  hipKernelNodeParams np;
  memset(&np, 0, sizeof(np));
  np.func = (void*)k_copy_many;
  np.gridDim = dim3(gx, n);
  np.blockDim = dim3(256);
  np.sharedMemBytes = 0;
  np.kernelParams = args;
  np.extra = nullptr;
  e = hipGraphExecKernelNodeSetParams((hipGraphExec_t)exec, node, &np);
  if (e != hipSuccess) return zk_set_error((int)e, "hipGraphExecKernelNodeSetParams: %s", hipGetErrorString(e));
  return 0;
}

#ifdef ZK_EXPERIMENTS   // the LayerNorm-free forward: measured, no gain (profiles/r04_negative_results.txt)
// descs: DEVICE array of nprob FoldDesc (64 bytes: W, gamma, beta, b, Wf, c, d pointers; K, N, block_start = running sum
// of N / 64, pad); total_blocks = that sum.  N % 64 == 0, 16-byte aligned rows.
int zk_ln_fold(const void* descs, int nprob, int total_blocks, hipStream_t stream) {
  if (nprob == 0 || total_blocks == 0) return 0;
  ZK_CHECK_ARG(descs != nullptr && nprob > 0 && total_blocks > 0, "zk_ln_fold: bad arguments");
  hipLaunchKernelGGL(k_ln_fold, dim3(total_blocks), dim3(256), 0, stream, (const FoldDesc*)descs, nprob);
  ZK_LAUNCH_CHECK();
  return 0;
}
#endif  // ZK_EXPERIMENTS
}  // extern "C"
