// zk_decode.hip -- beam-search decode step tail and cache plumbing (gfx950).
//
// search.py:143-176: logits / temperature -> log-softmax -> (step 0: forbid EOS by -1e8)
// -> + previous beam log-prob -> / length penalty -> top-2K over the K*V candidates of each
// sentence (ties -> lower flat index, the tf.nn.top_k contract).  One fused kernel: the
// [B*K, V] log-prob tensor and the [B, K*V] score tensor are never materialised.
// search.py:198-210: beam reordering of the per-beam caches = row gather.
// transformer_aan.py:110-112: decode-time cumulative average, cache in HBM (fp32).
#include "zk_common.h"
#include "zk_lndec_dev.h"
#include <alloca.h>

#define TOPK_MAX 16

__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
}

// block-wide selection of the k2 best (score, index) pairs out of per-thread sorted lists in LDS
// (ls/li: [256][TOPK_MAX+1], cnt entries each).  Every round: arg-max over the list heads.
__device__ __forceinline__ void block_select(float (*ls)[TOPK_MAX + 1], int (*li)[TOPK_MAX + 1], int cnt, int k2,
                                             float* out_s, int* out_i, float* ws, int* wi, int* wt) {
  const int tid = threadIdx.x;
  int head = 0;
  for (int r = 0; r < k2; ++r) {
    float s = (head < cnt) ? ls[tid][head] : -INFINITY;
    int i = (head < cnt) ? li[tid][head] : 0x7fffffff;
    int t = tid;
    wave_argbest(s, i, t, [](float s2, int i2, float s1, int i1) { return better(s2, i2, s1, i1); });
    __syncthreads();
    if ((tid & 63) == 0) { ws[tid >> 6] = s; wi[tid >> 6] = i; wt[tid >> 6] = t; }
    __syncthreads();
    float bs = ws[0]; int bi = wi[0], bt = wt[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (better(ws[w], wi[w], bs, bi)) { bs = ws[w]; bi = wi[w]; bt = wt[w]; }
    if (tid == bt) ++head;
    if (tid == 0) { out_s[r] = bs; out_i[r] = bi; }
  }
}

// stage 1: one block per (sentence, beam) ROW: log-sum-exp of the row, candidate scores
// (prev + log-prob) / penalty, the row's own top-k2 -> cand_s / cand_i [B*K, k2] (flat index k*V+v).
// penalty / forbid_id may come from device memory (scal_dev: {penalty as float bits, forbid_id})
// so that a captured decode-step graph picks up the per-step values.
__global__ void __launch_bounds__(256) k_beam_topk_rows(const float* __restrict__ logits, const float* __restrict__ prev_lp,
                                                        float* __restrict__ cand_s, int* __restrict__ cand_i, int K,
                                                        int V, int ld, int k2, float inv_temp, float penalty,
                                                        int forbid_id, float forbid_value,
                                                        const int* __restrict__ scal_dev) {
  __shared__ float sm[8];
  __shared__ float ls[256][TOPK_MAX + 1];
  __shared__ int li[256][TOPK_MAX + 1];
  __shared__ float ws[4];
  __shared__ int wi[4];
  __shared__ int wt[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int k = row % K;
  if (scal_dev != nullptr) { penalty = __int_as_float(scal_dev[0]); forbid_id = scal_dev[1]; }
  const float* z = logits + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, z[c] * inv_temp);
  m = block_max<4>(m, sm);
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += __expf(z[c] * inv_temp - m);
  s = block_sum<4>(s, sm);
  const float lse = m + __logf(s);
  const float prev = prev_lp[row];
  int cnt = 0;
  for (int t = 0; t < k2; ++t) { ls[tid][t] = -INFINITY; li[tid][t] = 0x7fffffff; }
  for (int v = tid; v < V; v += 256) {
    float lp = z[v] * inv_temp - lse;
    if (v == forbid_id) lp += -forbid_value;
    const float sc = (prev + lp) / penalty;
    if (cnt < k2 || sc > ls[tid][k2 - 1]) {
      int pos = cnt < k2 ? cnt : k2 - 1;
      while (pos > 0 && sc > ls[tid][pos - 1]) {   // strict: equal scores keep index order
        ls[tid][pos] = ls[tid][pos - 1];
        li[tid][pos] = li[tid][pos - 1];
        --pos;
      }
      ls[tid][pos] = sc;
      li[tid][pos] = k * V + v;
      if (cnt < k2) ++cnt;
    }
  }
  block_select(ls, li, cnt, k2, cand_s + (size_t)row * k2, cand_i + (size_t)row * k2, ws, wi, wt);
}

// ---- chunked form (default): every (sentence, beam) row is split into `nchunks` column chunks, one block
// each, so that B*K*nchunks blocks read the logits with many loads in flight (the one-block-per-row kernel
// above keeps ONE dependent load per thread outstanding: 149 us for 128 rows x 32000 at 4 waves per CU).
// A chunk keeps its keys z/T (minus the EOS ban) in LDS, reduces {max, sum exp} and picks its k2 best by k2
// block-wide arg-max rounds; within a row the order of the keys is the order of the final scores
// ((prev + key - lse) / penalty is increasing in key), so nothing is lost before the merge.
#define TOPK_CHUNK_MAX 8192
// NV float4 of keys per thread, register-resident: element e of the chunk belongs to thread (e / 4) % 256, slot
// (e / 1024) * 4 + e % 4, so a thread's slots are in increasing element order (ties must keep the lower index).
// Timeline of the LDS version (s_memtime stamps, 4000-key chunks): 10 900 cycles of 4-byte loads, 8 arg-max rounds
// of 3 600 cycles each -- dominated by the owner's 16 dependent LDS reads when it rescans its keys.
template <int NV>
__global__ void __launch_bounds__(256) k_beam_topk_chunks(const float* __restrict__ logits, float* __restrict__ part_ms,
                                                          float* __restrict__ cand_key, int* __restrict__ cand_v,
                                                          int V, int ld, int k2, int chunk, int nchunks,
                                                          float inv_temp, int forbid_id, float forbid_value,
                                                          const int* __restrict__ scal_dev) {
  __shared__ float sm[8];
  __shared__ float ws[8];
  __shared__ int wi[8];
  const int row = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  if (scal_dev != nullptr) forbid_id = scal_dev[1];
  const int v0 = c * chunk, n = max(0, min(chunk, V - v0));
  const float* z = logits + (size_t)row * ld + v0;
  const bool vec = (((uintptr_t)z) & 15) == 0;
  float key[NV * 4];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int e = 4 * (tid + 256 * u);
    if (vec && e + 3 < n) {
      const float4 x = *reinterpret_cast<const float4*>(z + e);
      key[u * 4 + 0] = x.x * inv_temp; key[u * 4 + 1] = x.y * inv_temp;
      key[u * 4 + 2] = x.z * inv_temp; key[u * 4 + 3] = x.w * inv_temp;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) key[u * 4 + q] = (e + q < n) ? z[e + q] * inv_temp : -INFINITY;
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV * 4; ++i) m = fmaxf(m, key[i]);        // the EOS ban applies after the log-softmax (search.py:148-155)
  m = block_max<4>(m, sm);
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < NV; ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * (tid + 256 * u) + q < n) s += __expf(key[u * 4 + q] - m);
  s = block_sum<4>(s, sm);
  if (tid == 0) {
    part_ms[((size_t)row * nchunks + c) * 2] = m;
    part_ms[((size_t)row * nchunks + c) * 2 + 1] = s;
  }
  if (forbid_id >= v0 && forbid_id < v0 + n) {
    const int e = forbid_id - v0;
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * (tid + 256 * u) + q == e) key[u * 4 + q] -= forbid_value;
  }
  // k2 arg-max rounds over the per-thread bests; only the winner's owner rescans (registers)
  float my_bs;
  int my_bi;
  auto rescan = [&]() {
    my_bs = -INFINITY; my_bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (key[u * 4 + q] > my_bs) { my_bs = key[u * 4 + q]; my_bi = 4 * (tid + 256 * u) + q; }   // strict: ties keep the lower index
  };
  rescan();
  for (int r = 0; r < k2; ++r) {
    float bs = my_bs;
    int bi = my_bi;
    {
      int unused = 0;
      wave_argbest(bs, bi, unused, [](float s2, int i2, float s1, int i1) { return better(s2, i2, s1, i1); });
    }
    float* wsr = ws + (r & 1) * 4;                    // double-buffered: one barrier per round
    int* wir = wi + (r & 1) * 4;
    if ((tid & 63) == 0) { wsr[tid >> 6] = bs; wir[tid >> 6] = bi; }
    __syncthreads();
    bs = wsr[0]; bi = wir[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (better(wsr[w], wir[w], bs, bi)) { bs = wsr[w]; bi = wir[w]; }
    if (tid == 0) {
      const size_t o = ((size_t)row * nchunks + c) * k2 + r;
      cand_key[o] = bs;
      cand_v[o] = bi < n ? v0 + bi : 0x7fffffff;
    }
    if (bi < n && ((bi >> 2) & 255) == tid) {
#pragma unroll
      for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * (tid + 256 * u) + q == bi) key[u * 4 + q] = -INFINITY;
      rescan();
    }
  }
}

// merge of the chunked form: one block per sentence; thread t owns candidate t of the K*nchunks*k2 (<= 256)
__device__ __forceinline__ void beam_merge_chunks_block(const float* __restrict__ part_ms, const float* __restrict__ cand_key,
                                                        const int* __restrict__ cand_v, const float* __restrict__ prev_lp,
                                                        float* __restrict__ out_s, int* __restrict__ out_i, int K, int V,
                                                        int k2, int nchunks, float penalty,
                                                        const int* __restrict__ scal_dev) {
  __shared__ float ls[256][TOPK_MAX + 1];
  __shared__ int li[256][TOPK_MAX + 1];
  __shared__ float ws[4];
  __shared__ int wi[4];
  __shared__ int wt[4];
  __shared__ float lse[TOPK_MAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (scal_dev != nullptr) penalty = __int_as_float(scal_dev[0]);
  // every global load of the block is requested up front (nchunks <= 8: a rolled loop over the chunk partials is one
  // dependent round trip per chunk)
  const int per_row = nchunks * k2, n = K * per_row;
  const int ck = min(tid, n - 1) / per_row;
  const size_t co = (size_t)(b * K + ck) * per_row + (min(tid, n - 1) - ck * per_row);
  const int cv_r = cand_v[co];
  const float ckey_r = cand_key[co], cprev_r = prev_lp[b * K + ck];
  if (tid < K) {
    const float2* pm = reinterpret_cast<const float2*>(part_ms + ((size_t)(b * K + tid) * nchunks) * 2);
    float2 pv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) pv[c] = pm[min(c, nchunks - 1)];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nchunks) m = fmaxf(m, pv[c].x);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nchunks && pv[c].x > -INFINITY) s += pv[c].y * __expf(pv[c].x - m);
    lse[tid] = m + __logf(s);
  }
  __syncthreads();
  int cnt = 0;
  if (tid < n) {
    const int k = ck;
    const int v = cv_r;
    if (v != 0x7fffffff) {
      ls[tid][0] = (cprev_r + (ckey_r - lse[k])) / penalty;
      li[tid][0] = k * V + v;
      cnt = 1;
    }
  }
  block_select(ls, li, cnt, k2, out_s + (size_t)b * k2, out_i + (size_t)b * k2, ws, wi, wt);
}
__global__ void __launch_bounds__(256) k_beam_topk_merge_chunks(const float* __restrict__ part_ms,
                                                                const float* __restrict__ cand_key,
                                                                const int* __restrict__ cand_v,
                                                                const float* __restrict__ prev_lp,
                                                                float* __restrict__ out_s, int* __restrict__ out_i,
                                                                int K, int V, int k2, int nchunks, float penalty,
                                                                const int* __restrict__ scal_dev) {
  beam_merge_chunks_block(part_ms, cand_key, cand_v, prev_lp, out_s, out_i, K, V, k2, nchunks, penalty, scal_dev);
}

// stage 2: one block per sentence merges its K*k2 candidates (ties -> lower flat index)
__global__ void __launch_bounds__(256) k_beam_topk_merge(const float* __restrict__ cand_s, const int* __restrict__ cand_i,
                                                         float* __restrict__ out_s, int* __restrict__ out_i, int K,
                                                         int k2) {
  __shared__ float ls[256][TOPK_MAX + 1];
  __shared__ int li[256][TOPK_MAX + 1];
  __shared__ float ws[4];
  __shared__ int wi[4];
  __shared__ int wt[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = K * k2;                    // <= 256
  int cnt = 0;
  if (tid < n) {
    ls[tid][0] = cand_s[(size_t)b * n + tid];
    li[tid][0] = cand_i[(size_t)b * n + tid];
    cnt = 1;
  }
  block_select(ls, li, cnt, k2, out_s + (size_t)b * k2, out_i + (size_t)b * k2, ws, wi, wt);
}

__global__ void __launch_bounds__(256) k_gather_rows(const uint4* __restrict__ src, size_t src_stride16,
                                                     const int* __restrict__ index, uint4* __restrict__ dst,
                                                     size_t dst_stride16, size_t row16, int period) {
  const int r = blockIdx.x;
  // period > 0: `rows / period` stacked tables share one index of length `period` (all layers' caches)
  const size_t s = index ? (period > 0 ? (size_t)(r / period) * period + index[r % period] : (size_t)index[r])
                         : (size_t)r;
  for (size_t c = (size_t)blockIdx.y * 256 + threadIdx.x; c < row16; c += (size_t)gridDim.y * 256)
    dst[(size_t)r * dst_stride16 + c] = src[s * src_stride16 + c];
}

// search.py:143-145 + util.py:189-195: logits += -log(-log(u + eps) + eps), u ~ U[0,1) from the
// counter-based generator (TF's stream cannot be matched; the distribution is)
__global__ void __launch_bounds__(256) k_add_gumbel(float* __restrict__ logits, int rows, int V, int ld, float eps,
                                                    const uint64_t* __restrict__ seedp, uint32_t sid) {
  const uint64_t seed = *seedp;
  const size_t n = (size_t)rows * V;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / V, c = i % V;
    const float u = (float)zk_rand_u32(seed, sid, i) * 2.3283064365386963e-10f;   // * 2^-32
    logits[r * ld + c] += -__logf(-__logf(u + eps) + eps);
  }
}

// k/v cache maintenance with the step counter read on the device (see zk_cache_rows)
__global__ void __launch_bounds__(256) k_cache_rows(const uint4* __restrict__ src, size_t src_stride16,
                                                    const int* __restrict__ index, uint4* __restrict__ dst,
                                                    size_t dst_stride16, size_t unit16,
                                                    const int* __restrict__ time_dev, int mode, int period) {
  const int r = blockIdx.x;
  const size_t t = (size_t)*time_dev;
  if (mode == 0) {
    for (size_t c = (size_t)blockIdx.y * 256 + threadIdx.x; c < unit16; c += (size_t)gridDim.y * 256)
      dst[(size_t)r * dst_stride16 + t * unit16 + c] = src[(size_t)r * src_stride16 + c];
  } else {
    const size_t s = index ? (period > 0 ? (size_t)(r / period) * period + index[r % period] : (size_t)index[r])
                           : (size_t)r;
    const size_t n16 = t * unit16;
    for (size_t c = (size_t)blockIdx.y * 256 + threadIdx.x; c < n16; c += (size_t)gridDim.y * 256)
      dst[(size_t)r * dst_stride16 + c] = src[s * src_stride16 + c];
  }
}

// cache += x (fp32 running sum); cat = [x | cache * inv_count]
__global__ void __launch_bounds__(256) k_aan_decode(const bf16_t* __restrict__ x, float* __restrict__ cache,
                                                    bf16_t* __restrict__ cat, int rows, int H, float inv_count,
                                                    const int* __restrict__ time_dev) {
  if (time_dev != nullptr) inv_count = 1.f / (float)(*time_dev + 1);
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nc = H / 8;
  if (idx >= (size_t)rows * nc) return;
  const size_t r = idx / nc;
  const int c = (int)(idx % nc) * 8;
  const uint4 xv = *reinterpret_cast<const uint4*>(x + r * H + c);
  float v[8], o[8];
  unpack8(xv, v);
  float* cp = cache + r * H + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float s = cp[j] + v[j]; cp[j] = s; o[j] = s * inv_count; }
  *reinterpret_cast<uint4*>(cat + r * 2 * H + c) = xv;
  *reinterpret_cast<uint4*>(cat + r * 2 * H + H + c) = pack8(o);
}

// Input of the decoder at one decode position in ONE launch (was zk_all_equal + zk_embed_fwd + zk_aan_decode):
// transformer.py:88-119 -- every fed id == pad (the first step) -> zero embedding, else table[id] * scale + bias; plus the
// timing signal of the position; cache != NULL: the first layer's average-attention update from the row just written
// (transformer_aan.py:110-112: cache += x; cat = [x | cache / (time + 1)]).  One wave per row; same arithmetic as the
// separate kernels (x is rounded to bf16 before it enters the running sum, as when it went through memory).
__global__ void __launch_bounds__(256) k_dec_embed(const int* __restrict__ ids, int pad_id, const bf16_t* __restrict__ table,
                                                   const float* __restrict__ bias, const float* __restrict__ timing,
                                                   bf16_t* __restrict__ out, int rows, int H, float scale, int pos0,
                                                   const int* __restrict__ pos_dev, float* __restrict__ cache,
                                                   bf16_t* __restrict__ cat, float inv_count,
                                                   const float* __restrict__ gsrc, const int* __restrict__ gidx, int nl) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= rows) return;
  // beam reorder of the running sums of every layer (search.py:206-209): cache[l][r] <- gsrc[l][gidx[r]]; layer 0's row
  // is updated below
  const int parent = gsrc != nullptr ? gidx[r] : r;
  if (blockIdx.y > 0) {                       // grid.y = nl: the rows of layers 1 .. nl-1 are plain copies
    const int l = blockIdx.y;
    for (int c = lane * 4; c < H; c += 256)
      *reinterpret_cast<float4*>(cache + ((size_t)l * rows + r) * H + c) =
          *reinterpret_cast<const float4*>(gsrc + ((size_t)l * rows + parent) * H + c);
    return;
  }
  const float* cin = gsrc != nullptr ? gsrc + (size_t)parent * H : cache + (size_t)r * H;
  int differs = 0;
  for (int i = lane; i < rows; i += 64) differs |= (ids[i] != pad_id);
  const bool zero_all = !__any(differs);
  if (pos_dev != nullptr) { pos0 = *pos_dev; inv_count = 1.f / (float)(pos0 + 1); }
  const int id = zero_all ? -1 : ids[r];
  const float* tim = timing + (size_t)pos0 * H;
  for (int c = lane * 8; c < H; c += 64 * 8) {
    float v[8];
    if (id >= 0) {
      unpack8(*reinterpret_cast<const uint4*>(table + (size_t)id * H + c), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * scale + bias[c + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += tim[c + j];
    const uint4 xv = pack8(v);
    *reinterpret_cast<uint4*>(out + (size_t)r * H + c) = xv;
    if (cache != nullptr) {
      float o[8];
      unpack8(xv, v);
      float* cp = cache + (size_t)r * H + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float s = cin[c + j] + v[j]; cp[j] = s; o[j] = s * inv_count; }
      *reinterpret_cast<uint4*>(cat + (size_t)r * 2 * H + c) = xv;
      *reinterpret_cast<uint4*>(cat + (size_t)r * 2 * H + H + c) = pack8(o);
    }
  }
}

// transformer_fuse decode step (func.py:262-272): cache += vq;  att += cache / (time + 1)
__global__ void __launch_bounds__(256) k_fuse_decode(const bf16_t* __restrict__ vq, float* __restrict__ cache,
                                                     bf16_t* __restrict__ att, int rows, int H, float inv_count,
                                                     const int* __restrict__ time_dev) {
  if (time_dev != nullptr) inv_count = 1.f / (float)(*time_dev + 1);
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nc = H / 8;
  if (idx >= (size_t)rows * nc) return;
  const size_t r = idx / nc;
  const int c = (int)(idx % nc) * 8;
  float v[8], o[8];
  unpack8(*reinterpret_cast<const uint4*>(vq + r * H + c), v);
  unpack8(*reinterpret_cast<const uint4*>(att + r * H + c), o);
  float* cp = cache + r * H + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float s = cp[j] + v[j]; cp[j] = s; o[j] += s * inv_count; }
  *reinterpret_cast<uint4*>(att + r * H + c) = pack8(o);
}

// Decode-step fusions around the residual + LayerNorm of the decoder (transformer_aan.py:165-192, 92-117): a
// decode step runs on 128 rows, every launch costs its latency chain (~4.6 us for k_add_ln_fwd, k_aan_gate_fwd,
// k_aan_decode alike), so the row-local neighbours of a LayerNorm ride in its launch (zk_lndec_dev.h).  One wave per row.
template <int MAXC>
__global__ void __launch_bounds__(256) k_ln_decode(LnDecArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= a.rows) return;
  uint4 outp[MAXC];
  ln_decode_row<MAXC>(a, r, lane, true, outp);
}

extern "C" {

// logits: fp32 [B*K, ld]; prev_log_probs: fp32 [B*K]; outputs fp32/int32 [B, k2]
// (topk_index = beam*V + symbol).  forbid_id < 0 disables the EOS ban.  scal_dev (device int[2] =
// {float bits of the length penalty, forbid_id}, or NULL) overrides the two per-step scalars at
// run time.  workspace: zk_beam_topk_workspace(B, K, k2) bytes.
static int topk_chunks(int K, int V, int k2) {
  int nc = 256 / (K * k2);                       // merge block: one candidate per thread
  if (nc > 8) nc = 8;
  if (nc < 1 || (V + nc - 1) / nc + 3 > TOPK_CHUNK_MAX) return 0;   // fall back to one block per row
  return nc;
}
size_t zk_beam_topk_workspace(int B, int K, int k2) { return (size_t)B * K * (8 * k2 * 8 + 8 * 8); }
int zk_beam_topk(const float* logits, const float* prev_log_probs, float* topk_scores, int* topk_index, int B,
                 int K, int V, int ld, int k2, float temperature, float length_penalty, int forbid_id,
                 float forbid_value, const int* scal_dev, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG(k2 >= 1 && k2 <= TOPK_MAX && K >= 1 && K <= TOPK_MAX, "zk_beam_topk: K=%d, k2=%d out of range (<=%d)",
               K, k2, TOPK_MAX);
  ZK_CHECK_ARG((long)K * V >= k2 && V >= k2, "zk_beam_topk: fewer candidates than k2");
  ZK_CHECK_ARG(ws_bytes >= zk_beam_topk_workspace(B, K, k2), "zk_beam_topk: workspace too small");
  if (B == 0) return 0;
  const int nc = topk_chunks(K, V, k2);
  if (nc > 0) {
    const int chunk = ((V + nc - 1) / nc + 3) & ~3;       // multiple of 4: every chunk starts 16-byte aligned
    float* part = (float*)workspace;
    float* ck = part + (size_t)B * K * nc * 2;
    int* cv = (int*)(ck + (size_t)B * K * nc * k2);
    if (chunk <= 4096)
      hipLaunchKernelGGL(k_beam_topk_chunks<4>, dim3(B * K, nc), dim3(256), 0, stream, logits, part, ck, cv, V, ld, k2,
                         chunk, nc, 1.f / temperature, forbid_id, forbid_value, scal_dev);
    else
      hipLaunchKernelGGL(k_beam_topk_chunks<8>, dim3(B * K, nc), dim3(256), 0, stream, logits, part, ck, cv, V, ld, k2,
                         chunk, nc, 1.f / temperature, forbid_id, forbid_value, scal_dev);
    ZK_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_beam_topk_merge_chunks, dim3(B), dim3(256), 0, stream, (const float*)part, (const float*)ck,
                       (const int*)cv, prev_log_probs, topk_scores, topk_index, K, V, k2, nc, length_penalty,
                       scal_dev);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  float* cs = (float*)workspace;
  int* ci = (int*)(cs + (size_t)B * K * k2);
  hipLaunchKernelGGL(k_beam_topk_rows, dim3(B * K), dim3(256), 0, stream, logits, prev_log_probs, cs, ci, K, V, ld,
                     k2, 1.f / temperature, length_penalty, forbid_id, forbid_value, scal_dev);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_beam_topk_merge, dim3(B), dim3(256), 0, stream, (const float*)cs, (const int*)ci, topk_scores,
                     topk_index, K, k2);
  ZK_LAUNCH_CHECK();
  return 0;
}

// dst row r <- src row index[r] (index NULL: r); strides and row_bytes in bytes, multiples of 16
int zk_gather_rows_ex(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                      size_t row_bytes, int period, hipStream_t stream);
int zk_gather_rows(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                   size_t row_bytes, hipStream_t stream) {
  return zk_gather_rows_ex(src, src_stride, index, dst, dst_stride, rows, row_bytes, 0, stream);
}
// period > 0: rows = n_tables * period; table t gathers src row t*period + index[r] (one launch for the caches
// of every decoder layer, search.py:206-209)
int zk_gather_rows_ex(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                      size_t row_bytes, int period, hipStream_t stream) {
  ZK_CHECK_ARG(src_stride % 16 == 0 && dst_stride % 16 == 0 && row_bytes % 16 == 0,
               "zk_gather_rows: strides / row_bytes must be multiples of 16");
  ZK_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "zk_gather_rows: pointers must be 16-byte aligned");
  if (rows == 0 || row_bytes == 0) return 0;
  const size_t row16 = row_bytes / 16;
  int gy = (int)((row16 + 255) / 256);
  if (gy > 64) gy = 64;
  ZK_CHECK_ARG(period >= 0 && (period == 0 || rows % period == 0), "zk_gather_rows: rows must be a multiple of period");
  hipLaunchKernelGGL(k_gather_rows, dim3(rows, gy), dim3(256), 0, stream, (const uint4*)src, src_stride / 16, index,
                     (uint4*)dst, dst_stride / 16, row16, period);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_add_gumbel(float* logits, int rows, int V, int ld, float eps, const uint64_t* seed, uint32_t sid,
                  hipStream_t stream) {
  ZK_CHECK_ARG(seed != nullptr && ld >= V, "zk_add_gumbel: seed pointer required, ld >= V");
  const size_t n = (size_t)rows * V;
  if (n == 0) return 0;
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_add_gumbel, dim3((unsigned)g), dim3(256), 0, stream, logits, rows, V, ld, eps, seed, sid);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_cache_rows(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                  size_t unit_bytes, int max_units, const int* time_dev, int mode, int period, hipStream_t stream) {
  ZK_CHECK_ARG(src_stride % 16 == 0 && dst_stride % 16 == 0 && unit_bytes % 16 == 0,
               "zk_cache_rows: strides / unit_bytes must be multiples of 16");
  ZK_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "zk_cache_rows: pointers must be 16-byte aligned");
  ZK_CHECK_ARG(time_dev != nullptr && (mode == 0 || mode == 1), "zk_cache_rows: time_dev required, mode 0 or 1");
  if (rows == 0 || unit_bytes == 0) return 0;
  const size_t unit16 = unit_bytes / 16;
  const size_t span = mode == 0 ? unit16 : unit16 * (size_t)(max_units > 0 ? max_units : 1);
  int gy = (int)((span + 255) / 256);
  if (gy > 64) gy = 64;
  ZK_CHECK_ARG(period >= 0 && (period == 0 || rows % period == 0), "zk_cache_rows: rows must be a multiple of period");
  hipLaunchKernelGGL(k_cache_rows, dim3(rows, gy), dim3(256), 0, stream, (const uint4*)src, src_stride / 16, index,
                     (uint4*)dst, dst_stride / 16, unit16, time_dev, mode, period);
  ZK_LAUNCH_CHECK();
  return 0;
}

// out = LayerNorm(x + y) with the optional neighbours described in zk_lndec_dev.h.  x / out dense [rows, H]; ybuf
// [rows, H] is the sub-layer output (read), or with z != NULL / parts != NULL a scratch row buffer y is written to first.
int zk_ln_decode(const void* x, void* ybuf, const float* gamma, const float* beta, void* out, int rows, int H, float eps,
                 const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                 float* cache, void* cat_out, float inv_count, const int* time_dev, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0 && H <= 2048, "zk_ln_decode: H=%d must be a multiple of 8 and <= 2048", H);
  ZK_CHECK_ARG((cat_in != nullptr || z == nullptr) && (cat_in == nullptr || z != nullptr || parts != nullptr) &&
               (cache == nullptr) == (cat_out == nullptr),
               "zk_ln_decode: the gate needs cat_in and z (or its partial sums); cache / cat_out go together");
  ZK_CHECK_ARG(parts == nullptr || (z == nullptr && nparts >= 1 && part_stride >= (long)rows * H * (cat_in ? 2 : 1)),
               "zk_ln_decode: partial sums exclude z and need nparts >= 1, part_stride >= rows * H (2H for the gate)");
  if (rows == 0) return 0;
  const dim3 grid((unsigned)((rows + 3) / 4));
  LnDecArgs a{(const bf16_t*)x, (bf16_t*)ybuf, gamma, beta, (bf16_t*)out, rows, H, eps, (const bf16_t*)z,
              (const bf16_t*)cat_in, parts, nparts, part_stride, bias, cache, (bf16_t*)cat_out, inv_count, time_dev};
  if (H <= 512) hipLaunchKernelGGL(k_ln_decode<1>, grid, dim3(256), 0, stream, a);
  else if (H <= 1024) hipLaunchKernelGGL(k_ln_decode<2>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_ln_decode<4>, grid, dim3(256), 0, stream, a);
  ZK_LAUNCH_CHECK();
  return 0;
}

// ids int32 [rows] (the tokens fed at this position), table bf16 [V, H], bias fp32 [H], timing fp32 [>= pos + 1, H];
// pos0 / inv_count are overridden by *pos_dev (then inv_count = 1 / (*pos_dev + 1)); cache / cat NULL: no AAN update.
// gather_src != NULL: the running sums of all nl layers ([nl, rows, H] fp32, `cache` = the destination of the same shape)
// are reordered in the same launch: cache[l][r] = gather_src[l][gather_idx[r]] (+ x for layer 0).
int zk_dec_embed(const int* ids, int pad_id, const void* table, const float* bias, const float* timing, void* out, int rows,
                 int H, float scale, int pos0, const int* pos_dev, float* cache, void* cat, float inv_count,
                 const float* gather_src, const int* gather_idx, int nl, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0 && rows >= 0, "zk_dec_embed: H=%d must be a multiple of 8", H);
  ZK_CHECK_ARG((cache == nullptr) == (cat == nullptr), "zk_dec_embed: cache and cat go together");
  ZK_CHECK_ARG(gather_src == nullptr || (cache != nullptr && gather_idx != nullptr && nl >= 1 && gather_src != cache),
               "zk_dec_embed: the reorder needs cache, an index and a source distinct from the destination");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_dec_embed, dim3((unsigned)((rows + 3) / 4), gather_src != nullptr ? nl : 1), dim3(256), 0, stream, ids,
                     pad_id, (const bf16_t*)table,
                     bias, timing, (bf16_t*)out, rows, H, scale, pos0, pos_dev, cache, (bf16_t*)cat, inv_count, gather_src,
                     gather_idx, nl);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_aan_decode(const void* x, float* cache, void* cat, int rows, int H, float inv_count, const int* time_dev,
                  hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_aan_decode: H=%d must be a multiple of 8", H);
  const size_t n = (size_t)rows * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aan_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, cache,
                     (bf16_t*)cat, rows, H, inv_count, time_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_fuse_decode(const void* vq, float* cache, void* att, int rows, int H, float inv_count, const int* time_dev,
                   hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_fuse_decode: H=%d must be a multiple of 8", H);
  const size_t n = (size_t)rows * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_fuse_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)vq, cache,
                     (bf16_t*)att, rows, H, inv_count, time_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Host side of one beam-search step (search.py:85-113 and 168-228) in C: the 2K survivors of every
// sentence come back from the GPU and the alive / finished sets are updated here in ~2 us instead of
// ~25 small numpy calls (~130 us, a fifth of a decode step).  Plain fp32 arithmetic in the reference's
// order; tf.nn.top_k ties -> lower index.  Pure host code.
// ---------------------------------------------------------------------------------------------
// Device-resident search bookkeeping: the same statements as zk_beam_host_step / zk_beam_host_should_stop
// below, run as the first (prepare) and last (advance) node of the decode-step graph, so a step needs no
// host round trip; the host only polls ctrl[1] every few replays.  Once the stop test fires the state is
// frozen (advance is a no-op), so replays issued past the stop change nothing.
//   ctrl int32[4]: [0] steps taken = next time step, [1] stopped, [2] error (time ran into the cache cap)
struct BeamDev {
  int* ctrl; int* stepbuf;               // stepbuf: [0] time, [1] penalty (fp32 bits), [2] banned symbol or -1
  const float* pen_table;                // ((5 + t + 1) / 6)^alpha for t < Tcap, computed on the host in fp32
  const float* max_lp; const int* mtl_i; // per sentence: ((5 + max_target_length) / 6)^alpha, int(max_target_length)
  const float* topk_scores; const int* topk_idx;   // [B, 2K] survivors of this step
  int* seq; int* fin_seq;                // [B, K, Tcap]
  float* log_probs; float* scores; float* fin_scores; int* fin_flags;   // [B, K]
  int* flat_idx; int* next_tok; float* prev;       // [B*K] inputs of the next step
  int B, K, V, Tcap, Tmax, eos_id, pad_id;
};
#define ZK_F32MIN (-3.4028234663852886e38f)

__global__ void __launch_bounds__(256) k_beam_prepare(BeamDev d) {
  __shared__ int s_bound_fail, s_length;
  if (d.ctrl[1]) return;
  if (threadIdx.x == 0) { s_bound_fail = 0; s_length = 0; }
  __syncthreads();
  const int time = d.ctrl[0];
  for (int b = threadIdx.x; b < d.B; b += blockDim.x) {       // search.py:85-113
    const float best_alive = d.log_probs[b * d.K] / d.max_lp[b];
    float worst = INFINITY;
    bool any = false;
    for (int k = 0; k < d.K; ++k) {
      const bool f = d.fin_flags[b * d.K + k] != 0;
      worst = fminf(worst, d.fin_scores[b * d.K + k] * (f ? 1.f : 0.f));
      any = any || f;
    }
    worst = worst + (1.f - (any ? 1.f : 0.f)) * ZK_F32MIN;
    if (!(worst > best_alive)) atomicOr(&s_bound_fail, 1);
    if (time < d.mtl_i[b]) atomicOr(&s_length, 1);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (!s_bound_fail || !s_length) { d.ctrl[1] = 1; return; }
  if (time >= d.Tmax) { d.ctrl[1] = 1; d.ctrl[2] = 1; return; }
  d.stepbuf[0] = time;
  d.stepbuf[1] = __float_as_int(d.pen_table[time]);
  d.stepbuf[2] = time < 1 ? d.eos_id : -1;
  d.ctrl[0] = time + 1;
}

// descending, ties -> lower index (np.argsort(-x, kind="stable")[:k])
__device__ inline void dev_top_k(const float* x, int n, int k, int* idx) {
  unsigned long long used = 0ull;
  for (int r = 0; r < k; ++r) {
    int best = -1;
    for (int i = 0; i < n; ++i) {
      if ((used >> i) & 1ull) continue;
      if (best < 0 || x[i] > x[best]) best = i;
    }
    idx[r] = best;
    used |= 1ull << best;
  }
}

// one block per sentence; dynamic LDS: 2 * K * Tcap ints (the sentence's alive and finished rows).
// The candidate fields are loaded and derived by 2K threads in parallel and the per-beam results written by K
// threads; only the two top-k selections (at most 16 of 48 values) run on one thread.
// the alive / finished sequences of the sentence -> LDS (does not depend on this step's candidates: the fused launch
// requests them before the top-k merge)
__device__ __forceinline__ void beam_advance_stage(const BeamDev& d, int* s_rows) {
  const int b = blockIdx.x, K = d.K, Tcap = d.Tcap, tid = threadIdx.x;
  const int len = d.stepbuf[0] + 1;
  const int* sq = d.seq + (size_t)b * K * Tcap;
  const int* fs = d.fin_seq + (size_t)b * K * Tcap;
  int* l_seq = s_rows;
  int* l_fin = s_rows + K * Tcap;
  for (int i = tid; i < K * len; i += blockDim.x) {
    const int k = i / len, t = i - k * len;
    l_seq[k * Tcap + t] = sq[k * Tcap + t];
    l_fin[k * Tcap + t] = fs[k * Tcap + t];
  }
}
__device__ __forceinline__ void beam_advance_block(const BeamDev& d, int* s_rows, bool staged) {
  __shared__ int s_beam[32], s_cur[32], s_cfin[32], s_aidx[16], s_fidx[16], s_allfl[48];
  __shared__ float s_masked[32], s_allsc[48];
  if (d.ctrl[1]) return;
  const int b = blockIdx.x, K = d.K, K2 = 2 * d.K, Tcap = d.Tcap, tid = threadIdx.x;
  const int time = d.stepbuf[0], len = time + 1;
  int* sq = d.seq + (size_t)b * K * Tcap;
  int* fs = d.fin_seq + (size_t)b * K * Tcap;
  int* l_seq = s_rows;
  int* l_fin = s_rows + K * Tcap;
  if (tid < K2) {                                          // candidates (search.py:168-190)
    const float ts = d.topk_scores[(size_t)b * K2 + tid];
    const int ti = d.topk_idx[(size_t)b * K2 + tid];
    const int cur = ti % d.V;
    const bool fin = (cur == d.eos_id) || (time >= d.mtl_i[b]);
    s_beam[tid] = ti / d.V; s_cur[tid] = cur; s_cfin[tid] = fin;
    s_masked[tid] = ts + (fin ? 1.f : 0.f) * ZK_F32MIN;
    s_allsc[K + tid] = ts + (1.f - (fin ? 1.f : 0.f)) * ZK_F32MIN;
    s_allfl[K + tid] = fin;
  } else if (tid >= 64 && tid < 64 + K) {                  // previous finished set
    s_allsc[tid - 64] = d.fin_scores[b * K + tid - 64];
    s_allfl[tid - 64] = d.fin_flags[b * K + tid - 64];
  }
  if (!staged) beam_advance_stage(d, s_rows);
  __syncthreads();
  if (tid == 0) {                                          // alive (search.py:192-210)
    float x[32]; int idx[16];
    for (int c = 0; c < K2; ++c) x[c] = s_masked[c];
    dev_top_k(x, K2, K, idx);
    for (int k = 0; k < K; ++k) s_aidx[k] = idx[k];
  } else if (tid == 64) {                                  // finished (search.py:212-228): previous K | 2K candidates
    float x[48]; int idx[16];
    for (int c = 0; c < K + K2; ++c) x[c] = s_allsc[c];
    dev_top_k(x, K + K2, K, idx);
    for (int k = 0; k < K; ++k) s_fidx[k] = idx[k];
  }
  __syncthreads();
  if (tid < K) {
    const int c = s_aidx[tid];
    const float penalty = __int_as_float(d.stepbuf[1]);
    d.scores[b * K + tid] = s_masked[c];
    const float lp = s_masked[c] * penalty;
    d.log_probs[b * K + tid] = lp;
    d.prev[b * K + tid] = lp;
    d.flat_idx[b * K + tid] = b * K + s_beam[c];
    d.next_tok[b * K + tid] = s_cur[c];
  } else if (tid >= 64 && tid < 64 + K) {
    const int j = s_fidx[tid - 64];
    d.fin_scores[b * K + tid - 64] = s_allsc[j];
    d.fin_flags[b * K + tid - 64] = s_allfl[j];
  }
  for (int i = tid; i < K * (len + 1); i += blockDim.x) {
    const int k = i / (len + 1), t = i - k * (len + 1);
    const int c = s_aidx[k];
    sq[k * Tcap + t] = t < len ? l_seq[s_beam[c] * Tcap + t] : s_cur[c];
    const int j = s_fidx[k];
    int v;
    if (j < K) v = t < len ? l_fin[j * Tcap + t] : d.pad_id;
    else v = t < len ? l_seq[s_beam[j - K] * Tcap + t] : s_cur[j - K];
    fs[k * Tcap + t] = v;
  }
}
__global__ void __launch_bounds__(128) k_beam_advance(BeamDev d) {
  extern __shared__ int s_rows[];
  beam_advance_block(d, s_rows, false);
}
// merge of the chunked top-k + the bookkeeping of the step in one launch (one block per sentence both ways)
__global__ void __launch_bounds__(256) k_beam_merge_advance(const float* __restrict__ part_ms,
                                                            const float* __restrict__ cand_key,
                                                            const int* __restrict__ cand_v, int k2, int nchunks,
                                                            BeamDev d) {
  extern __shared__ int s_rows[];
  beam_advance_stage(d, s_rows);
  beam_merge_chunks_block(part_ms, cand_key, cand_v, d.prev, const_cast<float*>(d.topk_scores),
                          const_cast<int*>(d.topk_idx), d.K, d.V, k2, nchunks, 1.f, d.stepbuf + 1);
  __threadfence_block();
  __syncthreads();
  beam_advance_block(d, s_rows, true);
}

// ---------------------------------------------------------------------------------------------
static const float kF32Min = -3.4028234663852886e38f;

// search.py:85-113: stop when every sentence's worst finished score beats its best alive bound, or when
// no sentence may grow any more.  Returns 1 to stop.
int zk_beam_host_should_stop(int B, int K, const float* log_probs, const float* fin_scores,
                             const unsigned char* fin_flags, const float* max_target_length, const int* mtl_i,
                             int time, float alpha) {
  bool bound = true, length = false;
  for (int b = 0; b < B; ++b) {
    const float max_lp = powf((5.f + max_target_length[b]) / 6.f, alpha);
    const float best_alive = log_probs[b * K] / max_lp;
    float worst = INFINITY;
    bool any = false;
    for (int k = 0; k < K; ++k) {
      const float f = fin_flags[b * K + k] ? 1.f : 0.f;
      worst = fminf(worst, fin_scores[b * K + k] * f);
      any = any || fin_flags[b * K + k];
    }
    worst = worst + (1.f - (any ? 1.f : 0.f)) * kF32Min;
    if (!(worst > best_alive)) bound = false;
    if (time < mtl_i[b]) length = true;
  }
  return (bound || !length) ? 1 : 0;
}

// descending, ties -> lower index (np.argsort(-x, kind="stable")[:k])
static void host_top_k(const float* x, int n, int k, int* idx) {
  bool used[64];
  for (int i = 0; i < n; ++i) used[i] = false;
  for (int r = 0; r < k; ++r) {
    int best = -1;
    for (int i = 0; i < n; ++i) {
      if (used[i]) continue;
      if (best < 0 || x[i] > x[best]) best = i;        // strict: the first of equal values wins
    }
    idx[r] = best;
    used[best] = true;
  }
}

// seq / fin_seq: int32 [B, K, Tcap], the first `len` entries of every row valid (len = time + 1 on entry,
// time + 2 on exit).  topk_idx = beam * V + symbol.  Outputs the next step's tokens, log-probs, scores and
// the flat beam index (b * K + source beam) of every alive hypothesis.
int zk_beam_host_step(int B, int K, int V, int Tcap, int time, const float* topk_scores, const int* topk_idx,
                      int* seq, int* fin_seq, float* log_probs, float* scores, float* fin_scores,
                      unsigned char* fin_flags, const int* mtl_i, int eos_id, int pad_id, float penalty,
                      int* flat_idx, int* next_tok) {
  const int K2 = 2 * K, len = time + 1;
  if (K < 1 || K2 + K > 64 || len + 1 > Tcap) return -1;
  int cur[64], beam[64], aidx[64], fidx[64];
  float masked[64], allsc[64];
  unsigned char cfin[64], allfl[64];
  int* tmp = (int*)alloca((size_t)(K2 + K) * (len + 1) * sizeof(int) * 2);
  int* new_alive = tmp;                                   // [K][len+1]
  int* new_fin = tmp + (size_t)K * (len + 1);             // [K][len+1]
  for (int b = 0; b < B; ++b) {
    const float* ts = topk_scores + (size_t)b * K2;
    const int* ti = topk_idx + (size_t)b * K2;
    int* sq = seq + (size_t)b * K * Tcap;
    int* fs = fin_seq + (size_t)b * K * Tcap;
    const bool at_cap = time >= mtl_i[b];
    for (int c = 0; c < K2; ++c) {
      beam[c] = ti[c] / V;
      cur[c] = ti[c] % V;
      cfin[c] = (cur[c] == eos_id) || at_cap;
      masked[c] = ts[c] + (cfin[c] ? 1.f : 0.f) * kF32Min;
    }
    // alive (search.py:192-210)
    host_top_k(masked, K2, K, aidx);
    for (int k = 0; k < K; ++k) {
      const int c = aidx[k];
      for (int t = 0; t < len; ++t) new_alive[k * (len + 1) + t] = sq[beam[c] * Tcap + t];
      new_alive[k * (len + 1) + len] = cur[c];
      scores[b * K + k] = masked[c];
      log_probs[b * K + k] = masked[c] * penalty;
      flat_idx[b * K + k] = b * K + beam[c];
      next_tok[b * K + k] = cur[c];
    }
    // finished (search.py:212-228): previous K finished | the 2K candidates
    for (int k = 0; k < K; ++k) { allsc[k] = fin_scores[b * K + k]; allfl[k] = fin_flags[b * K + k]; }
    for (int c = 0; c < K2; ++c) {
      allsc[K + c] = ts[c] + (1.f - (cfin[c] ? 1.f : 0.f)) * kF32Min;
      allfl[K + c] = cfin[c];
    }
    host_top_k(allsc, K + K2, K, fidx);
    for (int k = 0; k < K; ++k) {
      const int j = fidx[k];
      if (j < K) {
        for (int t = 0; t < len; ++t) new_fin[k * (len + 1) + t] = fs[j * Tcap + t];
        new_fin[k * (len + 1) + len] = pad_id;
      } else {
        const int c = j - K;
        for (int t = 0; t < len; ++t) new_fin[k * (len + 1) + t] = sq[beam[c] * Tcap + t];
        new_fin[k * (len + 1) + len] = cur[c];
      }
      masked[k] = allsc[j];                                // reuse as the new finished scores
      cfin[k] = allfl[j];
    }
    for (int k = 0; k < K; ++k) {
      fin_scores[b * K + k] = masked[k];
      fin_flags[b * K + k] = cfin[k];
      for (int t = 0; t <= len; ++t) {
        sq[k * Tcap + t] = new_alive[k * (len + 1) + t];
        fs[k * Tcap + t] = new_fin[k * (len + 1) + t];
      }
    }
  }
  return 0;
}


// Device-resident search state (see BeamDev above).  All pointers are device memory; `state` points at one
// int32/fp32 arena laid out by the caller:  the entry points take the pieces explicitly so the layout stays
// the caller's business.  prepare: <<<1, 256>>>, advance: <<<B, 128, 2*K*Tcap*4>>>.
static int beam_dev_fill(BeamDev* d, int* ctrl, int* stepbuf, const float* pen_table, const float* max_lp,
                         const int* mtl_i, const float* topk_scores, const int* topk_idx, int* seq, int* fin_seq,
                         float* log_probs, float* scores, float* fin_scores, int* fin_flags, int* flat_idx,
                         int* next_tok, float* prev, int B, int K, int V, int Tcap, int Tmax, int eos_id, int pad_id) {
  ZK_CHECK_ARG(B >= 1 && K >= 1 && K <= 16 && V >= 1 && Tcap >= 2, "zk_beam_dev: bad dims B=%d K=%d V=%d Tcap=%d", B, K, V, Tcap);
  ZK_CHECK_ARG((size_t)2 * K * Tcap * sizeof(int) <= 64 * 1024, "zk_beam_dev: K*Tcap=%d rows do not fit LDS", K * Tcap);
  ZK_CHECK_ARG(ctrl && stepbuf && pen_table && max_lp && mtl_i && topk_scores && topk_idx && seq && fin_seq && log_probs &&
               scores && fin_scores && fin_flags && flat_idx && next_tok && prev, "zk_beam_dev: null pointer");
  d->ctrl = ctrl; d->stepbuf = stepbuf; d->pen_table = pen_table; d->max_lp = max_lp; d->mtl_i = mtl_i;
  d->topk_scores = topk_scores; d->topk_idx = topk_idx; d->seq = seq; d->fin_seq = fin_seq;
  d->log_probs = log_probs; d->scores = scores; d->fin_scores = fin_scores; d->fin_flags = fin_flags;
  d->flat_idx = flat_idx; d->next_tok = next_tok; d->prev = prev;
  d->B = B; d->K = K; d->V = V; d->Tcap = Tcap; d->Tmax = Tmax; d->eos_id = eos_id; d->pad_id = pad_id;
  return 0;
}
#define ZK_BEAM_DEV_ARGS                                                                                         \
  int* ctrl, int* stepbuf, const float* pen_table, const float* max_lp, const int* mtl_i,                         \
  const float* topk_scores, const int* topk_idx, int* seq, int* fin_seq, float* log_probs, float* scores,        \
  float* fin_scores, int* fin_flags, int* flat_idx, int* next_tok, float* prev, int B, int K, int V, int Tcap,   \
  int Tmax, int eos_id, int pad_id
#define ZK_BEAM_DEV_PASS                                                                                         \
  ctrl, stepbuf, pen_table, max_lp, mtl_i, topk_scores, topk_idx, seq, fin_seq, log_probs, scores, fin_scores,   \
  fin_flags, flat_idx, next_tok, prev, B, K, V, Tcap, Tmax, eos_id, pad_id
int zk_beam_dev_prepare(ZK_BEAM_DEV_ARGS, hipStream_t stream) {
  BeamDev d;
  if (int rc = beam_dev_fill(&d, ZK_BEAM_DEV_PASS)) return rc;
  hipLaunchKernelGGL(k_beam_prepare, dim3(1), dim3(256), 0, stream, d);
  ZK_LAUNCH_CHECK();
  return 0;
}
// zk_beam_topk (k2 = 2K candidates per sentence from the logits, log-softmax + length penalty / EOS ban read from
// stepbuf[1..2]) followed by zk_beam_dev_advance, with the merge of the chunked top-k and the bookkeeping in ONE launch
// (both are one block per sentence).  Same results as the two calls.  Returns -2 (nothing launched) when the shape needs
// the unchunked top-k: call the two entry points then.
int zk_beam_topk_advance(const float* logits, int ld, float temperature, float forbid_value, void* workspace,
                         size_t ws_bytes, ZK_BEAM_DEV_ARGS, hipStream_t stream) {
  BeamDev d;
  if (int rc = beam_dev_fill(&d, ZK_BEAM_DEV_PASS)) return rc;
  const int k2 = 2 * K;
  ZK_CHECK_ARG(k2 <= TOPK_MAX, "zk_beam_topk_advance: K=%d too large", K);
  ZK_CHECK_ARG(ws_bytes >= zk_beam_topk_workspace(B, K, k2), "zk_beam_topk_advance: workspace too small");
  const int nc = topk_chunks(K, V, k2);
  if (nc <= 0) return -2;
  const int chunk = ((V + nc - 1) / nc + 3) & ~3;
  float* part = (float*)workspace;
  float* ck = part + (size_t)B * K * nc * 2;
  int* cv = (int*)(ck + (size_t)B * K * nc * k2);
  if (chunk <= 4096)
    hipLaunchKernelGGL(k_beam_topk_chunks<4>, dim3(B * K, nc), dim3(256), 0, stream, logits, part, ck, cv, V, ld, k2, chunk,
                       nc, 1.f / temperature, -1, forbid_value, (const int*)(stepbuf + 1));
  else
    hipLaunchKernelGGL(k_beam_topk_chunks<8>, dim3(B * K, nc), dim3(256), 0, stream, logits, part, ck, cv, V, ld, k2, chunk,
                       nc, 1.f / temperature, -1, forbid_value, (const int*)(stepbuf + 1));
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_beam_merge_advance, dim3(B), dim3(256), (size_t)2 * K * Tcap * sizeof(int), stream,
                     (const float*)part, (const float*)ck, (const int*)cv, k2, nc, d);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_beam_dev_advance(ZK_BEAM_DEV_ARGS, hipStream_t stream) {
  BeamDev d;
  if (int rc = beam_dev_fill(&d, ZK_BEAM_DEV_PASS)) return rc;
  hipLaunchKernelGGL(k_beam_advance, dim3(B), dim3(128), (size_t)2 * K * Tcap * sizeof(int), stream, d);
  ZK_LAUNCH_CHECK();
  return 0;
}

// The replay loop of the device-resident search (zero_amd/search.py _beam_search_device) without the interpreter: launch
// the two parity graphs of the decode step alternately, `poll` replays per group; behind every group copy the 16-byte
// control block {time, stop, overflow, -} of the search into a pinned slot and read it one group LATER, so the device
// never idles on the poll; stop when the flag is up or max_launch replays went out (the state is frozen from the stop
// on, replays past it change nothing).  One call per batch: a host thread that drives a batch holds no interpreter lock
// while it does -- with several batches in flight (evalu.decode_many) the per-step Python of the lanes no longer
// serialises them.  ctrl_pinned8: two pinned 4-int slots; *newest_slot_out: the slot of the last copy (complete when
// the call returns: the stream is drained); *launched_out: replays issued.
int zk_beam_dev_run(void* graph_even, void* graph_odd, int parity, const int* ctrl_dev, int* ctrl_pinned8, int max_launch,
                    int poll, hipStream_t stream, int* launched_out, int* newest_slot_out) {
  ZK_CHECK_ARG(graph_even != nullptr && graph_odd != nullptr && ctrl_dev != nullptr && ctrl_pinned8 != nullptr &&
               launched_out != nullptr && newest_slot_out != nullptr, "zk_beam_dev_run: null argument");
  ZK_CHECK_ARG(poll >= 1 && poll <= 64 && (parity == 0 || parity == 1), "zk_beam_dev_run: poll %d / parity %d", poll, parity);
  hipGraphExec_t g[2] = {(hipGraphExec_t)graph_even, (hipGraphExec_t)graph_odd};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
  int launched = 0, group = 0, pending = -1;
  volatile int* slots = ctrl_pinned8;
  while (e == hipSuccess) {
    for (int i = 0; i < poll && e == hipSuccess; ++i) {
      e = hipGraphLaunch(g[parity], stream);
      parity ^= 1;
      ++launched;
    }
    if (e != hipSuccess) break;
    const int slot = group & 1;
    e = hipMemcpyAsync(ctrl_pinned8 + 4 * slot, ctrl_dev, 16, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipEventRecord(ev[slot], stream);
    if (e != hipSuccess) break;
    if (pending >= 0) {
      e = hipEventSynchronize(ev[pending]);
      if (e != hipSuccess) break;
      if (slots[4 * pending + 1] != 0 || launched > max_launch) break;
    }
    pending = slot;
    ++group;
  }
  const hipError_t e2 = hipStreamSynchronize(stream);
  if (ev[0]) hipEventDestroy(ev[0]);
  if (ev[1]) hipEventDestroy(ev[1]);
  *launched_out = launched;
  *newest_slot_out = group & 1;
  if (e != hipSuccess) return zk_set_error((int)e, "zk_beam_dev_run: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return zk_set_error((int)e2, "zk_beam_dev_run: %s", hipGetErrorString(e2));
  return 0;
}
}  // extern "C"
