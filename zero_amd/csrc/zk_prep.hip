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
// k_embed_bwd_sorted.  One workgroup of 1024 threads, keys in LDS up to 16384 rows per side (128 KiB), in a
// caller-provided global scratch beyond that.  4096 rows: 78 compare-exchange rounds, ~6 us.
#include "zk_common.h"

#define ZK_PREP_NT 1024

struct PrepSide {
  const int* ids;      // [B, L]
  int* rows;           // [T]     out: token rows grouped by id
  int* seg;            // [T + 1] out: group boundaries
  int* uid;            // [T]     out: the id of each group
  int* n;              // [1]     out: number of groups
  unsigned long long* scratch;   // [npad] global keys when npad > the LDS capacity of the instantiation
  int L, shift, npad;  // npad: power of two >= B * L
};

struct PrepArgs {
  PrepSide side[2];
  float* smask;        // [B, Ls] or null
  float* tmask;        // [B, Lt] or null
  float* tw;           // [B, Lt] or null
  int B;
  float loss_scale;
};

template <bool GLOBAL>
__device__ __forceinline__ void prep_sort_side(const PrepSide& s, int B, unsigned long long* lds_keys) {
  const int tid = threadIdx.x;
  const int T = B * s.L, N = s.npad;
  unsigned long long* key = GLOBAL ? s.scratch : lds_keys;
  __shared__ int scan[ZK_PREP_NT];
  for (int i = tid; i < N; i += ZK_PREP_NT) {
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
      for (int i = tid; i < (N >> 1); i += ZK_PREP_NT) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1)), b = a + j;
        const unsigned long long x = key[a], y = key[b];
        const bool up = (a & k) == 0;
        if ((x > y) == up) { key[a] = y; key[b] = x; }
      }
      __syncthreads();
    }
  }
  // group heads -> uid / seg; every thread owns a contiguous chunk of the sorted order
  const int per = (N + ZK_PREP_NT - 1) / ZK_PREP_NT;
  const int p0 = tid * per, p1 = min(p0 + per, N);
  int heads = 0;
  for (int p = p0; p < p1; ++p) {
    const unsigned long long k = key[p];
    if (k == ~0ull) break;
    if (p == 0 || (unsigned)(k >> 32) != (unsigned)(key[p - 1] >> 32)) ++heads;
  }
  scan[tid] = heads;
  __syncthreads();
  for (int off = 1; off < ZK_PREP_NT; off <<= 1) {       // inclusive Hillis-Steele scan of the 1024 counts
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
  if (tid == ZK_PREP_NT - 1) {
    const int n = scan[ZK_PREP_NT - 1];
    s.seg[n] = s.shift ? B * max(s.L - 1, 0) : T;       // rows that carry an embedding
    s.n[0] = n;
  }
}

template <int NLDS>
__global__ void __launch_bounds__(ZK_PREP_NT) k_batch_prep(PrepArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NLDS];
  const int blk = blockIdx.x;
  if (blk < 2) {
    const PrepSide& s = a.side[blk];
    if (s.ids == nullptr || s.rows == nullptr) return;
    if (s.npad > NLDS) prep_sort_side<true>(s, a.B, keys);
    else prep_sort_side<false>(s, a.B, keys);
    return;
  }
  // masks and loss weights of sentence b (func.py:372-387; transformer.py:198-211: per-sentence mean, then batch mean)
  const int b = blk - 2, tid = threadIdx.x;
  const int Ls = a.side[0].L, Lt = a.side[1].L;
  if (a.smask != nullptr)
    for (int t = tid; t < Ls; t += ZK_PREP_NT) a.smask[b * Ls + t] = (a.side[0].ids[b * Ls + t] != 0) ? 1.f : 0.f;
  if (a.side[1].ids == nullptr || (a.tmask == nullptr && a.tw == nullptr)) return;
  float* sm = reinterpret_cast<float*>(keys);
  float cnt = 0.f;
  for (int t = tid; t < Lt; t += ZK_PREP_NT) cnt += (a.side[1].ids[b * Lt + t] != 0) ? 1.f : 0.f;
  const float len = block_sum<ZK_PREP_NT / 64>(cnt, sm);
  for (int t = tid; t < Lt; t += ZK_PREP_NT) {
    const float mk = (a.side[1].ids[b * Lt + t] != 0) ? 1.f : 0.f;
    if (a.tmask != nullptr) a.tmask[b * Lt + t] = mk;
    if (a.tw != nullptr) a.tw[b * Lt + t] = a.loss_scale * mk / (len * (float)a.B);
  }
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
                  float* tmask, float* tw, float loss_scale, void* scratch, size_t scratch_bytes, hipStream_t stream) {
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
  a.side[0] = PrepSide{src_ids, src_rows, src_seg, src_uid, src_n, (unsigned long long*)scratch, Ls, 0,
                       prep_npad((long)B * Ls)};
  a.side[1] = PrepSide{tgt_ids, tgt_ids ? tgt_rows : nullptr, tgt_seg, tgt_uid, tgt_n,
                       (unsigned long long*)((char*)scratch + need_s), Lt, 1, prep_npad((long)B * Lt)};
  ZK_CHECK_ARG(src_rows == nullptr || (src_seg && src_uid && src_n), "zk_batch_prep: source outputs incomplete");
  ZK_CHECK_ARG(a.side[1].rows == nullptr || (tgt_seg && tgt_uid && tgt_n), "zk_batch_prep: target outputs incomplete");
  int big = 0;
  if (src_rows) big = a.side[0].npad;
  if (a.side[1].rows && a.side[1].npad > big) big = a.side[1].npad;
  const dim3 grid(2 + B), blk(ZK_PREP_NT);
  if (big <= 4096) hipLaunchKernelGGL(k_batch_prep<4096>, grid, blk, 0, stream, a);
  else hipLaunchKernelGGL(k_batch_prep<16384>, grid, blk, 0, stream, a);
  ZK_LAUNCH_CHECK();
  return 0;
}
}  // extern "C"
