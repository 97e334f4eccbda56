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
                                                  const float* __restrict__ bias, int act, int tiles_n, int tiles_m) {
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
  // tiles_m > 0 (round 6; KSPLIT == 1, M <= N): tiles are dealt COLUMN-major -- the four waves of a workgroup, and the
  // workgroups next to it, are the row tiles of one column panel of B, the large operand: the logits product of a decode
  // step (128 x 32000 x 512) read its 65 MB of softmax embedding once per row tile, from four workgroups a thousand
  // blocks apart (72 us); now the panel's four readers issue the same loads side by side
  const int tm = tiles_m > 0 ? tile % tiles_m : tile / tiles_n, tn = tiles_m > 0 ? tile / tiles_m : tile % tiles_n;
  const bool live = tm * 32 < M && tn * 32 < N;  // (KSPLIT == 1: the last workgroup may hold tiles past the end)
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

// Round 6: the K-sliced form for products that do not fill the chip (a decode step's 128 rows; the encoder pass of one
// batch).  Measured on the round-5 kernel (profiles/r06_rocprof_decode_f32_1lane_before.txt): a 128-row product took
// 5 us + 2.2 us per ROUND of 64 k -- each wave held four chunks in flight, and a chunk's loads (1.5-2 us from the
// Infinity Cache / HBM) were requested only three chunks of MFMA work (0.6 us) before their use; the feed-forward output
// (K = 2048: eight dependent rounds on 256 waves) took 25 us.  Here a workgroup is KS waves that share one 32 x 32 tile,
// wave w owns the k range [w per, (w + 1) per) with per <= 128, i.e. AT MOST EIGHT CHUNKS, ALL requested before the first
// MFMA: one memory round trip per product, 512-2048 waves per launch instead of 256-1024.  Every output is still exact
// fp32: one fmaf chain per wave over its k in the fixed order of a chunk (k0, k0 + 8, k0 + 1, ...), then the KS partial
// sums added in wave order 0, 1, .. (by all threads of the workgroup: thread t finishes outputs t, t + 64 KS, ..).
template <bool TB, int KS>
__global__ void __launch_bounds__(64 * KS) k_f32_gemm_ks(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                         const float* __restrict__ bias, int act, int tiles_m, int per) {
  extern __shared__ float red[];                 // [KS][1024]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the row tiles of one column panel of B = consecutive slots of ONE XCD (workgroup b runs on XCD b % 8, each XCD has its
  // own L2: see k_f32_gemm_t16); the grid is padded to eight panels per round
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm = slot % tiles_m, tn = (slot / tiles_m) * 8 + xcd;
  if (tn * 32 >= N) return;
  const int kbeg = wave * per, kend = min(K, kbeg + per);
  const int arow = min(tm * 32 + (lane & 31), M - 1);
  const int bcol = min(tn * 32 + (lane & 31), N - 1);
  // sixteen waves = 1024 threads = four waves per SIMD: 128 registers each, room for four chunks in flight; a k range of
  // more than 64 (K = 2048) then takes a second round, requested chunk by chunk as the first is consumed
  constexpr int NCH = KS == 16 ? 4 : 8;
  F32Chunk ch[NCH];
#pragma unroll
  for (int p = 0; p < NCH; ++p)
    if (kbeg + 16 * p < kend) f32_load_chunk<TB>(ch[p], A, B, lda, ldb, arow, bcol, kbeg + 16 * p, kend, lane);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 16 * NCH) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      if (k0 + 16 * p < kend) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[p].a[j], ch[p].b[j], acc, 0, 0, 0);
        if (NCH < 8 && k0 + 16 * (p + NCH) < kend)
          f32_load_chunk<TB>(ch[p], A, B, lda, ldb, arow, bcol, k0 + 16 * (p + NCH), kend, lane);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16 / KS; ++i) {
    const int o = threadIdx.x + 64 * KS * i;     // output o = r * 64 + lane of the tile's accumulator layout
    const int r = o >> 6, l = o & 63;
    float v = red[o];
#pragma unroll
    for (int w = 1; w < KS; ++w) v += red[w * 1024 + o];
    const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = tn * 32 + (l & 31);
    if (row < M && col < N) {
      if (bias != nullptr) v += bias[col];
      if (act == 1) v = fmaxf(v, 0.f);
      C[(size_t)row * ldc + col] = v;
    }
  }
}

template <bool TB, int KS>
static int launch_f32_gemm_ks(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                              const float* bias, int act, int tiles_m, long tiles, int per, hipStream_t stream) {
  auto kern = k_f32_gemm_ks<TB, KS>;
  const size_t lds = (size_t)KS * 1024 * sizeof(float);
  if (lds >= 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return zk_set_error((int)e, "zk_f32_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long grid = (long)tiles_m * ((tiles / tiles_m + 7) / 8) * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * KS), lds, stream, A, B, C, M, N, K, lda, ldb, ldc, bias, act,
                     tiles_m, per);
  ZK_LAUNCH_CHECK();
  return 0;
}

// Round 6, 16 x 16 tiles for the products of a decode step (128 rows): with 32 x 32 tiles a 128 x 512 output is 64
// workgroups -- a quarter of the chip, and the KS waves of a workgroup share ONE CU (the K = 2048 feed-forward output:
// sixteen waves x 64 MFMAs on each of 64 CUs, 6.8 us of matrix-core time alone).  v_mfma_f32_16x16x4_f32 is the same exact
// fp32 arithmetic (an fmaf chain per output, bitwise; MI355X_MICROARCH.md) at the same rate per FLOP, on a tile a quarter
// the size: 256-1024 workgroups per product, every CU busy, 0.4-1.7 us of MFMA time.  Lane (i = lane & 15, g = lane >> 4)
// supplies A[i][k] and B[k][i] for k = 4 g' + j of instruction j (g' = g): a chunk of 16 k is ONE float4 of A per lane (16
// rows x 64 contiguous bytes: every fetched line used in full -- the 32 x 32 form touched each line of A twice) and four
// dwords (B [K, N]) or one float4 (B [N, K]) of B.  At most eight chunks per wave, all requested before the first MFMA.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool TB, int KS>
__global__ void __launch_bounds__(64 * KS) k_f32_gemm_t16(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                          const float* __restrict__ bias, int act, int tiles_m, int per) {
  __shared__ float red[KS * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  // Workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own L2: the row tiles of one column panel
  // of B must be workgroups of ONE XCD, or the panel crosses the fabric once per XCD that reads it (the first version
  // dealt the eight row tiles of a panel to eight XCDs: 4 MB of weights fetched 8 times, 10 us for a product whose
  // 1-MB siblings took 5).  XCD x owns the panels tn = x, x + 8, ..; its consecutive slots are the row tiles of a panel.
  // A 16-column panel of B [K, N] is 64 bytes of every weight row, half a 128-byte L2 line: an XCD owns PAIRS of
  // neighbouring panels (32 columns = whole lines), so that no line is fetched into two L2s.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm = slot % tiles_m, sub = (slot / tiles_m) & 1, pair = slot / (2 * tiles_m);
  const int tn = (pair * 8 + xcd) * 2 + sub;
  if (tn * 16 >= N) return;                                 // (the grid is padded to eight panel pairs per round)
  const int kbeg = wave * per, kend = min(K, kbeg + per);
  const int arow = min(tm * 16 + i, M - 1);
  const int bcol = min(tn * 16 + i, N - 1);
  constexpr int NCH = 8;
  float4 av[NCH], bv[NCH];
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int k = kbeg + 16 * p + 4 * g;                   // (K % 4 == 0: a float4 is inside the row or outside it)
    av[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    bv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend) {
      av[p] = *reinterpret_cast<const float4*>(A + (size_t)arow * lda + k);
      if (TB) bv[p] = *reinterpret_cast<const float4*>(B + (size_t)bcol * ldb + k);
      else {
        const float* bp = B + (size_t)k * ldb + bcol;
        bv[p].x = bp[0]; bv[p].y = bp[ldb]; bv[p].z = bp[2 * (size_t)ldb]; bv[p].w = bp[3 * (size_t)ldb];
      }
    }
  }
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    if (kbeg + 16 * p < kend) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p].x, bv[p].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p].y, bv[p].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p].z, bv[p].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p].w, bv[p].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lane] = acc[r];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int o = threadIdx.x, r = o >> 6, l = o & 63;
    float v = red[o];
#pragma unroll
    for (int w = 1; w < KS; ++w) v += red[w * 256 + o];
    const int row = tm * 16 + 4 * (l >> 4) + r, col = tn * 16 + (l & 15);
    if (row < M && col < N) {
      if (bias != nullptr) v += bias[col];
      if (act == 1) v = fmaxf(v, 0.f);
      C[(size_t)row * ldc + col] = v;
    }
  }
}

// Round 6, the logits product of a decode step (transformer.py:182-196: 128 rows x 32000 words x 512, B = the softmax
// embedding [V, H], K-contiguous): 4.2 GFLOP of exact-fp32 MFMA = 26.7 us at the chip's 157 TF; the one-wave-per-tile kernel
// took 70 us -- every wave pulled its own 32 x 16 fragments of A and B through the CU's L1 (a 64-byte piece of each of
// 32 + 32 rows per chunk: as many L1 line accesses as MFMA cycles) and every workgroup re-read all of A that way.
// Here a workgroup of eight waves owns 128 rows x 64 columns (wave = row tile w & 3, column tile w >> 2) and stages A
// [128 x 32 k] and B [64 x 32 k] through LDS with full-line loads (eight lanes per 128-byte row piece), double-buffered,
// one barrier per 32 k; the fragments come from LDS (row stride 36 floats: the 16 rows a ds_read_b128 phase touches fall
// on distinct banks).  Per output the SAME fmaf chain as k_f32_gemm (k0, k0 + 8, k0 + 1, ... over chunks of 16, no K
// split): the logits are bit-identical to the round-5 kernel's.
template <int DUMMY>
__global__ void __launch_bounds__(512) k_f32_gemm_tb_lds(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                         int ldc, const float* __restrict__ bias, int act) {
  constexpr int S = 36;                               // LDS row stride (floats): 32 k + 4 pad
  __shared__ __align__(16) float sA[2][128 * S];
  __shared__ __align__(16) float sB[2][64 * S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = wave & 3, ct = wave >> 2;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 64;
  // staging roles: thread -> (row, float4 column) of the A / B slab
  const int ar0 = tid >> 3, ac = (tid & 7) * 4;       // A rows ar0 and ar0 + 64
  const int br = tid >> 3;                            // B row (of 64)
  const float* ap0 = A + (size_t)min(m0 + ar0, M - 1) * lda + ac;
  const float* ap1 = A + (size_t)min(m0 + ar0 + 64, M - 1) * lda + ac;
  const float* bp = B + (size_t)min(n0 + br, N - 1) * ldb + ac;
  float4 ra0, ra1, rb;
  auto gload = [&](int k0) {
    ra0 = *reinterpret_cast<const float4*>(ap0 + k0);
    ra1 = *reinterpret_cast<const float4*>(ap1 + k0);
    rb = *reinterpret_cast<const float4*>(bp + k0);
  };
  auto lstore = [&](int st) {
    *reinterpret_cast<float4*>(&sA[st][ar0 * S + ac]) = ra0;
    *reinterpret_cast<float4*>(&sA[st][(ar0 + 64) * S + ac]) = ra1;
    *reinterpret_cast<float4*>(&sB[st][br * S + ac]) = rb;
  };
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int i = lane & 31, h = lane >> 5;
  gload(0);
  lstore(0);
  __syncthreads();
  const int nk = K / 32;
  for (int kb = 0; kb < nk; ++kb) {
    const int st = kb & 1;
    if (kb + 1 < nk) gload((kb + 1) * 32);
    const float* fa = &sA[st][(rt * 32 + i) * S + 8 * h];
    const float* fb = &sB[st][(ct * 32 + i) * S + 8 * h];
#pragma unroll
    for (int c = 0; c < 2; ++c) {                     // the two 16-k chunks of the slab, in k order
      const float4 a0 = *reinterpret_cast<const float4*>(fa + 16 * c), a1 = *reinterpret_cast<const float4*>(fa + 16 * c + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(fb + 16 * c), b1 = *reinterpret_cast<const float4*>(fb + 16 * c + 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    if (kb + 1 < nk) lstore(st ^ 1);
    __syncthreads();
  }
  const int col = n0 + ct * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M) {
      float v = acc[r] + bv;
      if (act == 1) v = fmaxf(v, 0.f);
      C[(size_t)row * ldc + col] = v;
    }
  }
}

static int g_f32_gemm_legacy = 0;
static int g_f32_gemm_t16 = 1;
// A/B switch: 1 = the round-5 kernels for every shape (returns the old value; negative: query only)
extern "C" int zk_f32_gemm_legacy(int on) {
  const int old = g_f32_gemm_legacy;
  if (on >= 0) { g_f32_gemm_legacy = on == 1; g_f32_gemm_t16 = on != 2; }     // 2: round-6 kernels without the 16 x 16 tiles
  return old;
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
  // fewer than two waves per SIMD chip-wide: K-sliced workgroups, every operand chunk requested up front (round 6).
  // KS = 4, 8 or 16 waves per tile: the fewest that bring a wave's k range to <= 128 (eight chunks) and the launch to
  // >= 1024 waves, while a wave keeps at least two chunks.  (g_f32_gemm_legacy: the round-5 kernel, for A/B runs.)
  if (tb && N >= 2048 && K % 32 == 0 && ldb % 4 == 0 && !g_f32_gemm_legacy && g_f32_gemm_t16) {
    hipLaunchKernelGGL(k_f32_gemm_tb_lds<0>, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 127) / 128)), dim3(512), 0, stream,
                       A, B, C, M, N, K, lda, ldb, ldc, bias, act);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  const long tiles16 = (long)((M + 15) / 16) * ((N + 15) / 16);
  if (tiles16 <= 2048 && K >= 64 && K <= 2048 && !g_f32_gemm_legacy && g_f32_gemm_t16) {
    // 16 x 16 tiles: KS = the fewest waves per tile that bring a wave's k range to <= 128 and the launch to >= 2048 waves
    int ks = 4;
    while (ks < 16 && (((K + ks * 16 - 1) / (ks * 16)) * 16 > 128 || tiles16 * ks < 2048) && K / (ks * 2) >= 32) ks *= 2;
    const int per = ((K + ks * 16 - 1) / (ks * 16)) * 16;
    if (per <= 128) {
      const int tm16 = (M + 15) / 16;
      const long grid16 = (long)tm16 * ((((N + 15) / 16) + 15) / 16) * 16;    // eight panel pairs (one per XCD) per round
#define ZK_F32_T16(TB_, KS_)                                                                                              \
  hipLaunchKernelGGL((k_f32_gemm_t16<TB_, KS_>), dim3((unsigned)grid16), dim3(64 * KS_), 0, stream, A, B, C, M, N, K, lda, \
                     ldb, ldc, bias, act, tm16, per)
      if (ks == 4) { if (tb) ZK_F32_T16(true, 4); else ZK_F32_T16(false, 4); }
      else if (ks == 8) { if (tb) ZK_F32_T16(true, 8); else ZK_F32_T16(false, 8); }
      else { if (tb) ZK_F32_T16(true, 16); else ZK_F32_T16(false, 16); }
#undef ZK_F32_T16
      ZK_LAUNCH_CHECK();
      return 0;
    }
  }
  if (tiles < 2048 && K >= 64 && K <= 2048 && !g_f32_gemm_legacy) {
    int ks = 4;
    while (ks < 16 && (((K + ks * 16 - 1) / (ks * 16)) * 16 > 128 || tiles * ks < 1024) && K / (ks * 2) >= 32) ks *= 2;
    const int per = ((K + ks * 16 - 1) / (ks * 16)) * 16;
    if (per <= 128) {
#define ZK_F32_KS(TB_, KS_) launch_f32_gemm_ks<TB_, KS_>(A, B, C, M, N, K, lda, ldb, ldc, bias, act, tiles_m, tiles, per, stream)
      if (ks == 4) return tb ? ZK_F32_KS(true, 4) : ZK_F32_KS(false, 4);
      if (ks == 8) return tb ? ZK_F32_KS(true, 8) : ZK_F32_KS(false, 8);
      return tb ? ZK_F32_KS(true, 16) : ZK_F32_KS(false, 16);
#undef ZK_F32_KS
    }
  }
  // fewer than two waves per SIMD chip-wide and a K loop worth splitting: the four waves of a workgroup share a tile
  const bool split = tiles < 2048 && K >= 256;
#define ZK_F32_GEMM(TB_, KS_, GRID_)                                                                                      \
  hipLaunchKernelGGL((k_f32_gemm<TB_, KS_>), dim3((unsigned)(GRID_)), dim3(256), 0, stream, A, B, C, M, N, K, lda, ldb, \
                     ldc, bias, act, tiles_n, (KS_ == 1 && M <= N && !g_f32_gemm_legacy) ? tiles_m : 0)
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
                                                   const int* __restrict__ pos_dev, const int* __restrict__ all_pad,
                                                   int pad_id, float* __restrict__ cache, float* __restrict__ cat_out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int id = ids[r];
  const int t0 = pos_dev != nullptr ? *pos_dev : pos0;
  const int pos = min(t0 + r % L, timing_rows - 1);
  bool zero = all_pad != nullptr && *all_pad != 0;
  if (pad_id >= 0) {
    // round 6: the all-pad test of transformer.py:113-115 inside this launch (every wave looks at all ids of the step:
    // 128 of them) instead of a launch of its own in front
    bool all = true;
    for (int i = lane; i < rows; i += 64) all = all && ids[i] == pad_id;
    zero = __all(all);
  }
  const float* e = table + (size_t)id * H;
  const float* t = timing + (size_t)pos * H;
  const float div = (float)(t0 + 1);
  for (int c = lane; c < H; c += 64) {
    float v = 0.f;
    if (!zero) { v = e[c] * scale; v = v + bias[c]; }
    v = v + t[c];
    out[(size_t)r * H + c] = v;
    if (cache != nullptr) {        // the first layer's average attention (transformer_aan.py:110-112), L == 1
      const float sc = v + cache[(size_t)r * H + c];
      cache[(size_t)r * H + c] = sc;
      cat_out[(size_t)r * 2 * H + c] = v;
      cat_out[(size_t)r * 2 * H + H + c] = sc / div;
    }
  }
}

extern "C" int zk_f32_embed(const int* ids, int rows, int L, const float* table, const float* bias, const float* timing,
                            int timing_rows, float* out, int H, float scale, int pos0, const int* pos_dev,
                            const int* all_pad, hipStream_t stream) {
  ZK_CHECK_ARG(ids != nullptr && table != nullptr && bias != nullptr && timing != nullptr && out != nullptr && L >= 1 &&
               H >= 1 && timing_rows >= 1, "zk_f32_embed: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_f32_embed, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, rows, L, table, bias, timing, timing_rows,
                     out, H, scale, pos0, pos_dev, all_pad, -1, (float*)nullptr, (float*)nullptr);
  ZK_LAUNCH_CHECK();
  return 0;
}

// The decoder input of ONE decode position (L = 1) with its neighbours in the same launch (round 6): the all-pad test of
// transformer.py:113-115 on the step's ids (pad_id >= 0: every id equal to it -> the embedding part is exact zeros) and,
// with cache / cat_out, the first layer's average attention (cache += x; cat_out = [x | cache / (t + 1)]).
extern "C" int zk_f32_embed_step(const int* ids, int rows, const float* table, const float* bias, const float* timing,
                                 int timing_rows, float* out, int H, float scale, int pos0, const int* pos_dev, int pad_id,
                                 float* cache, float* cat_out, hipStream_t stream) {
  ZK_CHECK_ARG(ids != nullptr && table != nullptr && bias != nullptr && timing != nullptr && out != nullptr && H >= 1 &&
               timing_rows >= 1 && rows <= 65536, "zk_f32_embed_step: bad arguments");
  ZK_CHECK_ARG((cache == nullptr) == (cat_out == nullptr), "zk_f32_embed_step: cache and cat_out go together");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_f32_embed, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, rows, 1, table, bias, timing, timing_rows,
                     out, H, scale, pos0, pos_dev, (const int*)nullptr, pad_id, cache, cat_out);
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

// Round 6: the row-local neighbours of a decode step's LayerNorm in the same launch (one wave per row, four consecutive
// channels per lane, 16-byte loads) -- a decode step is a chain of ~5 us launches, so every launch that is only a row-local
// pass over 128 x H floats is worth folding into its neighbour:
//   y        = ybuf                                       (func.py:321-324 residual input), or, with z / cat_in given,
//              sigmoid(z_i) x_c + sigmoid(z_f) y_c        (transformer_aan.py:186-189; cat_in = [x_c | y_c], and the
//                                                          residual x IS x_c: pass x = NULL)
//   out      = gamma (s - mean) / sqrt(var + eps) + beta,  s = x + y     (func.py:289-303, the two passes over a row)
//   cache / cat_out given: the NEXT layer's average attention from the normalised row (transformer_aan.py:110-112):
//              cache += out;  cat_out = [out | cache / (t + 1)],  t = time_dev ? *time_dev : time
// Same arithmetic as zk_f32_gate + zk_f32_add_ln + zk_f32_aan_step, value by value; the row sums run over four partial
// sums per lane instead of one (fp32 rounding of mean / variance).
#define ZK_F32_LN4_MAXU 8        // H <= 64 * 4 * 8 = 2048
__global__ void __launch_bounds__(256) k_f32_ln_fused(const float* __restrict__ x, const float* __restrict__ ybuf,
                                                      const float* __restrict__ z, const float* __restrict__ cat_in,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ out, int rows, int H, float eps,
                                                      float* __restrict__ cache, float* __restrict__ cat_out, int time,
                                                      const int* __restrict__ time_dev) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  float4 v[ZK_F32_LN4_MAXU], gm[ZK_F32_LN4_MAXU], bt[ZK_F32_LN4_MAXU], cv[ZK_F32_LN4_MAXU];
  float4 ya[ZK_F32_LN4_MAXU], zi[ZK_F32_LN4_MAXU], zf[ZK_F32_LN4_MAXU], yc[ZK_F32_LN4_MAXU];
  const bool gate = cat_in != nullptr, upd = cache != nullptr;
  // every load of the row is requested before the first use: one memory round trip
#pragma unroll
  for (int u = 0; u < ZK_F32_LN4_MAXU; ++u) {
    const int c = (u * 64 + lane) * 4;
    if (c < H) {
      gm[u] = *reinterpret_cast<const float4*>(gamma + c);
      bt[u] = *reinterpret_cast<const float4*>(beta + c);
      if (gate) {
        v[u] = *reinterpret_cast<const float4*>(cat_in + (size_t)r * 2 * H + c);
        yc[u] = *reinterpret_cast<const float4*>(cat_in + (size_t)r * 2 * H + H + c);
        zi[u] = *reinterpret_cast<const float4*>(z + (size_t)r * 2 * H + c);
        zf[u] = *reinterpret_cast<const float4*>(z + (size_t)r * 2 * H + H + c);
      } else {
        v[u] = *reinterpret_cast<const float4*>(x + (size_t)r * H + c);
        if (ybuf != nullptr) ya[u] = *reinterpret_cast<const float4*>(ybuf + (size_t)r * H + c);
      }
      if (upd) cv[u] = *reinterpret_cast<const float4*>(cache + (size_t)r * H + c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < ZK_F32_LN4_MAXU; ++u) {
    const int c = (u * 64 + lane) * 4;
    if (c < H) {
      if (gate) {
        float4 g;
        g.x = (1.0f / (1.0f + expf(-zi[u].x))) * v[u].x + (1.0f / (1.0f + expf(-zf[u].x))) * yc[u].x;
        g.y = (1.0f / (1.0f + expf(-zi[u].y))) * v[u].y + (1.0f / (1.0f + expf(-zf[u].y))) * yc[u].y;
        g.z = (1.0f / (1.0f + expf(-zi[u].z))) * v[u].z + (1.0f / (1.0f + expf(-zf[u].z))) * yc[u].z;
        g.w = (1.0f / (1.0f + expf(-zi[u].w))) * v[u].w + (1.0f / (1.0f + expf(-zf[u].w))) * yc[u].w;
        v[u].x += g.x; v[u].y += g.y; v[u].z += g.z; v[u].w += g.w;
      } else if (ybuf != nullptr) {
        v[u].x += ya[u].x; v[u].y += ya[u].y; v[u].z += ya[u].z; v[u].w += ya[u].w;
      }
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < ZK_F32_LN4_MAXU; ++u) {
    const int c = (u * 64 + lane) * 4;
    if (c < H) {
      const float d0 = v[u].x - mean, d1 = v[u].y - mean, d2 = v[u].z - mean, d3 = v[u].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float var = wave_sum(q) / (float)H;
  const float rs = 1.0f / sqrtf(var + eps);
  const float div = (float)((time_dev != nullptr ? *time_dev : time) + 1);
#pragma unroll
  for (int u = 0; u < ZK_F32_LN4_MAXU; ++u) {
    const int c = (u * 64 + lane) * 4;
    if (c < H) {
      float4 o;
      o.x = gm[u].x * (v[u].x - mean) * rs + bt[u].x;
      o.y = gm[u].y * (v[u].y - mean) * rs + bt[u].y;
      o.z = gm[u].z * (v[u].z - mean) * rs + bt[u].z;
      o.w = gm[u].w * (v[u].w - mean) * rs + bt[u].w;
      *reinterpret_cast<float4*>(out + (size_t)r * H + c) = o;
      if (upd) {
        float4 sc = make_float4(o.x + cv[u].x, o.y + cv[u].y, o.z + cv[u].z, o.w + cv[u].w);
        *reinterpret_cast<float4*>(cache + (size_t)r * H + c) = sc;
        *reinterpret_cast<float4*>(cat_out + (size_t)r * 2 * H + c) = o;
        *reinterpret_cast<float4*>(cat_out + (size_t)r * 2 * H + H + c) = make_float4(sc.x / div, sc.y / div, sc.z / div, sc.w / div);
      }
    }
  }
}

extern "C" int zk_f32_ln_fused(const float* x, const float* ybuf, const float* z, const float* cat_in, const float* gamma,
                               const float* beta, float* out, int rows, int H, float eps, float* cache, float* cat_out,
                               int time, const int* time_dev, hipStream_t stream) {
  ZK_CHECK_ARG(gamma != nullptr && beta != nullptr && out != nullptr && H >= 4 && H % 4 == 0 && H <= 256 * ZK_F32_LN4_MAXU,
               "zk_f32_ln_fused: H=%d must be a multiple of 4, at most %d", H, 256 * ZK_F32_LN4_MAXU);
  ZK_CHECK_ARG((cat_in != nullptr) ? (z != nullptr && x == nullptr && ybuf == nullptr) : (x != nullptr && z == nullptr),
               "zk_f32_ln_fused: either (x, ybuf) or the gate form (z, cat_in; x = ybuf = NULL)");
  ZK_CHECK_ARG((cache == nullptr) == (cat_out == nullptr), "zk_f32_ln_fused: cache and cat_out go together");
  ZK_CHECK_ARG((((uintptr_t)x | (uintptr_t)ybuf | (uintptr_t)z | (uintptr_t)cat_in | (uintptr_t)gamma | (uintptr_t)beta |
                 (uintptr_t)out | (uintptr_t)cache | (uintptr_t)cat_out) & 15) == 0, "zk_f32_ln_fused: 16-byte alignment");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_f32_ln_fused, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ybuf, z, cat_in, gamma, beta, out, rows, H,
                     eps, cache, cat_out, time, time_dev);
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
