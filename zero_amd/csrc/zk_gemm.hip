// zk_gemm.hip -- bf16 MFMA GEMM with fused epilogues for the dense contractions of the
// Transformer hot path (func.py:14-65 `linear`, func.py:327-338 FFN, transformer.py:182-196
// logits, and their dgrad / wgrad mirrors), gfx950 only.
//
//   C[M,N] = alpha * op(A)[M,K] x op(B)[K,N]   (+bias[N]) (+residual[M,N]) (act) (dropout)
//
//   ta=0: A stored [M,lda], K contiguous      ta=1: A stored [K,lda], M contiguous
//   tb=0: B stored [K,ldb], N contiguous      tb=1: B stored [N,ldb], K contiguous
//
//   forward  x@W      : ta=0 tb=0        dgrad  dY@W^T : ta=0 tb=1
//   wgrad    X^T@dY   : ta=1 tb=0        logits h@E^T  : ta=0 tb=1
//
// Kernel structure (v1): BMxBNx64 block tile, 4 wave64 as 2x2, v_mfma_f32_32x32x16_bf16,
// single LDS stage + register prefetch of the next K tile.  LDS tiles are [rows][64+8] bf16
// (K contiguous, 144-byte rows -> conflict-free ds_read_b128 fragments).  Operands whose
// contraction dim is NOT contiguous in HBM (ta=1 / tb=0) are transposed on the way into
// LDS: each thread loads 4 k-rows x 8 contiguous rows (4x16 B, coalesced), and writes
// 8 x ds_write_b64; the tile's rows are stored in a permuted physical order
// (phys = (r%8)*(R/8) + r/8) so that those writes spread over LDS banks; the permutation is
// undone in the epilogue's row/column indices.
#include "zk_common.h"

#include "zk_gemm.h"
#include "zk_prog.h"

// ------------------------------------------------------------------ reference kernel
// one thread per output element; any shape / alignment.  Used for parity checks of the
// MFMA kernel and for shapes the MFMA kernel does not accept.
__global__ void __launch_bounds__(256) k_gemm_naive(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                    int M, int N, int K, int lda, int ldb, int ta, int tb,
                                                    GemmEpi e) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = bf2f(ta ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]);
    const float b = bf2f(tb ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]);
    acc += a * b;
  }
  const uint64_t seed = e.thr ? *e.seed : 0;
  epi_store(e, acc, m, n, N, seed);
}

// ------------------------------------------------------------------ MFMA kernel
template <int R, bool TRANS>
__device__ __forceinline__ void g2r(uint4 (&reg)[4], const bf16_t* __restrict__ src, int ld, int row0,
                                    int rows_total, int k0, int kend, int tid) {
  if (!TRANS) {
#pragma unroll
    for (int i = 0; i < (R * 8) / 256; ++i) {
      const int t = tid + i * 256;
      const int r = t >> 3, kc = t & 7;
      const int grow = row0 + r, gk = k0 + kc * 8;
      if (grow < rows_total && gk < kend)
        reg[i] = *reinterpret_cast<const uint4*>(src + (size_t)grow * ld + gk);
      else
        reg[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
    constexpr int RC = R / 8;
    if (tid < RC * (BK / 4)) {
      const int rc = tid % RC, kq = tid / RC;
      const int grow = row0 + rc * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int gk = k0 + kq * 4 + kk;
        if (gk < kend && grow < rows_total)
          reg[kk] = *reinterpret_cast<const uint4*>(src + (size_t)gk * ld + grow);
        else
          reg[kk] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
}

__device__ __forceinline__ uint32_t half_of(const uint4& v, int i) {
  const uint32_t w = (i >> 1) == 0 ? v.x : ((i >> 1) == 1 ? v.y : ((i >> 1) == 2 ? v.z : v.w));
  return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

template <int R, bool TRANS>
__device__ __forceinline__ void r2s(const uint4 (&reg)[4], bf16_t* sT, int tid) {
  if (!TRANS) {
#pragma unroll
    for (int i = 0; i < (R * 8) / 256; ++i) {
      const int t = tid + i * 256;
      const int r = t >> 3, kc = t & 7;
      *reinterpret_cast<uint4*>(sT + r * LDS_LD + kc * 8) = reg[i];
    }
  } else {
    constexpr int RC = R / 8;
    if (tid < RC * (BK / 4)) {
      const int rc = tid % RC, kq = tid / RC;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint2 o;
        o.x = half_of(reg[0], i) | (half_of(reg[1], i) << 16);
        o.y = half_of(reg[2], i) | (half_of(reg[3], i) << 16);
        const int phys = i * RC + rc;
        *reinterpret_cast<uint2*>(sT + phys * LDS_LD + kq * 4) = o;
      }
    }
  }
}

// physical LDS row -> logical tile row
template <int R, bool TRANS>
__device__ __forceinline__ int logical_row(int q) {
  if (!TRANS) return q;
  constexpr int RC = R / 8;
  return (q % RC) * 8 + q / RC;
}

template <int BM, int BN, bool TA, bool TB>
__global__ void __launch_bounds__(256) k_gemm_mfma(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                   int M, int N, int K, int lda, int ldb, int kchunk,
                                                   float* __restrict__ slabs, TileSched ts, GemmEpi e) {
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  __shared__ __attribute__((aligned(16))) bf16_t sA[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) bf16_t sB[BN * LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm_, tn_, z_;
  tile_of_block(ts, tm_, tn_, z_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int kbeg = z_ * kchunk;
  const int kend = min(K, kbeg + kchunk);

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  // A tile rows index M; "transposed staging" when M is the contiguous dim (ta=1).
  // B tile rows index N; "transposed staging" when N is the contiguous dim (tb=0).
  g2r<BM, TA>(ra, A, lda, m0, M, kbeg, kend, tid);
  g2r<BN, !TB>(rb, B, ldb, n0, N, kbeg, kend, tid);

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    r2s<BM, TA>(ra, sA, tid);
    r2s<BN, !TB>(rb, sB, tid);
    __syncthreads();
    if (k0 + BK < kend) {
      g2r<BM, TA>(ra, A, lda, m0, M, k0 + BK, kend, tid);
      g2r<BN, !TB>(rb, B, ldb, n0, N, k0 + BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(
            sA + (wm * WTM + i * 32 + (lane & 31)) * LDS_LD + kk * 16 + (lane >> 5) * 8);
        af[i] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(
            sB + (wn * WTN + j * 32 + (lane & 31)) * LDS_LD + kk * 16 + (lane >> 5) * 8);
        bfr[j] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of v_mfma_f32_32x32x16: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)
  const uint64_t seed = e.thr ? *e.seed : 0;
  const bool to_slab = slabs != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int qn = wn * WTN + j * 32 + (lane & 31);
      const int gn = n0 + logical_row<BN, !TB>(qn);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qm = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int gm = m0 + logical_row<BM, TA>(qm);
        if (gm < M && gn < N) {
          if (to_slab) slabs[((size_t)z_ * M + gm) * N + gn] = acc[i][j][r];
          else epi_store(e, acc[i][j][r], gm, gn, N, seed);
        }
      }
    }
}

// split-K finish: C = epilogue(sum_z slabs[z])
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ slabs, int splits, int M, int N,
                                                       GemmEpi e) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  float t = 0.f;
  for (int z = 0; z < splits; ++z) t += slabs[(size_t)z * M * N + idx];
  const uint64_t seed = e.thr ? *e.seed : 0;
  epi_store(e, t, (int)(idx / N), (int)(idx % N), N, seed);
}

template <int BM, int BN>
static int launch_mfma(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                       int splits, int kchunk, float* slabs, const GemmEpi& e, int sched_flags,
                       hipStream_t stream) {
  TileSched ts;
  ts.tiles_m = (M + BM - 1) / BM;
  ts.tiles_n = (N + BN - 1) / BN;
  ts.n_major = ((long)N > (long)M) ? 1 : 0;   // keep the panels of the larger operand L2-resident
  if (sched_flags & 2) ts.n_major ^= 1;
  ts.xcd_remap = (sched_flags & 1) ? 0 : 1;
  dim3 grid((unsigned)((long)ts.tiles_m * ts.tiles_n * splits));
  if (!ta && !tb)
    hipLaunchKernelGGL((k_gemm_mfma<BM, BN, false, false>), grid, dim3(256), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e);
  else if (!ta && tb)
    hipLaunchKernelGGL((k_gemm_mfma<BM, BN, false, true>), grid, dim3(256), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e);
  else if (ta && !tb)
    hipLaunchKernelGGL((k_gemm_mfma<BM, BN, true, false>), grid, dim3(256), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e);
  else
    hipLaunchKernelGGL((k_gemm_mfma<BM, BN, true, true>), grid, dim3(256), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e);
  ZK_LAUNCH_CHECK();
  return 0;
}

static bool mfma_ok(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int ta, int tb) {
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return false;
  if (lda % 8 || ldb % 8) return false;
  if (!ta && K % 8) return false;   // A: K contiguous
  if (ta && M % 8) return false;    // A: M contiguous
  if (tb && K % 8) return false;    // B: K contiguous
  if (!tb && N % 8) return false;   // B: N contiguous
  return true;
}

// Tile / split heuristic (measured on MI355X, scripts/gemm_bench.py, scripts/gemm_ns_bench.py).
// Finding of round 1: the K loop of these small GEMMs is bound by the per-wave cost of ISSUING
// loads (one 1-KiB LDS-DMA per ~100-150 cycles), not by HBM, L2 (82 % hits) or ring depth; what
// helps is (a) more co-resident workgroups per CU and (b) more MFMAs per load (bigger tiles) --
// which conflict for outputs as small as 4096x512.  These GEMMs are small
// for a 256-CU chip: with <= 1 workgroup per CU the K loop is exposed-latency bound, so the
// goal is >= ~3 co-resident workgroups per CU (thread-level parallelism hides the L2 latency)
// before tile area (arithmetic intensity) is considered.  Large outputs use 128x128; mid-size
// ones 64x128 / 128x64; small ones 64x64 plus split-K when the epilogue allows it.
extern int g_tune[24];
static void pick_config(int M, int N, int K, int allow_split, int* bm, int* bn, int* splits, int* wide = nullptr) {
  if (wide) *wide = 0;
  // Measured on MI355X (scripts/gemm_bench.py at base and big widths, profiles/r01_gemm_microbench*.txt):
  //  * >= 8 M outputs (or a very long K): 128x128 -- most MFMAs per LDS byte, still >= 512 workgroups;
  //  * 3 M .. 8 M outputs (4096x1024, 4096x1536): 128x64 / 64x128 -- 64x64 leaves 20-35 % on the table
  //    there (4096x1024x4096: 46 vs 63 us);
  //  * smaller outputs (4096x512): 64x64 -- more co-resident workgroups beat bigger tiles;
  //  * split-K when the tiles cannot fill even a quarter of the CUs (with >= 128 tiles and K <= 4096 every
  //    split measured slower than none: 1024x1024x4096 17.8 us unsplit, 27.4 us split 4), and for the very
  //    long K of dlogits x E (K = 32000: 2-way split to 1024 workgroups is worth 0.15 ms per step).
  const long out = (long)M * N;
  if (g_tune[5] && out >= (long)g_tune[5] * 1024 * 1024 && M >= 256) { *bm = 256; *bn = 128; }   // A/B: 256x128 macro tile
  else if (wide && !g_tune[5] && K >= 8192 && M >= 128 && N >= 256 && allow_split) {
    // dlogits x E (K = 32000): 128x256 tiles of four 128x64 register tiles + producer waves, ring depth 3
    // (scripts/gemm_big_bench.py: 245 -> 203 us)
    *bm = 128; *bn = 256; *wide = 1;
  }
  else if (out >= 8L * 1024 * 1024 || (K >= 8192 && M >= 128 && N >= 128)) { *bm = 128; *bn = 128; }
  else if (out >= 3L * 1024 * 1024) { if (N >= M) { *bm = 64; *bn = 128; } else { *bm = 128; *bn = 64; } }
  else if (g_tune[2] == 1 || g_tune[2] == 3) { if (N >= M) { *bm = 64; *bn = 128; } else { *bm = 128; *bn = 64; } }   // A/B switches (3: deep ring, see dispatch)
  else if (g_tune[2] == 2) { *bm = 128; *bn = 128; }
  else if (g_tune[11] > 0 && K >= g_tune[11] && out >= 1024L * 1024) {   // A/B: long-K products with a small output on 128x64
    if (N >= M) { *bm = 64; *bn = 128; } else { *bm = 128; *bn = 64; }
  }
  else { *bm = 64; *bn = 64; }
  const long tiles = (long)((M + *bm - 1) / *bm) * ((N + *bn - 1) / *bn);
  int s = 1;
  if (wide && *wide && allow_split && tiles < 256) s = (int)((256 + tiles - 1) / tiles);   // one 144-KiB workgroup per CU
  else if (g_tune[1] && allow_split && tiles < 1024 && K >= 1024) s = (int)((1024 + tiles - 1) / tiles);   // A/B: round-1 rule
  else if (allow_split && tiles < 1024 && K >= 8192) s = (int)((1024 + tiles - 1) / tiles);   // dlogits x E: K = 32000
  else if (allow_split && tiles <= 96 && K >= 1024) s = (int)((256 + tiles - 1) / tiles);
  if (s > 1) {
    const int maxs = K / (4 * BK);            // keep >= 4 K-tiles per split
    if (s > maxs) s = maxs;
    if (s > 8) s = 8;
    const long slab_cap = ((K >= 8192 ? 64L : 16L) << 20) / ((long)M * N * 4);   // keep the fp32 slab traffic <= 16 MiB (64 MiB behind a very long K)
    if (s > slab_cap) s = (int)slab_cap;
    if (s < 1) s = 1;
  }
  *splits = s;
}

int zk_gemm_dlds_pw(int bm, int bn);
int zk_gemm_dlds_ln_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int bm, int bn,
                             const GemmEpi& e, hipStream_t stream);
int zk_gemm_dlds_sync_ln_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int bm,
                                  const GemmEpi& e, hipStream_t stream);
int zk_gemm_dlds_sync_ln_bwd_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb,
                                      const GemmEpi& e, hipStream_t stream);
#ifdef ZK_EXPERIMENTS
int zk_ffn_pair_launch(const bf16_t* x, const bf16_t* W1, bf16_t* h, const bf16_t* W2, float* parts, int M, int F, int H, int K1,
                       int ldx, int ldw1, int ldw2, int kchunk, int nparts, const GemmEpi& e1, const GemmEpi& e2,
                       unsigned long long* cnt, int* err, hipStream_t stream);
#endif
int zk_gemm_dlds_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                          int bm, int bn, int splits, int kchunk, float* slabs, const GemmEpi& e, int sched_flags,
                          hipStream_t stream);
static int g_default_gen = 2;   // generation used by impl=0 (auto) / impl=2: 1 = k_gemm_mfma, 2 = k_gemm_dlds

extern "C" {

// which MFMA kernel generation `auto` uses (tuning / A-B tests); returns the previous value
int zk_gemm_set_generation(int gen) {
  const int old = g_default_gen;
  if (gen == 1 || gen == 2) g_default_gen = gen;
  return old;
}

// workspace bytes zk_gemm may need for (M,N,K) (split-K slabs), upper bound
size_t zk_gemm_workspace(int M, int N, int K) {
  int bm, bn, s, wide;
  pick_config(M, N, K, 1, &bm, &bn, &s, &wide);
  if (s < 2 && K >= 1024) s = 2;
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
// Which kernel would impl=0 pick?  Returns gen | (bm/8)<<8 | (bn/8)<<16 | splits<<24 | producer waves<<28 (for
// labelling measurements with the kernel instance that actually runs).
int zk_gemm_plan(int M, int N, int K, int out_f32, int plain) {
  int bm, bn, s, wide;
  pick_config(M, N, K, plain ? 1 : 0, &bm, &bn, &s, &wide);
  int gen = g_default_gen;
  if (!g_tune[7] && gen == 2 && out_f32 && K <= 512 && (long)M * N >= 64L * 1024 * 1024) gen = 1;
  // tile sizes in units of 8 (256 does not fit a byte); wide register tiles always carry four producer waves
  return gen | ((bm / 8) << 8) | ((bn / 8) << 16) | ((s & 15) << 24) | ((gen == 2 ? (wide ? 4 : zk_gemm_dlds_pw(bm, bn)) : 0) << 28);
}

// workspace for an explicit split-K override (tuning)
size_t zk_gemm_workspace_split(int M, int N, int splits) { return (size_t)splits * M * N * sizeof(float); }

int zk_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int ta,
            int tb, int out_f32, float alpha, const float* bias, const void* residual, int ldr, int act,
            const void* aux, int ldaux, float aux_scale, float drop_p, const uint64_t* seed, uint32_t sid,
            int impl, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "zk_gemm: negative dims");
  ZK_CHECK_ARG(act >= 0 && act <= 2, "zk_gemm: act=%d unknown", act);
  ZK_CHECK_ARG(act != 2 || aux != nullptr, "zk_gemm: act=2 needs aux");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_gemm: dropout needs a seed pointer");
  // impl bits: [1:0] 0 auto / 1 reference / 2 mfma (default generation) / 3 mfma generation 1;
  // [11:8] tile override (1:128x128 2:128x64
  // 3:64x128 4:64x64); [23:16] split-K override (tuning / tests)
  const int tile_ovr = (impl >> 8) & 15, split_ovr = (impl >> 16) & 255;
  const int sched_flags = ((impl >> 12) & 3) | (((impl >> 24) & 15) << 4);   // bit0: no XCD remap, bit1: flip panel order, [7:4] ring depth override
  impl &= 3;
  int gen = g_default_gen;
  if (impl == 3) { gen = 1; impl = 2; }
  // measured (scripts/gemm_bench.py): a huge fp32 output with a short K loop (the logits GEMM) is
  // epilogue-bound; the register-staged kernel keeps 2 workgroups per CU and overlaps one's
  // epilogue with the other's K loop
  else if (!g_tune[7] && impl == 0 && gen == 2 && out_f32 && K <= 512 && (long)M * N >= 64L * 1024 * 1024) gen = 1;
  if (M == 0 || N == 0) return 0;
  GemmEpi e;
  e.C = C; e.ldc = ldc; e.out_f32 = out_f32; e.alpha = alpha; e.bias = bias;
  e.res = (const bf16_t*)residual; e.ldr = ldr; e.act = act; e.aux = (const bf16_t*)aux; e.ldaux = ldaux;
  e.aux_scale = aux_scale;
  e.thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  e.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  e.seed = seed; e.sid = sid;
  const bool ok = mfma_ok(A, B, M, N, K, lda, ldb, ta, tb);
  ZK_CHECK_ARG(impl != 2 || ok, "zk_gemm: shape/alignment not supported by the MFMA kernel "
               "(M=%d N=%d K=%d lda=%d ldb=%d ta=%d tb=%d)", M, N, K, lda, ldb, ta, tb);
  if (zk_prog_active() && (impl == 1 || (impl == 0 && !ok)))
    return zk_prog_reject("gemm on the reference kernel");
  if (impl == 1 || (impl == 0 && !ok)) {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(k_gemm_naive, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)A,
                       (const bf16_t*)B, M, N, K, lda, ldb, ta, tb, e);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  int bm, bn, splits;
  const bool plain = (bias == nullptr && residual == nullptr && act == 0 && drop_p == 0.f);
  int wide = 0;
  pick_config(M, N, K, plain ? 1 : 0, &bm, &bn, &splits, &wide);
     // tile overrides 6 / 7: 256x128 / 128x256 with 128x64 register tiles per wave
  if (tile_ovr && tile_ovr <= 7) {
    const int tb_[8][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}, {256, 128}, {128, 256}};
    bm = tb_[tile_ovr][0]; bn = tb_[tile_ovr][1]; wide = tile_ovr >= 6;
  }
  if (split_ovr && plain) splits = split_ovr;
  if (splits > 1 && ws_bytes < (size_t)splits * M * N * sizeof(float)) splits = 1;
  if (zk_prog_active()) {
    if (gen != 2 || splits > 1 || wide) return zk_prog_reject("gemm variant (generation / split-K / wide tile) not available in the layer program");
    return zk_prog_record_gemm((const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, ta, tb, bm, bn, e);
  }
  int kchunk = K;
  float* slabs = nullptr;
  if (splits > 1) {
    kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (K + kchunk - 1) / kchunk;
    if (splits > 1) slabs = (float*)workspace;
    else kchunk = K;
  }
  if (kchunk < 1) kchunk = 1;
  int rc;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
  if (gen == 2) rc = zk_gemm_dlds_dispatch(a, b, M, N, K, lda, ldb, ta, tb, bm, bn, splits, kchunk, slabs, e, sched_flags | (wide ? 0x100 : 0), stream);
  else if (bm == 128 && bn == 128) rc = launch_mfma<128, 128>(a, b, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  else if (bm == 128 && bn == 64) rc = launch_mfma<128, 64>(a, b, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  else if (bm == 64 && bn == 128) rc = launch_mfma<64, 128>(a, b, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  else rc = launch_mfma<64, 64>(a, b, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (rc) return rc;
  if (slabs != nullptr) {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, splits, M,
                       N, e);
    ZK_LAUNCH_CHECK();
  }
  return 0;
}

#ifdef ZK_EXPERIMENTS   // measured: no gain over the LayerNorm launches (profiles/r04_negative_results.txt): make EXPERIMENTS=1
// Forward linear with the residual + LayerNorm of the sub-layer folded into GEMM epilogues (GemmEpi, round 4): no
// LayerNorm launch.  C bf16 [M, ldc] = epilogue(A [M, K] x B [K, N]), ta = tb = 0, gen-2 tile kernels only.
//   stat_out  != NULL  PRODUCER: C is the un-normalised sum  residual + dropout(A B + bias)  and stat_out [M][N/64][2]
//                      receives {sum, M2} of each (row, 64-column group) of the stored bf16 values;
//   res_part  != NULL  the residual operand is itself an un-normalised sum: residual <- bf16(LN(residual)) with the
//                      statistics res_part [M][np][2] and res_gamma / res_beta [N];
//   in_c      != NULL  CONSUMER: A is an un-normalised sum with statistics in_part [M][np][2] (np = K / 64), B the weight
//                      with gamma folded in and in_c / bias the vectors zk_ln_fold made:
//                      C = act(rstd (A B - mu in_c) + bias), then dropout.
int zk_gemm_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const float* bias,
               const void* residual, int ldr, int act, float drop_p, const uint64_t* seed, uint32_t sid, float* stat_out,
               const float* in_part, const float* in_c, const float* res_part, const float* res_gamma,
               const float* res_beta, int np, float eps, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 0 && N >= 1 && K >= 1, "zk_gemm_ln: bad dims");
  ZK_CHECK_ARG(act == 0 || act == 1, "zk_gemm_ln: act=%d (0 none, 1 ReLU)", act);
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_gemm_ln: dropout needs a seed pointer");
  ZK_CHECK_ARG(N % 64 == 0 && ldc % 8 == 0 && (residual == nullptr || ldr % 8 == 0), "zk_gemm_ln: N=%d must be a multiple of 64, ldc / ldr of 8", N);
  ZK_CHECK_ARG(np >= 2 && np <= ZK_LN_MAXP && np % 2 == 0, "zk_gemm_ln: np=%d partials per row (even, 2..%d)", np, ZK_LN_MAXP);
  ZK_CHECK_ARG(stat_out == nullptr || N == np * 64, "zk_gemm_ln: a producer writes N/64 = np partials per row");
  ZK_CHECK_ARG((in_c == nullptr) == (in_part == nullptr), "zk_gemm_ln: in_part and in_c come together");
  ZK_CHECK_ARG(in_c == nullptr || (stat_out == nullptr && res_part == nullptr && residual == nullptr),
               "zk_gemm_ln: a GEMM is a consumer or a producer, not both");
  ZK_CHECK_ARG(in_c == nullptr || (K == np * 64 && bias != nullptr), "zk_gemm_ln: a consumer needs K = 64 np and the folded bias");
  ZK_CHECK_ARG(res_part == nullptr || (residual != nullptr && res_gamma != nullptr && res_beta != nullptr && N == np * 64),
               "zk_gemm_ln: a lazy residual needs the sum, gamma, beta and N = 64 np");
  const uintptr_t al = (uintptr_t)C | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)stat_out | (uintptr_t)in_part |
                       (uintptr_t)in_c | (uintptr_t)res_part | (uintptr_t)res_gamma | (uintptr_t)res_beta;
  ZK_CHECK_ARG((al & 15) == 0, "zk_gemm_ln: operands must be 16-byte aligned");
  ZK_CHECK_ARG(mfma_ok(A, B, M, N, K, lda, ldb, 0, 0), "zk_gemm_ln: shape/alignment not supported by the MFMA kernel "
               "(M=%d N=%d K=%d lda=%d ldb=%d)", M, N, K, lda, ldb);
  ZK_CHECK_ARG(!zk_prog_active(), "zk_gemm_ln cannot be part of a layer program");
  if (M == 0) return 0;
  GemmEpi e;
  e.C = C; e.ldc = ldc; e.out_f32 = 0; e.alpha = 1.f; e.bias = bias;
  e.res = (const bf16_t*)residual; e.ldr = ldr; e.act = act; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f;
  e.thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  e.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  e.seed = seed; e.sid = sid;
  e.ln_stat_out = stat_out; e.ln_in_part = in_part; e.ln_c = in_c;
  e.res_part = res_part; e.res_gamma = res_gamma; e.res_beta = res_beta;
  e.ln_np = np; e.res_after_drop = 1; e.ln_eps = eps; e.ln_invh = 1.f / (float)(np * 64);
  int bm, bn, splits;
  pick_config(M, N, K, 0, &bm, &bn, &splits);
  (void)splits;                 // never split: the epilogue needs the whole sum
  if (bm > 128 || bn > 128) { bm = 128; bn = 128; }
  return zk_gemm_dlds_ln_dispatch((const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, bm, bn, e, stream);
}

#endif  // ZK_EXPERIMENTS

// The tail of a post-LN sub-layer in ONE launch (func.py:321-324 residual_fn, func.py:289-303 layer_norm, the order of
// transformer.py:57-58):   s = residual + dropout(bf16(A B + bias));   y = LN(s) = gamma (s - mu) rstd + beta.
// The N/64 workgroups that hold one block of rows exchange the {sum, M2} of their 64 columns through `slots`
// (zk_gemm_add_ln_workspace(rows, N) bytes, zero-filled once, shared by every call on the stream) and each normalises its
// own columns -- the LayerNorm launch, its read of the product and of the residual, and a kernel boundary disappear.
// s_out / mean / rstd (what the backward reads) may be null.  `epoch` is a device word that zk_ln_epoch_bump advances
// once per forward pass; `site` (1..255) must differ between the calls of one pass that share `slots`.  *err (device
// int, may be null) becomes 1 if a workgroup gave up waiting (tens of milliseconds; never observed).
// Same values as zk_gemm followed by zk_add_ln_fwd up to the last bit of the statistics (Chan's combination of the
// per-64-column partials instead of two passes over the row).
size_t zk_gemm_add_ln_workspace(int rows, int N) { return (size_t)((rows + 127) / 128 * 128) * (size_t)(N / 64) * 16; }

__global__ void k_ln_epoch_bump(uint32_t* epoch) { if (threadIdx.x == 0) { const uint32_t v = *epoch + 1; *epoch = (v & 0xffffffu) ? v : 1; } }
int zk_ln_epoch_bump(uint32_t* epoch, hipStream_t stream) {
  ZK_CHECK_ARG(epoch != nullptr, "zk_ln_epoch_bump: null epoch word");
  hipLaunchKernelGGL(k_ln_epoch_bump, dim3(1), dim3(64), 0, stream, epoch);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_gemm_add_ln(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias,
                   const void* residual, int ldr, float drop_p, const uint64_t* seed, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, const uint32_t* epoch, uint32_t site, int* err, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 0 && N >= 64 && K >= 1, "zk_gemm_add_ln: bad dims");
  ZK_CHECK_ARG(N % 64 == 0 && N / 64 <= 16, "zk_gemm_add_ln: N=%d must be a multiple of 64 and <= 1024", N);
  ZK_CHECK_ARG(residual != nullptr && ldr % 8 == 0 && gamma != nullptr && beta != nullptr && y != nullptr,
               "zk_gemm_add_ln: residual (row stride a multiple of 8), gamma, beta and y are required");
  ZK_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "zk_gemm_add_ln: mean / rstd must both be given");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_gemm_add_ln: dropout needs a seed pointer");
  ZK_CHECK_ARG(slots != nullptr && epoch != nullptr && site >= 1 && site <= 255, "zk_gemm_add_ln: slots, epoch and a site in 1..255 are required");
  ZK_CHECK_ARG(slots_bytes >= zk_gemm_add_ln_workspace(M, N), "zk_gemm_add_ln: slots too small (zk_gemm_add_ln_workspace)");
  const uintptr_t al = (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)s_out |
                       (uintptr_t)y | (uintptr_t)slots;
  ZK_CHECK_ARG((al & 15) == 0, "zk_gemm_add_ln: operands must be 16-byte aligned");
  ZK_CHECK_ARG(mfma_ok(A, B, M, N, K, lda, ldb, 0, 0), "zk_gemm_add_ln: shape/alignment not supported by the MFMA kernel "
               "(M=%d N=%d K=%d lda=%d ldb=%d)", M, N, K, lda, ldb);
  ZK_CHECK_ARG(!zk_prog_active(), "zk_gemm_add_ln cannot be part of a layer program");
  if (M == 0) return 0;
  GemmEpi e;
  e.C = s_out; e.ldc = N; e.out_f32 = 0; e.alpha = 1.f; e.bias = bias;
  e.res = (const bf16_t*)residual; e.ldr = ldr; e.act = 0; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f;
  e.thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  e.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  e.seed = seed; e.sid = sid;
  e.ln_eps = eps; e.ln_invh = 1.f / (float)N;
  e.sy_slots = (unsigned long long*)slots; e.sy_epoch = epoch; e.sy_site = site;
  e.sy_gamma = gamma; e.sy_beta = beta; e.sy_y = (bf16_t*)y; e.sy_ldy = N;
  e.sy_mean = mean; e.sy_rstd = rstd; e.sy_err = err;
  // always 64 x 64 tiles, never split: the epilogue needs the whole sum of a 64-column group, and 128-row tiles were
  // measured slower for these products (profiles/r04_negative_results.txt item 6)
  return zk_gemm_dlds_sync_ln_dispatch((const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, 64, e, stream);
}

// The backward of that tail inside the dgrad launch that completes its input gradient (autodiff of transformer.py:57-58
// through func.py:289-303 / 321-324):   dout = bf16(dY W^T + residual)   [never stored]
//   g = dout gamma, xh = (s - mean) rstd, ds = rstd (g - mean_row(g) - xh mean_row(g xh)) -> dsum (bf16)
//   dy = bf16(ds) x dropout mask of (seed, sid) -> dy_out (null: no dropout, dy == ds)
//   partials [ceil(M/64)][3][N] = column sums over each 64-row block of {dout xh, dout, dy}: the dgamma / dbeta /
//   previous-bias partials zk_add_ln_bwd leaves for its reduction (nblk = ceil(M/64) here).
// dY [M, lda] (K contiguous), W [N, ldb] (K contiguous: the forward weight of a layer with N inputs).  Replaces zk_gemm
// (tb = 1, residual) followed by zk_add_ln_bwd(defer_reduce = 1); the exchange of the two row means between the N/64
// workgroups of a row block uses the slots / epoch / site scheme of zk_gemm_add_ln.
size_t zk_gemm_ln_bwd_partials(int rows, int N) { return (size_t)((rows + 63) / 64) * 3 * (size_t)N * sizeof(float); }

int zk_gemm_ln_bwd(const void* dY, const void* W, int M, int N, int K, int lda, int ldb, const void* residual, int ldr,
                   const void* s, const float* mean, const float* rstd, const float* gamma, float drop_p,
                   const uint64_t* seed, uint32_t sid, void* dsum, void* dy_out, float* partials, void* slots,
                   size_t slots_bytes, const uint32_t* epoch, uint32_t site, int* err, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 0 && N >= 64 && K >= 1, "zk_gemm_ln_bwd: bad dims");
  ZK_CHECK_ARG(N % 64 == 0 && N / 64 <= 16, "zk_gemm_ln_bwd: N=%d must be a multiple of 64 and <= 1024", N);
  ZK_CHECK_ARG(s != nullptr && mean != nullptr && rstd != nullptr && gamma != nullptr && dsum != nullptr && partials != nullptr,
               "zk_gemm_ln_bwd: s, mean, rstd, gamma, dsum and partials are required");
  ZK_CHECK_ARG(residual == nullptr || ldr % 8 == 0, "zk_gemm_ln_bwd: residual row stride must be a multiple of 8");
  ZK_CHECK_ARG(drop_p == 0.f || (seed != nullptr && dy_out != nullptr), "zk_gemm_ln_bwd: dropout needs a seed pointer and dy_out");
  ZK_CHECK_ARG(slots != nullptr && epoch != nullptr && site >= 1 && site <= 255, "zk_gemm_ln_bwd: slots, epoch and a site in 1..255 are required");
  ZK_CHECK_ARG(slots_bytes >= zk_gemm_add_ln_workspace(M, N), "zk_gemm_ln_bwd: slots too small (zk_gemm_add_ln_workspace)");
  const uintptr_t al = (uintptr_t)residual | (uintptr_t)s | (uintptr_t)gamma | (uintptr_t)dsum | (uintptr_t)dy_out |
                       (uintptr_t)slots | (uintptr_t)partials;
  ZK_CHECK_ARG((al & 15) == 0, "zk_gemm_ln_bwd: operands must be 16-byte aligned");
  ZK_CHECK_ARG(mfma_ok(dY, W, M, N, K, lda, ldb, 0, 1), "zk_gemm_ln_bwd: shape/alignment not supported by the MFMA kernel "
               "(M=%d N=%d K=%d lda=%d ldb=%d)", M, N, K, lda, ldb);
  ZK_CHECK_ARG(!zk_prog_active(), "zk_gemm_ln_bwd cannot be part of a layer program");
  if (M == 0) return 0;
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
  e.sy_dy = (bf16_t*)dy_out; e.sy_part = partials;
  return zk_gemm_dlds_sync_ln_bwd_dispatch((const bf16_t*)dY, (const bf16_t*)W, M, N, K, lda, ldb, e, stream);
}

// Split-K product left as its partial sums: parts[z] (fp32 [M, N] each, z < splits, part z at parts + z*M*N) =
// A[:, K_z] B[K_z, :] over the z-th K range; no epilogue and no reduction launch -- the consumer adds them in the
// fixed order z = 0, 1, .. (zk_ln_decode with parts / nparts / part_stride = M*N and the bias).  For products with few
// rows and a long K (the decode step's FFN output projection: 128 x 512 x 2048 is 16 tiles of 64 x 64 -- 16 of 256
// CUs streaming 512 KB each; with splits = 4 it is 64 workgroups of 128 KB).  Returns the number of parts written (the
// K ranges are multiples of 64, so it can be smaller than `splits`), or < 0 / a hipError_t as a negative-free error.
int zk_gemm_parts(const void* A, const void* B, float* parts, int M, int N, int K, int lda, int ldb, int ta, int tb,
                  int splits, int* nparts_out, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 0 && N >= 0 && K >= 1 && splits >= 1 && splits <= 64, "zk_gemm_parts: bad sizes");
  ZK_CHECK_ARG(parts != nullptr && nparts_out != nullptr, "zk_gemm_parts: parts and nparts_out are required");
  ZK_CHECK_ARG(mfma_ok(A, B, M, N, K, lda, ldb, ta, tb), "zk_gemm_parts: shape/alignment not supported by the MFMA kernel "
               "(M=%d N=%d K=%d lda=%d ldb=%d ta=%d tb=%d)", M, N, K, lda, ldb, ta, tb);
  ZK_CHECK_ARG(!zk_prog_active(), "zk_gemm_parts cannot be part of a layer program");
  int kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  const int n = (K + kchunk - 1) / kchunk;
  *nparts_out = n;
  if (M == 0 || N == 0) return 0;
  GemmEpi e;
  e.C = nullptr; e.ldc = N; e.out_f32 = 1; e.alpha = 1.f; e.bias = nullptr; e.res = nullptr; e.ldr = 0; e.act = 0;
  e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f; e.thr = 0; e.inv_keep = 1.f; e.seed = nullptr; e.sid = 0;
  // the slab path of the tile kernels writes part z only when there is more than one; a single part is a plain product
  if (n == 1) {
    e.C = parts;
    return zk_gemm_dlds_dispatch((const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, ta, tb, 64, 64, 1, K, nullptr, e, 0,
                                 stream);
  }
  return zk_gemm_dlds_dispatch((const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, ta, tb, 64, 64, n, kchunk, parts, e, 0,
                               stream);
}

#ifdef ZK_EXPERIMENTS   // measured: slower than the two launches (profiles/r04_negative_results.txt item 9)
// The two products of a feed-forward sub-layer on few rows in one launch (see k_ffn_pair, zk_gemm2.hip):
//   h = relu(x W1 + b1)  (bf16 [M, F], written: ldh = F),   parts[z] = h[:, K_z] W2[K_z, :]  (fp32 [M, H] at parts + z M H)
// for the z-th of *nparts_out <= splits K ranges (multiples of 64) -- exactly what zk_gemm(act = 1) followed by zk_gemm_parts
// leaves (same tile function, same K order).  x [M, ldx] (K1 = the model width contiguous), W1 [K1, ldw1], W2 [F, ldw2].
// counter: a zeroed device uint64 that only calls with the same (ceil(M/64), F) may share (the barrier's arrival count);
// *err (device int, may be null) is set if a workgroup gave up waiting.  Returns 2 without launching when the shape is not
// covered (more workgroups than are resident at once, or fewer phase-1 than phase-2 tiles): call the two entry points.
int zk_ffn_pair(const void* x, const void* W1, const float* b1, void* h, const void* W2, float* parts, int M, int F, int H, int K1,
                int ldx, int ldw1, int ldw2, int splits, int* nparts_out, void* counter, int* err, hipStream_t stream) {
  ZK_CHECK_ARG(M >= 1 && F >= 64 && H >= 64 && K1 >= 64 && splits >= 2 && splits <= 64, "zk_ffn_pair: bad sizes");
  ZK_CHECK_ARG(h != nullptr && parts != nullptr && nparts_out != nullptr && counter != nullptr, "zk_ffn_pair: h, parts, nparts_out and counter are required");
  if (F % 64 != 0 || H % 64 != 0 || K1 % 64 != 0) return 2;
  if (!mfma_ok(x, W1, M, F, K1, ldx, ldw1, 0, 0) || !mfma_ok(h, W2, M, H, F, F, ldw2, 0, 0) || zk_prog_active()) return 2;
  if ((((uintptr_t)b1 | (uintptr_t)h | (uintptr_t)parts | (uintptr_t)counter) & 15) != 0) return 2;
  const int kchunk = ((F + splits - 1) / splits + BK - 1) / BK * BK;
  const int n = (F + kchunk - 1) / kchunk;
  const int tiles_m = (M + 63) / 64, T1 = tiles_m * (F / 64), T2 = tiles_m * (H / 64) * n;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 2;
  if (n < 2 || T2 > T1 || T1 > 2 * cus) return 2;               // every workgroup must be resident while it waits
  *nparts_out = n;
  GemmEpi e1, e2;
  e1.C = h; e1.ldc = F; e1.out_f32 = 0; e1.alpha = 1.f; e1.bias = b1; e1.res = nullptr; e1.ldr = 0; e1.act = 1;
  e1.aux = nullptr; e1.ldaux = 0; e1.aux_scale = 1.f; e1.thr = 0; e1.inv_keep = 1.f; e1.seed = nullptr; e1.sid = 0;
  e2.C = nullptr; e2.ldc = H; e2.out_f32 = 1; e2.alpha = 1.f; e2.bias = nullptr; e2.res = nullptr; e2.ldr = 0; e2.act = 0;
  e2.aux = nullptr; e2.ldaux = 0; e2.aux_scale = 1.f; e2.thr = 0; e2.inv_keep = 1.f; e2.seed = nullptr; e2.sid = 0;
  return zk_ffn_pair_launch((const bf16_t*)x, (const bf16_t*)W1, (bf16_t*)h, (const bf16_t*)W2, parts, M, F, H, K1, ldx, ldw1, ldw2,
                            kchunk, n, e1, e2, (unsigned long long*)counter, err, stream);
}

#endif  // ZK_EXPERIMENTS

}  // extern "C"
