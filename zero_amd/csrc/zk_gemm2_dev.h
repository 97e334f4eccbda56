// zk_gemm2_dev.h -- device side of the second-generation bf16 MFMA GEMM (LDS-DMA ring + hardware-transpose reads):
// everything up to `gemm_tile`, one BMxBN output tile over raw pointers.  Included by zk_gemm2.hip (the launch-per-op
// kernels) and zk_layer.hip (the per-XCD layer program, which calls the same tile function phase after phase, so
// the two paths give bit-identical results).  See zk_gemm2.hip for the design notes.
#pragma once
#include "zk_gemm.h"

typedef short v4s_t __attribute__((ext_vector_type(4)));

static __device__ __attribute__((aligned(16))) uint4 zk_zero_page[4];   // source of out-of-range pieces (one copy per translation unit)
// In-kernel timeline (diagnostic build only: make TRACE=1, scripts/trace_gemm.py): wave 0 of workgroup 301
// stamps s_memtime at the phase boundaries of the K loop (ZK_T, one row per K step) and of the kernel (ZK_E).
#ifdef ZK_GEMM_TRACE
static __device__ unsigned long long zk_trace_buf[8192];
#define ZK_E(slot) do { if (blockIdx.x == 301 && threadIdx.x == 0) zk_trace_buf[4096 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define ZK_T(slot) do { if (tr_on) zk_trace_buf[tr_i * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define ZK_E(slot) do { } while (0)
#define ZK_T(slot) do { } while (0)
#endif

// LDS-DMA issued from inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every
// ds_read that may alias an LDS-DMA it knows about, which would drain the ring each K step.  The
// asm form is invisible to that pass; completion is tracked by the counted vmcnt waits below.
// LDS destination = M0 (wave-uniform byte address) + lane*16.  Nothing else in these kernels reads
// M0 (gfx9+ DS instructions do not), so it is written and left.
// FRESH: the source was written by another CU earlier in this launch (layer program): bypass the vector L1 (sc1)
template <bool FRESH = false>
__device__ __forceinline__ void glds16(const bf16_t* gsrc, uint32_t lds_byte_addr) {
  if (FRESH)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1"
                 :
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
  else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// The same with the address as a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset (saddr form): a K loop
// whose pieces advance by a uniform stride keeps ONE offset register per piece and advances the base with scalar adds --
// no vector instruction per piece (the pointer form costs a 64-bit add and, with the K-tail select, ~7 VALU per piece).
template <bool FRESH = false>
__device__ __forceinline__ void glds16s(uint32_t voff, const void* sbase, uint32_t lds_byte_addr) {
  if (FRESH)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) unsigned char*)p);
}

// chunk-position swizzles (16-byte chunks)
__device__ __forceinline__ int swz_direct(int row) { return (row >> 1) & 7; }            // 128-B rows
template <int R>
__device__ __forceinline__ int swz_trans(int k) { return R >= 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2); }

// Per-lane LDS-DMA plan of one operand: for each of the wave's NINSTR pieces the running source
// pointer (advanced by one K tile after every issue) and the k index inside the tile that decides
// the K-tail predicate.  Tile rows / column chunks outside the matrix are CLAMPED to a valid
// row / chunk 0: what they bring in only feeds output rows / columns the epilogue never stores.
// Only a K tile that crosses kend needs zero fill (both operands), done by the TAIL variant.
// (round 5: the plan holds 32-bit BYTE OFFSETS against a wave-uniform tile base instead of running 64-bit pointers: the
// K loop advances the base with scalar adds and a piece costs no vector instruction -- glds16s.  An operand must span less
// than 4 GB from its first element to the last byte a tile touches: checked by the launchers' callers' shapes, every
// matrix of the path is far below.)
template <int R, int NW = 4>
struct DmaPlan {
  static constexpr int PER_WAVE = R * 8 / NW;     // 16-byte chunks per wave
  static constexpr int NINSTR = PER_WAVE / 64;
  uint32_t off[NINSTR];                           // byte offset of the lane's 16 bytes against the K tile's base
  int kofs[NINSTR];
};

// K tile t of an operand starts at  src + kbeg + 64 t  (k contiguous)  or  src + (kbeg + 64 t) ld  (TRANS: k-major rows)
template <int R, bool TRANS, int NW>
__device__ __forceinline__ void dma_plan(DmaPlan<R, NW>& pl, int ld, int row0, int rows_total, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < DmaPlan<R, NW>::NINSTR; ++j) {
    const int P = wave * DmaPlan<R, NW>::PER_WAVE + j * 64 + lane;
    if (!TRANS) {
      const int row = P >> 3, pos = P & 7;
      const int c = pos ^ swz_direct(row);
      const int grow = min(row0 + row, rows_total - 1);
      pl.kofs[j] = c * 8;
      pl.off[j] = (uint32_t)(((size_t)grow * ld + c * 8) * 2);
    } else {
      constexpr int CPR = R / 8;             // chunks per k row
      const int k = P / CPR, pos = P % CPR;
      const int c = pos ^ swz_trans<R>(k);
      int grow = row0 + c * 8;
      grow = grow < rows_total ? grow : 0;
      pl.kofs[j] = k;
      pl.off[j] = (uint32_t)(((size_t)k * ld + grow) * 2);
    }
  }
}

// issue the LDS-DMA of K tile `t` of one operand (its base: `tile_base`, wave-uniform) into `stage`.
// TAIL: the tile crosses (or lies past) kend -- pointer form with a per-lane select against the zero page.
template <int R, bool TRANS, bool TAIL, int NW, bool FRESH = false>
__device__ __forceinline__ void dma_tile(const DmaPlan<R, NW>& pl, const bf16_t* tile_base, int t, int klen, uint32_t stage_addr,
                                         int wave) {
#pragma unroll
  for (int j = 0; j < DmaPlan<R, NW>::NINSTR; ++j) {
    const uint32_t dst = stage_addr + (uint32_t)(wave * DmaPlan<R, NW>::PER_WAVE + j * 64) * 16u;
    if (TAIL) {
      const bf16_t* g = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(tile_base) + pl.off[j]);
      g = (t * 64 + pl.kofs[j] < klen) ? g : reinterpret_cast<const bf16_t*>(zk_zero_page);
      glds16<FRESH>(g, dst);
    } else {
      glds16s<FRESH>(pl.off[j], tile_base, dst);
    }
  }
}

// MFMA 32x32x16 operand fragment of rows r0 + (lane&31), k = kk*16 + (lane>>5)*8 .. +7
template <int R, bool TRANS>
__device__ __forceinline__ bf16x8_t load_frag(const bf16_t* stage, int r0, int kk, int lane) {
  if (!TRANS) {
    const int row = r0 + (lane & 31);
    const int c = kk * 2 + (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(stage + row * 64 + ((c ^ swz_direct(row)) << 3));
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    // ds_read_b64_tr_b16: in each 16-lane group, lane p supplies 4 contiguous elements of k-row p/4
    // at column (p%4)*4 and receives column p of the 4(k) x 16 block
    const int p = lane & 15;
    const int rr = r0 + ((lane >> 4) & 1) * 16 + (p & 3) * 4;
    const int kb = kk * 16 + (lane >> 5) * 8 + (p >> 2);
    const int chunk = rr >> 3, within = rr & 7;
    v4s_t lo, hi;
    {
      const int k = kb;
      const bf16_t* a = stage + k * R + ((chunk ^ swz_trans<R>(k)) << 3) + within;
      lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)a);
    }
    {
      const int k = kb + 4;
      const bf16_t* a = stage + k * R + ((chunk ^ swz_trans<R>(k)) << 3) + within;
      hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)a);
    }
    typedef short v8s_t __attribute__((ext_vector_type(8)));
    const v8s_t both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
  }
}

extern int g_tune[24];   // A/B switches (zk_tune, zk_elem.hip)

struct EpiVec {
  int vec_ok;   // 16-byte vector epilogue allowed (alignment checked on the host)
};

template <int BM, int BN, int NS>
struct DldsCfg {
  static constexpr int STAGE = (BM + BN) * 64;                     // bf16 elements per ring stage
  static constexpr int CLD = BN + 4;                               // fp32 epilogue tile row stride
  static constexpr int RING_BYTES = NS * STAGE * 2, EPI_BYTES = BM * CLD * 4;
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
};

// K-segmented GEMM: C = sum_s A_s B_s with every A_s / B_s its own matrix (same shape and leading dimension), e.g.
// the gradient of the encoder output, which every decoder layer's cross-attention K and V projection feeds
// (12 segments of K = 512 in one launch instead of 12 dependent GEMMs accumulating in place).
#define ZK_KSEG_MAX 16
struct KSegDesc {
  const bf16_t* A[ZK_KSEG_MAX];
  const bf16_t* B[ZK_KSEG_MAX];
  int nseg, tps;               // segments, 64-deep K tiles per segment
};

// NW compute waves per workgroup: 4 (2 x 2 over the tile), 2 (2 x 1: each wave a BM/2 x BN slab) or 8 (4 x 2: the
// 256x128 macro tile -- 25 % fewer L1->LDS bytes and DMA issues per MFMA than two 128x128 tiles).
// PW > 0: PW extra PRODUCER waves (wave index >= NW) issue every LDS-DMA of the workgroup and the NW compute
// waves issue none.  Reason (profiles/r01_gemm_kloop_trace.txt): a global_load_lds stalls its wave ~100 cycles
// while the CU's texture-address path is busy, and a stalled wave cannot issue its MFMAs, so with PW = 0 the
// DMA-issue time and the MFMA time of a K step add up inside a workgroup; with producer waves they overlap.
// K loop of one BMxBN tile over k in [kbeg, kend); leaves the fp32 tile in LDS (sC[BM][CLD], smem reused)
// behind a workgroup barrier, ready for a row-wise epilogue.
// FRESH: the A operand (activations) and the epilogue's residual / mask tensors were written by other CUs earlier in
// the same launch (layer program) -> L1-bypassing loads; B (weights) always takes the ordinary path.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// HOOK: called once by the waves that run the epilogue's row loop (every wave without producer waves, the compute waves
// with them) right after the first ring stages have been requested -- work that only needs global memory (the row
// statistics of a lazy LayerNorm) rides behind the ring fill instead of adding a round trip of its own.
template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0, bool KSEG = false, bool FRESH = false,
          typename HOOK = NoHook>
__device__ __forceinline__ void gemm_tile_to_lds(unsigned char* smem, const bf16_t* __restrict__ A,
                                                 const bf16_t* __restrict__ B, int M, int N, int lda, int ldb,
                                                 int kbeg, int kend, int m0, int n0, const KSegDesc* ks = nullptr,
                                                 float* __restrict__ colsum = nullptr, HOOK hook = HOOK()) {
  constexpr int NWM = NW == 8 ? 4 : 2, NWN = NW / NWM;       // wave grid over the tile: 2x2, 2x1 or 4x2
  constexpr int WTM = BM / NWM, WTN = BN / NWN, TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE = DldsCfg<BM, BN, NS>::STAGE;
  constexpr int NDW = PW ? PW : NW;                          // waves that issue the DMA
  constexpr int PER_STAGE = (BM * 8 / NDW + BN * 8 / NDW) / 64;  // DMA instructions per issuing wave per stage
  constexpr int CLD = DldsCfg<BM, BN, NS>::CLD;
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PW > 0 && wave >= NW;
  const int dwave = PW ? wave - NW : wave;                   // index among the issuing waves
  const int wm = wave / NWN, wn = wave % NWN;
  const int nk = (kend - kbeg + 63) >> 6;

  ZK_E(0);
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  DmaPlan<BM, NDW> planA;
  DmaPlan<BN, NDW> planB;
  if (PW == 0 || producer) {                       // (the offsets do not depend on the K segment: one plan per tile)
    dma_plan<BM, TA, NDW>(planA, lda, m0, M, dwave, lane);
    dma_plan<BN, !TB, NDW>(planB, ldb, n0, N, dwave, lane);
  }
  [[maybe_unused]] int seg_left = 0, seg_id = 0;   // KSEG: K tiles left in the current segment, next segment
  const int klen = kend - kbeg;
  const size_t stepA = TA ? (size_t)64 * lda : (size_t)64;
  const size_t stepB = !TB ? (size_t)64 * ldb : (size_t)64;
  // wave-uniform bases of the NEXT K tile to issue (tiles are issued in order), advanced by scalar adds
  const bf16_t* baseA = KSEG ? nullptr : A + (TA ? (size_t)kbeg * lda : (size_t)kbeg);
  const bf16_t* baseB = KSEG ? nullptr : B + (!TB ? (size_t)kbeg * ldb : (size_t)kbeg);
  const uint32_t ring_addr = lds_addr(ring);
  auto issue = [&](int t) {
    if (KSEG) {                                   // tiles are issued in order: new bases at every segment start
      if (seg_left == 0) {
        if (seg_id < ks->nseg) {
          baseA = ks->A[seg_id];
          baseB = ks->B[seg_id];
          ++seg_id;
          seg_left = ks->tps;
        } else {
          seg_left = 0x40000000;                  // past the last segment: only all-zero tail pieces follow
        }
      }
      --seg_left;
    }
    const uint32_t st = ring_addr + (uint32_t)((t % NS) * STAGE * 2);
    if (t * 64 + 64 <= klen) {
      dma_tile<BM, TA, false, NDW, FRESH>(planA, baseA, t, klen, st, dwave);
      dma_tile<BN, !TB, false, NDW>(planB, baseB, t, klen, st + BM * 128, dwave);
    } else {
      dma_tile<BM, TA, true, NDW, FRESH>(planA, baseA, t, klen, st, dwave);
      dma_tile<BN, !TB, true, NDW>(planB, baseB, t, klen, st + BM * 128, dwave);
    }
    baseA += stepA;
    baseB += stepB;
  };
  // MFMAs of K tile kt out of its ring stage
  auto compute = [&](int kt) {
    const bf16_t* sA = ring + (kt % NS) * STAGE;
    const bf16_t* sB = sA + BM * 64;
    // fragments of k-slice kk+1 are read while the MFMAs of slice kk run (two register sets)
    bf16x8_t af[2][TM], bfr[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = load_frag<BM, TA>(sA, wm * WTM + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = load_frag<BN, !TB>(sB, wn * WTN + j * 32, 0, lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = load_frag<BM, TA>(sA, wm * WTM + i * 32, kk + 1, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[(kk + 1) & 1][j] = load_frag<BN, !TB>(sB, wn * WTN + j * 32, kk + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
    }
  };
  [[maybe_unused]] const bool tr_on = (blockIdx.x == 301) && (tid == 0);
  if (PW > 0) {
    // invariant at the barrier of step kt: tile kt has landed (the producers waited for it) and every compute
    // wave is done with tile kt-1, whose stage the producers refill next.  Both roles pass nk barriers.
    if (producer) {
      // colsum != nullptr (B stored [k][n], i.e. !TB: the dY operand of a weight gradient): the producer waves, idle
      // between two DMA issues, also sum the columns of every B tile they brought in -- the bias gradient
      // sum_k dY[k][n] (func.py:16, 58-60) -- so that no separate pass re-reads dY (k_colsum_grouped: 430 MB per step).
      // Thread p of the producers owns column n0 + p; tile kt has landed for everybody at barrier kt and its stage is
      // refilled by this wave itself only after these reads.
      float cs = 0.f;
      const int pcol = dwave * 64 + lane;
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) issue(s);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_STAGE) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(kt + NS - 1);
        if (!TB && colsum != nullptr && pcol < BN) {
          const bf16_t* sBk = ring + (kt % NS) * STAGE + BM * 64;
          const int chunk = pcol >> 3, within = pcol & 7;
#pragma unroll 8
          for (int k = 0; k < 64; ++k) cs += bf2f(sBk[k * BN + ((chunk ^ swz_trans<BN>(k)) << 3) + within]);
        }
      }
      if (!TB && colsum != nullptr && pcol < BN && n0 + pcol < N) colsum[n0 + pcol] = cs;
    } else {
      ZK_E(1);
      hook();
      for (int kt = 0; kt < nk; ++kt) {
        [[maybe_unused]] const int tr_i = kt;
        ZK_T(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ZK_T(2);
        compute(kt);
        __builtin_amdgcn_sched_barrier(0);
        ZK_T(4);
      }
    }
  } else {
    // prologue: tiles 0 .. NS-2 (tiles past the end are all-zero pieces: keeps the DMA count uniform)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    __builtin_amdgcn_sched_barrier(0);
    ZK_E(1);
    hook();
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
      [[maybe_unused]] const int tr_i = kt;
      ZK_T(0);
      // tile kt has landed once at most NS-2 later tiles of this wave are still in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_STAGE) : "memory");
      ZK_T(1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      ZK_T(2);
      issue(kt + NS - 1);                         // refill the stage everybody finished reading
      __builtin_amdgcn_sched_barrier(0);
      ZK_T(3);
      compute(kt);
      __builtin_amdgcn_sched_barrier(0);
      ZK_T(4);
    }
  }
  // ---- epilogue through LDS
  ZK_E(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (all-zero) pieces have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  float* sC = reinterpret_cast<float*>(smem);
  if (!producer) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * WTN + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          sC[row * CLD + col] = acc[i][j][r];
        }
      }
  }
  __syncthreads();
  ZK_E(3);
}

// one BMxBN output tile over k in [kbeg, kend); slab != null: write the fp32 partial tile there
// LN: the lazy-LayerNorm epilogue forms of GemmEpi (zk_gemm_ln) are compiled in (separate instantiations: the ordinary
// kernels pay nothing for them).  The row statistics (mu, rstd) of the tile's BM rows -- combined from the per-64-column
// partials -- are computed once, by the first BM threads, behind the ring fill, and wait in 2 x BM float2 of LDS.
template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0, bool KSEG = false, bool FRESH = false,
          int LN = 0>      // LN: 0 none, 1 consumer form (ln_c), 2 producer forms (ln_stat_out, res_part, res_after_drop),
                           //     3 residual + LayerNorm in the launch (sy_*: the row's tiles exchange their statistics)
__device__ __forceinline__ void gemm_tile(unsigned char* smem, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, int M, int N, int lda, int ldb, int kbeg,
                                          int kend, int m0, int n0, float* __restrict__ slab, const GemmEpi& e,
                                          int vec_ok, const KSegDesc* ks = nullptr, float* __restrict__ colsum = nullptr) {
  constexpr int CLD = DldsCfg<BM, BN, NS>::CLD;
  constexpr int NT = (NW + PW) * 64;
  const int tid = threadIdx.x;
  constexpr int CPRW = BN / 8;
  // Fast path (interior tile, 16-byte epilogue, no split-K slab): every thread's chunks sit in the same 8
  // columns (NT is a multiple of CPRW), so the loop is fully unrolled with all LDS reads first, then all
  // residual / mask loads, then the arithmetic and the stores -- the latencies overlap instead of adding up
  // once per chunk (the rolled loop below cost ~1200 cycles per chunk, profiles/r01_gemm_kloop_trace.txt).
  static_assert(NT % CPRW == 0, "a thread's chunks must share their columns");
  constexpr int CH = BM * CPRW, ITER = (CH + NT - 1) / NT;
  constexpr int RSTEP = NT / CPRW;
  // (LN = 3: zk_gemm_add_ln has checked the 16-byte path and N % 64 == 0; rows past M are clamped for the loads)
  const bool fast = LN >= 3 ? true : (slab == nullptr && vec_ok && m0 + BM <= M && n0 + BN <= N);
  const int cc = (tid % CPRW) * 8, gn = n0 + cc, row0 = tid / CPRW;
  // The epilogue's inputs (residual, ReLU mask, bias, dropout seed) do not depend on the product: on the small tiles,
  // where a launch is a chain of a few memory round trips, they are requested BEFORE the K loop instead of after it
  // (one round trip less per launch; the registers are few: ITER <= 2).
  constexpr bool PRE = ITER <= 2;
  uint4 rres[ITER], raux[ITER];
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float lcv[8], lgv[8], lbv[8];            // LN: this thread's 8 columns of ln_c / res_gamma / res_beta
  [[maybe_unused]] uint32_t sy_ep = 0;
  uint64_t seed = 0;
  auto ld8 = [&](const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  };
  auto load_epi = [&]() {
    if constexpr (LN == 1) { if (e.ln_c) ld8(e.ln_c + gn, lcv); }
    if constexpr (LN == 2) { if (e.res_part) { ld8(e.res_gamma + gn, lgv); ld8(e.res_beta + gn, lbv); } }
    if constexpr (LN == 3) { ld8(e.sy_gamma + gn, lgv); ld8(e.sy_beta + gn, lbv); sy_ep = *e.sy_epoch; }
    if constexpr (LN == 4) {
      ld8(e.sy_gamma + gn, lgv);
      sy_ep = *e.sy_epoch;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int gm = min(m0 + min(row0 + it * RSTEP, BM - 1), M - 1);
        raux[it] = zk_ld16<FRESH>(e.sy_s + (size_t)gm * e.sy_lds + gn);
        lbv[2 * it] = e.sy_mean_in[gm];
        lbv[2 * it + 1] = e.sy_rstd_in[gm];
      }
    }
    if (e.res) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        int gm = m0 + min(row0 + it * RSTEP, BM - 1);
        if constexpr (LN >= 3) gm = min(gm, M - 1);
        rres[it] = zk_ld16<FRESH>(e.res + (size_t)gm * e.ldr + gn);
      }
    }
    if (e.act == 2) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int gm = m0 + min(row0 + it * RSTEP, BM - 1);
        raux[it] = zk_ld16<FRESH>(e.aux + (size_t)gm * e.ldaux + gn);
      }
    }
    if (e.bias) {
      const float4 a = *reinterpret_cast<const float4*>(e.bias + gn);
      const float4 b = *reinterpret_cast<const float4*>(e.bias + gn + 4);
      bv[0] = a.x; bv[1] = a.y; bv[2] = a.z; bv[3] = a.w; bv[4] = b.x; bv[5] = b.y; bv[6] = b.z; bv[7] = b.w;
    }
  };
  if (PRE && !FRESH) {
    seed = e.thr ? *e.seed : 0;
    if (fast) load_epi();
  }
  __shared__ float2 s_lnstat[LN ? BM : 1];     // (mu, rstd) of the tile's rows: of the A operand (LN = 1) / of the residual (LN = 2)
  // LN >= 3: a wait of the in-launch exchange ran out of its spin budget somewhere in this workgroup.  Set by the lanes that
  // watch the peers, read by everybody behind the barrier that follows: the remaining waits of the workgroup return at once
  // (one budget per launch instead of one per row and peer) and the rows leave as NaN, so that the step's own non-finite
  // guard (k_adam / k_norm_final2: flag, sticky count, skipped update) trips on THIS step, not at the next host read.
  __shared__ int s_giveup;
  if constexpr (LN >= 3) { if (tid == 0) s_giveup = 0; }   // (the K loop's barriers order this before any use)
  if constexpr (LN != 0) {
    auto ln_rows = [&]() {
      const float* part = LN == 1 ? e.ln_in_part : e.res_part;
      if (tid < BM && part != nullptr) {
        const size_t gr = (size_t)min(m0 + tid, M - 1);
        float mu, rs;
        zk_ln_row_stats(part, e.ln_np, e.ln_invh, e.ln_eps, gr, mu, rs);
        s_lnstat[tid] = make_float2(mu, rs);
      }
    };
    gemm_tile_to_lds<BM, BN, NS, TA, TB, NW, PW, KSEG, FRESH>(smem, A, B, M, N, lda, ldb, kbeg, kend, m0, n0, ks, colsum, ln_rows);
  } else {
    gemm_tile_to_lds<BM, BN, NS, TA, TB, NW, PW, KSEG, FRESH>(smem, A, B, M, N, lda, ldb, kbeg, kend, m0, n0, ks, colsum);
  }
  const float* sC = reinterpret_cast<const float*>(smem);
  if (!(PRE && !FRESH)) seed = e.thr ? *e.seed : 0;
  if constexpr (LN >= 3) {
    // residual + LayerNorm inside this launch (GemmEpi.sy_*): s = res + dropout(bf16(acc + bias)) exactly as the two
    // launches form it (the product rounded to bf16 where it used to be stored, the same dropout index), the statistics
    // of the STORED sum as {sum, M2} of this tile's 64 columns, published; then every thread fetches one peer's partial
    // of its row (its own tile's comes back the same way), the eight lanes of a row combine them (Chan), and the row's 64
    // columns are normalised from the registers they are still in.
    static_assert(BN == 64 && CH % NT == 0, "the in-launch LayerNorm is instantiated for 64-column tiles");
    if (!(PRE && !FRESH)) load_epi();
    const int np = N >> 6, tn = n0 >> 6, j8 = tid & 7;
    const uint32_t tag = (sy_ep << 8) | e.sy_site;
    // sy_local (the row block's workgroups share an XCD: grid a multiple of 64 under the XCD remap): their L2 is the point
    // of coherence -- ordinary stores (the L1 writes through), and polls by an atomic OR of zero, which executes in the L2
    // whatever the L1 holds.  [Group-scope (sc0) loads may hit the CU's own L1, which still holds the line an earlier
    // poll fetched: every launch ran into the spin budget.  Ordinary loads behind an L1 invalidation (buffer_inv sc1) are
    // correct but the invalidations cost the CU's other workgroup its K loop: 61.6 us a launch.]  Otherwise agent-scope
    // (sc1) stores and loads: through memory.
    const bool local = e.sy_local != 0;
    auto slot_st = [&](unsigned long long* p, unsigned long long v) {
      if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // one slot of a peer, valid once both halves carry this launch's tag.  plain_first (local): the line has not been
    // touched by this CU in this launch, an ordinary load misses the L1 and is served by the L2
    bool gave_up = false;
    auto slot_wait = [&](unsigned long long* p, unsigned long long& a, unsigned long long& b, bool plain_first) {
      int spins = 0;
      if (gave_up) { a = b = 0; return; }
      if (local && plain_first) {
        a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) return;
      }
      for (;;) {
        if (local) {
          // (written out: the compiler turns an atomic OR of zero into a group-scope LOAD, which may hit the L1)
          const unsigned long long z = 0;
          asm volatile("global_atomic_or_x2 %0, %2, %3, off sc0\n\tglobal_atomic_or_x2 %1, %2, %3, off offset:8 sc0\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(a), "=&v"(b) : "v"(p), "v"(z) : "memory");
        } else {
          a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) return;
        if (++spins > (1 << 15)) {             // tens of milliseconds: a peer that never came (must not happen)
          if (e.sy_err != nullptr) __hip_atomic_store(e.sy_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gave_up = true;
          a = b = 0;
          return;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    };
    if constexpr (LN == 3) {
    const int rlim = e.sy_rows > 0 ? e.sy_rows : BM;
    float r8[ITER][8], own1[ITER], own2[ITER];
    uint4 pk[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = row0 + it * RSTEP, gm = m0 + row;
      const bool ok = gm < M && row < rlim;
      float v[8];
      {
        const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
        const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * e.alpha + bv[j];
      unpack8(pack8(v), v);
      if (e.thr) {
        float dm[8];
        zk_drop_scale8(seed, e.sid, (uint64_t)gm * N + gn, e.thr, e.inv_keep, dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= dm[j];
      }
      if (e.res) {
        float rv[8];
        unpack8(rres[it], rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[j] + v[j];
      }
      pk[it] = pack8(v);
      // training (e.C given): statistics of the STORED bf16 sum, as the backward will read it.  Inference (round 6): nothing
      // is stored and the sum stays fp32, as zk_add_ln_fwd without sum_out keeps it (zk_ln_dev.h)
      if (e.C != nullptr) unpack8(pk[it], r8[it]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[it][j] = v[j];
      }
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += r8[it][j];
      s1 = zk_sum8(s1);
      const float mt = s1 * (1.f / 64.f);
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = r8[it][j] - mt; s2 += d * d; }
      s2 = zk_sum8(s2);
      own1[it] = s1; own2[it] = s2;
      if (ok && j8 == 0 && !(e.sy_fault && tn == 1)) {       // published first: the peers are waiting for this, nobody for the sum below
        unsigned long long* p = e.sy_slots + ((size_t)gm * np + tn) * 2;
        slot_st(p, ((unsigned long long)tag << 32) | __float_as_uint(s1));
        slot_st(p + 1, ((unsigned long long)tag << 32) | __float_as_uint(s2));
      }
    }
    ZK_E(5);
    if (e.C != nullptr) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int gm = m0 + row0 + it * RSTEP;
        if (gm < M && row0 + it * RSTEP < rlim) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)gm * e.ldc + gn) = pk[it];
      }
    }
    // a few lanes watch ONE row's slots (the block's last row, one lane per peer) until every peer has published; only
    // then does every thread fetch its own slots.  [All threads polling from the start: 30.7 us a launch against
    // 13.9 + 7.0 for the two launches; a few lanes of every wave: 23.4 us; this: 20.0 us through memory.]
    {
      const int grep = min(m0 + rlim, M) - 1;
      if (tid < np && tid != tn) {
        unsigned long long a, b;
        slot_wait(e.sy_slots + ((size_t)grep * np + tid) * 2, a, b, false);
        if (gave_up) __hip_atomic_store(&s_giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      gave_up = __hip_atomic_load(&s_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
    }
    ZK_E(6);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = row0 + it * RSTEP, gm = m0 + row;
      const bool ok = gm < M && row < rlim;
      float ps[2] = {0.f, 0.f}, pm[2] = {0.f, 0.f};
      bool have[2] = {false, false};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = j8 + 8 * u;
        if (ok && q < np) {
          if (q == tn) { ps[u] = own1[it]; pm[u] = own2[it]; }        // this tile's own partial never leaves the registers
          else {
            unsigned long long a, b;
            slot_wait(e.sy_slots + ((size_t)gm * np + q) * 2, a, b, true);   // (a peer's other rows land within its store burst)
            ps[u] = __uint_as_float((uint32_t)a);
            pm[u] = __uint_as_float((uint32_t)b);
          }
          have[u] = true;
        }
      }
      const float mu = zk_sum8(ps[0] + ps[1]) * e.ln_invh;
      const float d0 = ps[0] * (1.f / 64.f) - mu, d1 = ps[1] * (1.f / 64.f) - mu;
      const float m2 = zk_sum8((have[0] ? pm[0] + 64.f * d0 * d0 : 0.f) + (have[1] ? pm[1] + 64.f * d1 * d1 : 0.f));
      const float rs = rsqrtf(m2 * e.ln_invh + e.ln_eps);
      ZK_E(7);
      if (ok) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = gave_up ? __builtin_nanf("") : lgv[j] * (r8[it][j] - mu) * rs + lbv[j];
        *reinterpret_cast<uint4*>(e.sy_y + (size_t)gm * e.sy_ldy + gn) = pack8(o);
        if (tn == 0 && j8 == 0 && e.sy_mean != nullptr) { e.sy_mean[gm] = mu; e.sy_rstd[gm] = rs; }
      }
    }
    }
    if constexpr (LN == 4) {
      // backward of the residual + LayerNorm whose input gradient this product completes (GemmEpi.sy_s ..): see zk_gemm.h
      static_assert(ITER <= 4, "lbv holds (mu, rstd) of at most four rows");
      // sentence-aligned row tiles (zk_attn_bwd_ln): only the first sy_rows rows are this tile's, one partial row per tile
      const int rlim = e.sy_rows > 0 ? e.sy_rows : BM, pidx = e.sy_rows > 0 ? m0 / e.sy_rows : m0 / BM;
      float* red = reinterpret_cast<float*>(smem);         // [3][BM][64] column-partial staging (the fp32 tile is consumed first)
      float d[ITER][8], xh[ITER][8], g[ITER][8], own1[ITER], own2[ITER];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = row0 + it * RSTEP, gm = m0 + row;
        const bool ok = gm < M && row < rlim;
        float v[8];
        {
          const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
          const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= e.alpha;
        if (e.res) {
          float rv[8];
          unpack8(rres[it], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        unpack8(pack8(v), d[it]);                           // the gradient as the dgrad launch used to store it
        float sv[8];
        unpack8(raux[it], sv);
        const float mu = lbv[2 * it], rs = lbv[2 * it + 1];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (!ok) d[it][j] = 0.f;
          xh[it][j] = (sv[j] - mu) * rs;
          g[it][j] = d[it][j] * lgv[j];
          sg += g[it][j];
          sgx += g[it][j] * xh[it][j];
        }
        sg = zk_sum8(sg);
        sgx = zk_sum8(sgx);
        own1[it] = sg; own2[it] = sgx;
        if (ok && j8 == 0 && !(e.sy_fault && tn == 1)) {
          unsigned long long* p = e.sy_slots + ((size_t)gm * np + tn) * 2;
          slot_st(p, ((unsigned long long)tag << 32) | __float_as_uint(sg));
          slot_st(p + 1, ((unsigned long long)tag << 32) | __float_as_uint(sgx));
        }
      }
      // the two column sums that need nothing from the peers, while they arrive
      __syncthreads();                                      // every thread has taken its part of the fp32 tile
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = row0 + it * RSTEP;
        float* r0 = red + row * 64 + cc;
        float* r1 = red + (BM + row) * 64 + cc;
        *reinterpret_cast<float4*>(r0) = make_float4(d[it][0] * xh[it][0], d[it][1] * xh[it][1], d[it][2] * xh[it][2], d[it][3] * xh[it][3]);
        *reinterpret_cast<float4*>(r0 + 4) = make_float4(d[it][4] * xh[it][4], d[it][5] * xh[it][5], d[it][6] * xh[it][6], d[it][7] * xh[it][7]);
        *reinterpret_cast<float4*>(r1) = make_float4(d[it][0], d[it][1], d[it][2], d[it][3]);
        *reinterpret_cast<float4*>(r1 + 4) = make_float4(d[it][4], d[it][5], d[it][6], d[it][7]);
      }
      __syncthreads();
      if (tid < 128) {
        const int q = tid >> 6, col = tid & 63;
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < BM; ++r) t += red[(q * BM + r) * 64 + col];
        e.sy_part[((size_t)pidx * 3 + q) * N + n0 + col] = t;
      }
      {
        const int grep = min(m0 + rlim, M) - 1;
        if (tid < np && tid != tn) {
          unsigned long long a, b;
          slot_wait(e.sy_slots + ((size_t)grep * np + tid) * 2, a, b, false);
          if (gave_up) __hip_atomic_store(&s_giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        gave_up = __hip_atomic_load(&s_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
      }
      float* red2 = red + 2 * BM * 64;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = row0 + it * RSTEP, gm = m0 + row;
        const bool ok = gm < M && row < rlim;
        float ps[2] = {0.f, 0.f}, pm[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = j8 + 8 * u;
          if (ok && q < np) {
            if (q == tn) { ps[u] = own1[it]; pm[u] = own2[it]; }
            else {
              unsigned long long a, b;
              slot_wait(e.sy_slots + ((size_t)gm * np + q) * 2, a, b, true);
              ps[u] = __uint_as_float((uint32_t)a);
              pm[u] = __uint_as_float((uint32_t)b);
            }
          }
        }
        const float mg = zk_sum8(ps[0] + ps[1]) * e.ln_invh;
        const float mgx = zk_sum8(pm[0] + pm[1]) * e.ln_invh;
        const float rs = lbv[2 * it + 1];
        float o[8], oy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = gave_up ? __builtin_nanf("") : rs * (g[it][j] - mg - xh[it][j] * mgx);
        const uint4 po = pack8(o);
        if (ok) *reinterpret_cast<uint4*>(e.sy_y + (size_t)gm * e.sy_ldy + gn) = po;
        unpack8(po, o);                                     // dy derives from the stored (rounded) ds
        if (e.thr) {
          float dm[8];
          zk_drop_scale8(seed, e.sid, (uint64_t)gm * N + gn, e.thr, e.inv_keep, dm);
#pragma unroll
          for (int j = 0; j < 8; ++j) oy[j] = o[j] * dm[j];
          const uint4 py = pack8(oy);
          if (ok && e.sy_dy != nullptr) *reinterpret_cast<uint4*>(e.sy_dy + (size_t)gm * e.sy_ldy + gn) = py;
          unpack8(py, oy);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) oy[j] = o[j];
        }
        float* r2 = red2 + row * 64 + cc;
        *reinterpret_cast<float4*>(r2) = ok ? make_float4(oy[0], oy[1], oy[2], oy[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(r2 + 4) = ok ? make_float4(oy[4], oy[5], oy[6], oy[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
      if (tid < 64) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < BM; ++r) t += red2[r * 64 + tid];
        e.sy_part[((size_t)pidx * 3 + 2) * N + n0 + tid] = t;
      }
    }
    return;
  }
  if (fast) {
    float v[ITER][8];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = min(row0 + it * RSTEP, BM - 1);
      const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
      const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
      v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
      v[it][4] = b.x; v[it][5] = b.y; v[it][6] = b.z; v[it][7] = b.w;
    }
    if (!(PRE && !FRESH)) load_epi();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = row0 + it * RSTEP;
      if (CH % NT != 0 && row >= BM) break;
      const int gm = m0 + row;
      if (LN == 1 && e.ln_c) {       // consumer of a lazy LayerNorm: rstd (s . (gamma o W) - mu colsum(gamma o W)) + (beta . W + b)
        const float2 st = s_lnstat[row];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = st.y * (v[it][j] * e.alpha - st.x * lcv[j]) + bv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = v[it][j] * e.alpha + bv[j];
      }
      [[maybe_unused]] float rv[8];
      if (e.res) {
        unpack8(rres[it], rv);
        if (LN == 2 && e.res_part) {   // the residual is LN(previous sum), formed as k_add_ln_fwd forms it and rounded as it stores it
          const float2 st = s_lnstat[row];
#pragma unroll
          for (int j = 0; j < 8; ++j) rv[j] = bf2f(f2bf(lgv[j] * (rv[j] - st.x) * st.y + lbv[j]));
        }
        if (!(LN == 2 && e.res_after_drop)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[it][j] += rv[j];
        }
      }
      if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = fmaxf(v[it][j], 0.f);
      } else if (e.act == 2) {
        float av[8];
        unpack8(raux[it], av);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = av[j] > 0.f ? v[it][j] * e.aux_scale : 0.f;
      }
      if (e.thr) {
        float dm8[8];
        zk_drop_scale8(seed, e.sid, (uint64_t)gm * N + gn, e.thr, e.inv_keep, dm8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] *= dm8[j];
      }
      if constexpr (LN == 2) {
        if (e.res && e.res_after_drop) {     // residual_fn: x + dropout(y) (func.py:321-324)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[it][j] += rv[j];
        }
        if (e.ln_stat_out) {
          // statistics of the STORED (bf16) sum, as k_add_ln_fwd takes them: {sum, M2} of this row's 64-column group
          const uint4 pk = pack8(v[it]);
          float r8[8];
          unpack8(pk, r8);
          float s1 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s1 += r8[j];
          s1 = zk_sum8(s1);
          const float mt = s1 * (1.f / 64.f);
          float s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = r8[j] - mt; s2 += d * d; }
          s2 = zk_sum8(s2);
          if ((tid & 7) == 0)
            *reinterpret_cast<float2*>(e.ln_stat_out + ((size_t)gm * (N >> 6) + (gn >> 6)) * 2) = make_float2(s1, s2);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)gm * e.ldc + gn) = pk;
          continue;
        }
      }
      if (e.out_f32) {
        // fp32 outputs are the large write-once tensors (weight gradients, logits): streaming stores, so that the
        // L2 does not fetch the lines it is about to overwrite (FETCH_SIZE == output size with ordinary stores)
        float* d = reinterpret_cast<float*>(e.C) + (size_t)gm * e.ldc + gn;
        zk_f32x4 lo, hi;
        lo.x = v[it][0]; lo.y = v[it][1]; lo.z = v[it][2]; lo.w = v[it][3];
        hi.x = v[it][4]; hi.y = v[it][5]; hi.z = v[it][6]; hi.w = v[it][7];
        __builtin_nontemporal_store(lo, reinterpret_cast<zk_f32x4*>(d));
        __builtin_nontemporal_store(hi, reinterpret_cast<zk_f32x4*>(d) + 1);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)gm * e.ldc + gn) = pack8(v[it]);
      }
    }
    return;
  }
  for (int c = tid; c < BM * CPRW; c += NT) {
    const int row = c / CPRW, cc = (c % CPRW) * 8;
    const int gm = m0 + row, gn = n0 + cc;
    if (gm >= M || gn >= N) continue;
    float v[8];
    {
      const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
      const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    if (slab != nullptr) {
      float* d = slab + (size_t)gm * N + gn;
      if (gn + 8 <= N && (N & 3) == 0) {
        reinterpret_cast<float4*>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        for (int j = 0; j < 8 && gn + j < N; ++j) d[j] = v[j];
      }
      continue;
    }
    if (vec_ok && gn + 8 <= N) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= e.alpha;
      if (LN == 1 && e.ln_c) {
        const float2 st = s_lnstat[row];
        float cv8[8];
        ld8(e.ln_c + gn, cv8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = st.y * (v[j] - st.x * cv8[j]);
      }
      if (e.bias) {
        const float4 a = *reinterpret_cast<const float4*>(e.bias + gn);
        const float4 b = *reinterpret_cast<const float4*>(e.bias + gn + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
      [[maybe_unused]] float rv[8];
      if (e.res) {
        unpack8(zk_ld16<FRESH>(e.res + (size_t)gm * e.ldr + gn), rv);
        if (LN == 2 && e.res_part) {
          const float2 st = s_lnstat[row];
          float g8[8], b8[8];
          ld8(e.res_gamma + gn, g8);
          ld8(e.res_beta + gn, b8);
#pragma unroll
          for (int j = 0; j < 8; ++j) rv[j] = bf2f(f2bf(g8[j] * (rv[j] - st.x) * st.y + b8[j]));
        }
        if (!(LN == 2 && e.res_after_drop)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
      }
      if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (e.act == 2) {
        float av[8];
        unpack8(zk_ld16<FRESH>(e.aux + (size_t)gm * e.ldaux + gn), av);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = av[j] > 0.f ? v[j] * e.aux_scale : 0.f;
      }
      if (e.thr) {
        { float dm[8]; zk_drop_scale8(seed, e.sid, (uint64_t)gm * N + gn, e.thr, e.inv_keep, dm);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= dm[j]; }
      }
      if constexpr (LN == 2) {
        if (e.res && e.res_after_drop) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        if (e.ln_stat_out) {
          // (N % 64 == 0 and the 8 threads of a 64-column group are consecutive: the group is active as a whole)
          const uint4 pk = pack8(v);
          float r8[8];
          unpack8(pk, r8);
          float s1 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s1 += r8[j];
          s1 = zk_sum8(s1);
          const float mt = s1 * (1.f / 64.f);
          float s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = r8[j] - mt; s2 += d * d; }
          s2 = zk_sum8(s2);
          if ((c & 7) == 0)
            *reinterpret_cast<float2*>(e.ln_stat_out + ((size_t)gm * (N >> 6) + (gn >> 6)) * 2) = make_float2(s1, s2);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)gm * e.ldc + gn) = pk;
          continue;
        }
      }
      if (e.out_f32) {
        float* d = reinterpret_cast<float*>(e.C) + (size_t)gm * e.ldc + gn;
        reinterpret_cast<float4*>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.C) + (size_t)gm * e.ldc + gn) = pack8(v);
      }
    } else {
      for (int j = 0; j < 8 && gn + j < N; ++j) epi_store(e, v[j], gm, gn + j, N, seed);
    }
  }
}

