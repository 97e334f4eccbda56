// zk_gemm2.hip -- second-generation bf16 MFMA GEMM: LDS-DMA ring + hardware-transpose reads.
//
// Same contract as k_gemm_mfma (zk_gemm.hip).  What changed, and why (rocprof of round 1: the
// step's GEMMs are small -- 4096x512x512 is 0.9 us of MFMA work per CU -- so the K loop was
// bound by exposed L2/fabric latency, one tile ahead was not enough):
//   * operands go HBM/L2 -> LDS by `global_load_lds_dwordx4` (16 B per lane, no VGPR staging)
//     into an NS-stage ring; NS-1 K tiles are in flight per workgroup, tracked with a counted
//     `s_waitcnt vmcnt(N)` and ONE raw s_barrier per K step;
//   * LDS-DMA writes lane-linear 1 KiB pieces, so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and undone in the fragment read (same XOR both sides);
//   * operands whose contraction dim is not contiguous in HBM (W[K,N] forward, both wgrad
//     operands) are stored UN-transposed ([k][rows]) and read with ds_read_b64_tr_b16, the
//     hardware 4x16 transpose read (semantics probed on the device: zk_probe_tr16) -- no
//     register transposes, no row permutation;
//   * the epilogue goes through LDS: every thread finishes 8 contiguous outputs with 16-byte
//     loads/stores (bias, residual, ReLU, ReLU-backward mask, dropout fused as before).
#include "zk_gemm.h"

typedef short v4s_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) uint4 zk_zero_page[4];   // source of out-of-range pieces
// In-kernel timeline (diagnostic build only: make TRACE=1, scripts/trace_gemm.py): wave 0 of workgroup 301
// stamps s_memtime at the phase boundaries of the K loop (ZK_T, one row per K step) and of the kernel (ZK_E).
#ifdef ZK_GEMM_TRACE
__device__ unsigned long long zk_trace_buf[8192];
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
__device__ __forceinline__ void glds16(const bf16_t* gsrc, uint32_t lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gsrc), "s"(lds_byte_addr)
               : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) unsigned char*)p);
}

// chunk-position swizzles (16-byte chunks)
__device__ __forceinline__ int swz_direct(int row) { return (row >> 1) & 7; }            // 128-B rows
template <int R>
__device__ __forceinline__ int swz_trans(int k) { return R == 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2); }

// Per-lane LDS-DMA plan of one operand: for each of the wave's NINSTR pieces the running source
// pointer (advanced by one K tile after every issue) and the k index inside the tile that decides
// the K-tail predicate.  Tile rows / column chunks outside the matrix are CLAMPED to a valid
// row / chunk 0: what they bring in only feeds output rows / columns the epilogue never stores.
// Only a K tile that crosses kend needs zero fill (both operands), done by the TAIL variant.
template <int R, int NW = 4>
struct DmaPlan {
  static constexpr int PER_WAVE = R * 8 / NW;     // 16-byte chunks per wave
  static constexpr int NINSTR = PER_WAVE / 64;
  const bf16_t* cur[NINSTR];
  int kofs[NINSTR];
};

template <int R, bool TRANS, int NW>
__device__ __forceinline__ void dma_plan(DmaPlan<R, NW>& pl, const bf16_t* __restrict__ src, int ld, int row0,
                                         int rows_total, int kbeg, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < DmaPlan<R, NW>::NINSTR; ++j) {
    const int P = wave * DmaPlan<R, NW>::PER_WAVE + j * 64 + lane;
    if (!TRANS) {
      const int row = P >> 3, pos = P & 7;
      const int c = pos ^ swz_direct(row);
      const int grow = min(row0 + row, rows_total - 1);
      pl.kofs[j] = c * 8;
      pl.cur[j] = src + (size_t)grow * ld + kbeg + c * 8;
    } else {
      constexpr int CPR = R / 8;             // chunks per k row
      const int k = P / CPR, pos = P % CPR;
      const int c = pos ^ swz_trans<R>(k);
      int grow = row0 + c * 8;
      grow = grow < rows_total ? grow : 0;
      pl.kofs[j] = k;
      pl.cur[j] = src + (size_t)(kbeg + k) * ld + grow;
    }
  }
}

// issue the LDS-DMA of the next K tile `t` (k range [kbeg + 64 t, ...)) of one operand into `stage`
// and advance the plan.  TAIL: the tile crosses (or lies past) kend.
template <int R, bool TRANS, bool TAIL, int NW>
__device__ __forceinline__ void dma_tile(DmaPlan<R, NW>& pl, size_t step, int t, int klen, uint32_t stage_addr, int wave) {
#pragma unroll
  for (int j = 0; j < DmaPlan<R, NW>::NINSTR; ++j) {
    const bf16_t* g = pl.cur[j];
    if (TAIL) g = (t * 64 + pl.kofs[j] < klen) ? g : reinterpret_cast<const bf16_t*>(zk_zero_page);
    glds16(g, stage_addr + (uint32_t)(wave * DmaPlan<R, NW>::PER_WAVE + j * 64) * 16u);
    pl.cur[j] += step;
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

extern int g_tune[8];   // A/B switches (zk_tune, zk_elem.hip)

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
template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0, bool KSEG = false>
__device__ __forceinline__ void gemm_tile_to_lds(unsigned char* smem, const bf16_t* __restrict__ A,
                                                 const bf16_t* __restrict__ B, int M, int N, int lda, int ldb,
                                                 int kbeg, int kend, int m0, int n0, const KSegDesc* ks = nullptr) {
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
  if (!KSEG && (PW == 0 || producer)) {
    dma_plan<BM, TA, NDW>(planA, A, lda, m0, M, kbeg, dwave, lane);
    dma_plan<BN, !TB, NDW>(planB, B, ldb, n0, N, kbeg, dwave, lane);
  }
  [[maybe_unused]] int seg_left = 0, seg_id = 0;   // KSEG: K tiles left in the current segment, next segment
  const int klen = kend - kbeg;
  const size_t stepA = TA ? (size_t)64 * lda : (size_t)64;
  const size_t stepB = !TB ? (size_t)64 * ldb : (size_t)64;
  const uint32_t ring_addr = lds_addr(ring);
  auto issue = [&](int t) {
    if (KSEG) {                                   // tiles are issued in order: re-plan at every segment start
      if (seg_left == 0) {
        if (seg_id < ks->nseg) {
          dma_plan<BM, TA, NDW>(planA, ks->A[seg_id], lda, m0, M, 0, dwave, lane);
          dma_plan<BN, !TB, NDW>(planB, ks->B[seg_id], ldb, n0, N, 0, dwave, lane);
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
      dma_tile<BM, TA, false, NDW>(planA, stepA, t, klen, st, dwave);
      dma_tile<BN, !TB, false, NDW>(planB, stepB, t, klen, st + BM * 128, dwave);
    } else {
      dma_tile<BM, TA, true, NDW>(planA, stepA, t, klen, st, dwave);
      dma_tile<BN, !TB, true, NDW>(planB, stepB, t, klen, st + BM * 128, dwave);
    }
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
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) issue(s);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_STAGE) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(kt + NS - 1);
      }
    } else {
      ZK_E(1);
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
template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0, bool KSEG = false>
__device__ __forceinline__ void gemm_tile(unsigned char* smem, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, int M, int N, int lda, int ldb, int kbeg,
                                          int kend, int m0, int n0, float* __restrict__ slab, const GemmEpi& e,
                                          int vec_ok, const KSegDesc* ks = nullptr) {
  gemm_tile_to_lds<BM, BN, NS, TA, TB, NW, PW, KSEG>(smem, A, B, M, N, lda, ldb, kbeg, kend, m0, n0, ks);
  constexpr int CLD = DldsCfg<BM, BN, NS>::CLD;
  constexpr int NT = (NW + PW) * 64;
  const int tid = threadIdx.x;
  const float* sC = reinterpret_cast<const float*>(smem);
  const uint64_t seed = e.thr ? *e.seed : 0;
  constexpr int CPRW = BN / 8;
  // Fast path (interior tile, 16-byte epilogue, no split-K slab): every thread's chunks sit in the same 8
  // columns (NT is a multiple of CPRW), so the loop is fully unrolled with all LDS reads first, then all
  // residual / mask loads, then the arithmetic and the stores -- the latencies overlap instead of adding up
  // once per chunk (the rolled loop below cost ~1200 cycles per chunk, profiles/r01_gemm_kloop_trace.txt).
  if (slab == nullptr && vec_ok && m0 + BM <= M && n0 + BN <= N) {
    static_assert(NT % CPRW == 0, "a thread's chunks must share their columns");
    constexpr int CH = BM * CPRW, ITER = (CH + NT - 1) / NT;
    const int cc = (tid % CPRW) * 8, gn = n0 + cc, row0 = tid / CPRW;
    constexpr int RSTEP = NT / CPRW;
    float v[ITER][8];
    uint4 rres[ITER], raux[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = min(row0 + it * RSTEP, BM - 1);
      const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
      const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
      v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
      v[it][4] = b.x; v[it][5] = b.y; v[it][6] = b.z; v[it][7] = b.w;
    }
    if (e.res) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int gm = m0 + min(row0 + it * RSTEP, BM - 1);
        rres[it] = *reinterpret_cast<const uint4*>(e.res + (size_t)gm * e.ldr + gn);
      }
    }
    if (e.act == 2) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int gm = m0 + min(row0 + it * RSTEP, BM - 1);
        raux[it] = *reinterpret_cast<const uint4*>(e.aux + (size_t)gm * e.ldaux + gn);
      }
    }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e.bias) {
      const float4 a = *reinterpret_cast<const float4*>(e.bias + gn);
      const float4 b = *reinterpret_cast<const float4*>(e.bias + gn + 4);
      bv[0] = a.x; bv[1] = a.y; bv[2] = a.z; bv[3] = a.w; bv[4] = b.x; bv[5] = b.y; bv[6] = b.z; bv[7] = b.w;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = row0 + it * RSTEP;
      if (CH % NT != 0 && row >= BM) break;
      const int gm = m0 + row;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = v[it][j] * e.alpha + bv[j];
      if (e.res) {
        float rv[8];
        unpack8(rres[it], rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] += rv[j];
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
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[it][j] *= zk_drop_scale(seed, e.sid, (uint64_t)gm * N + gn + j, e.thr, e.inv_keep);
      }
      if (e.out_f32) {
        float* d = reinterpret_cast<float*>(e.C) + (size_t)gm * e.ldc + gn;
        reinterpret_cast<float4*>(d)[0] = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
        reinterpret_cast<float4*>(d)[1] = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
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
      if (e.bias) {
        const float4 a = *reinterpret_cast<const float4*>(e.bias + gn);
        const float4 b = *reinterpret_cast<const float4*>(e.bias + gn + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
      if (e.res) {
        float rv[8];
        unpack8(*reinterpret_cast<const uint4*>(e.res + (size_t)gm * e.ldr + gn), rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (e.act == 2) {
        float av[8];
        unpack8(*reinterpret_cast<const uint4*>(e.aux + (size_t)gm * e.ldaux + gn), av);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = av[j] > 0.f ? v[j] * e.aux_scale : 0.f;
      }
      if (e.thr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= zk_drop_scale(seed, e.sid, (uint64_t)gm * N + gn + j, e.thr, e.inv_keep);
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

template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0>
__global__ void __launch_bounds__((NW + PW) * 64) k_gemm_dlds(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M,
                                                   int N, int K, int lda, int ldb, int kchunk,
                                                   float* __restrict__ slabs, TileSched ts, GemmEpi e, EpiVec ev) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];   // the ONLY LDS object
  int tm_, tn_, z_;
  tile_of_block(ts, tm_, tn_, z_);
  const int kbeg = z_ * kchunk;
  const int kend = min(K, kbeg + kchunk);
  gemm_tile<BM, BN, NS, TA, TB, NW, PW>(smem, A, B, M, N, lda, ldb, kbeg, kend, tm_ * BM, tn_ * BN,
                                    slabs ? slabs + (size_t)z_ * M * N : nullptr, e, ev.vec_ok);
  __builtin_amdgcn_sched_barrier(0);
  ZK_E(4);
}

template <int BM, int BN, int NS, bool TB, int PW>
__global__ void __launch_bounds__((4 + PW) * 64) k_gemm_kseg(KSegDesc ks, int M, int N, int lda, int ldb, TileSched ts,
                                                             GemmEpi e, EpiVec ev) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];
  int tm_, tn_, z_;
  tile_of_block(ts, tm_, tn_, z_);
  gemm_tile<BM, BN, NS, false, TB, 4, PW, true>(smem, nullptr, nullptr, M, N, lda, ldb, 0, ks.nseg * ks.tps * 64,
                                                tm_ * BM, tn_ * BN, nullptr, e, ev.vec_ok, &ks);
}

// Grouped launch: many independent GEMMs (same transposition flags) in ONE grid -- the deferred
// weight-gradient GEMMs of several layers, or the cross-attention K/V projections of all decoder
// layers.  Each problem is far too small to fill 256 CUs; together they do, with 128x128 tiles, the
// full K per tile and no split-K slabs.  Descriptors live in device memory.
struct GroupDesc {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias;
  const bf16_t* res;          // optional bf16 residual added in the epilogue (may alias C), row stride ldr
  int M, N, K, lda, ldb, ldc, out_f32, tile_start, tiles_n, ldr;
};                            // 80 bytes; mirrored by zero_amd/func.py:_GroupDesc

template <int BM, int BN, int NS, bool TA, bool TB, int PW = 0>
__global__ void __launch_bounds__((4 + PW) * 64) k_gemm_grouped(const GroupDesc* __restrict__ descs, int nprob) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  int p = 0;
  while (p + 1 < nprob && descs[p + 1].tile_start <= t) ++p;
  const GroupDesc d = descs[p];
  const int local = t - d.tile_start;
  const int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
  GemmEpi e;
  e.C = d.C; e.ldc = d.ldc; e.out_f32 = d.out_f32; e.alpha = 1.f; e.bias = d.bias;
  e.res = d.res; e.ldr = d.ldr; e.act = 0; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f;
  e.thr = 0; e.inv_keep = 1.f; e.seed = nullptr; e.sid = 0;
  const uintptr_t al = (uintptr_t)d.C | (uintptr_t)d.bias | (uintptr_t)d.res;
  const int vec_ok = ((al & 15) == 0) && (d.ldc % 8 == 0) && (d.res == nullptr || d.ldr % 8 == 0);
  gemm_tile<BM, BN, NS, TA, TB, 4, PW>(smem, d.A, d.B, d.M, d.N, d.lda, d.ldb, 0, d.K, tm * BM, tn * BN, nullptr, e,
                                       vec_ok);
}

// =====================================================================================
// Fused logits + label-smoothed cross entropy (transformer.py:182-216, util.py:88-103) for the training
// path: the [T, V] fp32 logits (524 MB at the bench shapes) are never written.
//   forward : every 128x128 logits tile is reduced in its epilogue to one float4 per row
//             {max, sum exp(z - max), sum z, z_gold (or 0)} over the tile's valid columns -> part[T][tiles_n];
//             k_ce_combine merges the tiles_n partials of a row into lse and
//             ce = lse - p z_gold - q (sum z - z_gold) - normalizer.
//   backward: the tile is recomputed (137 GFLOP is cheaper than reading 524 MB back) and its epilogue writes
//             dlogits = w_row (exp(z - lse_row) - soft) as bf16, soft = p on the gold column, q elsewhere,
//             0 in the padding columns >= V: the operand of the two logits-gradient GEMMs.
// =====================================================================================
struct CeEpi {
  const int* ids; const float* lse; const float* w;   // per token row: gold id, log-sum-exp, weight
  float4* part; bf16_t* dlogits;
  int V, ldd, tiles_n; float p, q;
};

template <int PHASE>
__global__ void __launch_bounds__(256) k_logits_ce(const bf16_t* __restrict__ feat, const bf16_t* __restrict__ E,
                                                   int T, int K, int ldf, int lde, TileSched ts, CeEpi c) {
  constexpr int BM = 128, BN = 128, NS = 2;
  constexpr int CLD = DldsCfg<BM, BN, NS>::CLD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];
  int tm_, tn_, z_;
  tile_of_block(ts, tm_, tn_, z_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  gemm_tile_to_lds<BM, BN, NS, false, true>(smem, feat, E, T, c.V, ldf, lde, 0, K, m0, n0);
  const float* sC = reinterpret_cast<const float*>(smem);
  const int tid = threadIdx.x;
  if (PHASE == 1) {
    const int row = tid >> 1, half = tid & 1;
    const int gm = m0 + row;
    const int cbeg = n0 + half * 64;
    const int nval = min(64, c.V - cbeg);              // valid columns of this half (may be <= 0)
    const float* zr = sC + row * CLD + half * 64;
    float m = -INFINITY, sz = 0.f;
    for (int j = 0; j < nval; ++j) { m = fmaxf(m, zr[j]); sz += zr[j]; }
    float se = 0.f;
    for (int j = 0; j < nval; ++j) se += __expf(zr[j] - m);
    float zg = 0.f;
    const int gold = gm < T ? c.ids[gm] : -1;
    if (gold >= cbeg && gold < cbeg + nval) zg = zr[gold - cbeg];
    // merge the two halves of the row
    const float m2 = __shfl_xor(m, 1, 64), se2 = __shfl_xor(se, 1, 64);
    const float mm = fmaxf(m, m2);
    float tot = 0.f;
    if (m > -INFINITY) tot += se * __expf(m - mm);
    if (m2 > -INFINITY) tot += se2 * __expf(m2 - mm);
    sz += __shfl_xor(sz, 1, 64);
    zg += __shfl_xor(zg, 1, 64);
    if (half == 0 && gm < T) c.part[(size_t)gm * c.tiles_n + tn_] = make_float4(mm, tot, sz, zg);
  } else {
    constexpr int CPRW = BN / 8;
    for (int t = tid; t < BM * CPRW; t += 256) {
      const int row = t / CPRW, cc = (t % CPRW) * 8;
      const int gm = m0 + row, gn = n0 + cc;
      if (gm >= T || gn >= c.ldd) continue;
      const float lse = c.lse[gm], wr = c.w[gm];
      const int gold = c.ids[gm];
      const float4 a = *reinterpret_cast<const float4*>(sC + row * CLD + cc);
      const float4 b = *reinterpret_cast<const float4*>(sC + row * CLD + cc + 4);
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = gn + j;
        v[j] = col < c.V ? wr * (__expf(v[j] - lse) - (col == gold ? c.p : c.q)) : 0.f;
      }
      *reinterpret_cast<uint4*>(c.dlogits + (size_t)gm * c.ldd + gn) = pack8(v);
    }
  }
}

// one wave per token row: merge the per-tile partials
__global__ void __launch_bounds__(256) k_ce_combine(const float4* __restrict__ part, int T, int tiles_n, float p,
                                                    float q, float normalizer, float* __restrict__ ce,
                                                    float* __restrict__ lse_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  float m = -INFINITY;
  for (int t = lane; t < tiles_n; t += 64) m = fmaxf(m, part[(size_t)row * tiles_n + t].x);
  m = wave_max(m);
  float se = 0.f, sz = 0.f, zg = 0.f;
  for (int t = lane; t < tiles_n; t += 64) {
    const float4 v = part[(size_t)row * tiles_n + t];
    if (v.x > -INFINITY) se += v.y * __expf(v.x - m);
    sz += v.z;
    zg += v.w;
  }
  se = wave_sum(se); sz = wave_sum(sz); zg = wave_sum(zg);
  if (lane == 0) {
    const float lse = m + __logf(se);
    if (lse_out != nullptr) lse_out[row] = lse;
    if (ce != nullptr) ce[row] = lse - p * zg - q * (sz - zg) - normalizer;
  }
}

template <int BM, int BN, int NS, int NW = 4, int PW = 0>
static int launch_dlds(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                       int splits, int kchunk, float* slabs, const GemmEpi& e, int sched_flags, hipStream_t stream) {
  TileSched ts;
  ts.tiles_m = (M + BM - 1) / BM;
  ts.tiles_n = (N + BN - 1) / BN;
  ts.n_major = ((long)N > (long)M) ? 1 : 0;
  if (sched_flags & 2) ts.n_major ^= 1;
  ts.xcd_remap = (sched_flags & 1) ? 0 : 1;
  EpiVec ev;
  const uintptr_t al = (uintptr_t)e.C | (uintptr_t)e.bias | (uintptr_t)e.res | (uintptr_t)e.aux;
  ev.vec_ok = ((al & 15) == 0) && (e.ldc % 8 == 0) && (e.res == nullptr || e.ldr % 8 == 0) &&
              (e.aux == nullptr || e.ldaux % 8 == 0);
  dim3 grid((unsigned)((long)ts.tiles_m * ts.tiles_n * splits));
  if (!ta && !tb)
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, false, false, NW, PW>), grid, dim3((NW + PW) * 64), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e, ev);
  else if (!ta && tb)
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, false, true, NW, PW>), grid, dim3((NW + PW) * 64), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e, ev);
  else if (ta && !tb)
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, true, false, NW, PW>), grid, dim3((NW + PW) * 64), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e, ev);
  else
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, true, true, NW, PW>), grid, dim3((NW + PW) * 64), 0, stream, A, B, M, N, K, lda, ldb, kchunk, slabs, ts, e, ev);
  ZK_LAUNCH_CHECK();
  return 0;
}

extern "C" {
// descs: device array of `nprob` GroupDesc (80 bytes each, see zk_gemm2.hip) whose tile_start
// fields are the running sum of ceil(M/bm)*ceil(N/bn); total_tiles = that sum.
int zk_gemm_grouped(const void* descs, int nprob, int total_tiles, int ta, int tb, int tile, hipStream_t stream) {
  ZK_CHECK_ARG(nprob >= 1 && total_tiles >= 1, "zk_gemm_grouped: empty group");
  ZK_CHECK_ARG(tile == 1 || tile == 4, "zk_gemm_grouped: tile must be 1 (128x128) or 4 (64x64)");
  const GroupDesc* d = (const GroupDesc*)descs;
  dim3 grid((unsigned)total_tiles);
#define ZK_GROUP_LAUNCH(BM_, BN_, NS_, PW_)                                                                     \
  do {                                                                                                         \
    const dim3 blk((4 + PW_) * 64);                                                                            \
    if (!ta && !tb) hipLaunchKernelGGL((k_gemm_grouped<BM_, BN_, NS_, false, false, PW_>), grid, blk, 0, stream, d, nprob); \
    else if (!ta && tb) hipLaunchKernelGGL((k_gemm_grouped<BM_, BN_, NS_, false, true, PW_>), grid, blk, 0, stream, d, nprob); \
    else if (ta && !tb) hipLaunchKernelGGL((k_gemm_grouped<BM_, BN_, NS_, true, false, PW_>), grid, blk, 0, stream, d, nprob); \
    else hipLaunchKernelGGL((k_gemm_grouped<BM_, BN_, NS_, true, true, PW_>), grid, blk, 0, stream, d, nprob);     \
  } while (0)
  const bool pw = (g_tune[6] >> 16) & 1;          // producer-wave workgroups (see gemm_tile_to_lds)
  if (tile == 1) { if (pw) ZK_GROUP_LAUNCH(128, 128, 2, 4); else ZK_GROUP_LAUNCH(128, 128, 2, 0); }
  else { if (pw) ZK_GROUP_LAUNCH(64, 64, 4, 4); else ZK_GROUP_LAUNCH(64, 64, 4, 0); }
#undef ZK_GROUP_LAUNCH
  ZK_LAUNCH_CHECK();
  return 0;
}
// C bf16 [M, ldc] = sum_s A_s [M, kseg] (lda) x B_s (+ residual bf16 [M, ldr], may alias C).
//   tb = 1: B_s is [N, ldb] with K contiguous (dgrad through W stored [in, out]);  tb = 0: B_s is [kseg, ldb].
// a_segs / b_segs: HOST arrays of nseg (<= 16) device pointers; kseg a multiple of 64; 16-byte aligned operands,
// leading dimensions multiples of 8.
int zk_gemm_kseg(const void* const* a_segs, const void* const* b_segs, int nseg, int kseg, void* C, int M, int N,
                 int lda, int ldb, int ldc, int tb, const void* residual, int ldr, hipStream_t stream) {
  ZK_CHECK_ARG(nseg >= 1 && nseg <= ZK_KSEG_MAX, "zk_gemm_kseg: nseg=%d out of range (<= %d)", nseg, ZK_KSEG_MAX);
  ZK_CHECK_ARG(kseg >= 64 && kseg % 64 == 0, "zk_gemm_kseg: kseg=%d must be a positive multiple of 64", kseg);
  ZK_CHECK_ARG(M >= 0 && N >= 1 && C != nullptr, "zk_gemm_kseg: bad output");
  ZK_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && (residual == nullptr || ldr % 8 == 0) && (tb || N % 8 == 0),
               "zk_gemm_kseg: leading dimensions must be multiples of 8");
  if (M == 0) return 0;
  KSegDesc ks;
  uintptr_t al = (uintptr_t)C | (uintptr_t)residual;
  for (int i = 0; i < nseg; ++i) {
    ks.A[i] = (const bf16_t*)a_segs[i]; ks.B[i] = (const bf16_t*)b_segs[i];
    ZK_CHECK_ARG(a_segs[i] != nullptr && b_segs[i] != nullptr, "zk_gemm_kseg: null segment %d", i);
    al |= (uintptr_t)a_segs[i] | (uintptr_t)b_segs[i];
  }
  ZK_CHECK_ARG((al & 15) == 0, "zk_gemm_kseg: operands must be 16-byte aligned");
  for (int i = nseg; i < ZK_KSEG_MAX; ++i) { ks.A[i] = ks.A[0]; ks.B[i] = ks.B[0]; }
  ks.nseg = nseg; ks.tps = kseg / 64;
  GemmEpi e;
  e.C = C; e.ldc = ldc; e.out_f32 = 0; e.alpha = 1.f; e.bias = nullptr; e.res = (const bf16_t*)residual; e.ldr = ldr;
  e.act = 0; e.aux = nullptr; e.ldaux = 0; e.aux_scale = 1.f; e.thr = 0; e.inv_keep = 1.f; e.seed = nullptr; e.sid = 0;
  TileSched ts;
  ts.tiles_m = (M + 63) / 64; ts.tiles_n = (N + 63) / 64; ts.n_major = ((long)N > (long)M) ? 1 : 0; ts.xcd_remap = 1;
  EpiVec ev;
  ev.vec_ok = 1;
  dim3 grid((unsigned)((long)ts.tiles_m * ts.tiles_n));
  if (tb) hipLaunchKernelGGL((k_gemm_kseg<64, 64, 4, true, 4>), grid, dim3(512), 0, stream, ks, M, N, lda, ldb, ts, e, ev);
  else hipLaunchKernelGGL((k_gemm_kseg<64, 64, 4, false, 4>), grid, dim3(512), 0, stream, ks, M, N, lda, ldb, ts, e, ev);
  ZK_LAUNCH_CHECK();
  return 0;
}

static int ce_check(const void* feat, const void* E, int T, int V, int K, int ldf, int lde) {
  ZK_CHECK_ARG(T >= 0 && V >= 1 && K >= 8, "zk_logits_ce: bad dims T=%d V=%d K=%d", T, V, K);
  ZK_CHECK_ARG(K % 8 == 0 && ldf % 8 == 0 && lde % 8 == 0, "zk_logits_ce: K, ldf, lde must be multiples of 8");
  ZK_CHECK_ARG((((uintptr_t)feat | (uintptr_t)E) & 15) == 0, "zk_logits_ce: operands must be 16-byte aligned");
  return 0;
}
static void ce_smoothing(int V, float label_smooth, float* p, float* q, float* normalizer) {
  *p = 1.f; *q = 0.f; *normalizer = 0.f;
  if (label_smooth > 0.f && label_smooth < 1.f) {        // util.py:90-97, fp32 arithmetic
    const float n = (float)(V - 1);
    *p = 1.f - label_smooth;
    *q = label_smooth / n;
    *normalizer = -(*p * logf(*p) + n * *q * logf(*q + 1e-20f));
  }
}
size_t zk_logits_ce_workspace(int T, int V) { return (size_t)T * ((V + 127) / 128) * sizeof(float4); }

// feat bf16 [T, K] (ldf), E bf16 [>= V rows, K] (lde; the softmax embedding), ids int32 [T].
// Writes ce fp32 [T] (may be NULL) and lse fp32 [T]; workspace >= zk_logits_ce_workspace(T, V).
int zk_logits_ce_fwd(const void* feat, const void* E, const int* ids, float* ce, float* lse, int T, int V, int K,
                     int ldf, int lde, float label_smooth, void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (int rc = ce_check(feat, E, T, V, K, ldf, lde)) return rc;
  ZK_CHECK_ARG(lse != nullptr && ids != nullptr, "zk_logits_ce_fwd: ids and lse are required");
  ZK_CHECK_ARG(ws_bytes >= zk_logits_ce_workspace(T, V), "zk_logits_ce_fwd: workspace too small");
  if (T == 0) return 0;
  CeEpi c;
  c.ids = ids; c.lse = nullptr; c.w = nullptr; c.part = (float4*)workspace; c.dlogits = nullptr;
  c.V = V; c.ldd = 0; c.tiles_n = (V + 127) / 128;
  float normalizer;
  ce_smoothing(V, label_smooth, &c.p, &c.q, &normalizer);
  TileSched ts;
  ts.tiles_m = (T + 127) / 128; ts.tiles_n = c.tiles_n; ts.n_major = 1; ts.xcd_remap = 1;
  hipLaunchKernelGGL(k_logits_ce<1>, dim3((unsigned)((long)ts.tiles_m * ts.tiles_n)), dim3(256), 0, stream,
                     (const bf16_t*)feat, (const bf16_t*)E, T, K, ldf, lde, ts, c);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ce_combine, dim3((T + 3) / 4), dim3(256), 0, stream, (const float4*)workspace, T, c.tiles_n,
                     c.p, c.q, normalizer, ce, lse);
  ZK_LAUNCH_CHECK();
  return 0;
}

// dlogits bf16 [T, ldd] (ldd >= V, multiple of 8; columns >= V are written 0) = w_row (softmax - soft labels)
int zk_logits_ce_bwd(const void* feat, const void* E, const int* ids, const float* w, const float* lse,
                     void* dlogits, int T, int V, int K, int ldf, int lde, int ldd, float label_smooth,
                     hipStream_t stream) {
  if (int rc = ce_check(feat, E, T, V, K, ldf, lde)) return rc;
  ZK_CHECK_ARG(ids && w && lse && dlogits, "zk_logits_ce_bwd: ids, w, lse, dlogits are required");
  ZK_CHECK_ARG(ldd >= V && ldd % 8 == 0 && ((uintptr_t)dlogits & 15) == 0, "zk_logits_ce_bwd: ldd=%d must be >= V and a multiple of 8", ldd);
  if (T == 0) return 0;
  CeEpi c;
  c.ids = ids; c.lse = lse; c.w = w; c.part = nullptr; c.dlogits = (bf16_t*)dlogits;
  c.V = V; c.ldd = ldd; c.tiles_n = (ldd + 127) / 128;
  float normalizer;
  ce_smoothing(V, label_smooth, &c.p, &c.q, &normalizer);
  TileSched ts;
  ts.tiles_m = (T + 127) / 128; ts.tiles_n = c.tiles_n; ts.n_major = 1; ts.xcd_remap = 1;
  hipLaunchKernelGGL(k_logits_ce<2>, dim3((unsigned)((long)ts.tiles_m * ts.tiles_n)), dim3(256), 0, stream,
                     (const bf16_t*)feat, (const bf16_t*)E, T, K, ldf, lde, ts, c);
  ZK_LAUNCH_CHECK();
  return 0;
}
#ifdef ZK_GEMM_TRACE
int zk_debug_trace_read(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_trace_buf), sizeof(unsigned long long) * n);
}
#endif
}  // extern "C"

// Producer waves of the gen-2 kernel for a bm x bn tile: tuning key 6 holds one nibble per tile class
// (64x64 | 128x128 << 4 | 128x64 and 64x128 << 12; bit 8: ring depth 3 for 128x128; bit 16: grouped launches).
// Default 0x44 -- in-step A/B on the bench configuration: 5.41 -> 5.32 ms (profiles/r01_gemm_producer_waves.txt).
int zk_gemm_dlds_pw(int bm, int bn) {
  const int t = g_tune[6];
  int pw = 0;
  if (bm == 64 && bn == 64) pw = t & 15;
  else if (bm == 128 && bn == 128) pw = (t >> 4) & 15;
  else if ((bm == 128 && bn == 64) || (bm == 64 && bn == 128)) pw = (t >> 12) & 15;
  const bool deep128 = bm == 128 && bn == 128 && ((t >> 8) & 1);
  if (pw == 2) return (bm == bn && !deep128) ? 2 : 0;      // instantiated: 64x64 and 128x128 with ring depth 2
  return (pw == 4 || pw == 8) ? pw : 0;
}

// entry used by zk_gemm (zk_gemm.hip)
int zk_gemm_dlds_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                          int bm, int bn, int splits, int kchunk, float* slabs, const GemmEpi& e, int sched_flags,
                          hipStream_t stream) {
  int ns = (sched_flags >> 4) & 15;   // ring-depth override (tuning)
  if (!ns && bm == 64 && bn == 64 && g_tune[3]) ns = g_tune[3];     // A/B: ring depth of the 64x64 tile in-step
  if (bm == 64 && bn == 64 && g_tune[4] == 2)                       // A/B: two-wave workgroups, ring depth 2
    return launch_dlds<64, 64, 2, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (bm == 64 && bn == 64 && g_tune[4] == 4)                       // two-wave workgroups, ring depth 4
    return launch_dlds<64, 64, 4, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (!ns) {                                                        // producer-wave workgroups (default for 64x64, 128x128)
    const int t = g_tune[6], pw = zk_gemm_dlds_pw(bm, bn), ns3 = (t >> 8) & 1;
#define ZK_PW(BM_, BN_, NS_, PW_) return launch_dlds<BM_, BN_, NS_, 4, PW_>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream)
    if (bm == 64 && bn == 64) { if (pw == 2) ZK_PW(64, 64, 4, 2); if (pw == 4) ZK_PW(64, 64, 4, 4); if (pw == 8) ZK_PW(64, 64, 4, 8); }
    if (bm == 128 && bn == 128 && !ns3) { if (pw == 2) ZK_PW(128, 128, 2, 2); if (pw == 4) ZK_PW(128, 128, 2, 4); if (pw == 8) ZK_PW(128, 128, 2, 8); }
    if (bm == 128 && bn == 128 && ns3) { if (pw == 4) ZK_PW(128, 128, 3, 4); if (pw == 8) ZK_PW(128, 128, 3, 8); }
    if (bm == 128 && bn == 64) { if (pw == 4) ZK_PW(128, 64, 2, 4); if (pw == 8) ZK_PW(128, 64, 2, 8); }
    if (bm == 64 && bn == 128) { if (pw == 4) ZK_PW(64, 128, 2, 4); if (pw == 8) ZK_PW(64, 128, 2, 8); }
#undef ZK_PW
  }
  if (ns) {
#define ZK_NS(BM_, BN_, NS_) return launch_dlds<BM_, BN_, NS_>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream)
    if (bm == 64 && bn == 64) { if (ns == 2) ZK_NS(64, 64, 2); if (ns == 6) ZK_NS(64, 64, 6); if (ns == 8) ZK_NS(64, 64, 8); }
    if (bm == 128 && bn == 64) { if (ns == 3) ZK_NS(128, 64, 3); if (ns == 6) ZK_NS(128, 64, 6); }
    if (bm == 128 && bn == 128) { if (ns == 3) ZK_NS(128, 128, 3); if (ns == 4) ZK_NS(128, 128, 4); }
#undef ZK_NS
  }
  // ring depth 2 for the larger tiles: the measured optimum is MORE workgroups per CU (64 KiB /
  // 48 KiB of LDS each), not deeper prefetch
  if (bm == 256 && bn == 128) return launch_dlds<256, 128, 2, 8>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (bm == 128 && bn == 128) return launch_dlds<128, 128, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (bm == 128 && bn == 64) return launch_dlds<128, 64, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (bm == 64 && bn == 128) return launch_dlds<64, 128, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  return launch_dlds<64, 64, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
}
