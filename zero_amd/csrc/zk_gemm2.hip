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
#include "zk_gemm2_dev.h"

template <int BM, int BN, int NS, bool TA, bool TB, int NW = 4, int PW = 0, int LN = 0>
__global__ void __launch_bounds__((NW + PW) * 64) k_gemm_dlds(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M,
                                                   int N, int K, int lda, int ldb, int kchunk,
                                                   float* __restrict__ slabs, TileSched ts, GemmEpi e, EpiVec ev) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];   // the ring (LN: + 2 x BM float2 of row statistics)
  int tm_, tn_, z_;
  tile_of_block(ts, tm_, tn_, z_);
  const int kbeg = z_ * kchunk;
  const int kend = min(K, kbeg + kchunk);
  gemm_tile<BM, BN, NS, TA, TB, NW, PW, false, false, LN>(smem, A, B, M, N, lda, ldb, kbeg, kend, tm_ * BM, tn_ * BN,
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
  float* colsum;              // optional fp32 [N]: column sums of B (tb = 0) over K, written by the producer waves of the
                              // tm = 0 tiles (bias gradient beside a weight gradient): by the producer waves of the
  long pad_;                  // wide tiles, by two extra MFMAs per eight on the 256x256 tiles (gemm256_acc<.., CS>)
};                            // 96 bytes; mirrored by zero_amd/func.py:_GroupDesc

template <int BM, int BN, int NS, bool TA, bool TB, int PW = 0>
__global__ void __launch_bounds__((4 + PW) * 64) k_gemm_grouped(const GroupDesc* __restrict__ descs, int nprob) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<BM, BN, NS>::LDS_BYTES];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  int p = 0;
  {   // the last problem whose first tile is <= t: binary search (a linear scan is one dependent global load per
      // problem -- 6 us before the first K tile with the 49 problems of a weight-gradient group)
    int hi = nprob - 1;
    while (p < hi) {
      const int mid = (p + hi + 1) >> 1;
      if (descs[mid].tile_start <= t) p = mid; else hi = mid - 1;
    }
  }
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
                                       vec_ok, nullptr, (PW > 0 && tm == 0) ? d.colsum : nullptr);
}

// =====================================================================================
// 256x256 tiles for the large fp32-output GEMMs (grouped weight gradients incl. the logits problem: 40 % of the
// step's FLOPs).  scripts/gemm_big_bench.py + the K-step arithmetic: these GEMMs are bound by the L2 -> LDS path at
// ~10-13 TB/s chip-wide (the same rate the best published gfx950 kernels sustain), so the lever is bytes per FLOP:
// 256x256x64 moves 64 KB for 8.4 MFLOP (7.8 KB/MFLOP; 128x256: 11.7, 128x128: 15.6).
// Eight waves as 2 x 4, each a 128x64 register tile (TM = 4, TN = 2: 128 accumulator registers), all of them issue
// LDS-DMA (two pieces after every 16-deep K slice's MFMAs, so the issue stalls sit behind matrix work), two ring
// stages of 64 KiB.  The tile leaves straight from the accumulators (a 32x32 MFMA tile stores two full 128-byte
// lines per instruction for fp32) -- the fp32 tile would not fit in LDS.  fp32 output only, no epilogue options.
// =====================================================================================
// accumulators of one 256x256 tile: acc[i][j][e] = element (row m0 + wm*128 + i*32 + (e&3) + 8*(e>>2) + 4*(lane>>5),
// column n0 + wn*64 + j*32 + (lane&31)) with wave = wm * 4 + wn.  smem: the 128 KiB ring (free again on return, after a
// barrier by the caller).
// CS (tb = 0 only): acc_cs[j] additionally accumulates ones[32 x 16] x B-fragment, i.e. every row of it holds the column
// sums over k of this wave's 32 columns of B -- the bias gradient sum_k dY[k][n] (func.py:16, 58-60) of a weight-gradient
// problem, from the fragments the wave has in registers anyway (2 extra MFMAs per 8; waves with cs_on only).
// HP (round 5 experiment, tuning key 14 bit 5): only the FIRST half of the workgroup (waves 0-3) issues LDS-DMA -- all 16
// pieces of a wave pair -- and the second half starts multiplying right behind the barrier: the two waves of a SIMD are then
// never in their issue stalls at the same time.
template <bool TA, bool TB, bool SPREAD, bool CS = false, bool HP = false>
__device__ __forceinline__ void gemm256_acc(unsigned char* smem, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                            int lda, int ldb, int M, int N, int K, int m0, int n0, f32x16_t (&acc)[4][2],
                                            bool cs_on = false, f32x16_t* acc_cs = nullptr, int sched = 0) {
  constexpr int BM = 256, BN = 256, NS = 2, NW = 8, NWN = 4, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int STAGE = (BM + BN) * 64;
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int nk = (K + 63) >> 6;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8_t ones;
  if (CS) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc_cs[j][e] = 0.f;
    typedef short v8s_ __attribute__((ext_vector_type(8)));
    const v8s_ o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};     // bf16 1.0
    ones = __builtin_bit_cast(bf16x8_t, o);
  }
  constexpr int NDW = HP ? 4 : NW;                 // waves that issue LDS-DMA
  constexpr int NI = DmaPlan<BM, NDW>::NINSTR;    // pieces per issuing wave and operand (4, HP: 8)
  // (sched bit 6, HP only: the SECOND-dispatched half issues and the first, older, half multiplies first)
  const bool issuer = !HP || ((sched & 64) ? wave >= 4 : wave < 4);
  const int dw = HP ? (wave & 3) : wave;             // index among the issuing waves
  DmaPlan<BM, NDW> planA;
  DmaPlan<BN, NDW> planB;
  if (issuer) {
    dma_plan<BM, TA, NDW>(planA, lda, m0, M, dw, lane);
    dma_plan<BN, !TB, NDW>(planB, ldb, n0, N, dw, lane);
  }
  const size_t stepA = TA ? (size_t)64 * lda : (size_t)64;
  const size_t stepB = !TB ? (size_t)64 * ldb : (size_t)64;
  const uint32_t ring_addr = lds_addr(ring);
  // LDS-DMA of K tile t, pieces [j0, j1) of this wave's NI + NI (HP: full calls -- j0 = 0, j1 = 8 -- issue all 16).  The
  // pieces are 32-bit offsets against the tile's wave-uniform base (glds16s): no vector instruction per piece.
  auto issue_part = [&](int tt, int j0, int j1) {
    if (!issuer) return;
    if (HP) { j0 *= 2; j1 *= 2; }
    if (tt * 64 >= K) return;                       // (past the end: nothing to bring in; the waits are vmcnt(0))
    const uint32_t st = ring_addr + (uint32_t)((tt % NS) * STAGE * 2);
    const bf16_t* bA = A + (size_t)tt * stepA;
    const bf16_t* bB = B + (size_t)tt * stepB;
    const bool tail = tt * 64 + 64 > K;              // the one tile that crosses K: pointer form with the per-lane select
#pragma unroll
    for (int j = 0; j < 2 * NI; ++j) {
      if (j < j0 || j >= j1) continue;
      const bool isA = j < NI;
      const int jj = isA ? j : j - NI;
      const uint32_t dst = isA ? st + (uint32_t)(dw * DmaPlan<BM, NDW>::PER_WAVE + jj * 64) * 16u
                               : st + BM * 128 + (uint32_t)(dw * DmaPlan<BN, NDW>::PER_WAVE + jj * 64) * 16u;
      const uint32_t off = isA ? planA.off[jj] : planB.off[jj];
      const bf16_t* base = isA ? bA : bB;
      if (!tail) {
        glds16s(off, base, dst);
      } else {
        const int kofs = isA ? planA.kofs[jj] : planB.kofs[jj];
        const bf16_t* g = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(base) + off);
        g = (tt * 64 + kofs < K) ? g : reinterpret_cast<const bf16_t*>(zk_zero_page);
        glds16(g, dst);
      }
    }
  };
  // `sched` (round 5, tuning key 14; !SPREAD only): WHEN the second-dispatched half of the workgroup (waves 4-7, the
  // partner of waves 0-3 on each SIMD) issues its eight LDS-DMA pieces of the next K tile.  With every wave issuing right
  // behind the barrier, both waves of a SIMD sit in their DMA-issue stalls (~100-185 cycles a piece while the CU's
  // texture-address path is busy) at the same time and the matrix pipe idles; shifted, one wave's issue stalls run under
  // its partner's MFMAs (MI355X_MICROARCH.md "Two waves per SIMD": pair matrix with memory, not matrix with matrix).
  //   bits 0-1: 0 = with the first half (rounds 2-4); 1 = four pieces after the first and four after the second 16-deep
  //             slice's MFMAs; 2 = all eight after the second slice; 3 = two after every slice
  //   bit 2:    the second half runs at s_setprio 1 (the arbitration loser otherwise)
  const bool g1 = wave >= 4;
  const int sched_hp = sched;
  if (HP) sched &= 4;                               // (the placements below are about the second half's OWN pieces)
  //   bit 7: the placements of bits 0-1 apply to the FIRST half instead (the second half issues behind the barrier and
  //          the first, older, half multiplies first)
  const bool late = (sched & 128) ? !g1 : g1;
  int place = late ? (sched & 3) : 0;
  //   bits 3-4: the eight pieces of a wave go out in ONE slot of the step (slot 0 = behind the barrier, slot s = behind
  //             the MFMAs of slice s - 1), the slots dealt by SIMD so that few waves queue on the CU's one texture-address
  //             path at a time (64 pieces a step at ~16 cycles each: a wave that issues ALONE is through in ~130 cycles,
  //             eight at once block each other for ~1000): 1 = slot w % 4 for both waves of a SIMD, 2 = w % 4 for the
  //             first half and (w + 2) % 4 for the second (a SIMD always has one wave multiplying), 3 = w % 4 / (w + 1) % 4
  const int stag = (sched >> 3) & 3;
  int slot = -1;
  if (stag == 1) slot = wave & 3;
  else if (stag == 2) slot = g1 ? ((wave + 2) & 3) : (wave & 3);
  else if (stag == 3) slot = g1 ? ((wave + 1) & 3) : (wave & 3);
  if (slot >= 0) place = slot == 0 ? 0 : 4;      // 4: none of the fixed placements below
  if (g1 && (sched & 4)) __builtin_amdgcn_s_setprio(1);
  issue_part(0, 0, 8);
  __builtin_amdgcn_sched_barrier(0);
  // make TRACE=1 (scripts/trace_gemm256.py): waves 0 and 4 of workgroup 100 stamp the 100 MHz clock at the phase
  // boundaries of their first 64 K steps: [0] loop top, [1] LDS-DMA of this tile landed, [2] barrier passed, [3] next
  // tile's LDS-DMA issued (schedule 0), [4..7] the 16-deep slices' MFMAs issued
#ifdef ZK_GEMM_TRACE
  const bool tr256 = blockIdx.x == 100 && (tid == 0 || tid == 256);
#define ZK_T256(slot) do { if (tr256 && kt < 64) zk_trace_buf[kt * 16 + (tid ? 8 : 0) + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define ZK_T256(slot) do { } while (0)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    ZK_T256(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt has landed (it was issued one compute phase ago)
    ZK_T256(1);
    __builtin_amdgcn_s_barrier();                          // ... for every wave, and everybody is done with tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    ZK_T256(2);
    if (!SPREAD) { if (place == 0) issue_part(kt + 1, 0, 8); __builtin_amdgcn_sched_barrier(0); }
    ZK_T256(3);
    const bf16_t* sA = ring + (kt % NS) * STAGE;
    const bf16_t* sB = sA + BM * 64;
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
      if (CS && cs_on) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc_cs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[kk & 1][j], acc_cs[j], 0, 0, 0);
      }
#ifdef ZK_GEMM_TRACE
      __builtin_amdgcn_sched_barrier(0);
      ZK_T256(4 + kk);
#endif
      if (SPREAD) issue_part(kt + 1, kk * 2, kk * 2 + 2);   // the other stage is free since this step's barrier
      else if (place != 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (place == 1 && kk < 2) issue_part(kt + 1, kk * 4, kk * 4 + 4);
        if (place == 2 && kk == 1) issue_part(kt + 1, 0, 8);
        if (place == 3) issue_part(kt + 1, kk * 2, kk * 2 + 2);
        if (place == 4 && kk == slot - 1) issue_part(kt + 1, 0, 8);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (all-zero) pieces
  if (g1 && (sched & 4)) __builtin_amdgcn_s_setprio(0);
}

#ifdef ZK_EXPERIMENTS   // measured equal to the two-stage ring (profiles/r03_pmc_stall_split.txt): make EXPERIMENTS=1
// The same 256 x 256 tile on a FOUR-stage ring of 32-deep K tiles (4 x 32 KiB = the same 128 KiB) for the weight-gradient
// form (ta = 1, tb = 0: both operands stored [k][rows], so a stage is simply 32 k rows instead of 64).  Why: with two
// 64-deep stages the LDS-DMA of K tile kt+1 has ONE compute step (~2048 MFMA cycles per SIMD, 0.85 us) to come back from
// L2, and under load it takes longer -- the MFMA pipe of the one-launch weight-gradient group was 40 % busy
// (profiles/r03_pmc_mfma.json), the waves parked at s_waitcnt.  Four stages keep three tiles (1.3 us of compute) in
// flight behind counted vmcnt waits, at the price of twice the barriers.
template <bool CS>
__device__ __forceinline__ void gemm256_acc_k32(unsigned char* smem, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                int lda, int ldb, int M, int N, int K, int m0, int n0, f32x16_t (&acc)[4][2],
                                                bool cs_on, f32x16_t* acc_cs) {
  constexpr int BM = 256, BN = 256, NS = 4, KT = 32, NW = 8, NWN = 4, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int STAGE = (BM + BN) * KT;                       // bf16 elements per ring stage
  constexpr int PER_WAVE = BM * KT / 8 / NW, NINSTR = PER_WAVE / 64, CPR = BM / 8;   // (BM == BN)
  constexpr int PER_STAGE = 2 * NINSTR;                       // LDS-DMA pieces per wave and stage
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int nk = (K + KT - 1) / KT;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8_t ones;
  if (CS) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc_cs[j][e] = 0.f;
    typedef short v8s_ __attribute__((ext_vector_type(8)));
    const v8s_ o = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};     // bf16 1.0
    ones = __builtin_bit_cast(bf16x8_t, o);
  }
  // per-lane plan of the wave's pieces: running source pointer and k index inside the tile (K-tail predicate)
  const bf16_t* curA[NINSTR]; const bf16_t* curB[NINSTR];
  int kofs[NINSTR];
#pragma unroll
  for (int j = 0; j < NINSTR; ++j) {
    const int P = wave * PER_WAVE + j * 64 + lane;
    const int k = P / CPR, pos = P % CPR;
    const int c = pos ^ swz_trans<BM>(k);
    int ga = m0 + c * 8, gb = n0 + c * 8;
    ga = ga < M ? ga : 0;
    gb = gb < N ? gb : 0;
    kofs[j] = k;
    curA[j] = A + (size_t)k * lda + ga;
    curB[j] = B + (size_t)k * ldb + gb;
  }
  const size_t stepA = (size_t)KT * lda, stepB = (size_t)KT * ldb;
  const uint32_t ring_addr = lds_addr(ring);
  auto issue = [&](int tt) {
    const uint32_t st = ring_addr + (uint32_t)((tt % NS) * STAGE * 2);
    const bool tail = tt * KT + KT > K;
#pragma unroll
    for (int j = 0; j < NINSTR; ++j) {
      const bf16_t* g = curA[j];
      if (tail) g = (tt * KT + kofs[j] < K) ? g : reinterpret_cast<const bf16_t*>(zk_zero_page);
      glds16(g, st + (uint32_t)(wave * PER_WAVE + j * 64) * 16u);
      curA[j] += stepA;
    }
#pragma unroll
    for (int j = 0; j < NINSTR; ++j) {
      const bf16_t* g = curB[j];
      if (tail) g = (tt * KT + kofs[j] < K) ? g : reinterpret_cast<const bf16_t*>(zk_zero_page);
      glds16(g, st + BM * KT * 2 + (uint32_t)(wave * PER_WAVE + j * 64) * 16u);
      curB[j] += stepB;
    }
  };
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_) issue(s_);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most NS-2 later tiles of this wave are still in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_STAGE) : "memory");
    __builtin_amdgcn_s_barrier();                          // ... for every wave, and everybody is done with tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    issue(kt + NS - 1);                                    // refill the stage everybody finished reading
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* sA = ring + (kt % NS) * STAGE;
    const bf16_t* sB = sA + BM * KT;
    bf16x8_t af[2][TM], bfr[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = load_frag<BM, true>(sA, wm * WTM + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = load_frag<BN, true>(sB, wn * WTN + j * 32, 0, lane);
#pragma unroll
    for (int kk = 0; kk < KT / 16; ++kk) {
      if (kk + 1 < KT / 16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = load_frag<BM, true>(sA, wm * WTM + i * 32, kk + 1, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[(kk + 1) & 1][j] = load_frag<BN, true>(sB, wn * WTN + j * 32, kk + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
      if (CS && cs_on) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc_cs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[kk & 1][j], acc_cs[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (all-zero) pieces
}

#endif  // ZK_EXPERIMENTS

// UPD (round 4): the optimiser update of a weight INSIDE the launch that produces its gradient.  A tile of a problem whose
// descriptor carries pad_ & 1 does not store its fp32 gradient tile; every lane runs TF1 Adam (cycle.py:94-101 with
// clip_grad_norm = 0.0: the norm-free update; main.py:178-181) on the elements it holds in its accumulators -- reads
// theta, m, v, writes theta, m, v and the bf16 shadow, at the gradient's offset inside the flat buffers -- and leaves its
// wave's sums of squares (scaled gradient, parameters before the update) in `sq` for the norm the step reports.  The
// gradient never travels to HBM and back (8 B per parameter) and the HBM-bound update of the weights (22 B per
// parameter) runs under the MFMA / LDS-bound K loops of the other workgroups instead of in a pass of its own.
struct UpdArgs {
  float* master; float* m; float* v; bf16_t* shadow;   // flat buffers (zero_amd/variables.py)
  const float* grad_base;                              // the flat gradient buffer the descriptors' C pointers point into
  const float* hyper;                                  // [0] lr_t [1] beta1 [2] beta2 [3] eps [4] gradient scale
  float* sq;                                           // [tiles][8 waves][2]: sum g^2, sum theta^2 per wave (zeros for other tiles)
  int stagger;                                         // phases | us per phase << 8 (0: all workgroups start together)
  int sched;                                           // gemm256_acc's LDS-DMA issue schedule (tuning key 14), every variant
};

template <bool TA, bool TB, bool SPREAD, bool CS = false, bool K32 = false, bool UPD = false, bool HP = false>
__global__ void __launch_bounds__(512) k_gemm_grouped256(const GroupDesc* __restrict__ descs, int nprob, UpdArgs ua) {
  constexpr int BM = 256, BN = 256, NS = 2, NWN = 4, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int STAGE = (BM + BN) * 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE * 2];   // the ONLY LDS object (128 KiB)
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  int p = 0;
  {   // the last problem whose first tile is <= t: binary search (a linear scan is one dependent global load per
      // problem -- 6 us before the first K tile with the 49 problems of a weight-gradient group)
    int hi = nprob - 1;
    while (p < hi) {
      const int mid = (p + hi + 1) >> 1;
      if (descs[mid].tile_start <= t) p = mid; else hi = mid - 1;
    }
  }
  const GroupDesc d = descs[p];
  const int local = t - d.tile_start;
  if constexpr (UPD) {
    // Every tile ends with ~1.7 MB of optimiser traffic.  Workgroups that start together finish their K loops together
    // and then all pull on HBM at once (measured: 794 us for the launch, the K loops idle meanwhile, against 500 us
    // without the update).  The first wave of workgroups therefore starts in PHASES a fraction of a K loop apart
    // (tuning key 12: phases | delay in us per phase << 8): later workgroups inherit the phase of the CU they land on, so
    // that some CUs stream their update while the others multiply.
    const int phases = ua.stagger & 255, step_us = ua.stagger >> 8;
    if (phases > 1 && bid < 256) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
      const unsigned long long wait = (unsigned long long)((bid >> 3) % phases) * step_us * 100ull;
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
  }
  // tile order inside a problem: blocks of 4 x 8 tiles, the blocks of one 8-tile column group one after the other.  An XCD
  // runs 32 consecutive tiles at a time (its chunk of the launch is contiguous): a block keeps 4 A row blocks + 8 B column
  // blocks = 3 MB at K = 512 in its 4-MB L2, and the next block reuses the 8 column blocks.  Row-major order made the
  // logits forward (16 x 125 tiles) fetch the whole 33-MB softmax table once per row of tiles: FETCH_SIZE 543 MB against
  // 37 MB of operands (profiles/r03_pmc_traffic.json), as much HBM traffic again as the 525 MB of logits it writes.
  // Problems with fewer than 4 x 8 tiles (every weight gradient) keep the row-major order.
  int tm, tn;
  {
    constexpr int GM = 4, GN = 8;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int nbm = tiles_m / GM, nbn = d.tiles_n / GN;
    const int full = nbm * nbn * GM * GN;
    if (nbm == 0 || nbn == 0) {
      tm = local / d.tiles_n; tn = local - tm * d.tiles_n;
    } else if (local < full) {
      const int blk = local / (GM * GN), within = local - blk * (GM * GN);
      const int bn = blk / nbm, bm = blk - bn * nbm;
      tm = bm * GM + within / GN; tn = bn * GN + within % GN;
    } else {
      // what the blocks leave: the right stripe (all rows, the last tiles_n % 8 columns), then the bottom stripe
      int r = local - full;
      const int wr = d.tiles_n - nbn * GN, right = wr * tiles_m;
      if (r < right) { tm = r / wr; tn = nbn * GN + r - tm * wr; }
      else { r -= right; const int wb = nbn * GN; tm = nbm * GM + r / wb; tn = r % wb; }
    }
  }
  const int m0 = tm * BM, n0 = tn * BN, M = d.M, N = d.N;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  f32x16_t acc[TM][TN];
  if (CS) {
    // fp32 tile + bias gradient: the waves of the first row of waves (wm = 0) of the tm = 0 tiles carry the column sums
    f32x16_t acc_cs[TN];
    const bool cs_on = d.colsum != nullptr && tm == 0 && wm == 0;
#ifdef ZK_EXPERIMENTS
    if (K32) gemm256_acc_k32<true>(smem, d.A, d.B, d.lda, d.ldb, M, N, d.K, m0, n0, acc, cs_on, acc_cs);
    else
#endif
    gemm256_acc<TA, TB, SPREAD, true, HP>(smem, d.A, d.B, d.lda, d.ldb, M, N, d.K, m0, n0, acc, cs_on, acc_cs, ua.sched);
    if (cs_on && lane < 32) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + lane;
        if (col < N) d.colsum[col] = acc_cs[j][0];
      }
    }
#ifdef ZK_EXPERIMENTS
  } else if (K32) {
    gemm256_acc_k32<false>(smem, d.A, d.B, d.lda, d.ldb, M, N, d.K, m0, n0, acc, false, nullptr);
#endif
  } else {
    gemm256_acc<TA, TB, SPREAD, false, HP>(smem, d.A, d.B, d.lda, d.ldb, M, N, d.K, m0, n0, acc, false, nullptr, ua.sched);
  }
  float* C = reinterpret_cast<float*>(d.C);
  if constexpr (UPD) {
    float gacc = 0.f, pacc = 0.f;
    if (d.pad_ & 1) {
      const size_t off = (size_t)(reinterpret_cast<const float*>(d.C) - ua.grad_base);
      float* __restrict__ P = ua.master + off;
      float* __restrict__ Mo = ua.m + off;
      float* __restrict__ Vo = ua.v + off;
      bf16_t* __restrict__ S = ua.shadow + off;
      const float lr = ua.hyper[0], b1 = ua.hyper[1], b2 = ua.hyper[2], eps = ua.hyper[3], gs = ua.hyper[4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WTN + j * 32 + (lane & 31);
          // 24 loads (half a 32 x 32 piece) are requested before the first is used
#pragma unroll
          for (int eh = 0; eh < 16; eh += 8) {
            float pv[8], mv[8], vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int e = eh + q;
              const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
              const bool ok = row < M && col < N;
              const size_t idx = (size_t)row * d.ldc + col;
              pv[q] = ok ? P[idx] : 0.f;
              mv[q] = ok ? Mo[idx] : 0.f;
              vv[q] = ok ? Vo[idx] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int e = eh + q;
              const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
              if (row < M && col < N) {
                const size_t idx = (size_t)row * d.ldc + col;
                // the arithmetic of k_adam<GSQ = true> (zk_elem.hip), element for element
                const float gj = acc[i][j][e] * gs;
                gacc += gj * gj;
                pacc += pv[q] * pv[q];
                const float mn = b1 * mv[q] + (1.f - b1) * gj;
                const float vn = b2 * vv[q] + (1.f - b2) * gj * gj;
                const float pn = pv[q] - lr * mn / (sqrtf(vn) + eps);
                P[idx] = pn; Mo[idx] = mn; Vo[idx] = vn;
                S[idx] = f2bf(pn);
              }
            }
          }
        }
    }
    // the wave's sums of squares (zeros from the tiles that store a gradient): fixed slots, summed in a fixed order later
    gacc = wave_sum(gacc);
    pacc = wave_sum(pacc);
    if (lane == 0) { ua.sq[((size_t)t * 8 + wave) * 2] = gacc; ua.sq[((size_t)t * 8 + wave) * 2 + 1] = pacc; }
    if (d.pad_ & 1) return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        // streaming store: the tile is written once and read by a later kernel -- with an ordinary store the L2
        // fetches every line it is about to overwrite (FETCH_SIZE == output size, profiles/r02_pmc_traffic.json)
        if (row < M && col < N) __builtin_nontemporal_store(acc[i][j][e], &C[(size_t)row * d.ldc + col]);
      }
    }
}

#ifdef ZK_EXPERIMENTS   // fused logits + cross entropy kernels (profiles/r02_fused_ce_256_tile.txt: 449 vs 392 us)
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
    const float m2 = zk_dpp<ZK_DPP_XOR1, 0xf>(m, m), se2 = zk_dpp<ZK_DPP_XOR1, 0xf>(se, se);
    const float mm = fmaxf(m, m2);
    float tot = 0.f;
    if (m > -INFINITY) tot += se * __expf(m - mm);
    if (m2 > -INFINITY) tot += se2 * __expf(m2 - mm);
    sz += zk_dpp<ZK_DPP_XOR1, 0xf>(0.f, sz);
    zg += zk_dpp<ZK_DPP_XOR1, 0xf>(0.f, zg);
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

// ---- the same on 256x256 tiles (gemm256_acc: ~1.0 PF on these shapes against ~0.6 for the 128x128 tile above)
// forward: the TRANSPOSED product Z^T = E . feat^T, so that a token is a COLUMN of the tile: its 64 vocabulary entries of a
// wave sit in one lane's accumulator registers (rows spread over e, i) and max / sum exp / sum z are in-lane VALU work --
// with tokens as rows every row statistic is a 32-lane reduction per accumulator register.  z_gold is left to the
// combine kernel (one K-length dot product per token).  part[T][tiles_v] = {max, sum exp, sum z, 0}.
__global__ void __launch_bounds__(512) k_logits_ce256_fwd(const bf16_t* __restrict__ feat, const bf16_t* __restrict__ E,
                                                          int T, int K, int ldf, int lde, int tiles_t, CeEpi c) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 512 * 64 * 2];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tv = t / tiles_t, tt = t - tv * tiles_t;       // an XCD walks the token tiles of one vocabulary panel
  const int m0 = tv * 256, n0 = tt * 256;                  // rows: vocabulary, columns: tokens
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  f32x16_t acc[4][2];
  gemm256_acc<false, true, true>(smem, E, feat, lde, ldf, c.V, T, K, m0, n0, acc);
  __syncthreads();                                         // the ring is free
  float* sStat = reinterpret_cast<float*>(smem);           // [2 wm][256 tokens][4]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int vrow = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (vrow < c.V) m = fmaxf(m, acc[i][j][e]);
      }
    float se = 0.f, sz = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int vrow = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (vrow < c.V) { se += __expf(acc[i][j][e] - m); sz += acc[i][j][e]; }
      }
    // the other half of the token's rows lives in lane ^ 32
    const float m2 = __shfl_xor(m, 32, 64), se2 = __shfl_xor(se, 32, 64), sz2 = __shfl_xor(sz, 32, 64);
    const float mm = fmaxf(m, m2);
    float tot = 0.f;
    if (m > -INFINITY) tot += se * __expf(m - mm);
    if (m2 > -INFINITY) tot += se2 * __expf(m2 - mm);
    if (lane < 32) {
      float4* d = reinterpret_cast<float4*>(sStat + ((size_t)(wm * 256 + wn * 64 + j * 32 + lane)) * 4);
      *d = make_float4(mm, tot, sz + sz2, 0.f);
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int tok = n0 + threadIdx.x;
    if (tok < T) {
      const float4 a = *reinterpret_cast<const float4*>(sStat + (size_t)threadIdx.x * 4);
      const float4 b = *reinterpret_cast<const float4*>(sStat + (size_t)(256 + threadIdx.x) * 4);
      const float mm = fmaxf(a.x, b.x);
      float tot = 0.f;
      if (a.x > -INFINITY) tot += a.y * __expf(a.x - mm);
      if (b.x > -INFINITY) tot += b.y * __expf(b.x - mm);
      c.part[(size_t)tok * c.tiles_n + tv] = make_float4(mm, tot, a.z + b.z, 0.f);
    }
  }
}

// backward: tokens as rows (the dlogits rows are vocabulary-contiguous: a half-wave stores 64 contiguous bytes); the per-row
// scalars wait in LDS
__global__ void __launch_bounds__(512) k_logits_ce256_bwd(const bf16_t* __restrict__ feat, const bf16_t* __restrict__ E,
                                                          int T, int K, int ldf, int lde, int tiles_m, CeEpi c) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 512 * 64 * 2];
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t / tiles_m, tm = t - tn * tiles_m;       // an XCD walks the token tiles of one vocabulary panel
  const int m0 = tm * 256, n0 = tn * 256;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  f32x16_t acc[4][2];
  gemm256_acc<false, true, true>(smem, feat, E, ldf, lde, T, c.V, K, m0, n0, acc);
  __syncthreads();
  float* sL = reinterpret_cast<float*>(smem);
  float* sW = sL + 256;
  int* sG = reinterpret_cast<int*>(sW + 256);
  if (threadIdx.x < 256) {
    const int gm = min(m0 + (int)threadIdx.x, T - 1);
    sL[threadIdx.x] = c.lse[gm]; sW[threadIdx.x] = c.w[gm]; sG[threadIdx.x] = c.ids[gm];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int gm = m0 + rl;
      const float lse = sL[rl], wr = sW[rl];
      const int gold = sG[rl];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        const float v = col < c.V ? wr * (__expf(acc[i][j][e] - lse) - (col == gold ? c.p : c.q)) : 0.f;
        if (gm < T && col < c.ldd) c.dlogits[(size_t)gm * c.ldd + col] = f2bf(v);
      }
    }
}

// z_gold of the 256-tile forward: one wave per token, feat[t] . E[gold[t]] (bf16 products, fp32 sum)
__global__ void __launch_bounds__(256) k_ce_combine_gold(const float4* __restrict__ part, int T, int tiles_n, float p,
                                                         float q, float normalizer, float* __restrict__ ce,
                                                         float* __restrict__ lse_out, const bf16_t* __restrict__ feat,
                                                         const bf16_t* __restrict__ E, const int* __restrict__ ids, int K,
                                                         int ldf, int lde) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int gold = ids[row];
  float zg = 0.f;
  for (int c0 = lane * 8; c0 < K; c0 += 512) {
    float a[8], b[8];
    unpack8(*reinterpret_cast<const uint4*>(feat + (size_t)row * ldf + c0), a);
    unpack8(*reinterpret_cast<const uint4*>(E + (size_t)gold * lde + c0), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) zg += a[j] * b[j];
  }
  float m = -INFINITY;
  for (int t = lane; t < tiles_n; t += 64) m = fmaxf(m, part[(size_t)row * tiles_n + t].x);
  m = wave_max(m);
  float se = 0.f, sz = 0.f;
  for (int t = lane; t < tiles_n; t += 64) {
    const float4 v = part[(size_t)row * tiles_n + t];
    if (v.x > -INFINITY) se += v.y * __expf(v.x - m);
    sz += v.z;
  }
  se = wave_sum(se); sz = wave_sum(sz); zg = wave_sum(zg);
  if (lane == 0) {
    const float lse = m + __logf(se);
    if (lse_out != nullptr) lse_out[row] = lse;
    if (ce != nullptr) ce[row] = lse - p * zg - q * (sz - zg) - normalizer;
  }
}
#endif  // ZK_EXPERIMENTS

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
// descs: device array of `nprob` GroupDesc (96 bytes each, see zk_gemm2.hip) whose tile_start
// fields are the running sum of ceil(M/bm)*ceil(N/bn); total_tiles = that sum.
int zk_gemm_grouped(const void* descs, int nprob, int total_tiles, int ta, int tb, int tile, hipStream_t stream) {
  ZK_CHECK_ARG(nprob >= 1 && total_tiles >= 1, "zk_gemm_grouped: empty group");
  // bit 8 of `tile` (256x256 tiles, ta = 1, tb = 0): some problem carries a column-sum output (GroupDesc.colsum)
  const bool cs = (tile & 256) != 0;
  const bool k32 = (tile & 512) != 0;         // bit 9 (256x256, ta = 1, tb = 0): four-stage ring of 32-deep K tiles
  tile &= 255;
  ZK_CHECK_ARG(tile == 1 || tile == 4 || tile == 5 || tile == 6 || tile == 7 || tile == 8,
               "zk_gemm_grouped: tile must be 1 (128x128), 4 (64x64), 5 (256x128), 6 (128x256) or 7 / 8 (256x256)");
  ZK_CHECK_ARG(!cs || ((tile == 7 || tile == 8) && ta && !tb), "zk_gemm_grouped: column sums on 256x256 tiles need ta = 1, tb = 0");
  ZK_CHECK_ARG(!k32 || ((tile == 7 || tile == 8) && ta && !tb), "zk_gemm_grouped: the 32-deep ring exists for ta = 1, tb = 0 on 256x256 tiles");
#ifndef ZK_EXPERIMENTS
  ZK_CHECK_ARG(!k32, "zk_gemm_grouped: the 32-deep ring is an experiment (make EXPERIMENTS=1)");
#endif
  const GroupDesc* d = (const GroupDesc*)descs;
  dim3 grid((unsigned)total_tiles);
  if (tile == 7 || tile == 8) {     // fp32 outputs without epilogue options only (checked on the host copy by the caller)
    const dim3 blk(512);
    UpdArgs ua0 = UpdArgs();
    ua0.sched = g_tune[14];          // LDS-DMA issue schedule of the second half of the workgroup (gemm256_acc)
#ifdef ZK_EXPERIMENTS
#define ZK_G256_K32                                                                                                          \
      else if (ta && !tb && k32 && cs) hipLaunchKernelGGL((k_gemm_grouped256<true, false, false, true, true>), grid, blk, 0, stream, d, nprob, ua0); \
      else if (ta && !tb && k32) hipLaunchKernelGGL((k_gemm_grouped256<true, false, false, false, true>), grid, blk, 0, stream, d, nprob, ua0);
#else
#define ZK_G256_K32
#endif
#define ZK_G256(SP_)                                                                                         \
    do {                                                                                                     \
      if (!ta && !tb) hipLaunchKernelGGL((k_gemm_grouped256<false, false, SP_>), grid, blk, 0, stream, d, nprob, ua0);   \
      else if (!ta && tb) hipLaunchKernelGGL((k_gemm_grouped256<false, true, SP_>), grid, blk, 0, stream, d, nprob, ua0); \
      ZK_G256_K32                                                                                          \
      else if (ta && !tb && cs) hipLaunchKernelGGL((k_gemm_grouped256<true, false, SP_, true>), grid, blk, 0, stream, d, nprob, ua0); \
      else if (ta && !tb) hipLaunchKernelGGL((k_gemm_grouped256<true, false, SP_>), grid, blk, 0, stream, d, nprob, ua0); \
      else hipLaunchKernelGGL((k_gemm_grouped256<true, true, SP_>), grid, blk, 0, stream, d, nprob, ua0);               \
    } while (0)
#ifdef ZK_EXPERIMENTS   // half-producer form (HP): measured, no robust gain (profiles/r05_gemm256_half_producer_sweep.txt)
    if (tile == 8 && (g_tune[14] & 32) && ((ta && !tb) || (!ta && tb))) {
      if (ta && cs) hipLaunchKernelGGL((k_gemm_grouped256<true, false, false, true, false, false, true>), grid, blk, 0, stream, d, nprob, ua0);
      else if (ta) hipLaunchKernelGGL((k_gemm_grouped256<true, false, false, false, false, false, true>), grid, blk, 0, stream, d, nprob, ua0);
      else hipLaunchKernelGGL((k_gemm_grouped256<false, true, false, false, false, false, true>), grid, blk, 0, stream, d, nprob, ua0);
    } else
#endif
    if (tile == 7) ZK_G256(true); else ZK_G256(false);
#undef ZK_G256
    ZK_LAUNCH_CHECK();
    return 0;
  }
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
  // 256x128 / 128x256: four compute waves with a 128x64 register tile each (half the LDS fragment bytes and 3/4
  // of the L1->LDS bytes per MFMA of the 128x128 tile) + four producer waves, three-stage ring (144 KiB: one
  // workgroup per CU)
  else if (tile == 5) { if (g_tune[6] & (1 << 17)) ZK_GROUP_LAUNCH(256, 128, 2, 4); else ZK_GROUP_LAUNCH(256, 128, 3, 4); }
  else if (tile == 6) { if (g_tune[6] & (1 << 17)) ZK_GROUP_LAUNCH(128, 256, 2, 4); else ZK_GROUP_LAUNCH(128, 256, 3, 4); }
  else { if (pw) ZK_GROUP_LAUNCH(64, 64, 4, 4); else ZK_GROUP_LAUNCH(64, 64, 4, 0); }
#undef ZK_GROUP_LAUNCH
  ZK_LAUNCH_CHECK();
  return 0;
}
#ifdef ZK_EXPERIMENTS   // the update inside the weight-gradient launch: measured slower (profiles/r04_negative_results.txt)
// Every weight gradient of the step AND the update of the weights it belongs to, in one launch (UpdArgs above): the grouped
// weight-gradient launch of 256 x 256 tiles (ta = 1, tb = 0, bias column sums riding along) whose descriptors carry
// pad_ & 1 for the problems whose output is a whole variable of the flat buffers -- those tiles run TF1 Adam on their
// accumulators instead of storing them.  sq: fp32 [total_tiles * 16], written by every tile.
int zk_gemm_grouped_update(const void* descs, int nprob, int total_tiles, float* master, float* m, float* v, void* shadow,
                           const float* grad_base, const float* hyper, float* sq, hipStream_t stream) {
  ZK_CHECK_ARG(nprob >= 1 && total_tiles >= 1, "zk_gemm_grouped_update: empty group");
  ZK_CHECK_ARG(master && m && v && shadow && grad_base && hyper && sq, "zk_gemm_grouped_update: null pointer");
  UpdArgs ua;
  ua.master = master; ua.m = m; ua.v = v; ua.shadow = (bf16_t*)shadow; ua.grad_base = grad_base; ua.hyper = hyper; ua.sq = sq;
  ua.stagger = g_tune[12];         // tuning key 12 (default 4 phases, 28 us apart)
  ua.sched = g_tune[14];
  hipLaunchKernelGGL((k_gemm_grouped256<true, false, false, true, false, true>), dim3((unsigned)total_tiles), dim3(512), 0,
                     stream, (const GroupDesc*)descs, nprob, ua);
  ZK_LAUNCH_CHECK();
  return 0;
}
#endif  // ZK_EXPERIMENTS
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

#ifdef ZK_EXPERIMENTS   // fused logits + cross entropy: measured slower than GEMM + zk_ce_fused, make EXPERIMENTS=1
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
// 256x256 tiles when the problem fills them (tuning key 8 = 1: always the 128x128 kernels)
static bool ce_use_256(int T, int V, int K) { return g_tune[8] != 1 && T >= 256 && V >= 1024 && K % 8 == 0; }
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
  if (ce_use_256(T, V, K)) {
    c.tiles_n = (V + 255) / 256;
    const int tiles_t = (T + 255) / 256;
    hipLaunchKernelGGL(k_logits_ce256_fwd, dim3((unsigned)((long)c.tiles_n * tiles_t)), dim3(512), 0, stream,
                       (const bf16_t*)feat, (const bf16_t*)E, T, K, ldf, lde, tiles_t, c);
    ZK_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ce_combine_gold, dim3((T + 3) / 4), dim3(256), 0, stream, (const float4*)workspace, T, c.tiles_n,
                       c.p, c.q, normalizer, ce, lse, (const bf16_t*)feat, (const bf16_t*)E, ids, K, ldf, lde);
    ZK_LAUNCH_CHECK();
    return 0;
  }
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
  if (ce_use_256(T, V, K)) {
    c.tiles_n = (ldd + 255) / 256;
    const int tiles_m = (T + 255) / 256;
    hipLaunchKernelGGL(k_logits_ce256_bwd, dim3((unsigned)((long)c.tiles_n * tiles_m)), dim3(512), 0, stream,
                       (const bf16_t*)feat, (const bf16_t*)E, T, K, ldf, lde, tiles_m, c);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  TileSched ts;
  ts.tiles_m = (T + 127) / 128; ts.tiles_n = c.tiles_n; ts.n_major = 1; ts.xcd_remap = 1;
  hipLaunchKernelGGL(k_logits_ce<2>, dim3((unsigned)((long)ts.tiles_m * ts.tiles_n)), dim3(256), 0, stream,
                     (const bf16_t*)feat, (const bf16_t*)E, T, K, ldf, lde, ts, c);
  ZK_LAUNCH_CHECK();
  return 0;
}
#endif  // ZK_EXPERIMENTS
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

#ifdef ZK_EXPERIMENTS   // the LayerNorm-free forward: measured, no gain (profiles/r04_negative_results.txt)
// the lazy-LayerNorm forms (zk_gemm_ln; A [M,K] x B [K,N], no split-K): the tile kernels of the default dispatch with the
// LN epilogue compiled in -- 64x64 and 128x128 with four producer waves, 128x64 / 64x128 without (tuning key 6 = 0x44)
template <int BM, int BN, int NS, int PW>
static int launch_dlds_ln(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpi& e,
                          hipStream_t stream) {
  // two instantiations per tile: the consumer form (LN = 1) and the producer forms (LN = 2) -- a GEMM is never both, and
  // keeping them apart keeps the 128x128 consumer at two workgroups per CU (<= 128 VGPRs)
  TileSched ts;
  ts.tiles_m = (M + BM - 1) / BM;
  ts.tiles_n = (N + BN - 1) / BN;
  ts.n_major = ((long)N > (long)M) ? 1 : 0;
  ts.xcd_remap = 1;
  EpiVec ev;
  ev.vec_ok = 1;                     // checked by zk_gemm_ln: 16-byte aligned operands, ldc / ldr multiples of 8
  dim3 grid((unsigned)((long)ts.tiles_m * ts.tiles_n));
  if (e.ln_c != nullptr)
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, false, false, 4, PW, 1>), grid, dim3((4 + PW) * 64), 0, stream, A, B, M, N,
                       K, lda, ldb, K, (float*)nullptr, ts, e, ev);
  else
    hipLaunchKernelGGL((k_gemm_dlds<BM, BN, NS, false, false, 4, PW, 2>), grid, dim3((4 + PW) * 64), 0, stream, A, B, M, N,
                       K, lda, ldb, K, (float*)nullptr, ts, e, ev);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_gemm_dlds_ln_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int bm, int bn,
                             const GemmEpi& e, hipStream_t stream) {
  if (bm == 128 && bn == 128) return launch_dlds_ln<128, 128, 2, 4>(A, B, M, N, K, lda, ldb, e, stream);
  if (bm == 128 && bn == 64) return launch_dlds_ln<128, 64, 2, 0>(A, B, M, N, K, lda, ldb, e, stream);
  if (bm == 64 && bn == 128) return launch_dlds_ln<64, 128, 2, 0>(A, B, M, N, K, lda, ldb, e, stream);
  return launch_dlds_ln<64, 64, 4, 4>(A, B, M, N, K, lda, ldb, e, stream);
}

#endif  // ZK_EXPERIMENTS

// residual + LayerNorm inside the producing launch (zk_gemm_add_ln; A [M,K] x B [K,N], no split-K): the 64x64 tile with
// four producer waves that the default dispatch gives the sub-layer output products, with the LN = 3 epilogue.  Tile
// order: row-block major, so that the N/64 workgroups that wait for each other are consecutive in the dispatch order (and, with the XCD remap, on one XCD whenever the grid is a multiple
// of 64): a workgroup only ever waits for workgroups that are already resident or next in line.
template <int BM, int NS, int PW>
static int launch_dlds_sync_ln(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpi& e,
                               hipStream_t stream) {
  TileSched ts;
  ts.tiles_m = (M + BM - 1) / BM;
  ts.tiles_n = N / 64;
  ts.n_major = 0;
  ts.xcd_remap = 1;
  EpiVec ev;
  ev.vec_ok = 1;
  const long nwg = (long)ts.tiles_m * ts.tiles_n;
  dim3 grid((unsigned)nwg);
  GemmEpi el = e;
  // every XCD owns a contiguous range of nwg / 8 tiles: with that a multiple of the group size no group straddles two
  // XCDs and the exchange stays in their L2 (tuning key 15 = 1: never assume it -- exchange through memory)
  el.sy_local = (nwg % (8 * ts.tiles_n) == 0 && !(g_tune[15] & 1)) ? 1 : 0;
  el.sy_fault = (g_tune[15] & 2) ? 1 : 0;       // fault injection for tests/test_gpu_sync_ln.py: a peer that never publishes
  hipLaunchKernelGGL((k_gemm_dlds<BM, 64, NS, false, false, 4, PW, 3>), grid, dim3((4 + PW) * 64), 0, stream, A, B, M, N, K,
                     lda, ldb, K, (float*)nullptr, ts, el, ev);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_gemm_dlds_sync_ln_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int bm,
                                  const GemmEpi& e, hipStream_t stream) {
  (void)bm;
  return launch_dlds_sync_ln<64, 4, 4>(A, B, M, N, K, lda, ldb, e, stream);
}

// the backward form (zk_gemm_ln_bwd): dgrad A [M,K] x B[N,K]^T on 64x64 tiles with the LN = 4 epilogue
int zk_gemm_dlds_sync_ln_bwd_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb,
                                      const GemmEpi& e, hipStream_t stream) {
  TileSched ts;
  ts.tiles_m = (M + 63) / 64;
  ts.tiles_n = N / 64;
  ts.n_major = 0;
  ts.xcd_remap = 1;
  EpiVec ev;
  ev.vec_ok = 1;
  const long nwg = (long)ts.tiles_m * ts.tiles_n;
  GemmEpi el = e;
  el.sy_local = (nwg % (8 * ts.tiles_n) == 0 && !(g_tune[15] & 1)) ? 1 : 0;
  el.sy_fault = (g_tune[15] & 2) ? 1 : 0;       // fault injection for tests/test_gpu_sync_ln.py: a peer that never publishes
  hipLaunchKernelGGL((k_gemm_dlds<64, 64, 4, false, true, 4, 4, 4>), dim3((unsigned)nwg), dim3(512), 0, stream, A, B, M, N, K,
                     lda, ldb, K, (float*)nullptr, ts, el, ev);
  ZK_LAUNCH_CHECK();
  return 0;
}

#ifdef ZK_EXPERIMENTS   // measured: slower than the two launches, alone and with four lanes (profiles/r04_negative_results.txt item 9)
// ---- the two products of a feed-forward sub-layer on few rows in ONE launch (zk_ffn_pair; the decode step's
// func.py:327-338 ffn_layer at 128 rows): phase 1  h = relu(x W1 + b1)  on 64x64 tiles, a barrier among the launch's
// workgroups (all resident: the grid is at most two per CU), phase 2  parts[z] = h[:, K_z] W2[K_z, :]  (split-K partial
// sums as zk_gemm_parts leaves them for zk_ln_decode) on 64x64 tiles.  The barrier is a 64-bit arrival counter that is
// never reset: every launch with the same number T1 of phase-1 tiles adds exactly T1, so the window a launch waits for is
// (count at its start / T1 + 1) T1.  h crosses XCDs: agent-scope release (L2 write-back) before the arrival, acquire
// (invalidate) after the wait.
__global__ void __launch_bounds__(512) k_ffn_pair(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W1, bf16_t* __restrict__ h,
                                                   const bf16_t* __restrict__ W2, float* __restrict__ parts, int M, int F, int H,
                                                   int K1, int ldx, int ldw1, int ldw2, int kchunk, int tiles_m, int T1, int T2,
                                                   GemmEpi e1, GemmEpi e2, unsigned long long* __restrict__ cnt,
                                                   int* __restrict__ err) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DldsCfg<64, 64, 4>::LDS_BYTES];
  const int bid = blockIdx.x, tid = threadIdx.x;
  __shared__ unsigned long long s_target;
  if (tid == 0) {
    const unsigned long long v0 = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_target = (v0 / (unsigned long long)T1 + 1ull) * (unsigned long long)T1;
  }
  {
    const int tm = bid % tiles_m, tn = bid / tiles_m;                 // bid < T1: the grid IS the phase-1 tiles
    gemm_tile<64, 64, 4, false, false, 4, 4>(smem, x, W1, M, F, ldx, ldw1, 0, K1, tm * 64, tn * 64, nullptr, e1, 1);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                  // this thread's part of h is written back
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bid < T2) {
      const unsigned long long target = s_target;
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1 << 15)) {
          if (err != nullptr) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
  }
  if (bid >= T2) return;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int per = tiles_m * (H / 64);
  const int z = bid / per, rem = bid - z * per;
  const int tm = rem % tiles_m, tn = rem / tiles_m;
  gemm_tile<64, 64, 4, false, false, 4, 4>(smem, h, W2, M, H, F, ldw2, z * kchunk, min(F, (z + 1) * kchunk), tm * 64, tn * 64,
                                           parts + (size_t)z * M * H, e2, 0);
}

int zk_ffn_pair_launch(const bf16_t* x, const bf16_t* W1, bf16_t* h, const bf16_t* W2, float* parts, int M, int F, int H, int K1,
                       int ldx, int ldw1, int ldw2, int kchunk, int nparts, const GemmEpi& e1, const GemmEpi& e2,
                       unsigned long long* cnt, int* err, hipStream_t stream) {
  const int tiles_m = (M + 63) / 64, T1 = tiles_m * (F / 64), T2 = tiles_m * (H / 64) * nparts;
  hipLaunchKernelGGL(k_ffn_pair, dim3((unsigned)T1), dim3(512), 0, stream, x, W1, h, W2, parts, M, F, H, K1, ldx, ldw1, ldw2,
                     kchunk, tiles_m, T1, T2, e1, e2, cnt, err);
  ZK_LAUNCH_CHECK();
  return 0;
}

#endif  // ZK_EXPERIMENTS

// entry used by zk_gemm (zk_gemm.hip)
int zk_gemm_dlds_dispatch(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                          int bm, int bn, int splits, int kchunk, float* slabs, const GemmEpi& e, int sched_flags,
                          hipStream_t stream) {
  if (sched_flags & 0x100) {          // wide register tiles (tile overrides 6 / 7): 4 compute waves of 128x64 + 4 producers
    const int ns_w = (sched_flags >> 4) & 15;
    if (bm == 256 && bn == 128) {
      if (ns_w == 2) return launch_dlds<256, 128, 2, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
      return launch_dlds<256, 128, 3, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
    }
    if (bm == 128 && bn == 256) {
      if (ns_w == 2) return launch_dlds<128, 256, 2, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
      return launch_dlds<128, 256, 3, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
    }
  }
  if (g_tune[2] == 3 && bm == 128 && bn == 64) {
    // A/B: ONE 128x64 workgroup per CU with a deep ring (tuning key 3 = 4 / 5 / 6 stages of 24 KiB) + producer waves:
    // 25 % fewer L2 -> LDS bytes than two 64x64 workgroups and more of them in flight
    if (g_tune[3] == 6) return launch_dlds<128, 64, 6, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
    if (g_tune[3] == 5) return launch_dlds<128, 64, 5, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
    return launch_dlds<128, 64, 4, 4, 4>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  }
  int ns = (sched_flags >> 4) & 15;   // ring-depth override (tuning)
  if (!ns && bm == 64 && bn == 64 && g_tune[3] && g_tune[3] != 5) ns = g_tune[3];     // A/B: ring depth of the 64x64 tile in-step
  if (bm == 64 && bn == 64 && g_tune[4] == 2)                       // A/B: two-wave workgroups, ring depth 2
    return launch_dlds<64, 64, 2, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (bm == 64 && bn == 64 && g_tune[4] == 4)                       // two-wave workgroups, ring depth 4
    return launch_dlds<64, 64, 4, 2>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream);
  if (!ns) {                                                        // producer-wave workgroups (default for 64x64, 128x128)
    const int t = g_tune[6], pw = zk_gemm_dlds_pw(bm, bn), ns3 = (t >> 8) & 1;
#define ZK_PW(BM_, BN_, NS_, PW_) return launch_dlds<BM_, BN_, NS_, 4, PW_>(A, B, M, N, K, lda, ldb, ta, tb, splits, kchunk, slabs, e, sched_flags, stream)
    // tuning key 3 = 5: five ring stages (80 KiB: two workgroups fill the CU's 160 KiB of LDS, 128 KiB in flight)
    if (bm == 64 && bn == 64 && g_tune[3] == 5 && pw == 4) ZK_PW(64, 64, 5, 4);
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
