// zk_layer.hip -- the layer program: a run of dependent, sentence-local ops executed by ONE persistent launch.
//
// Why (profiles/r02_*: every microsecond of the step is inside a kernel, and the ~100 small kernels of the
// encoder / decoder chain cost 7-20 us each for 2-9 GFLOP -- fixed cost: launch ramp, first tiles fetched from the
// memory side because a kernel boundary writes the per-XCD L2s back, drain):  the reference's layer stack
// (transformer.py:35-69, 121-181; func.py:194-338) is data parallel over sentences all the way down -- linear and
// LayerNorm act per row, attention per sentence -- so the B sentences are dealt to the 8 XCDs and every XCD walks
// the whole op list on ITS sentences: workgroup b belongs to group b % 8 (the XCD it is observed to run on), a
// phase = one op restricted to the group's rows, phases are separated by a barrier among the group's 64
// workgroups only (~1 us; a barrier among all 256 CUs costs more than a kernel boundary), and activations stay in
// the group's 4-MiB L2 from one phase to the next.  Weights stream from the memory side into all eight L2s.
//
// The ops are the SAME tile functions the launch-per-op kernels run (zk_gemm2_dev.h gemm_tile, zk_attn_dev.h
// attn_fwd_tile / attn_bwd_fused64_tile, zk_ln_dev.h add_ln_fwd_row) with the same tile shapes, so the results are
// bit-identical to the launch-per-op path (tests/test_gpu_program.py).
//
// Visibility inside a group: stores are complete (s_waitcnt vmcnt(0)) before a workgroup arrives; the XCD's L2 is
// coherent for its own CUs; the only cache between a CU and its L2 that another CU's stores do not refresh is the
// CU's vector L1, so every load of a tensor that an earlier op of the SAME launch wrote bypasses the L1 (FRESH
// variants of the tile functions: nt loads, sc1 LDS-DMA; weights keep the ordinary path).  With that a barrier is an
// arrival count and nothing else (measured: an agent-scope acquire per workgroup per barrier -- buffer_inv sc1 --
// cost ~7 us per barrier at two workgroups per CU and made the program 1.45x SLOWER than launch-per-op).
// Placement is never ASSUMED: every workgroup compares HW_REG_XCC_ID with its group id before the first phase, and a
// group with a stray workgroup runs all its barriers with agent-scope release + acquire fences (L2 write-back, L1
// invalidate), which is correct for any placement (MI355X_MICROARCH.md, inter-workgroup visibility).
// Every spin is bounded: a barrier that does not complete sets the abort flag (state[576 + 1]) and the launch drains.
#include "zk_prog.h"
#include "zk_gemm2_dev.h"
#include "zk_attn_dev.h"
#include "zk_ln_dev.h"
#include <vector>
#include <string.h>

#define LP_GROUPS 8
#define LP_THREADS 256
#define LP_WG_PER_GROUP 64          // 2 workgroups per CU on the 32 CUs of an XCD
#define LP_FLAGS (LP_GROUPS * 64)   // state layout (ints): [g*64] arrival counter of group g; [512 + g] stray flag of
                                    // group g; [576] strays seen, [577] abort, [578] phases completed by workgroup 0
#define LP_STATE_INTS (640 + 16 * 512)   // + two 64-bit cycle stamps per op of workgroup 0 (after the op, after its barrier)
#define LP_SPIN_LIMIT 400000

enum { LP_GEMM = 1, LP_ATTN_FWD = 2, LP_ADD_LN_FWD = 3, LP_ATTN_BWD64 = 4 };
enum { LP_T64 = 0, LP_T128 = 1, LP_T128x64 = 2, LP_T64x128 = 3 };

struct LpOp {
  int kind;
  int rps;                 // rows of the op's row space per sentence
  // ---- gemm
  int tile, ta, tb, M, N, K, lda, ldb, vec_ok, pad_;
  const bf16_t* A; const bf16_t* B;
  GemmEpi epi;
  // ---- attention (forward: out/lse; backward: o, dout, dq, dk, dv)
  AttnArgs attn;
  bf16_t* out; float* lse; int ldo, nkt;
  const bf16_t* o_in; const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv; int lddo, lddq, lddk, lddv;
  // ---- residual + LayerNorm forward
  const bf16_t* x; const bf16_t* y; const float* gamma; const float* beta; bf16_t* ln_out; bf16_t* sum_out;
  float* mean; float* rstd; int H, maxc; float eps; uint32_t thr; float inv_keep; uint32_t sid; const uint64_t* seed;
};

constexpr int lp_max(int a, int b) { return a > b ? a : b; }
constexpr int LP_LDS_BYTES = lp_max(DldsCfg<64, 64, 4>::LDS_BYTES, lp_max(AttnFwdLds<2>::BYTES, ATTN_BWD64_LDS_BYTES));
static_assert(LP_LDS_BYTES + 16 <= 80 * 1024, "two workgroups per CU must fit in 160 KiB of LDS");

// barrier among the workgroups of one group; `target` = arrivals expected so far (monotonic counter).
// returns false when the launch is being aborted.
__device__ __forceinline__ bool lp_barrier(int* cnt, int target, int* flags, bool safe, volatile int* lp_ok, unsigned long long* st = nullptr) {
  if (st) st[2] = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores (and LDS-DMA) are complete
  __syncthreads();
  if (threadIdx.x == 0) {
    if (st) st[3] = __builtin_readcyclecounter();
    if (safe) {                                             // placement-independent form: write the L2 back first
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st) st[4] = __builtin_readcyclecounter();
    int ok = 1, spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255) == 0 && __hip_atomic_load(&flags[LP_FLAGS + 64 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
      if (spins > LP_SPIN_LIMIT) {
        __hip_atomic_store(&flags[LP_FLAGS + 64 + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    if (st) { st[5] = __builtin_readcyclecounter(); st[6] = spins; }
    if (safe) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // placement-independent form only (see header)
    *lp_ok = ok;
  }
  __syncthreads();
  return *lp_ok != 0;
}

template <int BM, int BN, int NS, bool TA, bool TB>
__device__ __forceinline__ void lp_gemm(unsigned char* smem, const LpOp& op, int row0, int row1, int j, int nwg) {
  const int tiles_m = (row1 - row0 + BM - 1) / BM, tiles_n = (op.N + BN - 1) / BN;
  for (int t = j; t < tiles_m * tiles_n; t += nwg) {
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    gemm_tile<BM, BN, NS, TA, TB, 4, 0, false, true>(smem, op.A, op.B, op.M, op.N, op.lda, op.ldb, 0, op.K, row0 + tm * BM, tn * BN,
                                        nullptr, op.epi, op.vec_ok);
    __syncthreads();           // the epilogue's LDS reads are done before the next tile's LDS-DMA lands
  }
}

// BWD = false: the ops of a forward chain (gemm, attention forward, residual + LayerNorm forward);
// BWD = true : the ops of a backward chain (gemm, attention backward).  Two kernels because the union needs more
// scalar registers than a wave has (182 scalar spills with both attention tile functions in one kernel).
template <bool BWD>
__global__ void __launch_bounds__(LP_THREADS, 2) k_layer_program(const LpOp* __restrict__ ops, int nops, int B,
                                                                int* __restrict__ state) {
  // the ONLY LDS object: the ops' tile buffers + one word behind them for the barrier's verdict
  __shared__ __attribute__((aligned(16))) unsigned char smem[LP_LDS_BYTES + 16];
  volatile int* lp_ok = reinterpret_cast<volatile int*>(smem + LP_LDS_BYTES);
  const int g = blockIdx.x & (LP_GROUPS - 1), j = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  int* cnt = state + g * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xf) != g) {
      __hip_atomic_store(&state[LP_FLAGS + g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&state[LP_FLAGS + 64], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  int arrivals = nwg;
  if (!lp_barrier(cnt, arrivals, state, true, lp_ok)) return;
  const bool safe = __hip_atomic_load(&state[LP_FLAGS + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  const int spg = (B + LP_GROUPS - 1) / LP_GROUPS;
  const int s0 = min(B, g * spg), s1 = min(B, s0 + spg);
  for (int p = 0; p < nops; ++p) {
    const LpOp& op = ops[p];
    const int row0 = s0 * op.rps, row1 = s1 * op.rps;
    if (op.kind == LP_GEMM) {
#define LP_G(T, BM_, BN_, NS_)                                                                             \
  if (op.tile == T) {                                                                                      \
    if (!BWD && !op.ta && !op.tb) lp_gemm<BM_, BN_, NS_, false, false>(smem, op, row0, row1, j, nwg);      \
    else if (BWD && !op.ta && op.tb) lp_gemm<BM_, BN_, NS_, false, true>(smem, op, row0, row1, j, nwg);    \
  }
      LP_G(LP_T64, 64, 64, 4)
#undef LP_G
    } else if (!BWD && op.kind == LP_ATTN_FWD) {
      const int qts = (op.attn.Lq + TQ - 1) / TQ, per_s = op.attn.nh * qts;
      for (int t = j; t < (s1 - s0) * per_s; t += nwg) {
        const int b = s0 + t / per_s, r = t % per_s, h = r / qts, qt = r % qts;
        attn_fwd_tile<1, true>(smem, op.attn, op.out, op.ldo, op.lse, qt, h, b);
        __syncthreads();
      }
    } else if (BWD && op.kind == LP_ATTN_BWD64) {
      for (int t = j; t < (s1 - s0) * op.attn.nh; t += nwg) {
        const int b = s0 + t / op.attn.nh, h = t % op.attn.nh;
        attn_bwd_fused64_tile(smem, op.attn, op.o_in, op.ldo, op.dout, op.lddo, op.lse, op.dq, op.lddq, op.dk, op.lddk,
                              op.dv, op.lddv, h, b);
        __syncthreads();
      }
    } else if (!BWD && op.kind == LP_ADD_LN_FWD) {
      const uint64_t seed = op.thr ? *op.seed : 0;
      const float invH = 1.f / (float)op.H;
      if (blockIdx.x == 0 && tid == 0 && p < 512) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reinterpret_cast<unsigned long long*>(state + 640)[8 * p + 7] = __builtin_readcyclecounter() + (seed & 1) + (unsigned long long)(invH > 2.f);
      }
      for (int r = row0 + j * 4 + w; r < row1; r += nwg * 4) {
        add_ln_fwd_row<2, true>(op.x, op.y, op.gamma, op.beta, op.ln_out, op.sum_out, op.mean, op.rstd, r, op.H, invH, op.eps, op.thr, op.inv_keep, seed, op.sid, lane);
      }
    }
    unsigned long long* st = (blockIdx.x == 0 && tid == 0 && p < 512) ? reinterpret_cast<unsigned long long*>(state + 640) + 8 * p : nullptr;
    if (st) st[0] = __builtin_readcyclecounter();
    if (p + 1 < nops) {
      arrivals += nwg;
      if (!lp_barrier(cnt, arrivals, state, safe, lp_ok, st)) return;
    }
    if (st) st[1] = __builtin_readcyclecounter();
  }
}

__global__ void k_lp_reset(int* state) {
  for (int i = threadIdx.x; i < 640; i += blockDim.x)
    state[i] = 0;
}

// ------------------------------------------------------------------------------------------------ recording
namespace {
struct Recorder {
  bool active = false, failed = false;
  int B = 0;
  char why[256] = {0};
  std::vector<LpOp> ops;
};
thread_local Recorder g_rec;

LpOp blank(int kind) {
  LpOp op;
  memset(&op, 0, sizeof(op));
  op.kind = kind;
  return op;
}
int rows_per_sentence(int rows, const char* what) {
  if (g_rec.B <= 0 || rows % g_rec.B != 0) {
    zk_prog_reject(what);
    return -1;
  }
  return rows / g_rec.B;
}
}  // namespace

bool zk_prog_active() { return g_rec.active; }

int zk_prog_reject(const char* why) {
  if (g_rec.active && !g_rec.failed) {
    g_rec.failed = true;
    snprintf(g_rec.why, sizeof(g_rec.why), "%s", why);
  }
  return zk_set_error(-2, "layer program: %s", why);
}

int zk_prog_record_gemm(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                        int bm, int bn, const GemmEpi& e) {
  if (ta) return zk_prog_reject("gemm with a transposed A operand (a sum over rows) is not sentence-local");
  // forward chains multiply by W [K, N] (tb = 0), backward chains by W^T (tb = 1): one gemm instance per kernel
  // keeps the code of a program kernel inside the instruction cache
  const int rps = rows_per_sentence(M, "gemm rows are not a multiple of the sentence count");
  if (rps < 0) return -2;
  // Every gemm of a program runs on 64x64 tiles whatever the launch-per-op path would pick (bm x bn): a group's 512
  // rows x 512 columns are exactly its 64 workgroups' worth of tiles, wider outputs are whole rounds of them, and
  // the value of an output element does not depend on the tile it is computed in (same MFMA, same K order).
  (void)bn;
  const int tile = LP_T64;
  bm = 64;
  const int spg = (g_rec.B + LP_GROUPS - 1) / LP_GROUPS;
  if ((spg * rps) % bm != 0) return zk_prog_reject("a group's rows are not a whole number of gemm tiles");
  LpOp op = blank(LP_GEMM);
  op.rps = rps; op.tile = tile; op.ta = ta; op.tb = tb; op.M = M; op.N = N; op.K = K; op.lda = lda; op.ldb = ldb;
  op.A = A; op.B = B; op.epi = e;
  const uintptr_t al = (uintptr_t)e.C | (uintptr_t)e.bias | (uintptr_t)e.res | (uintptr_t)e.aux;
  op.vec_ok = ((al & 15) == 0) && (e.ldc % 8 == 0) && (e.res == nullptr || e.ldr % 8 == 0) &&
              (e.aux == nullptr || e.ldaux % 8 == 0);
  g_rec.ops.push_back(op);
  return 0;
}

int zk_prog_record_attn_fwd(const AttnArgs& a, bf16_t* out, int ldo, float* lse, int nkt) {
  if (a.B != g_rec.B || a.kv_group != 1 || a.pos_dev != nullptr || a.gq != nullptr || nkt > 1)
    return zk_prog_reject("attention call outside what the layer program covers (batch, kv_group, decode, rpr, Lk > 128)");
  LpOp op = blank(LP_ATTN_FWD);
  op.rps = a.Lq; op.attn = a; op.out = out; op.ldo = ldo; op.lse = lse; op.nkt = nkt;
  g_rec.ops.push_back(op);
  return 0;
}

int zk_prog_record_attn_bwd64(const AttnArgs& a, const bf16_t* o, int ldo, const bf16_t* dout, int lddo,
                              const float* lse, bf16_t* dq, int lddq, bf16_t* dk, int lddk, bf16_t* dv, int lddv) {
  if (a.B != g_rec.B || a.kv_group != 1 || a.gq != nullptr)
    return zk_prog_reject("attention backward call outside what the layer program covers");
  LpOp op = blank(LP_ATTN_BWD64);
  op.rps = a.Lq; op.attn = a; op.o_in = o; op.ldo = ldo; op.dout = dout; op.lddo = lddo; op.lse = const_cast<float*>(lse);
  op.dq = dq; op.lddq = lddq; op.dk = dk; op.lddk = lddk; op.dv = dv; op.lddv = lddv;
  g_rec.ops.push_back(op);
  return 0;
}

int zk_prog_record_add_ln_fwd(const bf16_t* x, const bf16_t* y, const float* gamma, const float* beta, bf16_t* out,
                              bf16_t* sum_out, float* mean, float* rstd, int rows, int H, float eps, uint32_t thr,
                              float inv_keep, const uint64_t* seed, uint32_t sid) {
  if (H > 1024) return zk_prog_reject("LayerNorm wider than 1024 channels");
  const int rps = rows_per_sentence(rows, "LayerNorm rows are not a multiple of the sentence count");
  if (rps < 0) return -2;
  LpOp op = blank(LP_ADD_LN_FWD);
  op.rps = rps; op.x = x; op.y = y; op.gamma = gamma; op.beta = beta; op.ln_out = out; op.sum_out = sum_out;
  op.mean = mean; op.rstd = rstd; op.H = H; op.maxc = H <= 512 ? 1 : 2; op.eps = eps; op.thr = thr;
  op.inv_keep = inv_keep; op.seed = seed; op.sid = sid;
  g_rec.ops.push_back(op);
  return 0;
}

extern "C" {
size_t zk_prog_op_bytes(void) { return sizeof(LpOp); }
size_t zk_prog_state_bytes(void) { return LP_STATE_INTS * sizeof(int); }

// Start recording on the calling thread: until zk_prog_end, zk_gemm / zk_attn_fwd / zk_attn_bwd / zk_add_ln_fwd
// called from this thread append an op instead of launching (their stream argument is ignored).  `sentences` = B:
// every recorded op must act on a row space of B equal sentence blocks.
int zk_prog_begin(int sentences) {
  ZK_CHECK_ARG(!g_rec.active, "zk_prog_begin: a program is already being recorded on this thread");
  ZK_CHECK_ARG(sentences >= 1, "zk_prog_begin: sentences must be >= 1");
  g_rec.active = true; g_rec.failed = false; g_rec.B = sentences; g_rec.ops.clear(); g_rec.why[0] = 0;
  return 0;
}
// Stop recording.  ops_out (HOST buffer, cap_bytes) receives *nops records of zk_prog_op_bytes() each; the caller
// uploads them to device memory it owns and passes that to zk_prog_launch.  Returns -2 (and *nops = 0) when some call
// could not be recorded -- the caller then issues the same calls again as ordinary launches.
int zk_prog_end(void* ops_out, size_t cap_bytes, int* nops, int* is_backward) {
  ZK_CHECK_ARG(g_rec.active, "zk_prog_end: nothing is being recorded");
  g_rec.active = false;
  if (nops) *nops = 0;
  if (g_rec.failed) return zk_set_error(-2, "layer program not recordable: %s", g_rec.why);
  const size_t need = g_rec.ops.size() * sizeof(LpOp);
  ZK_CHECK_ARG(ops_out != nullptr && nops != nullptr && cap_bytes >= need, "zk_prog_end: buffer too small (%zu needed)", need);
  bool fwd = false, bwd = false;
  for (const LpOp& op : g_rec.ops) {
    if (op.kind == LP_ATTN_FWD || op.kind == LP_ADD_LN_FWD) fwd = true;
    if (op.kind == LP_ATTN_BWD64) bwd = true;
  }
  if (fwd && bwd) return zk_set_error(-2, "layer program mixes forward and backward ops");
  memcpy(ops_out, g_rec.ops.data(), need);
  *nops = (int)g_rec.ops.size();
  if (is_backward) *is_backward = bwd ? 1 : 0;
  return 0;
}
// ops_dev: device copy of the recorded ops; state_dev: zk_prog_state_bytes() of device memory owned by the caller
// (reset by a small kernel in front of every launch).  512 workgroups of 256 threads, two per CU.
int zk_prog_launch(const void* ops_dev, int nops, int sentences, int backward, void* state_dev, hipStream_t stream) {
  ZK_CHECK_ARG(ops_dev != nullptr && state_dev != nullptr && nops >= 1 && sentences >= 1, "zk_prog_launch: bad arguments");
  static int resident = -1;
  if (resident < 0) {
    int nf = 0, nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, k_layer_program<false>, LP_THREADS, 0) != hipSuccess) nf = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_layer_program<true>, LP_THREADS, 0) != hipSuccess) nb = 0;
    resident = nf < nb ? nf : nb;
  }
  // every workgroup must be resident at once (they wait for each other): two per CU or no launch
  ZK_CHECK_ARG(resident >= 2, "zk_prog_launch: only %d workgroup(s) of the layer program fit on a CU (2 needed)", resident);
  hipLaunchKernelGGL(k_lp_reset, dim3(1), dim3(256), 0, stream, (int*)state_dev);
  ZK_LAUNCH_CHECK();
  if (backward)
    hipLaunchKernelGGL(k_layer_program<true>, dim3(LP_GROUPS * LP_WG_PER_GROUP), dim3(LP_THREADS), 0, stream,
                       (const LpOp*)ops_dev, nops, sentences, (int*)state_dev);
  else
    hipLaunchKernelGGL(k_layer_program<false>, dim3(LP_GROUPS * LP_WG_PER_GROUP), dim3(LP_THREADS), 0, stream,
                       (const LpOp*)ops_dev, nops, sentences, (int*)state_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}
}  // extern "C"
