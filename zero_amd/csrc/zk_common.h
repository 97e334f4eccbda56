// zk_common.h -- device helpers shared by every kernel of libzero_hip.so (gfx950 only).
//
// Conventions
//   * activations and matrix weights are bf16 stored as raw uint16 ("bf16_t");
//   * vectors (bias, LayerNorm scale/offset), statistics, losses, gradients of
//     parameters and optimizer state are fp32;
//   * every kernel is launched on the stream handed through the C-ABI; nothing
//     here allocates, synchronises or keeps global state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

#define ZK_WAVE 64

// ---------------------------------------------------------------- error plumbing
extern thread_local char zk_err_buf[512];
int zk_set_error(int code, const char* fmt, ...);
#define ZK_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return zk_set_error(-1, __VA_ARGS__);            \
  } while (0)
#define ZK_LAUNCH_CHECK()                                         \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess)                                        \
      return zk_set_error((int)e__, "%s:%d launch failed: %s",    \
                          __FILE__, __LINE__, hipGetErrorString(e__)); \
  } while (0)

// ---------------------------------------------------------------- bf16 <-> fp32
__device__ __forceinline__ float bf2f(bf16_t x) {
  return __uint_as_float(((uint32_t)x) << 16);
}
// round-to-nearest-even; NaN stays NaN.  gfx950 converts two values per instruction (v_cvt_pk_bf16_f32: the same
// rounding as the seven-instruction integer sequence of rounds 1-5 for every finite value and infinity; a NaN comes out as
// a quiet NaN with the hardware's payload) -- round 6: every epilogue, the cross entropy's 131 M and Adam's 77 M
// conversions per step are one instruction per pair
typedef __bf16 zk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float zk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  zk_f32x2 v; v.x = lo; v.y = hi;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, zk_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  return (bf16_t)(pack2bf(f, 0.f) & 0xffffu);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// ---------------------------------------------------------------- loads of data another CU wrote during THIS launch
// A CU's vector L1 is never refreshed by another CU's stores (MI355X_MICROARCH.md, inter-workgroup visibility).  Inside
// the persistent layer program (zk_layer.hip) an op reads what other workgroups of its XCD stored one barrier earlier:
// FRESH = true issues the load with the nt policy, which bypasses the L1 and is served by the XCD's (coherent) L2, so
// the barrier needs no L1 invalidate (~5 us per barrier at two workgroups per CU).  FRESH = false: ordinary load.
typedef unsigned int zk_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int zk_u32x2 __attribute__((ext_vector_type(2)));
typedef float zk_f32x4 __attribute__((ext_vector_type(4)));
template <bool FRESH>
__device__ __forceinline__ uint4 zk_ld16(const void* p) {
  if (FRESH) {
    const zk_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const zk_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const uint4*>(p);
}
template <bool FRESH>
__device__ __forceinline__ float zk_ld_f32(const float* p) {
  return FRESH ? __builtin_nontemporal_load(p) : *p;
}

// ---------------------------------------------------------------- wave / block reductions
// Cross-lane steps are DPP moves (a VALU operand modifier, a few cycles each), not __shfl_xor: that compiles to
// ds_bpermute_b32, one LDS round trip (~100+ cycles) per step, and a 64-lane reduction is six DEPENDENT steps -- ~0.3 us
// of every row-wise kernel whose whole launch is a 5 us latency chain.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float zk_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define ZK_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define ZK_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define ZK_DPP_HALF_MIRROR 0x141  // lane i <- lane 7-i of its half row: the other quad
#define ZK_DPP_MIRROR 0x140       // lane i <- lane 15-i of its row: the other half
#define ZK_DPP_BCAST15 0x142      // lane 15 of every row -> the next row
#define ZK_DPP_BCAST31 0x143      // lane 31 -> rows 2 and 3
// sum / max over the 4 lanes of a quad (lane ^ 1, lane ^ 2) and over the 16 lanes of a DPP row (lane >> 4 fixed);
// every lane of the group receives the result; the pairing tree is the xor-butterfly's (1, 2, 4, 8)
__device__ __forceinline__ float quad_sum(float v) {
  v += zk_dpp<ZK_DPP_XOR1, 0xf>(0.f, v);
  v += zk_dpp<ZK_DPP_XOR2, 0xf>(0.f, v);
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v = quad_sum(v);
  v += zk_dpp<ZK_DPP_HALF_MIRROR, 0xf>(0.f, v);
  v += zk_dpp<ZK_DPP_MIRROR, 0xf>(0.f, v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, zk_dpp<ZK_DPP_XOR1, 0xf>(v, v));
  v = fmaxf(v, zk_dpp<ZK_DPP_XOR2, 0xf>(v, v));
  v = fmaxf(v, zk_dpp<ZK_DPP_HALF_MIRROR, 0xf>(v, v));
  v = fmaxf(v, zk_dpp<ZK_DPP_MIRROR, 0xf>(v, v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += zk_dpp<ZK_DPP_BCAST15, 0xa>(0.f, v);      // rows 1, 3 += rows 0, 2
  v += zk_dpp<ZK_DPP_BCAST31, 0xc>(0.f, v);      // rows 2, 3 += row 1: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, zk_dpp<ZK_DPP_BCAST15, 0xa>(v, v));
  v = fmaxf(v, zk_dpp<ZK_DPP_BCAST31, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int zk_dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
// arg-best over the 64 lanes for a strict total order `better(s2, i2, s, i)` ((s2, i2) beats (s, i)); extra payload t
// travels with the winner.  Every lane receives the winner (same selection as the xor butterfly: exact, no rounding).
template <typename B>
__device__ __forceinline__ void wave_argbest(float& s, int& i, int& t, B better) {
#define ZK_ARG_STEP(CTRL, MASK)                                                       \
  {                                                                                   \
    const float s2 = zk_dpp<CTRL, MASK>(s, s);                                        \
    const int i2 = zk_dpp_i<CTRL, MASK>(i, i), t2 = zk_dpp_i<CTRL, MASK>(t, t);       \
    if (better(s2, i2, s, i)) { s = s2; i = i2; t = t2; }                             \
  }
  ZK_ARG_STEP(ZK_DPP_XOR1, 0xf)
  ZK_ARG_STEP(ZK_DPP_XOR2, 0xf)
  ZK_ARG_STEP(ZK_DPP_HALF_MIRROR, 0xf)
  ZK_ARG_STEP(ZK_DPP_MIRROR, 0xf)
  ZK_ARG_STEP(ZK_DPP_BCAST15, 0xa)
  ZK_ARG_STEP(ZK_DPP_BCAST31, 0xc)
#undef ZK_ARG_STEP
  s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
  i = __builtin_amdgcn_readlane(i, 63);
  t = __builtin_amdgcn_readlane(t, 63);
}

// ---------------------------------------------------------------- LayerNorm statistics as per-64-column partials (round 4)
// (mu, rstd) of row `row` from its per-64-column partials {sum, M2}: Chan's combination, robust for |mean| >> sigma
#define ZK_LN_MAXP 16
__device__ __forceinline__ void zk_ln_row_stats(const float* __restrict__ part, int np, float invh, float eps, size_t row,
                                                float& mu, float& rs) {
  const float4* p = reinterpret_cast<const float4*>(part + row * (size_t)np * 2);
  float4 v[ZK_LN_MAXP / 2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ZK_LN_MAXP / 2; ++i) {
    if (2 * i < np) { v[i] = p[i]; s += v[i].x + v[i].z; }
  }
  mu = s * invh;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < ZK_LN_MAXP / 2; ++i) {
    if (2 * i < np) {
      const float d0 = v[i].x * (1.f / 64.f) - mu, d1 = v[i].z * (1.f / 64.f) - mu;
      m2 += v[i].y + v[i].w + 64.f * (d0 * d0 + d1 * d1);
    }
  }
  rs = rsqrtf(m2 * invh + eps);
}
// the same for a row the whole WAVE works on: lane i < np loads partial i (one coalesced load instead of np/2 wave-wide
// broadcast loads, each of which costs the address path as much as a full row), two 16-lane DPP reductions
__device__ __forceinline__ void zk_ln_row_stats_wave(const float* __restrict__ part, int np, float invh, float eps, size_t row,
                                                     int lane, float& mu, float& rs) {
  float2 p = make_float2(0.f, 0.f);
  if (lane < np) p = *reinterpret_cast<const float2*>(part + (row * (size_t)np + lane) * 2);
  const float tot = row16_sum(p.x);
  mu = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tot))) * invh;
  const float d = p.x * (1.f / 64.f) - mu;
  const float m2 = row16_sum(lane < np ? p.y + 64.f * d * d : 0.f);
  rs = rsqrtf(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m2))) * invh + eps);
}
// sum over the 8 consecutive lanes (aligned to 8) that hold one row's 64-column group; all 8 receive it
__device__ __forceinline__ float zk_sum8(float v) {
  v = quad_sum(v);
  v += zk_dpp<ZK_DPP_HALF_MIRROR, 0xf>(0.f, v);
  return v;
}

// sum over a block of NW waves; every thread gets the result. sm: >= NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += sm[i];
  return r;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  float r = sm[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, sm[i]);
  return r;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep-decision for element `idx` of dropout site `sid` at step seed `seed`:
// two rounds of a 32-bit multiply-xorshift mixer over (seed, sid, idx).  Stateless, so the
// backward kernels regenerate the same mask instead of storing it.
__device__ __forceinline__ uint32_t zk_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t zk_rand_u32(uint64_t seed, uint32_t sid, uint64_t idx) {
  // a and b0 depend on (seed, sid) only: the compiler hoists them out of the per-element loops, so an
  // element with idx < 2^32 (every dropout site of the path) costs ONE mixer round; the rare high part
  // takes the extra round.  Values are identical to mixing (seed_hi + idx_hi * C + a) unconditionally.
  const uint32_t a = zk_mix32((uint32_t)seed ^ (sid * 0x9E3779B1u));
  const uint32_t hi = (uint32_t)(seed >> 32);
  uint32_t b = zk_mix32(hi + a);
  const uint32_t ih = (uint32_t)(idx >> 32);
  if (__builtin_expect(ih != 0u, 0)) b = zk_mix32(hi + ih * 0x85EBCA77u + a);
  return zk_mix32(((uint32_t)idx) * 0xC2B2AE3Du + b);
}
// Dropout decisions: the elements 2m and 2m+1 of a site share ONE mixer round (32 bits: 16 each, compared with the upper
// 16 bits of thr = p * 2^32 -- the keep probability is exact to 2^-16) and the element index enters the mixer by addition
// (the mixer avalanches sequential inputs).  Round 4: the hash was ~0.16 ms of VALU time per training step (300 M
// decisions x three quarter-rate 32-bit multiplies); kernels that hold 8 consecutive elements use zk_drop_scale8 (four
// rounds for eight decisions).  Stateless: the backward kernels regenerate the mask instead of storing it.
__device__ __forceinline__ uint32_t zk_drop_bits(uint64_t seed, uint32_t sid, uint64_t pair) {
  const uint32_t a = zk_mix32((uint32_t)seed ^ (sid * 0x9E3779B1u));      // (seed, sid) only: hoisted out of element loops
  const uint32_t hi = (uint32_t)(seed >> 32);
  uint32_t b = zk_mix32(hi + a);
  const uint32_t ih = (uint32_t)(pair >> 32);
  if (__builtin_expect(ih != 0u, 0)) b = zk_mix32(hi + ih * 0x85EBCA77u + a);
  return zk_mix32((uint32_t)pair + b);
}
// returns the multiplier to apply: 0 if dropped, 1/(1-p) if kept.  thr = p * 2^32 (clamped).
__device__ __forceinline__ float zk_drop_scale(uint64_t seed, uint32_t sid, uint64_t idx,
                                               uint32_t thr, float inv_keep) {
  const uint32_t r = zk_drop_bits(seed, sid, idx >> 1);
  const uint32_t h = (idx & 1) ? (r >> 16) : (r & 0xffffu);
  return h >= (thr >> 16) ? inv_keep : 0.f;
}
// the same decisions for the 8 consecutive elements base .. base + 7 (same values as eight zk_drop_scale calls)
__device__ __forceinline__ void zk_drop_scale8(uint64_t seed, uint32_t sid, uint64_t base, uint32_t thr, float inv_keep,
                                               float (&m)[8]) {
  if (base & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = zk_drop_scale(seed, sid, base + j, thr, inv_keep);
    return;
  }
  const uint32_t t16 = thr >> 16;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t r = zk_drop_bits(seed, sid, (base >> 1) + k);
    m[2 * k] = (r & 0xffffu) >= t16 ? inv_keep : 0.f;
    m[2 * k + 1] = (r >> 16) >= t16 ? inv_keep : 0.f;
  }
}
static inline uint32_t zk_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// the hipGraph an executable of zk_graph_end was instantiated from (kept alive until zk_graph_destroy); zk_elem.hip
hipGraph_t zk_graph_template_of(void* exec);
// drops what zk_prep.hip remembers about an executable's nodes (zk_graph_destroy calls it: handles are reused)
void zk_graph_forget_nodes(void* exec);
