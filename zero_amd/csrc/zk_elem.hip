// zk_elem.hip -- HBM-bound kernels of the Transformer hot path (gfx950).
//
// Every kernel here is bandwidth-bound: 16-byte vector accesses, one wave64 per
// token row with shuffle reductions, fp32 math on bf16 storage.  Reference
// semantics are cited per kernel (paths relative to bzhangGo/zero).
#include "zk_common.h"
#include "zk_ln_dev.h"
#include "zk_prog.h"
#ifdef ZK_STEP_STAMPS   // make STEPSTAMPS=1 (scripts/step_stamps.py): 100 MHz-clock stamps at the head and the tail of a training step
__device__ unsigned long long zk_step_stamp_buf[2 * 4096];
__device__ unsigned int zk_step_stamp_n[2];
extern "C" int zk_step_stamps_read(unsigned long long* out, unsigned int* n2) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_step_stamp_buf), sizeof(unsigned long long) * 2 * 4096) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(n2, HIP_SYMBOL(zk_step_stamp_n), sizeof(unsigned int) * 2);
}
extern "C" int zk_step_stamps_reset(void) {
  unsigned int z[2] = {0, 0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(zk_step_stamp_n), z, sizeof(z));
}
#endif
#include <cstring>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <unordered_map>

thread_local char zk_err_buf[512] = {0};
int zk_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(zk_err_buf, sizeof(zk_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

#define MAXC_LIMIT 4  // row kernels keep up to 4*64*8 = 2048 channels in registers (templated on the slab count)

// =====================================================================================
// K1  embedding + sqrt(H) scale + shared bias + timing signal (+ dropout)
//     transformer.py:16-33 (encoder), 88-119 (decoder: shifted, zero first input,
//     decode-time zeroing when every fed id is pad), func.py:341-369 (timing table is
//     precomputed on the host: [Lmax, H], first H/2 sin, last H/2 cos).
// =====================================================================================
struct EmbedFwdSide {
  const int* ids; const bf16_t* table; const float* bias; const float* timing; bf16_t* out;
  int rows, L, H; float scale; int shift, pos0; uint32_t thr; float inv_keep; uint32_t sid;
};

// one row of transformer.py:16-33 / 88-119 by one wave: out[r] = dropout(table[id] * scale + bias + timing[pos0 + r % L])
__device__ __forceinline__ void embed_fwd_row(const EmbedFwdSide& a, int r, int lane, bool zero_all, uint64_t seed) {
  const int t = r % a.L, H = a.H;
  int id = -1;
  if (!zero_all) {
    if (a.shift) { if (t > 0) id = a.ids[r - 1]; }
    else id = a.ids[r];
  }
  const float* tim = a.timing + (size_t)(a.pos0 + t) * H;
  for (int c = lane * 8; c < H; c += 64 * 8) {
    float v[8];
    if (id >= 0) {
      uint4 e = *reinterpret_cast<const uint4*>(a.table + (size_t)id * H + c);
      unpack8(e, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * a.scale + a.bias[c + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += tim[c + j];
    if (a.thr != 0) {
      float dm[8];
      zk_drop_scale8(seed, a.sid, (uint64_t)r * H + c, a.thr, a.inv_keep, dm);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= dm[j];
    }
    *reinterpret_cast<uint4*>(a.out + (size_t)r * H + c) = pack8(v);
  }
}

__global__ void __launch_bounds__(256) k_embed_fwd(
    const int* __restrict__ ids, const bf16_t* __restrict__ table, const float* __restrict__ bias,
    const float* __restrict__ timing, bf16_t* __restrict__ out, int rows, int L, int H,
    float scale, int shift, int pos0, const int* __restrict__ zero_flag, uint32_t thr,
    float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid, const int* __restrict__ pos0_dev) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
#ifdef ZK_STEP_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned int i_ = atomicAdd(&zk_step_stamp_n[0], 1u);
    if (i_ < 4096) zk_step_stamp_buf[i_] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const uint64_t seed = (thr != 0) ? *seedp : 0;
  const bool zero_all = (zero_flag != nullptr) && (*zero_flag != 0);
  if (pos0_dev != nullptr) pos0 = *pos0_dev;   // decode step read at run time (captured graphs)
  const EmbedFwdSide a{ids, table, bias, timing, out, rows, L, H, scale, shift, pos0, thr, inv_keep, sid};
  for (int r = wave; r < rows; r += nwaves) embed_fwd_row(a, r, lane, zero_all, seed);
}

// Round 6: the encoder's and the decoder's input embeddings of a training step in ONE launch (both depend on the ids
// alone; they were two ~8 us launches, the second one in the middle of the step): the first waves take side a's rows, the
// rest side b's.  Row by row the arithmetic of k_embed_fwd.
__global__ void __launch_bounds__(256) k_embed_fwd_pair(EmbedFwdSide a, EmbedFwdSide b, const uint64_t* __restrict__ seedp,
                                                        uint32_t* __restrict__ epoch) {
  // (the step's first launch also advances the epoch word of the in-launch LayerNorm exchanges -- zk_ln_epoch_bump's
  // one-thread launch of its own; nothing of this launch reads the word, every later launch runs behind it)
  if (epoch != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t v = *epoch + 1; *epoch = (v & 0xffffffu) ? v : 1; }
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
#ifdef ZK_STEP_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned int i_ = atomicAdd(&zk_step_stamp_n[0], 1u);
    if (i_ < 4096) zk_step_stamp_buf[i_] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const uint64_t seed = (a.thr != 0 || b.thr != 0) ? *seedp : 0;
  for (int r = wave; r < a.rows + b.rows; r += nwaves) {
    if (r < a.rows) embed_fwd_row(a, r, lane, false, seed);
    else embed_fwd_row(b, r - a.rows, lane, false, seed);
  }
}

// backward: dtable[id] += scale * dout (*dropmask); dbias += dout (*dropmask) on rows that
// had an embedding (transformer.py:29-30,104-110).  fp32 hardware atomics.
__global__ void __launch_bounds__(256) k_embed_bwd(
    const int* __restrict__ ids, const bf16_t* __restrict__ dout, float* __restrict__ dtable,
    float* __restrict__ dbias, int rows, int L, int H, float scale, int shift, uint32_t thr,
    float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t seed = (thr != 0) ? *seedp : 0;
  for (int c = lane * 8; c < H; c += 64 * 8) {
    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = wave; r < rows; r += nwaves) {
      const int t = r % L;
      int id;
      if (shift) { if (t == 0) continue; id = ids[r - 1]; }
      else id = ids[r];
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(dout + (size_t)r * H + c), v);
      if (thr != 0) {
        float dm[8];
        zk_drop_scale8(seed, sid, (uint64_t)r * H + c, thr, inv_keep, dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= dm[j];
      }
      float* dst = dtable + (size_t)id * H + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsafeAtomicAdd(dst + j, v[j] * scale);
        bsum[j] += v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) unsafeAtomicAdd(dbias + c + j, bsum[j]);
  }
}

// =====================================================================================
// K4  residual (+dropout) + LayerNorm, forward:  s = x + drop(y);  out = LN(s)
//     func.py:321-324 (residual_fn), func.py:289-303 (layer_norm: biased variance,
//     eps=1e-8 inside rsqrt), post-LN order of transformer.py:57-58.
//     s is stored (bf16) for the backward; statistics are taken from the stored values.
// =====================================================================================
template <int MAXC>
__global__ void __launch_bounds__(256) k_add_ln_fwd(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out, bf16_t* __restrict__ sum_out,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int H, float eps,
    uint32_t thr, float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t seed = (thr != 0) ? *seedp : 0;
  const float invH = 1.f / (float)H;
  for (int r = wave; r < rows; r += nwaves)
    add_ln_fwd_row<MAXC>(x, y, gamma, beta, out, sum_out, mean_out, rstd_out, r, H, invH, eps, thr, inv_keep, seed, sid, lane);
}

// backward of the above.  Per row: xhat=(s-mean)*rstd, g=dout*gamma,
//   ds = rstd*(g - mean(g) - xhat*mean(g*xhat));  dy = ds * dropmask/(1-p).
// Per column (reduced over rows): dgamma=sum dout*xhat, dbeta=sum dout, dbias_prev=sum dy
// (the bias of the linear layer that produced y).  Stage 1 writes one partial row per
// block to `partials` [gridDim.x][3][H]; k_partials_reduce finishes.
template <int MAXC>
__global__ void __launch_bounds__(256) k_add_ln_bwd(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ s, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, bf16_t* __restrict__ dsum,
    bf16_t* __restrict__ dy, float* __restrict__ partials, int rows, int H, uint32_t thr,
    float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid, const float* __restrict__ part, int np, float eps,
    const float* __restrict__ beta, bf16_t* __restrict__ y_out) {
  __shared__ float red[4][3][8 * 64];  // per wave, per quantity, one 512-column slab at a time
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + w;
  const int nwaves = gridDim.x * 4;
  const uint64_t seed = (thr != 0) ? *seedp : 0;
  const float invH = 1.f / (float)H;
  float acc[3][MAXC][8];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q][i][j] = 0.f;

  for (int r = wave; r < rows; r += nwaves) {
    float mu, rs;
    if (part != nullptr) zk_ln_row_stats_wave(part, np, invH, eps, (size_t)r, lane, mu, rs);   // lazy LayerNorm: statistics left by the producing GEMM
    else { mu = mean[r]; rs = rstd[r]; }
    float xh[MAXC][8], g[MAXC][8];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
        float d[8], sv[8];
        unpack8(*reinterpret_cast<const uint4*>(dout + (size_t)r * H + c), d);
        unpack8(*reinterpret_cast<const uint4*>(s + (size_t)r * H + c), sv);
        if (y_out != nullptr) {     // the normalised rows the forward never wrote: operand of the deferred weight gradients
          float yv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) yv[j] = gamma[c + j] * (sv[j] - mu) * rs + beta[c + j];
          *reinterpret_cast<uint4*>(y_out + (size_t)r * H + c) = pack8(yv);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (sv[j] - mu) * rs;
          g[i][j] = d[j] * gamma[c + j];
          sg += g[i][j];
          sgx += g[i][j] * xh[i][j];
          acc[0][i][j] += d[j] * xh[i][j];
          acc[1][i][j] += d[j];
        }
      }
    }
    const float mg = wave_sum(sg) * invH;
    const float mgx = wave_sum(sgx) * invH;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
        float o[8], oy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - mg - xh[i][j] * mgx);
        uint4 p = pack8(o);
        *reinterpret_cast<uint4*>(dsum + (size_t)r * H + c) = p;
        unpack8(p, o);  // dy derives from the stored (rounded) ds
        if (thr != 0) {
          float dm[8];
          zk_drop_scale8(seed, sid, (uint64_t)r * H + c, thr, inv_keep, dm);
#pragma unroll
          for (int j = 0; j < 8; ++j) oy[j] = o[j] * dm[j];
          if (dy != nullptr) {
            uint4 py = pack8(oy);
            *reinterpret_cast<uint4*>(dy + (size_t)r * H + c) = py;
            unpack8(py, oy);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) oy[j] = o[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[2][i][j] += oy[j];
      }
    }
  }
  // cross-wave reduction, one 512-column slab (i) at a time
  for (int i = 0; i < MAXC; ++i) {
    if (i * 512 >= H) break;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[w][q][lane * 8 + j] = acc[q][i][j];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * 512; idx += 256) {
      const int q = idx / 512, cc = idx % 512;
      const int c = i * 512 + cc;
      if (c < H) {
        const float t = red[0][q][cc] + red[1][q][cc] + red[2][q][cc] + red[3][q][cc];
        partials[((size_t)blockIdx.x * 3 + q) * H + c] = t;
      }
    }
  }
}

// Wide variant for H <= 512: 1024 threads = 16 waves, every wave owns whole rows (8 channels per
// lane), so 16 rows are in flight per block and the per-row latency chain (two wave reductions)
// is overlapped 16 ways; column partials are reduced across the 16 waves through LDS, one
// quantity at a time.
// make ATTNTRACE=1: block 0 of the wide LayerNorm backward stamps the 100 MHz clock at its phase boundaries (scripts/ln_bwd_trace.py)
#ifdef ZK_ATTN_TRACE
__device__ unsigned long long zk_ln_trace_buf[8];
extern "C" int zk_ln_trace_read(unsigned long long* out8) {
  return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(zk_ln_trace_buf), sizeof(unsigned long long) * 8);
}
#define ZK_LT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) zk_ln_trace_buf[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ZK_LT(i)
#endif
__global__ void __launch_bounds__(1024) k_add_ln_bwd_wide(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ s, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, bf16_t* __restrict__ dsum,
    bf16_t* __restrict__ dy, float* __restrict__ partials, int rows, int H, uint32_t thr,
    float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid, const float* __restrict__ part, int np, float eps,
    const float* __restrict__ beta, bf16_t* __restrict__ y_out) {
  __shared__ float red[3][16][512 + 8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane * 8;
  const bool on = c < H;
  const uint64_t seed = (thr != 0) ? *seedp : 0;
  const float invH = 1.f / (float)H;
  float acc[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
  float gam[8];
  ZK_LT(0);
#pragma unroll
  for (int j = 0; j < 8; ++j) gam[j] = on ? gamma[c + j] : 0.f;
  float bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bet[j] = (on && y_out != nullptr) ? beta[c + j] : 0.f;
  for (int r = blockIdx.x * 16 + w; r < rows; r += gridDim.x * 16) {
    // the row's data is requested BEFORE the statistics are combined (the combination waits for its own loads: behind
    // it, the two round trips would add up)
    uint4 raw_d = make_uint4(0, 0, 0, 0), raw_s = make_uint4(0, 0, 0, 0);
    if (on) {
      raw_d = *reinterpret_cast<const uint4*>(dout + (size_t)r * H + c);
      raw_s = *reinterpret_cast<const uint4*>(s + (size_t)r * H + c);
    }
    float mu, rs;
    if (part != nullptr) zk_ln_row_stats_wave(part, np, invH, eps, (size_t)r, lane, mu, rs);   // lazy LayerNorm: statistics left by the producing GEMM
    else { mu = mean[r]; rs = rstd[r]; }
    float xh[8], g[8], d[8];
    float sg = 0.f, sgx = 0.f;
    if (on) {
      float sv[8];
      unpack8(raw_d, d);
      unpack8(raw_s, sv);
      if (y_out != nullptr) {       // the normalised rows the forward never wrote: operand of the deferred weight gradients
        float yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) yv[j] = gam[j] * (sv[j] - mu) * rs + bet[j];
        *reinterpret_cast<uint4*>(y_out + (size_t)r * H + c) = pack8(yv);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = (sv[j] - mu) * rs;
        g[j] = d[j] * gam[j];
        sg += g[j];
        sgx += g[j] * xh[j];
        acc[0][j] += d[j] * xh[j];
        acc[1][j] += d[j];
      }
    }
    ZK_LT(1);
    const float mg = wave_sum(sg) * invH;
    const float mgx = wave_sum(sgx) * invH;
    ZK_LT(2);
    if (on) {
      float o[8], oy[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rs * (g[j] - mg - xh[j] * mgx);
      uint4 p = pack8(o);
      *reinterpret_cast<uint4*>(dsum + (size_t)r * H + c) = p;
      unpack8(p, o);
      if (thr != 0) {
        float dm[8];
        zk_drop_scale8(seed, sid, (uint64_t)r * H + c, thr, inv_keep, dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) oy[j] = o[j] * dm[j];
        if (dy != nullptr) {
          uint4 py = pack8(oy);
          *reinterpret_cast<uint4*>(dy + (size_t)r * H + c) = py;
          unpack8(py, oy);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) oy[j] = o[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[2][j] += oy[j];
    }
  }
  // the three column partials of the 16 waves through LDS in ONE pass (one barrier instead of six)
  ZK_LT(3);
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[q][w][c + j] = acc[q][j];
  __syncthreads();
  ZK_LT(4);
  for (int e = threadIdx.x; e < 3 * H; e += 1024) {
    const int q = e / H, col = e - q * H;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[q][i][col];
    partials[((size_t)blockIdx.x * 3 + q) * H + col] = t;
  }
  ZK_LT(5);
}

// out_q[c] = sum_b partials[b][q][c]   (q < nq; out pointers may be null)
// block = 16 columns of one quantity x 16 row groups (short dependent chains, 64-B segments).
__global__ void __launch_bounds__(256) k_partials_reduce(const float* __restrict__ partials, int nblk,
                                                         int nq, int H, float* o0, float* o1,
                                                         float* o2, int accumulate) {
  __shared__ float red[16][17];
  const int q = blockIdx.y;
  float* o = q == 0 ? o0 : (q == 1 ? o1 : o2);
  if (o == nullptr) return;
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float t = 0.f;
  if (c < H)
    for (int b = g; b < nblk; b += 16) t += partials[((size_t)b * nq + q) * H + c];
  red[g][cl] = t;
  __syncthreads();
  if (g == 0 && c < H) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][cl];
    o[c] = accumulate ? o[c] + v : v;
  }
}

// =====================================================================================
// column sum of a bf16 [rows, N] matrix (bias gradients; func.py:58-60 bias_add backward)
// stage 1: block = 64 columns x row-chunk -> partials[gridDim.y][N]
// =====================================================================================
__global__ void __launch_bounds__(256) k_colsum(const bf16_t* __restrict__ a, int rows, int N, int lda,
                                                float* __restrict__ partials, int skip_L, uint32_t thr,
                                                float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid) {
  __shared__ float red[32][64 + 1];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + tx * 8;
  const int rpb = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rpb;
  const int r1 = min(rows, r0 + rpb);
  const uint64_t seed = thr ? *seedp : 0;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < N) {
    for (int r = r0 + ty; r < r1; r += 32) {
      if (skip_L > 0 && (r % skip_L) == 0) continue;   // rows without an embedding (shifted input)
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(a + (size_t)r * lda + c0), v);
      if (thr) {
        float dm[8];
        zk_drop_scale8(seed, sid, (uint64_t)r * N + c0, thr, inv_keep, dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= dm[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) {
      float t = 0.f;
      for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
      partials[(size_t)blockIdx.y * N + c] = t;
    }
  }
}

// Round 6: two column sums into ONE output (the shared input bias `bias` receives rows from the decoder's input gradient --
// without the shifted rows, skip_L -- and from the encoder's): blockIdx.z = the side, partial rows of side b behind side
// a's in the workspace; one k_partials_reduce then adds all of them in that order.
struct ColsumSide { const bf16_t* a; int rows, lda, skip_L; uint32_t sid; int gy; };
__global__ void __launch_bounds__(256) k_colsum_pair(ColsumSide sa, ColsumSide sb, int N, float* __restrict__ partials, uint32_t thr,
                                                     float inv_keep, const uint64_t* __restrict__ seedp) {
  __shared__ float red[32][64 + 1];
  const ColsumSide& s_ = blockIdx.z == 0 ? sa : sb;
  if ((int)blockIdx.y >= s_.gy) return;
  const bf16_t* __restrict__ a = s_.a;
  const int rows = s_.rows, lda = s_.lda, skip_L = s_.skip_L;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + tx * 8;
  const int rpb = (rows + s_.gy - 1) / s_.gy;
  const int r0 = blockIdx.y * rpb;
  const int r1 = min(rows, r0 + rpb);
  const uint64_t seed = thr ? *seedp : 0;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < N) {
    for (int r = r0 + ty; r < r1; r += 32) {
      if (skip_L > 0 && (r % skip_L) == 0) continue;
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(a + (size_t)r * lda + c0), v);
      if (thr) {
        float dm[8];
        zk_drop_scale8(seed, s_.sid, (uint64_t)r * N + c0, thr, inv_keep, dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= dm[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) {
      float t = 0.f;
      for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
      partials[(size_t)((blockIdx.z == 0 ? 0 : sa.gy) + blockIdx.y) * N + c] = t;
    }
  }
}

// ---- grouped column reductions: every bias gradient / LayerNorm parameter gradient of a group of
// layers in TWO launches instead of two per tensor (each is a few microseconds of work).
// stage 1 (bf16 matrices -> per-row-chunk partial sums), one block = 64 columns x one row chunk
struct ColsumDesc {
  const bf16_t* a; float* partials;   // partials: [gy][N] fp32, private to this problem
  int rows, N, lda, gy, block_start, pad_;
};
__global__ void __launch_bounds__(256) k_colsum_grouped(const ColsumDesc* __restrict__ descs, int nprob) {
  __shared__ float red[32][64 + 1];
  const int bid = blockIdx.x;
  int p = 0;
  {
    int hi = nprob - 1;              // binary search: see k_reduce_grouped
    while (p < hi) {
      const int mid = (p + hi + 1) >> 1;
      if (descs[mid].block_start <= bid) p = mid; else hi = mid - 1;
    }
  }
  const ColsumDesc d = descs[p];
  const int local = bid - d.block_start;
  const int bx = local / d.gy, by = local - bx * d.gy;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = bx * 64 + tx * 8;
  const int rpb = (d.rows + d.gy - 1) / d.gy;
  const int r0 = by * rpb, r1 = min(d.rows, r0 + rpb);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < d.N) {
    for (int r = r0 + ty; r < r1; r += 32) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(d.a + (size_t)r * d.lda + c0), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = bx * 64 + threadIdx.x;
    if (c < d.N) {
      float t = 0.f;
      for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
      d.partials[(size_t)by * d.N + c] = t;
    }
  }
}
// stage 2: out[c] = sum_b partials[b][q][c]; one block = ZK_RED_COLS columns of one quantity of one problem (32 columns:
// every row of partials is read as a full 128-byte line; with 16 the L2 fetched twice the bytes it delivered)
#define ZK_RED_COLS 32
struct ReduceDesc {
  const float* partials; float* out[3];
  int nblk, nq, H, block_start;
  int bstride, qstride;       // element strides between blocks / quantities; 0 = packed (nq * H, H)
};
__global__ void __launch_bounds__(256) k_reduce_grouped(const ReduceDesc* __restrict__ descs, int nprob) {
  constexpr int NG = 256 / ZK_RED_COLS;
  __shared__ float red[NG][ZK_RED_COLS + 1];
  const int bid = blockIdx.x;
  int p = 0;
  {   // the last problem whose first block is <= bid: binary search (a linear scan is one dependent global load per
      // problem -- ~8 us before the first partial is read with the ~80 problems of a whole-step group)
    int hi = nprob - 1;
    while (p < hi) {
      const int mid = (p + hi + 1) >> 1;
      if (descs[mid].block_start <= bid) p = mid; else hi = mid - 1;
    }
  }
  const ReduceDesc d = descs[p];
  const int local = bid - d.block_start;
  const int ncb = (d.H + ZK_RED_COLS - 1) / ZK_RED_COLS;
  const int q = local / ncb, cb = local - q * ncb;
  float* o = d.out[q];
  if (o == nullptr) return;
  const int cl = threadIdx.x % ZK_RED_COLS, g = threadIdx.x / ZK_RED_COLS;
  const int c = cb * ZK_RED_COLS + cl;
  float t = 0.f;
  if (c < d.H) {
    // eight independent loads in flight per thread (the loop is a chain of ~0.5-us round trips otherwise); the
    // additions keep the order b = g, g + NG, ...
    const float* src = d.partials + (size_t)q * (d.qstride ? d.qstride : d.H) + c;
    const size_t rs = d.bstride ? (size_t)d.bstride : (size_t)d.nq * d.H;
    int b = g;
    for (; b + 7 * NG < d.nblk; b += 8 * NG) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(b + i * NG) * rs];
#pragma unroll
      for (int i = 0; i < 8; ++i) t += v[i];
    }
    for (; b < d.nblk; b += NG) t += src[(size_t)b * rs];
  }
  red[g][cl] = t;
  __syncthreads();
  if (g == 0 && c < d.H) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) v += red[i][cl];
    o[c] = v;
  }
}
// Embedding-table gradient without atomics: the host sorts the token rows by id (it owns the
// ids anyway); one wave per DISTINCT id sums its rows in fp32 and does a single read-modify-write
// of the table row.  `rows_sorted` [n_used] = token-row indices grouped by id, `seg` [n_uniq+1]
// = group boundaries, `uid` [n_uniq] = the id of each group, `n_uniq_dev` = device int.
struct EmbedBwdSide {
  const int* rows_sorted; const int* seg; const int* uid; const int* n_uniq_dev; const bf16_t* dout; float* dtable;
  int accumulate; uint32_t sid;
};
__device__ __forceinline__ void embed_bwd_sorted_body(
    const int* __restrict__ rows_sorted, const int* __restrict__ seg, const int* __restrict__ uid,
    const int* __restrict__ n_uniq_dev, const bf16_t* __restrict__ dout, float* __restrict__ dtable, int H,
    float scale, int accumulate, uint32_t thr, float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid,
    int wave, int nwaves) {
  const int lane = threadIdx.x & 63;
  const int nu = *n_uniq_dev;
  const uint64_t seed = thr ? *seedp : 0;
  for (int u = wave; u < nu; u += nwaves) {
    const int s0 = seg[u], s1 = seg[u + 1];
    float* dst = dtable + (size_t)uid[u] * H;
    for (int cbase = 0; cbase < H; cbase += 64 * 8) {     // wave-uniform control flow: every lane takes part in the shuffles
      const int c = cbase + lane * 8;
      const bool active = c < H;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // A frequent id (the eos of every sentence, punctuation) owns a long run of rows.  Walking it one dependent
      // load chain at a time (index -> row) cost ~1 us per row -- 64 eos rows = the whole 55-us launch of the bench
      // batch.  Now: the run's row indices in ONE coalesced load per 64 rows, then the rows sixteen at a time with all
      // sixteen loads in flight; the additions keep the order of the run, so the sums are bit-identical to the serial walk.
      for (int k0 = s0; k0 < s1; k0 += 64) {
        const int nb = min(64, s1 - k0);
        const int my_r = (lane < nb) ? rows_sorted[k0 + lane] : 0;
        for (int j0 = 0; j0 < nb; j0 += 16) {
          uint4 raw[16];
          int rr[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            rr[j] = __shfl(my_r, (j0 + j) & 63);
            if (active && j0 + j < nb) raw[j] = *reinterpret_cast<const uint4*>(dout + (size_t)rr[j] * H + c);
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if (!active || j0 + j >= nb) break;
            float v[8];
            unpack8(raw[j], v);
            if (thr) {
              float dm[8];
              zk_drop_scale8(seed, sid, (uint64_t)rr[j] * H + c, thr, inv_keep, dm);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] *= dm[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += v[q];
          }
        }
      }
      if (!active) continue;
      float4* d4 = reinterpret_cast<float4*>(dst + c);
      float4 lo = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
      float4 hi = make_float4(acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale);
      if (accumulate) {
        const float4 a = d4[0], b = d4[1];
        lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w;
        hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
      }
      d4[0] = lo; d4[1] = hi;
    }
  }
}

__global__ void __launch_bounds__(256) k_embed_bwd_sorted(
    const int* __restrict__ rows_sorted, const int* __restrict__ seg, const int* __restrict__ uid,
    const int* __restrict__ n_uniq_dev, const bf16_t* __restrict__ dout, float* __restrict__ dtable, int H,
    float scale, int accumulate, uint32_t thr, float inv_keep, const uint64_t* __restrict__ seedp, uint32_t sid) {
  embed_bwd_sorted_body(rows_sorted, seg, uid, n_uniq_dev, dout, dtable, H, scale, accumulate, thr, inv_keep, seedp, sid,
                        (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6);
}
// Round 6: the gradient scatters of BOTH embedding tables in one launch (blockIdx.y = the side; different tables: the
// caller guarantees it).  Per side the arithmetic and the order of k_embed_bwd_sorted.
__global__ void __launch_bounds__(256) k_embed_bwd_sorted_pair(EmbedBwdSide a, EmbedBwdSide b, int H, float scale, uint32_t thr,
                                                               float inv_keep, const uint64_t* __restrict__ seedp) {
  const EmbedBwdSide& s_ = blockIdx.y == 0 ? a : b;
  embed_bwd_sorted_body(s_.rows_sorted, s_.seg, s_.uid, s_.n_uniq_dev, s_.dout, s_.dtable, H, scale, s_.accumulate, thr,
                        inv_keep, seedp, s_.sid, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6);
}

// =====================================================================================
// K6  label-smoothed cross entropy, fused forward + backward on fp32 logits
//     util.py:88-103 (soft labels p on gold, q=eps/(V-1) elsewhere, normaliser),
//     transformer.py:198-207.  Closed form (sum of soft labels is 1):
//        ce = lse - p*z_gold - q*(sum_z - z_gold) - normaliser
//        dlogits = w_row * (softmax(z) - soft)        (never materialises soft labels)
//     One block per token row; pass 1 streams the row from HBM (online max/sum), pass 2
//     re-reads it from L2 and writes bf16 dlogits.
// =====================================================================================
__global__ void __launch_bounds__(256) k_ce_fused(
    const float* __restrict__ logits, const int* __restrict__ ids, const float* __restrict__ w,
    float* __restrict__ ce_out, bf16_t* __restrict__ dlogits, int V, int ld, float p, float q,
    float normalizer) {
  __shared__ float sm[8];
  const int r = blockIdx.x;
  const float* z = logits + (size_t)r * ld;
  const float wr = (w != nullptr) ? w[r] : 0.f;
  const bool need_bwd = dlogits != nullptr;
  const int gold = ids[r];
  // pass 1
  float m = -INFINITY, s = 0.f, sz = 0.f;
  const int V4 = V & ~3;
  for (int c = threadIdx.x * 4; c < V4; c += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(z + c);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
    s += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
    sz += (v.x + v.y) + (v.z + v.w);
  }
  for (int c = V4 + threadIdx.x; c < V; c += 256) {
    const float v = z[c];
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
    sz += v;
  }
  const float gm = block_max<4>(m, sm);
  s = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = block_sum<4>(s, sm);
  const float gsz = block_sum<4>(sz, sm);
  const float lse = gm + __logf(gs);
  const float zg = z[gold];
  if (threadIdx.x == 0 && ce_out != nullptr)
    ce_out[r] = lse - p * zg - q * (gsz - zg) - normalizer;
  if (!need_bwd) return;
  bf16_t* d = dlogits + (size_t)r * ld;
  if (wr == 0.f) {
    for (int c = threadIdx.x * 4; c < ld; c += 256 * 4)
      *reinterpret_cast<uint2*>(d + c) = make_uint2(0u, 0u);
    return;
  }
  for (int c = threadIdx.x * 4; c < ld; c += 256 * 4) {
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c + j;
      if (cc < V) {
        const float sl = (cc == gold) ? p : q;
        o[j] = wr * (__expf(z[cc] - lse) - sl);
      } else o[j] = 0.f;
    }
    // streaming store: with an ordinary one the L2 fetches each line it is about to overwrite
    zk_u32x2 pk; pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]);
    __builtin_nontemporal_store(pk, reinterpret_cast<zk_u32x2*>(d + c));
  }
}

// Same computation with the whole row held in registers (V <= NV*4096): one HBM read of the
// logits row, no second pass -- 1024 threads (16 waves) per row.
template <int NV>
__global__ void __launch_bounds__(1024, 8) k_ce_fused_reg(
    const float* __restrict__ logits, const int* __restrict__ ids, const float* __restrict__ w,
    float* __restrict__ ce_out, bf16_t* __restrict__ dlogits, int V, int ld, float p, float q,
    float normalizer) {
  __shared__ float sm[16];
  __shared__ float bc;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* z = logits + (size_t)r * ld;
  const float wr = (w != nullptr) ? w[r] : 0.f;
  const int gold = ids[r];
  float4 v[NV];
  float m = -INFINITY, sz = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 1024 + tid) * 4;
    if (c + 3 < V) {
      // read once, never again: streaming load (no L2 allocation worth keeping)
      const zk_f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(z + c));
      v[i] = make_float4(t4.x, t4.y, t4.z, t4.w);
    } else {
      v[i].x = c < V ? z[c] : -INFINITY;
      v[i].y = c + 1 < V ? z[c + 1] : -INFINITY;
      v[i].z = c + 2 < V ? z[c + 2] : -INFINITY;
      v[i].w = -INFINITY;
    }
    m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    sz += (v[i].x > -INFINITY ? v[i].x : 0.f) + (v[i].y > -INFINITY ? v[i].y : 0.f) +
          (v[i].z > -INFINITY ? v[i].z : 0.f) + (v[i].w > -INFINITY ? v[i].w : 0.f);
  }
  // block reductions over 16 waves
  m = wave_max(m);
  if (lane == 0) sm[wv] = m;
  __syncthreads();
  if (tid == 0) { float t = sm[0]; for (int i = 1; i < 16; ++i) t = fmaxf(t, sm[i]); bc = t; }
  __syncthreads();
  const float gm = bc;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x = __expf(v[i].x - gm); v[i].y = __expf(v[i].y - gm);
    v[i].z = __expf(v[i].z - gm); v[i].w = __expf(v[i].w - gm);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = wave_sum(s);
  sz = wave_sum(sz);
  __syncthreads();
  if (lane == 0) sm[wv] = s;
  __syncthreads();
  if (tid == 0) { float t = 0.f; for (int i = 0; i < 16; ++i) t += sm[i]; bc = t; }
  __syncthreads();
  const float gs = bc;
  __syncthreads();
  if (lane == 0) sm[wv] = sz;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sm[i];
    const float lse = gm + __logf(gs);
    const float zg = z[gold];
    if (ce_out != nullptr) ce_out[r] = lse - p * zg - q * (t - zg) - normalizer;
  }
  if (dlogits == nullptr) return;
  bf16_t* d = dlogits + (size_t)r * ld;
  const float inv = wr / gs;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 1024 + tid) * 4;
    if (c >= ld) continue;
    float o[4] = {v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv};   // exp(-inf)=0 past V
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c + j;
      if (cc < V) o[j] -= wr * ((cc == gold) ? p : q);
      else o[j] = 0.f;
    }
    *reinterpret_cast<uint2*>(d + c) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
}

// target statistics: mask=(id!=0), w = loss_scale*mask/(len_b*B) (gradient weight of each
// token under loss = mean_b( sum_t ce*mask / sum_t mask ), transformer.py:209-211)
__global__ void __launch_bounds__(256) k_target_stats(const int* __restrict__ ids, float* __restrict__ mask,
                                                      float* __restrict__ w, int B, int L,
                                                      float loss_scale) {
  __shared__ float sm[8];
  const int b = blockIdx.x;
  float cnt = 0.f;
  for (int t = threadIdx.x; t < L; t += 256) cnt += (ids[b * L + t] != 0) ? 1.f : 0.f;
  const float len = block_sum<4>(cnt, sm);
  for (int t = threadIdx.x; t < L; t += 256) {
    const float mk = (ids[b * L + t] != 0) ? 1.f : 0.f;
    if (mask != nullptr) mask[b * L + t] = mk;
    if (w != nullptr) w[b * L + t] = loss_scale * mk / (len * (float)B);
  }
}

// per_sample[b] = sum_t ce*mask / sum_t mask (one block per sentence); loss = mean_b (0 if B==0)
__global__ void __launch_bounds__(256) k_per_sample(const float* __restrict__ ce, const int* __restrict__ ids,
                                                    float* __restrict__ per_sample, int L) {
  __shared__ float sm[8];
  const int b = blockIdx.x;
  float a = 0.f, c = 0.f;
  for (int t = threadIdx.x; t < L; t += 256) {
    const float mk = (ids[b * L + t] != 0) ? 1.f : 0.f;
    a += ce[b * L + t] * mk;
    c += mk;
  }
  a = block_sum<4>(a, sm);
  c = block_sum<4>(c, sm);
  if (threadIdx.x == 0) per_sample[b] = a / c;
}
__global__ void __launch_bounds__(256) k_mean(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float sm[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  a = block_sum<4>(a, sm);
  if (threadIdx.x == 0) out[0] = (n > 0) ? a / (float)n : 0.f;
}

// Round 6: the two launches above as ONE (a training step ends its forward with them; each was ~4.5 us of launch latency
// for a few hundred bytes): one workgroup of 16 waves, wave w takes sentences w, w + 16, ..: lane t adds tokens t, t + 64, ..
// and the wave reduces -- then the first four waves form the mean exactly as k_mean does (thread i adds sentences i,
// i + 256, ..; block_sum<4>).  For L <= 64 every per-sentence value has the bits k_per_sample gives (one addend per lane).
__global__ void __launch_bounds__(1024) k_loss_tail(const float* __restrict__ ce, const int* __restrict__ ids,
                                                    float* __restrict__ per_sample, float* __restrict__ loss, int B, int L) {
  __shared__ float sm[8];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = w; b < B; b += 16) {
    float a = 0.f, c = 0.f;
    for (int t = lane; t < L; t += 64) {
      const float mk = (ids[b * L + t] != 0) ? 1.f : 0.f;
      a += ce[b * L + t] * mk;
      c += mk;
    }
    a = wave_sum(a);
    c = wave_sum(c);
    if (lane == 0) per_sample[b] = a / c;
  }
  __syncthreads();                       // (the block's own global writes are visible to it behind the barrier)
  if (loss == nullptr || threadIdx.x >= 256) return;
  float a = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) a += per_sample[i];
  a = block_sum<4>(a, sm);
  if (threadIdx.x == 0) loss[0] = (B > 0) ? a / (float)B : 0.f;
}

__global__ void __launch_bounds__(256) k_make_mask(const int* __restrict__ ids, float* __restrict__ mask, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) mask[i] = (ids[i] != 0) ? 1.f : 0.f;
}

// flag = 1 iff every id equals `value` (transformer.py:113 reduce_all over the batch)
__global__ void __launch_bounds__(256) k_all_equal(const int* __restrict__ ids, int n, int value, int* flag) {
  __shared__ float sm[8];
  float bad = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) bad += (ids[i] != value) ? 1.f : 0.f;
  bad = block_sum<4>(bad, sm);
  if (threadIdx.x == 0) *flag = (bad == 0.f) ? 1 : 0;
}

// =====================================================================================
// K5  cumulative averages over the time axis (transformer_aan.py:92-108; merged attention func.py:258-275)
// One block per (sentence, group of 64 eight-channel chunks); its 256 threads split the L time steps into
// SCAN_SEG segments, so the dependent chain is L/4 long and 4x as many waves are in flight as with one
// thread per (sentence, chunk): pass 1 reduces each segment, the segment totals are exchanged through LDS,
// pass 2 re-walks the segment from the right starting value.  O(L*H) instead of the reference's
// [B,L,L]x[B,L,H] matmul.
//   forward :  avg_t = w_t * sum_{s<=t} u_s x_s / den(cnt_t)            cnt_t = sum_{s<=t} m_s
//   backward:  dx_s  = u_s * sum_{t>=s} (w_t / den(cnt_t)) g_t          (the transpose)
// use_mask (aan_mask, func.py:390-398): u = w = m, den = max(cnt, 1); otherwise (transformer_aan.py:103-108)
// u = w = 1, den = cnt (1 where cnt <= 0).
// =====================================================================================
#define SCAN_SEG 4
struct ScanPos {
  int b, c, sg, t0, t1;
  bool live;
};
__device__ __forceinline__ ScanPos scan_pos(int L, int H) {
  ScanPos p;
  p.b = blockIdx.x;
  const int chunk = blockIdx.y * 64 + (threadIdx.x & 63);
  p.c = chunk * 8;
  p.live = p.c < H;
  p.sg = threadIdx.x >> 6;
  const int len = (L + SCAN_SEG - 1) / SCAN_SEG;
  p.t0 = min(L, p.sg * len);
  p.t1 = min(L, p.t0 + len);
  return p;
}
__device__ __forceinline__ float scan_den(float cnt, int use_mask) {
  return use_mask ? fmaxf(cnt, 1.f) : (cnt <= 0.f ? 1.f : cnt);
}

// LOAD(row, c, v[8]) reads x of token row; EMIT(row, c, avg[8]) stores the result of one step
template <typename LOAD, typename EMIT>
__device__ __forceinline__ void scan_avg_fwd(const float* __restrict__ mask, int L, int H, int use_mask, LOAD load,
                                             EMIT emit) {
  __shared__ float s_part[SCAN_SEG][64][8];
  __shared__ float s_cnt[SCAN_SEG];
  const ScanPos p = scan_pos(L, H);
  const int cl = threadIdx.x & 63;
  float run[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float cnt = 0.f;
  for (int t = p.t0; t < p.t1; ++t) {
    const size_t r = (size_t)p.b * L + t;
    const float m = mask[r];
    cnt += m;
    if (p.live) {
      float v[8];
      load(r, p.c, v);
      const float u = use_mask ? m : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) run[j] += u * v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_part[p.sg][cl][j] = run[j];
  if (cl == 0) s_cnt[p.sg] = cnt;
  __syncthreads();
  cnt = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) run[j] = 0.f;
  for (int sg = 0; sg < p.sg; ++sg) {
    cnt += s_cnt[sg];
#pragma unroll
    for (int j = 0; j < 8; ++j) run[j] += s_part[sg][cl][j];
  }
  if (!p.live) return;
  for (int t = p.t0; t < p.t1; ++t) {
    const size_t r = (size_t)p.b * L + t;
    const float m = mask[r];
    float v[8], o[8];
    load(r, p.c, v);
    cnt += m;
    const float u = use_mask ? m : 1.f;
    const float a = u / scan_den(cnt, use_mask);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      run[j] += u * v[j];
      o[j] = a * run[j];
    }
    emit(r, p.c, o);
  }
}

// LOAD(row, c, g[8]) reads the gradient flowing into avg_t; EMIT(row, c, d[8]) receives u_s * reverse sum
template <typename LOAD, typename EMIT>
__device__ __forceinline__ void scan_avg_bwd(const float* __restrict__ mask, int L, int H, int use_mask, LOAD load,
                                             EMIT emit) {
  __shared__ float s_part[SCAN_SEG][64][8];
  const ScanPos p = scan_pos(L, H);
  const int cl = threadIdx.x & 63;
  float cnt = 0.f;                               // valid steps before this segment
  for (int t = 0; t < p.t0; ++t) cnt += mask[(size_t)p.b * L + t];
  float run[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = p.t0; t < p.t1; ++t) {
    const size_t r = (size_t)p.b * L + t;
    const float m = mask[r];
    cnt += m;
    if (p.live) {
      float g[8];
      load(r, p.c, g);
      const float a = (use_mask ? m : 1.f) / scan_den(cnt, use_mask);
#pragma unroll
      for (int j = 0; j < 8; ++j) run[j] += a * g[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_part[p.sg][cl][j] = run[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) run[j] = 0.f;
  for (int sg = p.sg + 1; sg < SCAN_SEG; ++sg)
#pragma unroll
    for (int j = 0; j < 8; ++j) run[j] += s_part[sg][cl][j];
  if (!p.live) return;
  // cnt holds the inclusive count at t1 - 1; walk the segment backwards
  for (int t = p.t1 - 1; t >= p.t0; --t) {
    const size_t r = (size_t)p.b * L + t;
    const float m = mask[r];
    float g[8], o[8];
    load(r, p.c, g);
    const float u = use_mask ? m : 1.f;
    const float a = u / scan_den(cnt, use_mask);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      run[j] += a * g[j];
      o[j] = u * run[j];
    }
    emit(r, p.c, o);
    cnt -= m;
  }
}

// transformer_aan.py:92-108: cat = [x | average]  ([T, 2H]) for the gate GEMM
__global__ void __launch_bounds__(256) k_aan_fwd(const bf16_t* __restrict__ x, const float* __restrict__ mask,
                                                 bf16_t* __restrict__ cat, int B, int L, int H,
                                                 int use_mask) {
  scan_avg_fwd(mask, L, H, use_mask,
               [&](size_t r, int c, float (&v)[8]) { unpack8(*reinterpret_cast<const uint4*>(x + r * H + c), v); },
               [&](size_t r, int c, const float (&o)[8]) {
                 *reinterpret_cast<uint4*>(cat + r * 2 * H + c) = *reinterpret_cast<const uint4*>(x + r * H + c);
                 *reinterpret_cast<uint4*>(cat + r * 2 * H + H + c) = pack8(o);
               });
}

// dx_total = ds + dx_gate + dcat[:, :H] + reverse_scan(dy_gate + dcat[:, H:])
// use_mask bit 0: aan_mask; bit 1: dcat[:, H:] is already folded into dyg by the caller (use_ffn)
__global__ void __launch_bounds__(256) k_aan_bwd(const bf16_t* __restrict__ dcat, const bf16_t* __restrict__ dxg,
                                                 const bf16_t* __restrict__ dyg, const bf16_t* __restrict__ ds,
                                                 const float* __restrict__ mask, bf16_t* __restrict__ dx,
                                                 int B, int L, int H, int use_mask) {
  const bool skip_dc2 = (use_mask & 2) != 0;
  scan_avg_bwd(mask, L, H, use_mask & 1,
               [&](size_t r, int c, float (&g)[8]) {
                 unpack8(*reinterpret_cast<const uint4*>(dyg + r * H + c), g);
                 if (!skip_dc2) {
                   float e[8];
                   unpack8(*reinterpret_cast<const uint4*>(dcat + r * 2 * H + H + c), e);
#pragma unroll
                   for (int j = 0; j < 8; ++j) g[j] += e[j];
                 }
               },
               [&](size_t r, int c, const float (&o)[8]) {
                 float g1[8], dc1[8], d0[8], out[8];
                 unpack8(*reinterpret_cast<const uint4*>(dcat + r * 2 * H + c), dc1);
                 unpack8(*reinterpret_cast<const uint4*>(dxg + r * H + c), g1);
                 unpack8(*reinterpret_cast<const uint4*>(ds + r * H + c), d0);
#pragma unroll
                 for (int j = 0; j < 8; ++j) out[j] = d0[j] + g1[j] + dc1[j] + o[j];
                 *reinterpret_cast<uint4*>(dx + r * H + c) = pack8(out);
               });
}

// transformer_fuse (func.py:258-275, training branch): the simplified average-attention term that is
// summed into the cross-attention heads before o_map:  out = att + average_mask(vq)   (out may alias att)
__global__ void __launch_bounds__(256) k_cumavg_add_fwd(const bf16_t* __restrict__ vq, const float* __restrict__ mask,
                                                        const bf16_t* att, bf16_t* out, int B, int L, int H) {
  scan_avg_fwd(mask, L, H, 1,
               [&](size_t r, int c, float (&v)[8]) { unpack8(*reinterpret_cast<const uint4*>(vq + r * H + c), v); },
               [&](size_t r, int c, const float (&o)[8]) {
                 float a[8];
                 unpack8(*reinterpret_cast<const uint4*>(att + r * H + c), a);
#pragma unroll
                 for (int j = 0; j < 8; ++j) a[j] += o[j];
                 *reinterpret_cast<uint4*>(out + r * H + c) = pack8(a);
               });
}
// its transpose
__global__ void __launch_bounds__(256) k_cumavg_bwd(const bf16_t* __restrict__ dy, const float* __restrict__ mask,
                                                    bf16_t* __restrict__ dvq, int B, int L, int H) {
  scan_avg_bwd(mask, L, H, 1,
               [&](size_t r, int c, float (&g)[8]) { unpack8(*reinterpret_cast<const uint4*>(dy + r * H + c), g); },
               [&](size_t r, int c, const float (&o)[8]) { *reinterpret_cast<uint4*>(dvq + r * H + c) = pack8(o); });
}

// out[i] (+)= sum_s in[s*n + i]
__global__ void __launch_bounds__(256) k_sum_slices(float* __restrict__ out, const float* __restrict__ in, int nslices,
                                                    size_t n, size_t stride, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float acc = accumulate ? out[i] : 0.f;
    for (int s = 0; s < nslices; ++s) acc += in[(size_t)s * stride + i];
    out[i] = acc;
  }
}

// out = a + b on bf16 row blocks with independent leading dimensions (cols % 8 == 0)
__global__ void __launch_bounds__(256) k_add_bf16(bf16_t* __restrict__ out, int ldo, const bf16_t* __restrict__ a,
                                                  int lda, const bf16_t* __restrict__ b, int ldb, int rows, int cols) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nc = cols / 8;
  if (idx >= (size_t)rows * nc) return;
  const size_t r = idx / nc;
  const int c = (int)(idx % nc) * 8;
  float x[8], y[8];
  unpack8(*reinterpret_cast<const uint4*>(a + r * lda + c), x);
  unpack8(*reinterpret_cast<const uint4*>(b + r * ldb + c), y);
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] += y[j];
  *reinterpret_cast<uint4*>(out + r * ldo + c) = pack8(x);
}

// gate (transformer_aan.py:186-189): i,f = split(z); out = sigmoid(i)*x + sigmoid(f)*y
// x = cat[:, :H], y = cat[:, H:]
__global__ void __launch_bounds__(256) k_aan_gate_fwd(const bf16_t* __restrict__ z, const bf16_t* __restrict__ cat,
                                                      bf16_t* __restrict__ out, int rows, int H) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nc = H / 8;
  if (idx >= (size_t)rows * nc) return;
  const size_t r = idx / nc;
  const int c = (int)(idx % nc) * 8;
  float zi[8], zf[8], xv[8], yv[8], o[8];
  unpack8(*reinterpret_cast<const uint4*>(z + r * 2 * H + c), zi);
  unpack8(*reinterpret_cast<const uint4*>(z + r * 2 * H + H + c), zf);
  unpack8(*reinterpret_cast<const uint4*>(cat + r * 2 * H + c), xv);
  unpack8(*reinterpret_cast<const uint4*>(cat + r * 2 * H + H + c), yv);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    o[j] = xv[j] / (1.f + __expf(-zi[j])) + yv[j] / (1.f + __expf(-zf[j]));
  *reinterpret_cast<uint4*>(out + r * H + c) = pack8(o);
}

__global__ void __launch_bounds__(256) k_aan_gate_bwd(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ z,
                                                      const bf16_t* __restrict__ cat, bf16_t* __restrict__ dz,
                                                      bf16_t* __restrict__ dxg, bf16_t* __restrict__ dyg,
                                                      int rows, int H) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int nc = H / 8;
  if (idx >= (size_t)rows * nc) return;
  const size_t r = idx / nc;
  const int c = (int)(idx % nc) * 8;
  float g[8], zi[8], zf[8], xv[8], yv[8], dzi[8], dzf[8], dx[8], dy[8];
  unpack8(*reinterpret_cast<const uint4*>(dg + r * H + c), g);
  unpack8(*reinterpret_cast<const uint4*>(z + r * 2 * H + c), zi);
  unpack8(*reinterpret_cast<const uint4*>(z + r * 2 * H + H + c), zf);
  unpack8(*reinterpret_cast<const uint4*>(cat + r * 2 * H + c), xv);
  unpack8(*reinterpret_cast<const uint4*>(cat + r * 2 * H + H + c), yv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float si = 1.f / (1.f + __expf(-zi[j]));
    const float sf = 1.f / (1.f + __expf(-zf[j]));
    dzi[j] = g[j] * xv[j] * si * (1.f - si);
    dzf[j] = g[j] * yv[j] * sf * (1.f - sf);
    dx[j] = g[j] * si;
    dy[j] = g[j] * sf;
  }
  *reinterpret_cast<uint4*>(dz + r * 2 * H + c) = pack8(dzi);
  *reinterpret_cast<uint4*>(dz + r * 2 * H + H + c) = pack8(dzf);
  *reinterpret_cast<uint4*>(dxg + r * H + c) = pack8(dx);
  *reinterpret_cast<uint4*>(dyg + r * H + c) = pack8(dy);
}

// =====================================================================================
// K7  optimiser: global norm, clip, TF1 Adam on flat fp32 buffers (utils/cycle.py:86-101,
//     tf.train.AdamOptimizer: m,v EMA, theta -= lr_t*m/(sqrt(v)+eps), eps outside sqrt,
//     lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host and read from `hyper`)
//     hyper (device floats): [0] lr_t  [1] beta1  [2] beta2  [3] eps  [4] grad_scale
//                            [5] clip_norm (0 = off)  [6] gnorm (in)  [7] skipped-flag (out)
//                            [8] EMA decay  [9] gnorm upper bound of safe_nan (0 = off)   -- 12 floats in all
//     The bf16 shadow copy used by the GEMMs is refreshed in the same pass
//     (utils/dtype.py:55-69 fp32 storage / low-precision compute contract).
// =====================================================================================
__global__ void __launch_bounds__(256) k_sumsq_partial(const float* __restrict__ x, size_t n,
                                                       float* __restrict__ partials) {
  __shared__ float sm[8];
  float acc = 0.f;
  const size_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = x4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += x[i] * x[i];
  acc = block_sum<4>(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_norm_final(const float* __restrict__ partials, int n, float scale,
                                                    float* __restrict__ out) {
  __shared__ float sm[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  acc = block_sum<4>(acc, sm);
  if (threadIdx.x == 0) out[0] = scale * sqrtf(acc);
}

// GSQ = false: the update depends on the global norm (clipping, safe_nan bound): hyper[6] holds it (zk_l2norm ran).
// GSQ = true : norm-free update (cycle.py:98-101 with clip_grad_norm = 0.0, no safe_nan -- the recipe): the gradient
//              norm is only reported (main.py:316-319, the log line), so its sum of squares is accumulated in THIS
//              pass over the gradient (gsq partials; k_norm_final2 writes hyper[6] and the flags afterwards) instead
//              of a pass of its own over the 308 MB.  The update is applied whatever the norm turns out to be --
//              exactly what the reference does without safe_nan (train_op and the norm are fetched together).
// seed (may be NULL): the per-step dropout seed is advanced here (one launch fewer per step).
template <bool GSQ, int U = 1, bool CONTIG = false>
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              bf16_t* __restrict__ shadow, size_t n,
                                              float* __restrict__ hyper, float* __restrict__ psq,
                                              float* __restrict__ gsq, uint64_t* __restrict__ seed,
                                              const int* __restrict__ skip_word = nullptr, int nt = 0) {
  // nt (tuning key 18, A/B): bit 0 = the fp32 parameter / moment STORES are non-temporal (nobody reads them before the next
  // step's update: they need not displace the bf16 shadow -- the next forward's weight operand -- from the L2 / Infinity
  // Cache); bit 1 = their loads as well
  __shared__ float sm_[8];
  float pacc = 0.f;   // sum of squares of the parameters BEFORE this update (tf.global_norm(variables))
  float gacc = 0.f;   // sum of squares of the scaled gradient (GSQ)
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  const float gs = hyper[4], clip = hyper[5], gnorm = hyper[6];
  if (seed != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *seed += 1;
  // skip_word (device int, may be NULL): non-zero = a launch of THIS step reported a fault of its own (the in-launch
  // LayerNorm exchange gave up waiting, GemmEpi.sy_err): the gradients are not to be trusted, so the update is not
  // applied -- parameters, moments and shadow stay as they are -- and the step is booked as skipped: the gradient
  // norm is reported as NaN (GSQ: k_norm_final2 turns it into hyper[7] / the sticky hyper[10]; !GSQ: set here).
  if (skip_word != nullptr && *skip_word != 0) {
    if (threadIdx.x == 0) {
      if (psq != nullptr) psq[blockIdx.x] = 0.f;
      if (GSQ && gsq != nullptr) gsq[blockIdx.x] = __builtin_nanf("");
      if (!GSQ && blockIdx.x == 0) { hyper[7] = 1.f; hyper[10] += 1.f; }
    }
    return;
  }
  float f = gs;
  if (!GSQ) {
    // NaN/Inf guard (main.py:316-319); hyper[9] > 0: also skip when gnorm exceeds it (safe_nan, main.py:325-329)
    if (!(gnorm == gnorm) || fabsf(gnorm) == INFINITY || (hyper[9] > 0.f && gnorm > hyper[9])) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { hyper[7] = 1.f; hyper[10] += 1.f; }   // [10]: sticky count
      return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hyper[7] = 0.f;   // the flag describes THIS update
    if (clip > 0.f) f *= clip / fmaxf(gnorm, clip);  // tf.clip_by_global_norm
  }
  const size_t n4 = n / 4;
  // U float4 per thread and round, all 4 U loads requested before the first use (tuning key 9 selects U)
  // CONTIG: every block streams ONE contiguous range of each array (fewer open DRAM pages than the grid-stride order)
  const bool contig = CONTIG;
  constexpr int UU = U;
  const size_t per_block = ((n4 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const size_t stride = contig ? 256 : (size_t)gridDim.x * 256;
  const size_t ibeg = contig ? (size_t)blockIdx.x * per_block + threadIdx.x : (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t iend = contig ? min(n4, (size_t)(blockIdx.x + 1) * per_block) : n4;
  for (size_t i0 = ibeg; i0 < iend; i0 += UU * stride) {
    float4 pp[UU], gg[UU], mm[UU], vv[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const size_t i = i0 + u * stride;
      if (i < iend) {
        const zk_f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(g) + i);   // read once
        gg[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
        if (nt & 2) {
          const zk_f32x4 a4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(p) + i);
          const zk_f32x4 b4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(m) + i);
          const zk_f32x4 c4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(v) + i);
          pp[u] = make_float4(a4.x, a4.y, a4.z, a4.w);
          mm[u] = make_float4(b4.x, b4.y, b4.z, b4.w);
          vv[u] = make_float4(c4.x, c4.y, c4.z, c4.w);
        } else {
          pp[u] = reinterpret_cast<float4*>(p)[i];
          mm[u] = reinterpret_cast<float4*>(m)[i];
          vv[u] = reinterpret_cast<float4*>(v)[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const size_t i = i0 + u * stride;
      if (i < iend) {
        pacc += pp[u].x * pp[u].x + pp[u].y * pp[u].y + pp[u].z * pp[u].z + pp[u].w * pp[u].w;
        float* P = &pp[u].x; const float* G = &gg[u].x; float* M = &mm[u].x; float* Vv = &vv[u].x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gj = G[j] * f;
          if (GSQ) gacc += gj * gj;
          M[j] = b1 * M[j] + (1.f - b1) * gj;
          Vv[j] = b2 * Vv[j] + (1.f - b2) * gj * gj;
          P[j] -= lr * M[j] / (sqrtf(Vv[j]) + eps);
        }
        if (nt & 1) {
          __builtin_nontemporal_store((zk_f32x4){pp[u].x, pp[u].y, pp[u].z, pp[u].w}, reinterpret_cast<zk_f32x4*>(p) + i);
          __builtin_nontemporal_store((zk_f32x4){mm[u].x, mm[u].y, mm[u].z, mm[u].w}, reinterpret_cast<zk_f32x4*>(m) + i);
          __builtin_nontemporal_store((zk_f32x4){vv[u].x, vv[u].y, vv[u].z, vv[u].w}, reinterpret_cast<zk_f32x4*>(v) + i);
        } else {
          reinterpret_cast<float4*>(p)[i] = pp[u];
          reinterpret_cast<float4*>(m)[i] = mm[u];
          reinterpret_cast<float4*>(v)[i] = vv[u];
        }
        if (shadow != nullptr) {
          if (nt & 4) {
            zk_u32x2 sh2; sh2.x = pack2bf(P[0], P[1]); sh2.y = pack2bf(P[2], P[3]);
            __builtin_nontemporal_store(sh2, reinterpret_cast<zk_u32x2*>(shadow) + i);
          } else {
            reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack2bf(P[0], P[1]), pack2bf(P[2], P[3]));
          }
        }
      }
    }
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gj = g[i] * f;
    if (GSQ) gacc += gj * gj;
    pacc += p[i] * p[i];
    m[i] = b1 * m[i] + (1.f - b1) * gj;
    v[i] = b2 * v[i] + (1.f - b2) * gj * gj;
    p[i] -= lr * m[i] / (sqrtf(v[i]) + eps);
    if (shadow != nullptr) shadow[i] = f2bf(p[i]);
  }
  if (psq != nullptr) {
    pacc = block_sum<4>(pacc, sm_);
    if (threadIdx.x == 0) psq[blockIdx.x] = pacc;
  }
  if (GSQ && gsq != nullptr) {
    gacc = block_sum<4>(gacc, sm_);
    if (threadIdx.x == 0) gsq[blockIdx.x] = gacc;
  }
}
#ifdef ZK_EXPERIMENTS
// The norm-free update on a SUBSET of the flat buffers: nseg segments [lo_s, lo_s + len_s) (float4 units, ascending),
// `prefix` their running lengths -- what is left when the weight matrices have been updated inside the launch that made
// their gradients (k_gemm_grouped256<.., UPD>): the embedding tables and the vectors between the matrices.  The blocks
// split the COMPACTED index space evenly (a split of the raw index space would leave the blocks that fall into the
// skipped ranges idle and the others with all the traffic); a thread finds its segment by binary search in LDS, almost
// always the segment of its previous element.  Same arithmetic as k_adam<true>.
#define ZK_ADAM_MAXSEG 512
__global__ void __launch_bounds__(256) k_adam_seg(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, bf16_t* __restrict__ shadow,
                                                  const long* __restrict__ seg_lo, const long* __restrict__ prefix, int nseg,
                                                  const float* __restrict__ hyper, float* __restrict__ psq,
                                                  float* __restrict__ gsq, uint64_t* __restrict__ seed) {
  __shared__ float sm_[8];
  __shared__ long s_lo[ZK_ADAM_MAXSEG], s_pre[ZK_ADAM_MAXSEG + 1];
  for (int i = threadIdx.x; i < nseg; i += 256) { s_lo[i] = seg_lo[i]; s_pre[i] = prefix[i]; }
  if (threadIdx.x == 0) s_pre[nseg] = prefix[nseg];
  __syncthreads();
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], f = hyper[4];
  if (seed != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *seed += 1;
  const long total = s_pre[nseg];
  const long per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long cbeg = (long)blockIdx.x * per_block, cend = min(total, cbeg + per_block);
  float pacc = 0.f, gacc = 0.f;
  int sgm = 0;
  for (long c0 = cbeg + threadIdx.x; c0 < cend; c0 += 2 * 256) {
    float4 pp[2], gg[2], mm[2], vv[2];
    long idx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long c = c0 + u * 256;
      idx[u] = -1;
      if (c < cend) {
        if (!(c >= s_pre[sgm] && c < s_pre[sgm + 1])) {
          int lo = 0, hi = nseg - 1;
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= c) lo = mid; else hi = mid - 1; }
          sgm = lo;
        }
        const long i = s_lo[sgm] + (c - s_pre[sgm]);
        idx[u] = i;
        pp[u] = reinterpret_cast<float4*>(p)[i];
        const zk_f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const zk_f32x4*>(g) + i);
        gg[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
        mm[u] = reinterpret_cast<float4*>(m)[i];
        vv[u] = reinterpret_cast<float4*>(v)[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (idx[u] < 0) continue;
      const long i = idx[u];
      pacc += pp[u].x * pp[u].x + pp[u].y * pp[u].y + pp[u].z * pp[u].z + pp[u].w * pp[u].w;
      float* P = &pp[u].x; const float* G = &gg[u].x; float* M = &mm[u].x; float* Vv = &vv[u].x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = G[j] * f;
        gacc += gj * gj;
        M[j] = b1 * M[j] + (1.f - b1) * gj;
        Vv[j] = b2 * Vv[j] + (1.f - b2) * gj * gj;
        P[j] -= lr * M[j] / (sqrtf(Vv[j]) + eps);
      }
      reinterpret_cast<float4*>(p)[i] = pp[u];
      reinterpret_cast<float4*>(m)[i] = mm[u];
      reinterpret_cast<float4*>(v)[i] = vv[u];
      reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack2bf(P[0], P[1]), pack2bf(P[2], P[3]));
    }
  }
  pacc = block_sum<4>(pacc, sm_);
  if (threadIdx.x == 0) psq[blockIdx.x] = pacc;
  gacc = block_sum<4>(gacc, sm_);
  if (threadIdx.x == 0) gsq[blockIdx.x] = gacc;
}
// finish of the split update: the block partials of k_adam_seg + the wave partials {sum g^2, sum theta^2} the updating
// weight-gradient launch left (extra [n_extra][2]), each summed in index order
__global__ void __launch_bounds__(256) k_norm_final3(const float* __restrict__ psq, const float* __restrict__ gsq, int n,
                                                     const float* __restrict__ extra, int n_extra,
                                                     float* __restrict__ hyper, float* __restrict__ pnorm_out) {
  __shared__ float sm[8];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += gsq[i]; b += psq[i]; }
  for (int i = threadIdx.x; i < n_extra; i += 256) { a += extra[2 * i]; b += extra[2 * i + 1]; }
  a = block_sum<4>(a, sm);
  b = block_sum<4>(b, sm);
  if (threadIdx.x == 0) {
    const float gn = sqrtf(a);
    hyper[6] = gn;
    const bool bad = !(gn == gn) || fabsf(gn) == INFINITY;
    hyper[7] = bad ? 1.f : 0.f;
    if (bad) hyper[10] += 1.f;
    if (pnorm_out != nullptr) pnorm_out[0] = sqrtf(b);
  }
}
#endif  // ZK_EXPERIMENTS
// finishes k_adam<true>: gnorm = sqrt(sum gsq) -> hyper[6], the per-update flag hyper[7] and the sticky count
// hyper[10] of non-finite norms; pnorm = sqrt(sum psq) -> pnorm_out.  One block.
__global__ void __launch_bounds__(256) k_norm_final2(const float* __restrict__ psq, const float* __restrict__ gsq,
                                                     int n, float* __restrict__ hyper, float* __restrict__ pnorm_out) {
  __shared__ float sm[8];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += gsq[i]; if (psq != nullptr) b += psq[i]; }
  a = block_sum<4>(a, sm);
  b = block_sum<4>(b, sm);
  if (threadIdx.x == 0) {
    const float gn = sqrtf(a);
    hyper[6] = gn;
    const bool bad = !(gn == gn) || fabsf(gn) == INFINITY;
    hyper[7] = bad ? 1.f : 0.f;
    if (bad) hyper[10] += 1.f;
    if (pnorm_out != nullptr) pnorm_out[0] = sqrtf(b);
#ifdef ZK_STEP_STAMPS
    {
      const unsigned int i_ = atomicAdd(&zk_step_stamp_n[1], 1u);
      if (i_ < 4096) zk_step_stamp_buf[4096 + i_] = __builtin_amdgcn_s_memrealtime();
    }
#endif
  }
}
// hyper[6] was written by zk_l2norm on a path whose update does not look at it (per-bucket updates of the
// data-parallel step): record a non-finite norm in the per-update flag and the sticky count.
__global__ void k_norm_flag(float* __restrict__ hyper) {
  const float gn = hyper[6];
  const bool bad = !(gn == gn) || fabsf(gn) == INFINITY;
  hyper[7] = bad ? 1.f : 0.f;
  if (bad) hyper[10] += 1.f;
}

__global__ void __launch_bounds__(256) k_cast_f32_bf16(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = f2bf(x[i]);
}
__global__ void __launch_bounds__(256) k_cast_bf16_f32(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = bf2f(x[i]);
}
// y += a * x  (gradient accumulation slots of utils/cycle.py:27-36,86-88)
__global__ void __launch_bounds__(256) k_axpy_f32(float* __restrict__ y, const float* __restrict__ x, float a,
                                                  float by, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = by * y[i] + a * x[i];
}
// tf.train.ExponentialMovingAverage update after train_op (utils/cycle.py:113-119):
// shadow -= (1 - d) * (shadow - p), d = hyper[8] (fed per step: min(decay, (1+t)/(10+t)));
// skipped together with the Adam update when the gradient norm is not finite.
__global__ void __launch_bounds__(256) k_ema(float* __restrict__ ema, const float* __restrict__ p,
                                             const float* __restrict__ hyper, size_t n) {
  const float gnorm = hyper[6], d = hyper[8];
  if (!(gnorm == gnorm) || fabsf(gnorm) == INFINITY || (hyper[9] > 0.f && gnorm > hyper[9])) return;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 e = reinterpret_cast<float4*>(ema)[i];
    const float4 w = reinterpret_cast<const float4*>(p)[i];
    e.x -= (1.f - d) * (e.x - w.x); e.y -= (1.f - d) * (e.y - w.y);
    e.z -= (1.f - d) * (e.z - w.z); e.w -= (1.f - d) * (e.w - w.w);
    reinterpret_cast<float4*>(ema)[i] = e;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    ema[i] -= (1.f - d) * (ema[i] - p[i]);
  }
}
__global__ void __launch_bounds__(256) k_dropout_mask(float* __restrict__ out, size_t n, uint32_t thr,
                                                      float inv_keep, const uint64_t* __restrict__ seedp,
                                                      uint32_t sid) {
  const uint64_t seed = *seedp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = thr ? zk_drop_scale(seed, sid, i, thr, inv_keep) : 1.f;
}
__global__ void k_seed_advance(uint64_t* seed, uint64_t inc) { *seed += inc; }

// =====================================================================================
// C-ABI
// =====================================================================================
static inline int row_grid(int rows) {
  int g = (rows + 3) / 4;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return g;
}
static inline int flat_grid(size_t n, int per_thread) {
  size_t g = (n / per_thread + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

const char* zk_last_error_string(void) { return zk_err_buf; }
int zk_version(void) { return 100; }

int zk_embed_fwd(const int* ids, const void* table, const float* bias, const float* timing, void* out,
                 int B, int L, int H, float scale, int shift, int pos0, const int* zero_flag,
                 float drop_p, const uint64_t* seed, uint32_t sid, const int* pos0_dev, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_embed_fwd: H=%d must be a multiple of 8", H);
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_embed_fwd: dropout needs a seed pointer");
  const int rows = B * L;
  if (rows == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  hipLaunchKernelGGL(k_embed_fwd, dim3(row_grid(rows)), dim3(256), 0, stream, ids, (const bf16_t*)table,
                     bias, timing, (bf16_t*)out, rows, L, H, scale, shift, pos0, zero_flag, thr, ik,
                     seed, sid, pos0_dev);
  ZK_LAUNCH_CHECK();
  return 0;
}

// both input embeddings of a training step (encoder: ids_a [B, La]; decoder: ids_b [B, Lb], shifted right by one,
// transformer.py:104-108) in one launch; arguments per side as zk_embed_fwd
int zk_embed_fwd_pair(const int* ids_a, const void* table_a, void* out_a, int La, uint32_t sid_a, const int* ids_b,
                      const void* table_b, void* out_b, int Lb, uint32_t sid_b, const float* bias, const float* timing, int B,
                      int H, float scale, float drop_p, const uint64_t* seed, uint32_t* ln_epoch, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_embed_fwd_pair: H=%d must be a multiple of 8", H);
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_embed_fwd_pair: dropout needs a seed pointer");
  const int ra = B * La, rb = B * Lb;
  if (ra + rb == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const EmbedFwdSide a{ids_a, (const bf16_t*)table_a, bias, timing, (bf16_t*)out_a, ra, La > 0 ? La : 1, H, scale, 0, 0, thr, ik, sid_a};
  const EmbedFwdSide b{ids_b, (const bf16_t*)table_b, bias, timing, (bf16_t*)out_b, rb, Lb > 0 ? Lb : 1, H, scale, 1, 0, thr, ik, sid_b};
  hipLaunchKernelGGL(k_embed_fwd_pair, dim3(row_grid(ra + rb)), dim3(256), 0, stream, a, b, seed, ln_epoch);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_embed_bwd(const int* ids, const void* dout, float* dtable, float* dbias, int B, int L, int H,
                 float scale, int shift, float drop_p, const uint64_t* seed, uint32_t sid,
                 hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_embed_bwd: H=%d must be a multiple of 8", H);
  const int rows = B * L;
  if (rows == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  int g = (rows + 3) / 4;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(k_embed_bwd, dim3(g), dim3(256), 0, stream, ids, (const bf16_t*)dout, dtable, dbias,
                     rows, L, H, scale, shift, thr, ik, seed, sid);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_add_ln_fwd(const void* x, const void* y, const float* gamma, const float* beta, void* out,
                  void* sum_out, float* mean, float* rstd, int rows, int H, float eps, float drop_p,
                  const uint64_t* seed, uint32_t sid, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0 && H <= MAXC_LIMIT * 512, "zk_add_ln_fwd: H=%d must be a multiple of 8 and <= %d", H,
               MAXC_LIMIT * 512);
  ZK_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "zk_add_ln_fwd: mean/rstd must both be given");
  if (rows == 0) return 0;
  const uint32_t thr = (drop_p > 0.f && y != nullptr) ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  if (zk_prog_active())
    return zk_prog_record_add_ln_fwd((const bf16_t*)x, (const bf16_t*)y, gamma, beta, (bf16_t*)out, (bf16_t*)sum_out, mean,
                                     rstd, rows, H, eps, thr, ik, seed, sid);
#define ZK_LN_FWD(NC)                                                                                   \
  hipLaunchKernelGGL(k_add_ln_fwd<NC>, dim3(row_grid(rows)), dim3(256), 0, stream, (const bf16_t*)x,    \
                     (const bf16_t*)y, gamma, beta, (bf16_t*)out, (bf16_t*)sum_out, mean, rstd, rows, H, \
                     eps, thr, ik, seed, sid)
  if (H <= 512) ZK_LN_FWD(1);
  else if (H <= 1024) ZK_LN_FWD(2);
  else ZK_LN_FWD(4);
#undef ZK_LN_FWD
  ZK_LAUNCH_CHECK();
  return 0;
}

size_t zk_add_ln_bwd_workspace(int rows, int H) {
  int g = (rows + 15) / 16;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (size_t)g * 3 * H * sizeof(float);
}

int g_tune[24] = {1, 0, 0, 0, 0, 0, 0x44, 0, 0, 2, 512, 0, 4 | (28 << 8), 0, 129, 0, 0, 0, 3, 0, 0, 0, 0, 0};   // [18] = 3 (round 6): Adam's fp32 parameter / moment loads and stores non-temporal (step 4.197 -> 4.184 ms, three interleaved runs)   // [16] (round 6): bit mask of merged small launches switched OFF (bit 0: per-sentence loss + mean as two launches)   // [14] = 129: 256x256 tile, the first half of the workgroup issues its LDS-DMA behind its first two slices (round 5)   // [12]: phases | us << 8 of the updating weight-gradient launch   // [0] wide LayerNorm backward kernel; [1] GEMM: legacy split-K rule (A/B)
static int ln_bwd_blocks(int rows) {
  int g = (rows + 15) / 16;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return g;
}

// tuning switches for A/B measurements (key 0: wide LayerNorm-backward kernel); returns the old value
int zk_tune(int key, int value) {
  if (key < 0 || key >= 24) return -1;
  const int old = g_tune[key];
  g_tune[key] = value;
  return old;
}

// second stage of zk_add_ln_bwd(defer_reduce=1): may run later and on another stream
int zk_add_ln_bwd_reduce(const void* workspace, int rows, int H, float* dgamma, float* dbeta, float* dbias_prev,
                         hipStream_t stream) {
  hipLaunchKernelGGL(k_partials_reduce, dim3((H + 15) / 16, 3), dim3(256), 0, stream, (const float*)workspace,
                     ln_bwd_blocks(rows), 3, H, dgamma, dbeta, dbias_prev, 0);
  ZK_LAUNCH_CHECK();
  return 0;
}

static int add_ln_bwd_launch(const void* dout, const void* sum, const float* mean, const float* rstd,
                             const float* gamma, void* dsum, void* dy, float* dgamma, float* dbeta, float* dbias_prev,
                             int rows, int H, float drop_p, const uint64_t* seed, uint32_t sid, void* workspace,
                             size_t ws_bytes, int defer_reduce, const float* part, int np, float eps, const float* beta,
                             void* y_out, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0 && H <= MAXC_LIMIT * 512, "zk_add_ln_bwd: H=%d must be a multiple of 8 and <= %d", H,
               MAXC_LIMIT * 512);
  ZK_CHECK_ARG(ws_bytes >= zk_add_ln_bwd_workspace(rows, H), "zk_add_ln_bwd: workspace too small");
  ZK_CHECK_ARG(drop_p == 0.f || dy != nullptr, "zk_add_ln_bwd: dropout needs a dy output");
  if (rows == 0) return 0;
  int g = (rows + 15) / 16;
  if (g > 256) g = 256;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#define ZK_LN_BWD(NC)                                                                                     \
  hipLaunchKernelGGL(k_add_ln_bwd<NC>, dim3(g), dim3(256), 0, stream, (const bf16_t*)dout, (const bf16_t*)sum, \
                     mean, rstd, gamma, (bf16_t*)dsum, (bf16_t*)dy, (float*)workspace, rows, H, thr, ik,     \
                     seed, sid, part, np, eps, beta, (bf16_t*)y_out)
  if (H <= 512 && g_tune[0])
    hipLaunchKernelGGL(k_add_ln_bwd_wide, dim3(g), dim3(1024), 0, stream, (const bf16_t*)dout, (const bf16_t*)sum,
                       mean, rstd, gamma, (bf16_t*)dsum, (bf16_t*)dy, (float*)workspace, rows, H, thr, ik, seed,
                       sid, part, np, eps, beta, (bf16_t*)y_out);
  else if (H <= 512) ZK_LN_BWD(1);
  else if (H <= 1024) ZK_LN_BWD(2);
  else ZK_LN_BWD(4);
#undef ZK_LN_BWD
  ZK_LAUNCH_CHECK();
  if (defer_reduce) return 0;
  return zk_add_ln_bwd_reduce(workspace, rows, H, dgamma, dbeta, dbias_prev, stream);
}

int zk_add_ln_bwd(const void* dout, const void* sum, const float* mean, const float* rstd,
                  const float* gamma, void* dsum, void* dy, float* dgamma, float* dbeta, float* dbias_prev,
                  int rows, int H, float drop_p, const uint64_t* seed, uint32_t sid, void* workspace,
                  size_t ws_bytes, int defer_reduce, hipStream_t stream) {
  return add_ln_bwd_launch(dout, sum, mean, rstd, gamma, dsum, dy, dgamma, dbeta, dbias_prev, rows, H, drop_p, seed, sid,
                           workspace, ws_bytes, defer_reduce, nullptr, 0, 0.f, nullptr, nullptr, stream);
}

#ifdef ZK_EXPERIMENTS   // the LayerNorm-free forward: measured, no gain (profiles/r04_negative_results.txt)
// The backward of a LayerNorm the forward never launched (zk_gemm_ln): the row statistics come from the per-64-column
// partials `part` [rows][H/64][2] the producing GEMM left beside the un-normalised sum, and y_out (optional) receives
// LN(sum) = bf16(gamma (sum - mu) rstd + beta) -- the rows k_add_ln_fwd would have written, needed only now, as the X
// operand of the deferred weight gradient of the layer that consumed them.
int zk_add_ln_bwd_lazy(const void* dout, const void* sum, const float* part, const float* gamma, const float* beta,
                       void* y_out, void* dsum, void* dy, float* dgamma, float* dbeta, float* dbias_prev, int rows, int H,
                       float eps, float drop_p, const uint64_t* seed, uint32_t sid, void* workspace, size_t ws_bytes,
                       int defer_reduce, hipStream_t stream) {
  ZK_CHECK_ARG(part != nullptr && H % 128 == 0 && H / 64 <= ZK_LN_MAXP, "zk_add_ln_bwd_lazy: H=%d must be a multiple of 128 and <= %d",
               H, ZK_LN_MAXP * 64);
  ZK_CHECK_ARG(y_out == nullptr || beta != nullptr, "zk_add_ln_bwd_lazy: y_out needs beta");
  return add_ln_bwd_launch(dout, sum, nullptr, nullptr, gamma, dsum, dy, dgamma, dbeta, dbias_prev, rows, H, drop_p, seed,
                           sid, workspace, ws_bytes, defer_reduce, part, H / 64, eps, beta, y_out, stream);
}
#endif  // ZK_EXPERIMENTS

size_t zk_colsum_workspace(int rows, int N) {
  int gy = (rows + 255) / 256;
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  return (size_t)gy * N * sizeof(float);
}

int zk_colsum_ex(const void* a, int rows, int N, int lda, float* out, int skip_L, int accumulate, float drop_p,
                 const uint64_t* seed, uint32_t sid, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG(N % 8 == 0 && lda % 8 == 0, "zk_colsum: N=%d, lda=%d must be multiples of 8", N, lda);
  ZK_CHECK_ARG(ws_bytes >= zk_colsum_workspace(rows, N), "zk_colsum: workspace too small");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_colsum: dropout needs a seed pointer");
  int gy = (rows + 255) / 256;
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, gy), dim3(256), 0, stream, (const bf16_t*)a, rows, N, lda,
                     (float*)workspace, skip_L, thr, ik, seed, sid);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_partials_reduce, dim3((N + 15) / 16, 1), dim3(256), 0, stream, (const float*)workspace,
                     gy, 1, N, out, (float*)nullptr, (float*)nullptr, accumulate);
  ZK_LAUNCH_CHECK();
  return 0;
}

// out[c] = sum_r a[r][c] (rows with r % skip_a == 0 left out when skip_a > 0) + sum_r b[r][c]  (+ dropout masks of the
// two sites): the bias gradient of the two input embeddings in two launches instead of four
int zk_colsum_pair(const void* a, int rows_a, int lda, int skip_a, uint32_t sid_a, const void* b, int rows_b, int ldb,
                   int skip_b, uint32_t sid_b, int N, float* out, float drop_p, const uint64_t* seed, void* workspace,
                   size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "zk_colsum_pair: N=%d, lda=%d, ldb=%d must be multiples of 8", N, lda, ldb);
  ZK_CHECK_ARG(ws_bytes >= zk_colsum_workspace(rows_a, N) + zk_colsum_workspace(rows_b, N), "zk_colsum_pair: workspace too small");
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_colsum_pair: dropout needs a seed pointer");
  auto chunks = [](int rows) { int gy = (rows + 255) / 256; return gy > 64 ? 64 : (gy < 1 ? 1 : gy); };
  const ColsumSide sa{(const bf16_t*)a, rows_a, lda, skip_a, sid_a, chunks(rows_a)};
  const ColsumSide sb{(const bf16_t*)b, rows_b, ldb, skip_b, sid_b, chunks(rows_b)};
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  hipLaunchKernelGGL(k_colsum_pair, dim3((N + 63) / 64, sa.gy > sb.gy ? sa.gy : sb.gy, 2), dim3(256), 0, stream, sa, sb, N,
                     (float*)workspace, thr, ik, seed);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_partials_reduce, dim3((N + 15) / 16, 1), dim3(256), 0, stream, (const float*)workspace, sa.gy + sb.gy,
                     1, N, out, (float*)nullptr, (float*)nullptr, 0);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_colsum(const void* a, int rows, int N, int lda, float* out, void* workspace, size_t ws_bytes,
              hipStream_t stream) {
  return zk_colsum_ex(a, rows, N, lda, out, 0, 0, 0.f, nullptr, 0, workspace, ws_bytes, stream);
}

// descs: DEVICE arrays (layouts: struct ColsumDesc / struct ReduceDesc in zk_elem.hip)
int zk_colsum_grouped(const void* descs, int nprob, int total_blocks, hipStream_t stream) {
  if (nprob == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(k_colsum_grouped, dim3(total_blocks), dim3(256), 0, stream, (const ColsumDesc*)descs, nprob);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_reduce_grouped(const void* descs, int nprob, int total_blocks, hipStream_t stream) {
  if (nprob == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(k_reduce_grouped, dim3(total_blocks), dim3(256), 0, stream, (const ReduceDesc*)descs, nprob);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_ln_bwd_blocks(int rows) { return ln_bwd_blocks(rows); }
int zk_colsum_rowchunks(int rows) {
  int gy = (rows + 255) / 256;
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  return gy;
}

int zk_embed_bwd_sorted(const int* rows_sorted, const int* seg, const int* uid, const int* n_uniq_dev,
                        int max_uniq, const void* dout, float* dtable, int H, float scale, int accumulate,
                        float drop_p, const uint64_t* seed, uint32_t sid, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_embed_bwd_sorted: H=%d must be a multiple of 8", H);
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_embed_bwd_sorted: dropout needs a seed pointer");
  if (max_uniq == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  int g = (max_uniq + 3) / 4;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_embed_bwd_sorted, dim3(g), dim3(256), 0, stream, rows_sorted, seg, uid, n_uniq_dev,
                     (const bf16_t*)dout, dtable, H, scale, accumulate, thr, ik, seed, sid);
  ZK_LAUNCH_CHECK();
  return 0;
}

// the gradient scatters of two DIFFERENT embedding tables in one launch (per side the arguments of zk_embed_bwd_sorted)
int zk_embed_bwd_sorted_pair(const int* rows_a, const int* seg_a, const int* uid_a, const int* n_a, int max_a, const void* dout_a,
                             float* dtable_a, int acc_a, uint32_t sid_a, const int* rows_b, const int* seg_b, const int* uid_b,
                             const int* n_b, int max_b, const void* dout_b, float* dtable_b, int acc_b, uint32_t sid_b, int H,
                             float scale, float drop_p, const uint64_t* seed, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_embed_bwd_sorted_pair: H=%d must be a multiple of 8", H);
  ZK_CHECK_ARG(drop_p == 0.f || seed != nullptr, "zk_embed_bwd_sorted_pair: dropout needs a seed pointer");
  ZK_CHECK_ARG(dtable_a != dtable_b, "zk_embed_bwd_sorted_pair: the two sides must write different tables");
  const int mx = max_a > max_b ? max_a : max_b;
  if (mx == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  int g = (mx + 3) / 4;
  if (g > 2048) g = 2048;
  const EmbedBwdSide a{rows_a, seg_a, uid_a, n_a, (const bf16_t*)dout_a, dtable_a, acc_a, sid_a};
  const EmbedBwdSide b{rows_b, seg_b, uid_b, n_b, (const bf16_t*)dout_b, dtable_b, acc_b, sid_b};
  hipLaunchKernelGGL(k_embed_bwd_sorted_pair, dim3(g, 2), dim3(256), 0, stream, a, b, H, scale, thr, ik, seed);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_ce_fused(const float* logits, const int* ids, const float* w, float* ce_out, void* dlogits, int rows,
                int V, int ld, float label_smooth, hipStream_t stream) {
  ZK_CHECK_ARG(ld % 4 == 0 && ld >= V, "zk_ce_fused: ld=%d must be a multiple of 4 and >= V=%d", ld, V);
  if (rows == 0) return 0;
  float p = 1.f, q = 0.f, normalizer = 0.f;
  if (label_smooth > 0.f && label_smooth < 1.f) {  // util.py:90-97, fp32 arithmetic
    const float n = (float)(V - 1);
    p = 1.f - label_smooth;
    q = label_smooth / n;
    normalizer = -(p * logf(p) + n * q * logf(q + 1e-20f));
  }
  if (ld <= 8 * 4096 && ld > 4 * 4096)
    hipLaunchKernelGGL(k_ce_fused_reg<8>, dim3(rows), dim3(1024), 0, stream, logits, ids, w, ce_out,
                       (bf16_t*)dlogits, V, ld, p, q, normalizer);
  else if (ld <= 4 * 4096 && ld > 4096)
    hipLaunchKernelGGL(k_ce_fused_reg<4>, dim3(rows), dim3(1024), 0, stream, logits, ids, w, ce_out,
                       (bf16_t*)dlogits, V, ld, p, q, normalizer);
  else
    hipLaunchKernelGGL(k_ce_fused, dim3(rows), dim3(256), 0, stream, logits, ids, w, ce_out, (bf16_t*)dlogits,
                       V, ld, p, q, normalizer);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_target_stats(const int* ids, float* mask, float* w, int B, int L, float loss_scale,
                    hipStream_t stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_target_stats, dim3(B), dim3(256), 0, stream, ids, mask, w, B, L, loss_scale);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_loss_reduce(const float* ce, const int* ids, float* per_sample, float* loss, int B, int L,
                   hipStream_t stream) {
  ZK_CHECK_ARG(per_sample != nullptr, "zk_loss_reduce: per_sample buffer required");
  if (B > 0 && B <= 4096 && !(g_tune[16] & 1)) {      // (tuning key 16 bit 0: the two-launch form, for A/B and the tests)
    hipLaunchKernelGGL(k_loss_tail, dim3(1), dim3(1024), 0, stream, ce, ids, per_sample, loss, B, L);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  if (B > 0) {
    hipLaunchKernelGGL(k_per_sample, dim3(B), dim3(256), 0, stream, ce, ids, per_sample, L);
    ZK_LAUNCH_CHECK();
  }
  if (loss != nullptr) {
    hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, stream, (const float*)per_sample, loss, B);
    ZK_LAUNCH_CHECK();
  }
  return 0;
}

int zk_make_mask(const int* ids, float* mask, int n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_make_mask, dim3((n + 255) / 256), dim3(256), 0, stream, ids, mask, n);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_all_equal(const int* ids, int n, int value, int* flag, hipStream_t stream) {
  hipLaunchKernelGGL(k_all_equal, dim3(1), dim3(256), 0, stream, ids, n, value, flag);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_aan_fwd(const void* x, const float* mask, void* cat, int B, int L, int H, int use_mask,
               hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_aan_fwd: H=%d must be a multiple of 8", H);
  const int n = B * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aan_fwd, dim3(B, (H / 8 + 63) / 64), dim3(256), 0, stream, (const bf16_t*)x, mask,
                     (bf16_t*)cat, B, L, H, use_mask);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_aan_bwd(const void* dcat, const void* dxg, const void* dyg, const void* ds, const float* mask, void* dx,
               int B, int L, int H, int use_mask, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_aan_bwd: H=%d must be a multiple of 8", H);
  const int n = B * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aan_bwd, dim3(B, (H / 8 + 63) / 64), dim3(256), 0, stream, (const bf16_t*)dcat,
                     (const bf16_t*)dxg, (const bf16_t*)dyg, (const bf16_t*)ds, mask, (bf16_t*)dx, B, L, H,
                     use_mask);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_cumavg_add_fwd(const void* vq, const float* mask, const void* att, void* out, int B, int L, int H,
                      hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_cumavg_add_fwd: H=%d must be a multiple of 8", H);
  const int n = B * (H / 8);
  if (n == 0 || L == 0) return 0;
  hipLaunchKernelGGL(k_cumavg_add_fwd, dim3(B, (H / 8 + 63) / 64), dim3(256), 0, stream, (const bf16_t*)vq, mask,
                     (const bf16_t*)att, (bf16_t*)out, B, L, H);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_cumavg_bwd(const void* dy, const float* mask, void* dvq, int B, int L, int H, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_cumavg_bwd: H=%d must be a multiple of 8", H);
  const int n = B * (H / 8);
  if (n == 0 || L == 0) return 0;
  hipLaunchKernelGGL(k_cumavg_bwd, dim3(B, (H / 8 + 63) / 64), dim3(256), 0, stream, (const bf16_t*)dy, mask,
                     (bf16_t*)dvq, B, L, H);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_sum_slices(float* out, const float* in, int nslices, size_t n, size_t stride, int accumulate,
                  hipStream_t stream) {
  if (n == 0 || nslices <= 0) return 0;
  ZK_CHECK_ARG(stride >= n, "zk_sum_slices: stride must be >= n");
  hipLaunchKernelGGL(k_sum_slices, dim3(flat_grid(n, 1)), dim3(256), 0, stream, out, in, nslices, n, stride,
                     accumulate);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_add_bf16(void* out, int ldo, const void* a, int lda, const void* b, int ldb, int rows, int cols,
                hipStream_t stream) {
  ZK_CHECK_ARG(cols % 8 == 0 && ldo % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0,
               "zk_add_bf16: cols=%d and leading dimensions must be multiples of 8", cols);
  const size_t n = (size_t)rows * (cols / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_add_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16_t*)out, ldo,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, rows, cols);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_aan_gate_fwd(const void* z, const void* cat, void* out, int rows, int H, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_aan_gate_fwd: H=%d must be a multiple of 8", H);
  const size_t n = (size_t)rows * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aan_gate_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)z,
                     (const bf16_t*)cat, (bf16_t*)out, rows, H);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_aan_gate_bwd(const void* dg, const void* z, const void* cat, void* dz, void* dxg, void* dyg, int rows,
                    int H, hipStream_t stream) {
  ZK_CHECK_ARG(H % 8 == 0, "zk_aan_gate_bwd: H=%d must be a multiple of 8", H);
  const size_t n = (size_t)rows * (H / 8);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aan_gate_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (const bf16_t*)dg, (const bf16_t*)z, (const bf16_t*)cat, (bf16_t*)dz, (bf16_t*)dxg,
                     (bf16_t*)dyg, rows, H);
  ZK_LAUNCH_CHECK();
  return 0;
}

#define ZK_NORM_BLOCKS 1024
size_t zk_norm_workspace(void) { return 2048 * sizeof(float); }

// out[0] = scale * ||x||_2   (tf.global_norm over the flat buffer, cycle.py:94-95)
int zk_l2norm(const float* x, size_t n, float scale, float* out, void* workspace, size_t ws_bytes,
              hipStream_t stream) {
  ZK_CHECK_ARG(ws_bytes >= zk_norm_workspace(), "zk_l2norm: workspace too small");
  ZK_CHECK_ARG(((uintptr_t)x & 15) == 0, "zk_l2norm: x must be 16-byte aligned");
  hipLaunchKernelGGL(k_sumsq_partial, dim3(ZK_NORM_BLOCKS), dim3(256), 0, stream, x, n, (float*)workspace);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, stream, (const float*)workspace, ZK_NORM_BLOCKS,
                     scale, out);
  ZK_LAUNCH_CHECK();
  return 0;
}

// pnorm_out (device float, may be NULL): ||p||_2 of the parameters BEFORE the update
// (cycle.py:95), accumulated in the same pass; workspace >= zk_norm_workspace() bytes then.
int zk_adam(float* p, const float* g, float* m, float* v, void* shadow, size_t n, float* hyper,
            float* pnorm_out, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "zk_adam: buffers must be 16-byte aligned");
  ZK_CHECK_ARG(pnorm_out == nullptr || ws_bytes >= 2048 * sizeof(float), "zk_adam: workspace too small");
  if (n == 0) return 0;
  const int grid = flat_grid(n, 4);
  float* psq = pnorm_out ? (float*)workspace : nullptr;
  hipLaunchKernelGGL(k_adam<false>, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow, n, hyper, psq,
                     (float*)nullptr, (uint64_t*)nullptr);
  ZK_LAUNCH_CHECK();
  if (pnorm_out) {
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, stream, (const float*)psq, grid, 1.f, pnorm_out);
    ZK_LAUNCH_CHECK();
  }
  return 0;
}
size_t zk_adam_step_workspace(void) { return 2 * 2048 * sizeof(float); }
// The whole update of one step in two launches.  norm_free = 1 (cycle.py:98-101 with clip_grad_norm 0.0 and no
// safe_nan): gradient norm (-> hyper[6], flag hyper[7], sticky count hyper[10]), TF1 Adam, bf16 shadow and parameter
// norm in ONE pass over the buffers + a one-block finish.  norm_free = 0: hyper[6] must already hold the gradient norm
// (zk_l2norm); the update is skipped when it is not finite / above hyper[9].  seed (device uint64, may be NULL) += 1.
int zk_adam_step(float* p, const float* g, float* m, float* v, void* shadow, size_t n, float* hyper,
                 float* pnorm_out, uint64_t* seed, int norm_free, const int* skip_word, void* workspace, size_t ws_bytes,
                 hipStream_t stream) {
  ZK_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "zk_adam_step: buffers must be 16-byte aligned");
  ZK_CHECK_ARG(ws_bytes >= zk_adam_step_workspace(), "zk_adam_step: workspace too small");
  ZK_CHECK_ARG(hyper != nullptr, "zk_adam_step: hyper is required");
  if (n == 0) return 0;
  const int grid = flat_grid(n, 4);
  float* psq = (float*)workspace;
  float* gsq = psq + 2048;
  if (norm_free) {
    // tuning key 9: 0 grid-stride, 1 / 2 / 4 contiguous range per block with that many float4 per thread and round
    // (default 2); tuning key 10: blocks (default 512; 0 = 2048).  scripts/adam_bench.py: 405 -> 391 us on one box
    // (fewer open DRAM pages); the box-to-box spread of this pass is larger than that (405 .. 477 us)
    const int gb = g_tune[10] > 0 && g_tune[10] <= 2048 ? (grid < g_tune[10] ? grid : g_tune[10]) : grid;
#define ZK_ADAM_L(...) hipLaunchKernelGGL((k_adam<__VA_ARGS__>), dim3(gb), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow, n, hyper, psq, gsq, seed, skip_word, g_tune[18])
    if (g_tune[9] == 1) ZK_ADAM_L(true, 1, true);
    else if (g_tune[9] == 2) ZK_ADAM_L(true, 2, true);
    else if (g_tune[9] == 4) ZK_ADAM_L(true, 4, true);
    else ZK_ADAM_L(true, 1, false);
#undef ZK_ADAM_L
    ZK_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_norm_final2, dim3(1), dim3(256), 0, stream, (const float*)psq, (const float*)gsq, gb, hyper,
                       pnorm_out);
  } else {
    hipLaunchKernelGGL(k_adam<false>, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow, n, hyper, psq,
                       (float*)nullptr, seed, skip_word);
    ZK_LAUNCH_CHECK();
    if (pnorm_out) hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, stream, (const float*)psq, grid, 1.f, pnorm_out);
  }
  ZK_LAUNCH_CHECK();
  return 0;
}
#ifdef ZK_EXPERIMENTS   // the update inside the weight-gradient launch: measured slower (profiles/r04_negative_results.txt)
// The rest of a step's update when the weight matrices were updated inside the weight-gradient launch
// (zk_gemm_grouped_update): TF1 Adam (norm-free form) on the nseg segments [seg_lo[s], seg_lo[s] + (prefix[s+1] -
// prefix[s])) of the flat buffers -- DEVICE int64 arrays in ELEMENTS, multiples of 4, ascending, prefix[0] = 0 --, then the
// norms over BOTH parts: extra [n_extra][2] are the wave partials of the fused launch.  seed (may be NULL) += 1.
int zk_adam_step_segments(float* p, const float* g, float* m, float* v, void* shadow, const long* seg_lo, const long* prefix,
                          int nseg, long total, float* hyper, float* pnorm_out, uint64_t* seed, const float* extra,
                          int n_extra, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)shadow) & 15) == 0,
               "zk_adam_step_segments: buffers must be 16-byte aligned");
  ZK_CHECK_ARG(ws_bytes >= zk_adam_step_workspace(), "zk_adam_step_segments: workspace too small");
  ZK_CHECK_ARG(hyper != nullptr && shadow != nullptr && nseg >= 1 && nseg <= ZK_ADAM_MAXSEG && total >= 0 && total % 4 == 0 &&
               seg_lo != nullptr && prefix != nullptr && (n_extra == 0 || extra != nullptr),
               "zk_adam_step_segments: bad arguments (nseg=%d, at most %d)", nseg, ZK_ADAM_MAXSEG);
  float* psq = (float*)workspace;
  float* gsq = psq + 2048;
  int gb = g_tune[10] > 0 && g_tune[10] <= 2048 ? g_tune[10] : 512;
  const long n4 = total / 4;
  if (n4 < (long)gb * 256) gb = (int)((n4 + 255) / 256);
  if (gb < 1) gb = 1;
  hipLaunchKernelGGL(k_adam_seg, dim3(gb), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow, seg_lo, prefix, nseg, hyper,
                     psq, gsq, seed);
  ZK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_norm_final3, dim3(1), dim3(256), 0, stream, (const float*)psq, (const float*)gsq, gb, extra, n_extra,
                     hyper, pnorm_out);
  ZK_LAUNCH_CHECK();
  return 0;
}
#endif  // ZK_EXPERIMENTS
#ifdef ZK_EXPERIMENTS   // Adam beside the encoder backward (5.01 vs 4.94 ms): make EXPERIMENTS=1
// The norm-free update in pieces (utils/parallel.py buckets; with ONE rank: the decoder-side parameters are updated
// on a side stream while the encoder backward is still running).  zk_adam_range: TF1 Adam + shadow refresh on n
// elements, per-block sums of squares of the scaled gradient and of the parameters into slot `slot` (< 16) of the
// workspace (always 2048 blocks, so every entry of the slot is written); zk_adam_finish: one block sums nslots slots
// -> hyper[6] gradient norm (+ flag hyper[7], sticky count hyper[10]), pnorm_out, seed += 1.
size_t zk_adam_range_workspace(void) { return 16 * 4096 * sizeof(float); }
int zk_adam_range(float* p, const float* g, float* m, float* v, void* shadow, size_t n, float* hyper, int slot,
                  void* workspace, size_t ws_bytes, hipStream_t stream) {
  ZK_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "zk_adam_range: buffers must be 16-byte aligned");
  ZK_CHECK_ARG(slot >= 0 && slot < 16 && ws_bytes >= zk_adam_range_workspace() && hyper != nullptr,
               "zk_adam_range: slot %d / workspace", slot);
  float* psq = (float*)workspace + (size_t)slot * 4096;
  // tuning key 13: blocks of a piece (default 2048).  A piece that runs BESIDE a latency-bound chain can be given few
  // blocks so that it trickles along on a fraction of the HBM bandwidth and of the wave slots (round-4 probe).
  const int nb = g_tune[13] > 0 && g_tune[13] <= 2048 ? g_tune[13] : 2048;
  hipLaunchKernelGGL(k_adam<true>, dim3(nb), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow, n, hyper, psq,
                     psq + 2048, (uint64_t*)nullptr);
  ZK_LAUNCH_CHECK();
  return 0;
}
__global__ void __launch_bounds__(256) k_adam_finish(const float* __restrict__ ws, int nslots, float* __restrict__ hyper,
                                                     float* __restrict__ pnorm_out, uint64_t* __restrict__ seed, int nb) {
  __shared__ float sm[8];
  float a = 0.f, b = 0.f;
  for (int s = 0; s < nslots; ++s)
    for (int i = threadIdx.x; i < nb; i += 256) { b += ws[s * 4096 + i]; a += ws[s * 4096 + 2048 + i]; }
  a = block_sum<4>(a, sm);
  b = block_sum<4>(b, sm);
  if (threadIdx.x == 0) {
    const float gn = sqrtf(a);
    hyper[6] = gn;
    const bool bad = !(gn == gn) || fabsf(gn) == INFINITY;
    hyper[7] = bad ? 1.f : 0.f;
    if (bad) hyper[10] += 1.f;
    if (pnorm_out != nullptr) pnorm_out[0] = sqrtf(b);
    if (seed != nullptr) *seed += 1;
  }
}
int zk_adam_finish(float* hyper, float* pnorm_out, uint64_t* seed, int nslots, const void* workspace, size_t ws_bytes,
                   hipStream_t stream) {
  ZK_CHECK_ARG(hyper != nullptr && nslots >= 1 && nslots <= 16 && ws_bytes >= zk_adam_range_workspace(),
               "zk_adam_finish: nslots %d / workspace", nslots);
  hipLaunchKernelGGL(k_adam_finish, dim3(1), dim3(256), 0, stream, (const float*)workspace, nslots, hyper, pnorm_out, seed,
                     g_tune[13] > 0 && g_tune[13] <= 2048 ? g_tune[13] : 2048);
  ZK_LAUNCH_CHECK();
  return 0;
}
#endif  // ZK_EXPERIMENTS
int zk_norm_flag(float* hyper, hipStream_t stream) {
  ZK_CHECK_ARG(hyper != nullptr, "zk_norm_flag: hyper is required");
  hipLaunchKernelGGL(k_norm_flag, dim3(1), dim3(1), 0, stream, hyper);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_cast_f32_bf16(const float* x, void* y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  ZK_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "zk_cast_f32_bf16: misaligned");
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3(flat_grid(n, 4)), dim3(256), 0, stream, x, (bf16_t*)y, n);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_cast_bf16_f32(const void* x, float* y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_cast_bf16_f32, dim3(flat_grid(n, 1)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_axpby_f32(float* y, const float* x, float a, float b, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpy_f32, dim3(flat_grid(n, 1)), dim3(256), 0, stream, y, x, a, b, n);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_ema(float* ema, const float* p, const float* hyper, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(ema) | reinterpret_cast<uintptr_t>(p)) & 15)
    return zk_set_error(-1, "zk_ema: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_ema, dim3(flat_grid(n, 4)), dim3(256), 0, stream, ema, p, hyper, n);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_dropout_mask(float* out, size_t n, float drop_p, const uint64_t* seed, uint32_t sid,
                    hipStream_t stream) {
  if (n == 0) return 0;
  const uint32_t thr = drop_p > 0.f ? zk_drop_threshold(drop_p) : 0;
  const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  hipLaunchKernelGGL(k_dropout_mask, dim3(flat_grid(n, 1)), dim3(256), 0, stream, out, n, thr, ik, seed, sid);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_seed_advance(uint64_t* seed, uint64_t inc, hipStream_t stream) {
  hipLaunchKernelGGL(k_seed_advance, dim3(1), dim3(1), 0, stream, seed, inc);
  ZK_LAUNCH_CHECK();
  return 0;
}

// Host-side CRC32C (Castagnoli, reflected 0x82f63b78), slicing-by-8: checksums of checkpoint
// tensors in the TensorFlow bundle format (zero_amd/utils/bundle.py; utils/saver.py:75,131-170).
// Pure host function: no device work, callable without a GPU.
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
  g_crc_ready = true;
}
uint32_t zk_crc32c(const void* data, size_t n, uint32_t crc) {
  if (!g_crc_ready) crc_init();
  const unsigned char* p = static_cast<const unsigned char*>(data);
  crc = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { crc = g_crc_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= crc;
    crc = g_crc_tab[7][w & 0xff] ^ g_crc_tab[6][(w >> 8) & 0xff] ^ g_crc_tab[5][(w >> 16) & 0xff] ^
          g_crc_tab[4][(w >> 24) & 0xff] ^ g_crc_tab[3][(w >> 32) & 0xff] ^ g_crc_tab[2][(w >> 40) & 0xff] ^
          g_crc_tab[1][(w >> 48) & 0xff] ^ g_crc_tab[0][(w >> 56) & 0xff];
    p += 8; n -= 8;
  }
  while (n--) crc = g_crc_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
  return ~crc;
}

// Measurement aid: keeps the stream busy for `usec` so that the launches enqueued behind it are
// already queued when they start (bench.py's per-launch HIP-event brackets then exclude the host
// launch latency).  One thread polling the 100 MHz wall clock.
__global__ void k_spin(uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int zk_spin(uint32_t usec, hipStream_t stream) {
  if (usec > 200000u) return zk_set_error(-1, "zk_spin: at most 200 ms");
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, stream, (uint64_t)usec * 100u);
  ZK_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_zero16(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int zk_zero(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  if ((((uintptr_t)p | bytes) & 15) == 0) {      // a kernel node, not a memset node (hipGraph replays)
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_zero16, dim3(flat_grid(n16, 1)), dim3(256), 0, stream, (uint4*)p, n16);
    ZK_LAUNCH_CHECK();
    return 0;
  }
  hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
  if (e != hipSuccess) return zk_set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
  return 0;
}

// ---- thin hipGraph wrappers (launch-bound step loop is captured once and replayed)
}  // extern "C"
static std::mutex g_graph_mu;
static std::unordered_map<void*, hipGraph_t> g_graph_of;      // executable -> the graph it was instantiated from
hipGraph_t zk_graph_template_of(void* exec) {                   // (zk_common.h; used by zk_prep.hip)
  std::lock_guard<std::mutex> lk(g_graph_mu);
  auto it = g_graph_of.find(exec);
  return it == g_graph_of.end() ? nullptr : it->second;
}
extern "C" {
int zk_graph_begin(hipStream_t stream) {
  hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return zk_set_error((int)e, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  return 0;
}
static int g_last_graph_nodes = 0;
int zk_graph_last_nodes(void) { return g_last_graph_nodes; }
int zk_graph_end(hipStream_t stream, void** exec_out) {
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(stream, &graph);
  if (e != hipSuccess) return zk_set_error((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
  size_t n_nodes = 0;
  if (graph == nullptr || hipGraphGetNodes(graph, nullptr, &n_nodes) != hipSuccess || n_nodes == 0) {
    if (graph) hipGraphDestroy(graph);      // nothing was captured: a null handle that launches as a no-op
    *exec_out = nullptr;
    return 0;
  }
  g_last_graph_nodes = (int)n_nodes;
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(graph);
    return zk_set_error((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  {   // the captured graph stays alive beside its executable: its node handles name the nodes whose parameters
      // zk_graph_set_copy_many rewrites before a replay (released by zk_graph_destroy)
    std::lock_guard<std::mutex> lk(g_graph_mu);
    g_graph_of[(void*)exec] = graph;
  }
  *exec_out = (void*)exec;
  return 0;
}
int zk_graph_launch(void* exec, hipStream_t stream) {
  if (exec == nullptr) return 0;
  hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, stream);
  if (e != hipSuccess) return zk_set_error((int)e, "hipGraphLaunch: %s", hipGetErrorString(e));
  return 0;
}
int zk_graph_destroy(void* exec) {
  if (exec == nullptr) return 0;
  hipGraph_t graph = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    auto it = g_graph_of.find(exec);
    if (it != g_graph_of.end()) { graph = it->second; g_graph_of.erase(it); }
  }
  zk_graph_forget_nodes(exec);
  hipGraphExecDestroy((hipGraphExec_t)exec);
  if (graph) hipGraphDestroy(graph);
  return 0;
}

}  // extern "C"
