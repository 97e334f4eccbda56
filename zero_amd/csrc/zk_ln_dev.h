// zk_ln_dev.h -- residual add + LayerNorm forward of ONE row by one wave (func.py:289-303, 321-324); shared by
// k_add_ln_fwd (zk_elem.hip) and the layer program (zk_layer.hip).
#pragma once
#include "zk_common.h"

template <int MAXC, bool FRESH = false>
__device__ __forceinline__ void add_ln_fwd_row(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out, bf16_t* __restrict__ sum_out,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int r, int H, float invH, float eps,
    uint32_t thr, float inv_keep, uint64_t seed, uint32_t sid, int lane) {
  float v[MAXC][8];
  float s1 = 0.f;
  // scale / offset are requested with the row, not after the two reductions (one memory round trip less per launch)
  float gm[MAXC][8], bt[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + c + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(beta + c + 4 * q);
        gm[i][4 * q] = g4.x; gm[i][4 * q + 1] = g4.y; gm[i][4 * q + 2] = g4.z; gm[i][4 * q + 3] = g4.w;
        bt[i][4 * q] = b4.x; bt[i][4 * q + 1] = b4.y; bt[i][4 * q + 2] = b4.z; bt[i][4 * q + 3] = b4.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
      float a[8], b[8];
      unpack8(zk_ld16<FRESH>(x + (size_t)r * H + c), a);
      if (y != nullptr) {
        unpack8(zk_ld16<FRESH>(y + (size_t)r * H + c), b);
        if (thr != 0) {
          float dm[8];
          zk_drop_scale8(seed, sid, (uint64_t)r * H + c, thr, inv_keep, dm);
#pragma unroll
          for (int j = 0; j < 8; ++j) b[j] *= dm[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
      }
      // training (sum_out given): the sum is saved as bf16 for the backward, and the statistics are taken from the SAVED
      // values so that forward and backward normalise the same rows.  Inference (round 6): nothing is saved, the sum stays fp32.
      if (sum_out != nullptr) {
        uint4 p = pack8(a);
        *reinterpret_cast<uint4*>(sum_out + (size_t)r * H + c) = p;
        unpack8(p, v[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = a[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += v[i][j];
    }
  }
  const float mean = wave_sum(s1) * invH;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; s2 += d * d; }
    }
  }
  const float var = wave_sum(s2) * invH;
  const float rstd = rsqrtf(var + eps);
  if (lane == 0 && mean_out != nullptr) { mean_out[r] = mean; rstd_out[r] = rstd; }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = gm[i][j] * (v[i][j] - mean) * rstd + bt[i][j];
      *reinterpret_cast<uint4*>(out + (size_t)r * H + c) = pack8(o);
    }
  }
}
