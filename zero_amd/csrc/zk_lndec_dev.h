// zk_lndec_dev.h -- the residual + LayerNorm of a decode step with its row-local neighbours, ONE row by one wave
// (transformer_aan.py:165-192, 92-117; func.py:289-303).  Shared by k_ln_decode (zk_decode.hip: one launch) and the
// prologue of the fused decode attention (zk_decfuse.hip: every workgroup of a sentence recomputes its rows).
//   y of the sub-layer is one of
//     ybuf                              (bf16 [rows, H], read)
//     gate(z, cat_in)   z != NULL       bf16(sigmoid(z_i) x_cat + sigmoid(z_f) y_cat) (k_aan_gate_fwd)
//     sum of partials   parts != NULL   bf16(sum_p parts[p][r][:] + bias) (the o_map / FFN output projection left as
//                                       per-head / per-slice fp32 partial products by the producer)
//   out = LayerNorm(x + y);
//   cache != NULL (and do_cache): the NEXT layer's average-attention input from the normalised row
//     (k_aan_decode: cache += out; cat_out = [out | cache / (time + 1)]).
#pragma once
#include "zk_ln_dev.h"

struct LnDecArgs {
  const bf16_t* x; bf16_t* ybuf; const float* gamma; const float* beta; bf16_t* out;
  int rows, H; float eps;
  const bf16_t* z; const bf16_t* cat_in;
  const float* parts; int nparts; long part_stride; const float* bias;
  float* cache; bf16_t* cat_out; float inv_count; const int* time_dev;
};

// The row stays in registers from the loads to the normalised output (the separate kernels went through memory between
// the gate, the residual sum and the LayerNorm; the values are rounded to bf16 at the same points, so the results are the
// same bits).  outp[i]: the normalised row as stored (8 bf16 of column (i*64 + lane)*8 ..), also written to a.out.
template <int MAXC>
__device__ __forceinline__ void ln_decode_row(const LnDecArgs& a, int r, int lane, bool do_cache, uint4 (&outp)[MAXC]) {
  const int H = a.H;
  const float invH = 1.f / (float)H;
  float v[MAXC][8];
  float s1 = 0.f;
  // every load of the row is requested before the first reduction: one memory round trip, not three
  float gm[MAXC][8], bt[MAXC][8], cv[MAXC][8];
  const bool upd = a.cache != nullptr && do_cache;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(a.gamma + c + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(a.beta + c + 4 * q);
        gm[i][4 * q] = g4.x; gm[i][4 * q + 1] = g4.y; gm[i][4 * q + 2] = g4.z; gm[i][4 * q + 3] = g4.w;
        bt[i][4 * q] = b4.x; bt[i][4 * q + 1] = b4.y; bt[i][4 * q + 2] = b4.z; bt[i][4 * q + 3] = b4.w;
        if (upd) {
          const float4 c4 = *reinterpret_cast<const float4*>(a.cache + (size_t)r * H + c + 4 * q);
          cv[i][4 * q] = c4.x; cv[i][4 * q + 1] = c4.y; cv[i][4 * q + 2] = c4.z; cv[i][4 * q + 3] = c4.w;
        }
      }
    }
  }
  float inv_count = a.inv_count;
  if (upd && a.time_dev != nullptr) inv_count = 1.f / (float)(*a.time_dev + 1);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < H) {
      float xa[8], y[8];
      const uint4 xraw = *reinterpret_cast<const uint4*>(a.x + (size_t)r * H + c);
      if (a.z != nullptr) {
        float zi[8], zf[8], xv[8], yv[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(a.z + (size_t)r * 2 * H + c), zi);
        unpack8(*reinterpret_cast<const uint4*>(a.z + (size_t)r * 2 * H + H + c), zf);
        unpack8(*reinterpret_cast<const uint4*>(a.cat_in + (size_t)r * 2 * H + c), xv);
        unpack8(*reinterpret_cast<const uint4*>(a.cat_in + (size_t)r * 2 * H + H + c), yv);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = xv[j] / (1.f + __expf(-zi[j])) + yv[j] / (1.f + __expf(-zf[j]));
        unpack8(pack8(o), y);
      } else if (a.parts != nullptr) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        const float* pp = a.parts + (size_t)r * H + c;
        for (int p = 0; p < a.nparts; ++p) {         // fixed order: deterministic
          const float4 u0 = *reinterpret_cast<const float4*>(pp + (size_t)p * a.part_stride);
          const float4 u1 = *reinterpret_cast<const float4*>(pp + (size_t)p * a.part_stride + 4);
          o[0] += u0.x; o[1] += u0.y; o[2] += u0.z; o[3] += u0.w;
          o[4] += u1.x; o[5] += u1.y; o[6] += u1.z; o[7] += u1.w;
        }
        if (a.bias != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += a.bias[c + j];
        }
        unpack8(pack8(o), y);
      } else {
        unpack8(*reinterpret_cast<const uint4*>(a.ybuf + (size_t)r * H + c), y);
      }
      unpack8(xraw, xa);
#pragma unroll
      for (int j = 0; j < 8; ++j) xa[j] += y[j];
      unpack8(pack8(xa), v[i]);
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
  const float rstd = rsqrtf(var + a.eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (i * 64 + lane) * 8;
    outp[i] = make_uint4(0u, 0u, 0u, 0u);
    if (c < H) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = gm[i][j] * (v[i][j] - mean) * rstd + bt[i][j];
      outp[i] = pack8(o);
      *reinterpret_cast<uint4*>(a.out + (size_t)r * H + c) = outp[i];
      if (upd) {
        float vv[8], av[8];
        unpack8(outp[i], vv);
        float* cp = a.cache + (size_t)r * H + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) { cv[i][j] += vv[j]; av[j] = cv[i][j] * inv_count; }
        *reinterpret_cast<float4*>(cp) = make_float4(cv[i][0], cv[i][1], cv[i][2], cv[i][3]);
        *reinterpret_cast<float4*>(cp + 4) = make_float4(cv[i][4], cv[i][5], cv[i][6], cv[i][7]);
        *reinterpret_cast<uint4*>(a.cat_out + (size_t)r * 2 * H + c) = outp[i];
        *reinterpret_cast<uint4*>(a.cat_out + (size_t)r * 2 * H + H + c) = pack8(av);
      }
    }
  }
}
