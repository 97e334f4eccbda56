// zk_lndec_dev.h -- the residual + LayerNorm of a decode step with its row-local neighbours, ONE row by one wave
// (transformer_aan.py:165-192, 92-117; func.py:289-303).  Shared by k_ln_decode (zk_decode.hip: one launch) and the
// prologue of the fused decode attention (zk_decfuse.hip: every workgroup of a sentence recomputes its rows).
//   y of the sub-layer is one of
//     ybuf                              (bf16 [rows, H], read)
//     gate(z, cat_in)   cat_in != NULL  bf16(sigmoid(z_i) x_cat + sigmoid(z_f) y_cat) (k_aan_gate_fwd); z bf16 [rows, 2H],
//                                       or z == NULL: z = bf16(sum_p parts[p][r][0 .. 2H) + bias) (the z_project GEMM
//                                       left as split-K partial products)
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

// The rows stay in registers from the loads to the normalised output (the separate kernels went through memory between
// the gate, the residual sum and the LayerNorm; y is rounded to bf16 where the launch-per-op path stores it, the residual
// sum x + y is NOT (round 6: it used to be, one rounding more per sub-layer than the checker's storage model places).  One wave handles NROW rows r0, r0 + rstep, .. (those >= rend are skipped) with every load of all of them
// requested before the first reduction: one memory round trip.  outp[n][i]: normalised row n as stored (8 bf16 of
// column (i*64 + lane)*8 ..), also written to a.out.
template <int MAXC, int NROW, typename F>
__device__ __forceinline__ void ln_decode_rows(const LnDecArgs& a, int r0, int rstep, int rend, int lane, bool do_cache,
                                               uint4 (&outp)[NROW][MAXC], F between) {
  const int H = a.H;
  const float invH = 1.f / (float)H;
  float v[NROW][MAXC][8];
  float gm[MAXC][8], bt[MAXC][8], cv[NROW][MAXC][8];
  const bool upd = a.cache != nullptr && do_cache;
  // ---- phase A: request everything (raw registers); `between()` then issues the caller's own prefetches, which
  // queue BEHIND these loads (a wave's loads return in order), and only then is anything waited for
  // (the partial-sum form keeps its eight fp32 sums in ya / yb: the forms are exclusive, the registers are shared)
  uint4 xraw[NROW][MAXC], ya[NROW][MAXC], yb[NROW][MAXC], yc[NROW][MAXC], yd[NROW][MAXC];
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
      }
    }
  }
  float inv_count = a.inv_count;
  if (upd && a.time_dev != nullptr) inv_count = 1.f / (float)(*a.time_dev + 1);
#pragma unroll
  for (int n = 0; n < NROW; ++n) {
    const int r = r0 + n * rstep;
    if (r >= rend) continue;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
        xraw[n][i] = *reinterpret_cast<const uint4*>(a.x + (size_t)r * H + c);
        if (upd) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 c4 = *reinterpret_cast<const float4*>(a.cache + (size_t)r * H + c + 4 * q);
            cv[n][i][4 * q] = c4.x; cv[n][i][4 * q + 1] = c4.y; cv[n][i][4 * q + 2] = c4.z; cv[n][i][4 * q + 3] = c4.w;
          }
        }
        if (a.cat_in != nullptr) {
          yc[n][i] = *reinterpret_cast<const uint4*>(a.cat_in + (size_t)r * 2 * H + c);
          yd[n][i] = *reinterpret_cast<const uint4*>(a.cat_in + (size_t)r * 2 * H + H + c);
          if (a.z != nullptr) {
            ya[n][i] = *reinterpret_cast<const uint4*>(a.z + (size_t)r * 2 * H + c);
            yb[n][i] = *reinterpret_cast<const uint4*>(a.z + (size_t)r * 2 * H + H + c);
          } else {
            // the gate's pre-activation z [rows, 2H] as split-K partial products (zk_gemm_parts) + bias, rounded to
            // bf16 as the GEMM epilogue would have stored it
            float zi[8], zf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { zi[j] = 0.f; zf[j] = 0.f; }
            const float* pp = a.parts + (size_t)r * 2 * H + c;
            for (int p0 = 0; p0 < a.nparts; p0 += 4) {
              float4 u[4][4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float* pq = pp + (size_t)min(p0 + q, a.nparts - 1) * a.part_stride;
                u[q][0] = *reinterpret_cast<const float4*>(pq);
                u[q][1] = *reinterpret_cast<const float4*>(pq + 4);
                u[q][2] = *reinterpret_cast<const float4*>(pq + H);
                u[q][3] = *reinterpret_cast<const float4*>(pq + H + 4);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (p0 + q < a.nparts) {
                  zi[0] += u[q][0].x; zi[1] += u[q][0].y; zi[2] += u[q][0].z; zi[3] += u[q][0].w;
                  zi[4] += u[q][1].x; zi[5] += u[q][1].y; zi[6] += u[q][1].z; zi[7] += u[q][1].w;
                  zf[0] += u[q][2].x; zf[1] += u[q][2].y; zf[2] += u[q][2].z; zf[3] += u[q][2].w;
                  zf[4] += u[q][3].x; zf[5] += u[q][3].y; zf[6] += u[q][3].z; zf[7] += u[q][3].w;
                }
              }
            }
            if (a.bias != nullptr) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { zi[j] += a.bias[c + j]; zf[j] += a.bias[H + c + j]; }
            }
            ya[n][i] = pack8(zi);
            yb[n][i] = pack8(zf);
          }
        } else if (a.parts != nullptr) {
          float ps[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) ps[j] = 0.f;
          const float* pp = a.parts + (size_t)r * H + c;
          // eight partial products per round, all sixteen loads requested before the first add (a rolled loop waits for
          // every pair of loads in turn: nparts dependent round trips); fixed summation order p = 0, 1, ..
          for (int p0 = 0; p0 < a.nparts; p0 += 8) {
            float4 u0[8], u1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int p = min(p0 + q, a.nparts - 1);
              u0[q] = *reinterpret_cast<const float4*>(pp + (size_t)p * a.part_stride);
              u1[q] = *reinterpret_cast<const float4*>(pp + (size_t)p * a.part_stride + 4);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              if (p0 + q < a.nparts) {
                ps[0] += u0[q].x; ps[1] += u0[q].y; ps[2] += u0[q].z; ps[3] += u0[q].w;
                ps[4] += u1[q].x; ps[5] += u1[q].y; ps[6] += u1[q].z; ps[7] += u1[q].w;
              }
            }
          }
          if (a.bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ps[j] += a.bias[c + j];
          }
          ya[n][i] = make_uint4(__float_as_uint(ps[0]), __float_as_uint(ps[1]), __float_as_uint(ps[2]), __float_as_uint(ps[3]));
          yb[n][i] = make_uint4(__float_as_uint(ps[4]), __float_as_uint(ps[5]), __float_as_uint(ps[6]), __float_as_uint(ps[7]));
        } else {
          ya[n][i] = *reinterpret_cast<const uint4*>(a.ybuf + (size_t)r * H + c);
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  between();
  __builtin_amdgcn_sched_barrier(0);
  // ---- phase B
#pragma unroll
  for (int n = 0; n < NROW; ++n) {
    const int r = r0 + n * rstep;
    if (r >= rend) continue;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
        float xa[8], y[8];
        if (a.cat_in != nullptr) {
          float zi[8], zf[8], xv[8], yv[8], o[8];
          unpack8(ya[n][i], zi);
          unpack8(yb[n][i], zf);
          unpack8(yc[n][i], xv);
          unpack8(yd[n][i], yv);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            o[j] = xv[j] / (1.f + __expf(-zi[j])) + yv[j] / (1.f + __expf(-zf[j]));
          unpack8(pack8(o), y);
        } else if (a.parts != nullptr) {
          const float ps[8] = {__uint_as_float(ya[n][i].x), __uint_as_float(ya[n][i].y), __uint_as_float(ya[n][i].z),
                               __uint_as_float(ya[n][i].w), __uint_as_float(yb[n][i].x), __uint_as_float(yb[n][i].y),
                               __uint_as_float(yb[n][i].z), __uint_as_float(yb[n][i].w)};
          unpack8(pack8(ps), y);
        } else {
          unpack8(ya[n][i], y);
        }
        unpack8(xraw[n][i], xa);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[n][i][j] = xa[j] + y[j];      // the residual sum stays fp32 (round 6): a decode step
                                                                    // saves nothing for a backward pass, and the bf16
                                                                    // storage model (oracle Cfg.store_bf16) has no rounding
                                                                    // between residual_fn and layer_norm (func.py:321-324, 289-303)
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NROW; ++n) {
    const int r = r0 + n * rstep;
    if (r >= rend) continue;
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += v[n][i][j];
      }
    }
    const float mean = wave_sum(s1) * invH;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[n][i][j] - mean; s2 += d * d; }
      }
    }
    const float var = wave_sum(s2) * invH;
    const float rstd = rsqrtf(var + a.eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (i * 64 + lane) * 8;
      outp[n][i] = make_uint4(0u, 0u, 0u, 0u);
      if (c < H) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = gm[i][j] * (v[n][i][j] - mean) * rstd + bt[i][j];
        outp[n][i] = pack8(o);
        *reinterpret_cast<uint4*>(a.out + (size_t)r * H + c) = outp[n][i];
        if (upd) {
          float vv[8], av[8];
          unpack8(outp[n][i], vv);
          float* cp = a.cache + (size_t)r * H + c;
#pragma unroll
          for (int j = 0; j < 8; ++j) { cv[n][i][j] += vv[j]; av[j] = cv[n][i][j] * inv_count; }
          *reinterpret_cast<float4*>(cp) = make_float4(cv[n][i][0], cv[n][i][1], cv[n][i][2], cv[n][i][3]);
          *reinterpret_cast<float4*>(cp + 4) = make_float4(cv[n][i][4], cv[n][i][5], cv[n][i][6], cv[n][i][7]);
          *reinterpret_cast<uint4*>(a.cat_out + (size_t)r * 2 * H + c) = outp[n][i];
          *reinterpret_cast<uint4*>(a.cat_out + (size_t)r * 2 * H + H + c) = pack8(av);
        }
      }
    }
  }
}

template <int MAXC>
__device__ __forceinline__ void ln_decode_row(const LnDecArgs& a, int r, int lane, bool do_cache, uint4 (&outp)[MAXC]) {
  uint4 o[1][MAXC];
  ln_decode_rows<MAXC, 1>(a, r, 1, r + 1, lane, do_cache, o, [] {});
#pragma unroll
  for (int i = 0; i < MAXC; ++i) outp[i] = o[0][i];
}
