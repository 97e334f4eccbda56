// zk_gemm.h -- pieces shared by the GEMM kernel generations (zk_gemm.hip, zk_gemm2.hip)
#pragma once
#include "zk_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define BK 64
#define LDS_LD (BK + 8)

struct GemmEpi {
  void* C; int ldc; int out_f32; float alpha;
  const float* bias;
  const bf16_t* res; int ldr;
  int act;                 // 0 none, 1 relu, 2 multiply by (aux>0)*aux_scale
  const bf16_t* aux; int ldaux; float aux_scale;
  uint32_t thr; float inv_keep; const uint64_t* seed; uint32_t sid;  // dropout on the output
  // ---- round 4: residual + LayerNorm WITHOUT a launch of its own (func.py:289-303, 321-324; post-LN order of
  // transformer.py:57-58).  The GEMM that produces a sub-layer's output adds the residual itself and leaves, beside the
  // un-normalised sum s (bf16), the row statistics of s as per-64-column partials; every reader of LN(s) applies the
  // normalisation where it reads (gen-2 tile kernels only, 16-byte epilogue, N % 64 == 0):
  //   producer   ln_stat_out != null: {sum, M2} of the bf16-ROUNDED outputs of each (row, 64-column group) ->
  //              ln_stat_out[(row * N/64 + group) * 2 ..]; res_after_drop: out = res + dropout(acc + bias);
  //   lazy residual   res_part != null: `res` holds the previous sub-layer's un-normalised sum, the residual is
  //              bf16(gamma (res - mu) rstd + beta) with (mu, rstd) combined from res_part [M][ln_np][2];
  //   consumer   ln_c != null: A was an un-normalised sum and B the weight with gamma folded in (zk_ln_fold):
  //              out = rstd_row (acc - mu_row ln_c[n]) + bias[n], bias = beta . W + b  (then act / dropout as usual).
  float* ln_stat_out = nullptr;
  const float* ln_in_part = nullptr; const float* ln_c = nullptr;
  const float* res_part = nullptr; const float* res_gamma = nullptr; const float* res_beta = nullptr;
  int ln_np = 0; int res_after_drop = 0; float ln_eps = 0.f; float ln_invh = 0.f;
  // ---- residual + LayerNorm INSIDE the producing launch (zk_gemm_add_ln; gemm_tile<.., LN = 3>): the N/64 workgroups
  // that hold one block of rows exchange their per-64-column {sum, M2} through `sy_slots` ([rows][N/64] x 16 bytes:
  // {sum, tag, M2, tag}, agent-scope 8-byte atomics, each half validated by its own tag) and every one of them
  // normalises its own 64 columns: C = s = res + dropout(bf16(acc + bias)) (may be null), sy_y = LN(s),
  // sy_mean / sy_rstd (may be null) by the tn = 0 tile.  tag = (*sy_epoch << 8) | sy_site: *sy_epoch is bumped once
  // per forward pass (zk_ln_epoch_bump), the site differs between the launches of one pass, so a slot left by an
  // earlier launch never validates.  *sy_err is set if a peer's partial did not arrive within the spin budget.
  unsigned long long* sy_slots = nullptr; const uint32_t* sy_epoch = nullptr; uint32_t sy_site = 0;
  const float* sy_gamma = nullptr; const float* sy_beta = nullptr; bf16_t* sy_y = nullptr; int sy_ldy = 0;
  float* sy_mean = nullptr; float* sy_rstd = nullptr; int* sy_err = nullptr; int sy_local = 0;
  // ---- the BACKWARD of a residual + LayerNorm inside the dgrad launch that produces its input gradient (zk_gemm_ln_bwd;
  // gemm_tile<.., LN = 4>): dout = bf16(acc + res) is the gradient of the normalised rows (never stored); with the saved
  // sum sy_s, its statistics sy_mean_in / sy_rstd_in and sy_gamma:  g = dout gamma, xh = (s - mu) rstd,
  // ds = rstd (g - mean(g) - xh mean(g xh)) -> sy_y (bf16), dy = bf16(ds) dropmask -> sy_dy (null without dropout),
  // column partials {sum dout xh, sum dout, sum dy} over the tile's rows -> sy_part [tiles_m][3][N].  The two row means
  // are exchanged between the row block's workgroups exactly as the forward's {sum, M2} (same slots, epoch and sites).
  const bf16_t* sy_s = nullptr; int sy_lds = 0; const float* sy_mean_in = nullptr; const float* sy_rstd_in = nullptr;
  bf16_t* sy_dy = nullptr; float* sy_part = nullptr;
  int sy_fault = 0;         // tests only (tuning key 15 bit 1): the tn = 1 tiles never publish -> every peer runs into its spin budget
  int sy_rows = 0;          // > 0: only the first sy_rows rows of the tile are this tile's (sentence-aligned row tiles, zk_attn_out_ln)
};

__device__ __forceinline__ void epi_store(const GemmEpi& e, float v, int gm, int gn, int N, uint64_t seed) {
  v *= e.alpha;
  if (e.bias) v += e.bias[gn];
  if (e.res) v += bf2f(e.res[(size_t)gm * e.ldr + gn]);
  if (e.act == 1) v = fmaxf(v, 0.f);
  else if (e.act == 2) v = (bf2f(e.aux[(size_t)gm * e.ldaux + gn]) > 0.f) ? v * e.aux_scale : 0.f;
  if (e.thr) v *= zk_drop_scale(seed, e.sid, (uint64_t)gm * N + gn, e.thr, e.inv_keep);
  if (e.out_f32) reinterpret_cast<float*>(e.C)[(size_t)gm * e.ldc + gn] = v;
  else reinterpret_cast<bf16_t*>(e.C)[(size_t)gm * e.ldc + gn] = f2bf(v);
}


// Tile schedule.  The launch is a 1-D grid of tiles_m*tiles_n*splits workgroups.  Workgroup b is
// observed to run on XCD b%8 (each XCD has a private 4 MiB L2), so the linear id is first
// remapped so that every XCD owns a CONTIGUOUS range of the tile order (bijective for any grid
// size); the tile order itself is split-major, then panel-major over the operand whose panels
// should stay L2-resident (n_major=0: consecutive tiles share the A row panel; n_major=1: they
// share the B panel).  Pure speed choice -- any placement gives the same result.
struct TileSched { int tiles_m, tiles_n, n_major, xcd_remap; };

__device__ __forceinline__ void tile_of_block(const TileSched& ts, int& tm, int& tn, int& z) {
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  const int t = ts.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc : bid;
  const int per = ts.tiles_m * ts.tiles_n;
  z = t / per;
  const int rem = t - z * per;
  const int d = ts.n_major ? ts.tiles_m : ts.tiles_n;
  const int hi = rem / d, lo = rem - hi * d;
  tm = ts.n_major ? lo : hi;
  tn = ts.n_major ? hi : lo;
}

