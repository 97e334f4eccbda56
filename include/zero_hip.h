/* zero_hip.h -- C-ABI of libzero_hip.so, the MI355X (gfx950) kernels behind the
 * Transformer training / decode hot path of bzhangGo/zero.
 *
 * The reference has NO native layer: every entry point below replaces a group of stock
 * TF1 ops issued by the cited reference lines (paths relative to the reference root).
 * A maintainer binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers unless noted; the caller owns every
 *     buffer including workspaces (query the size with the *_workspace function);
 *   - "bf16" = raw uint16 bfloat16 storage; activations and weight matrices are bf16,
 *     vectors / statistics / gradients of parameters / optimiser state are fp32;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t); no call
 *     synchronises, allocates or keeps global state, so the calls are re-entrant and can
 *     be captured into a hipGraph;
 *   - return 0 = ok, <0 = argument error, >0 = hipError_t; message in
 *     zk_last_error_string() (thread local);
 *   - dropout: `drop_p` in [0,1), `seed` = DEVICE pointer to a uint64 step seed (read at
 *     kernel run time so a captured graph sees fresh seeds), `sid` = site id; the mask is
 *     a pure function of (seed, sid, element index) and is regenerated in the backward;
 *   - `impl`: 0 = auto (MFMA kernel when the shape allows, else the reference HIP kernel),
 *     1 = reference HIP kernel, 2 = MFMA kernel (error if unsupported).
 */
#ifndef ZERO_HIP_H_
#define ZERO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* zk_stream_t; /* hipStream_t */

int zk_version(void);
const char* zk_last_error_string(void);

/* ---- func.py:14-65 linear / func.py:327-338 ffn_layer / transformer.py:182-196 logits,
 *      and the autodiff mirrors TF builds for them (main.py:28).
 * C[M,N] = alpha * op(A)[M,K] x op(B)[K,N] (+bias[N]) (+residual[M,N]) (act) (dropout)
 *   ta=0: A is [M,lda] (K contiguous), ta=1: A is [K,lda] (M contiguous)
 *   tb=0: B is [K,ldb] (N contiguous), tb=1: B is [N,ldb] (K contiguous)
 *   out_f32: C is fp32 (else bf16).  act: 0 none, 1 ReLU (func.py:332), 2 multiply by
 *   (aux>0)*aux_scale (ReLU+dropout backward through the saved activation).          */
size_t zk_gemm_workspace(int M, int N, int K);
size_t zk_gemm_workspace_split(int M, int N, int splits);
/* Split-K product left as its partial sums (no epilogue, no reduction launch): part z = A[:, K_z] B[K_z, :] as fp32
 * [M, N] at parts + z*M*N for the z-th of *nparts_out <= splits K ranges (multiples of 64).  The consumer adds them in the
 * order z = 0, 1, .. : zk_ln_decode(parts, nparts, part_stride = M*N, bias) for the decode step's FFN output projection
 * (func.py:327-338 ffn_layer "output" at 128 rows: 64 workgroups instead of 16). */
int zk_gemm_parts(const void* A, const void* B, float* parts, int M, int N, int K, int lda, int ldb, int ta, int tb,
                  int splits, int* nparts_out, zk_stream_t stream);
/* The tail of a post-LN sub-layer in ONE launch: s = residual + dropout(bf16(A B + bias)); y = LayerNorm(s)
 * (func.py:321-324 residual_fn + func.py:289-303 layer_norm in the order of transformer.py:57-58; replaces zk_gemm followed by
 * zk_add_ln_fwd).  A [M, lda] x B [K, ldb] (no transposition), N % 64 == 0, N <= 1024.  The N/64 workgroups holding a block
 * of rows exchange the {sum, M2} of their 64 columns through `slots` (zk_gemm_add_ln_workspace bytes, zero-filled once,
 * shared by all calls on a stream) and normalise their own columns.  s_out (bf16 [M, N]), mean, rstd (fp32 [M]): what
 * zk_add_ln_bwd reads, may be null.  `epoch`: device word advanced by zk_ln_epoch_bump once per forward pass (start it at
 * 0 and bump before the first use); `site` in 1..255 differs between the calls of one pass.  *err (device int, may be
 * null) is set to 1 if a workgroup gave up waiting for a peer. */
size_t zk_gemm_add_ln_workspace(int rows, int N);
int zk_ln_epoch_bump(uint32_t* epoch, zk_stream_t stream);
int zk_gemm_add_ln(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias,
                   const void* residual, int ldr, float drop_p, const uint64_t* seed, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, const uint32_t* epoch, uint32_t site, int* err, zk_stream_t stream);
/* The backward of that tail inside the dgrad launch that completes its input gradient: dout = bf16(dY W^T + residual) is
 * never stored; dsum = d(loss)/d(s) (bf16 [M, N]), dy_out = dsum x the dropout mask of (seed, sid) (null without dropout),
 * partials (zk_gemm_ln_bwd_partials bytes: [ceil(M/64)][3][N]) = the per-row-block column sums {dout xhat, dout, dy} that
 * zk_add_ln_bwd_reduce / zk_reduce_grouped finish with nblk = ceil(M/64).  dY [M, lda], W [N, ldb] (both K contiguous).
 * Replaces zk_gemm(tb = 1, residual) + zk_add_ln_bwd(defer_reduce = 1); slots / epoch / site / err as for zk_gemm_add_ln. */
/* Attention forward + output projection + residual + LayerNorm of a sub-layer in ONE launch: workgroup (sentence, head)
 * computes its attention output (func.py:218-256, as zk_attn_fwd's MFMA kernel: att and lse are still written), waits for
 * the other heads of its sentence and runs its 64-column tile of zk_gemm_add_ln over the sentence's rows.  d = 64,
 * Lq <= 64, Lk <= 256, no relative positions; W_o [nh*64, nh*64]; flags: zk_attn_out_ln_flags bytes, zero-filled once;
 * slots / epoch / site / err as for zk_gemm_add_ln.  Returns 2 (nothing launched) when the shape is not covered. */
size_t zk_attn_out_ln_flags(int B, int nh);
int zk_attn_out_ln(const void* q, const void* k, const void* v, void* att, float* lse, int B, int nh, int Lq, int Lk, int d,
                   int ldq, int ldk, int ldv, int ldatt, const float* kmask, int causal, float scale, float mask_inf,
                   float attn_drop_p, const uint64_t* seed, uint32_t attn_sid, int kv_group, const void* Wo, int ldw,
                   const float* bias, const void* residual, int ldr, float drop_p, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, void* flags, size_t flags_bytes, const uint32_t* epoch, uint32_t site, int* err,
                   zk_stream_t stream);
/* The same launch with the projection in FRONT of the attention inside it as well (func.py:206-216): workgroup (sentence,
 * head) first computes its own head's 64 columns of q (k, v) = x Wp + bp over its sentence's rows -- it is their only
 * consumer in the forward pass, so no exchange between workgroups is needed -- writes them (the backward reads them) and
 * goes on as zk_attn_out_ln.  pro = 3: Wp [Kp, 3 nh 64] = the merged qkv_map of a self-attention (k = q + nh*64 and
 * v = q + 2 nh*64 columns of one [B*Lq, ldq] matrix, Lk = Lq, kv_group = 1); pro = 1: Wp [Kp, nh 64] = the q_map of a
 * cross-attention (k, v given as before).  x [B*Lq, ldx] bf16, Kp a multiple of 64.  Replaces zk_gemm + zk_attn_out_ln with
 * bit-identical results (same tile function and K order).  Returns 2 (nothing launched) when the shape is not covered. */
int zk_proj_attn_out_ln(const void* x, int ldx, const void* Wp, int ldwp, const float* bp, int Kp, int pro,
                   const void* q, const void* k, const void* v, void* att, float* lse, int B, int nh, int Lq, int Lk, int d,
                   int ldq, int ldk, int ldv, int ldatt, const float* kmask, int causal, float scale, float mask_inf,
                   float attn_drop_p, const uint64_t* seed, uint32_t attn_sid, int kv_group, const void* Wo, int ldw,
                   const float* bias, const void* residual, int ldr, float drop_p, uint32_t sid, const float* gamma,
                   const float* beta, float eps, void* s_out, void* y, float* mean, float* rstd, void* slots,
                   size_t slots_bytes, void* flags, size_t flags_bytes, const uint32_t* epoch, uint32_t site, int* err,
                   zk_stream_t stream);
#ifdef ZK_EXPERIMENTS   /* measured, no gain over the two launches (profiles/r04_negative_results.txt item 8) */
/* Attention backward (single-tile path of zk_attn_bwd with the o_map dgrad folded in: d = 64, Lq, Lk <= 64, no relative
 * positions) + the dgrad dx = dA W^T + residual that consumes its dq / dk / dv (dA [B*Lq, K]: the matrix they are columns of)
 * + the LayerNorm backward of the sub-layer below (as zk_gemm_ln_bwd), one launch: workgroup (sentence, head) computes its
 * head's gradients, waits for the sentence's heads, runs its 64-column dgrad tile.  partials: [B][3][nh*64] (one row per
 * sentence).  flags: zk_attn_out_ln_flags bytes, zero-filled once.  Returns 2 (nothing launched) when not covered. */
int zk_attn_bwd_ln(const void* q, const void* k, const void* v, const void* out, const float* lse, void* dq, void* dk, void* dv,
                   int B, int nh, int Lq, int Lk, int d, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                   const float* kmask, int causal, float scale, float mask_inf, float attn_drop_p, const uint64_t* seed,
                   uint32_t attn_sid, const void* oproj_dy, int oproj_lddy, const void* oproj_w, int oproj_ldw, int oproj_n,
                   const void* dA, int lda, const void* W, int ldw, int K, const void* residual, int ldr, const void* s,
                   const float* mean, const float* rstd, const float* gamma, float drop_p, uint32_t sid, void* dsum,
                   void* dy_out, float* partials, void* slots, size_t slots_bytes, void* flags, size_t flags_bytes,
                   const uint32_t* epoch, uint32_t site, int* err, zk_stream_t stream);
#endif
size_t zk_gemm_ln_bwd_partials(int rows, int N);
int zk_gemm_ln_bwd(const void* dY, const void* W, int M, int N, int K, int lda, int ldb, const void* residual, int ldr,
                   const void* s, const float* mean, const float* rstd, const float* gamma, float drop_p,
                   const uint64_t* seed, uint32_t sid, void* dsum, void* dy_out, float* partials, void* slots,
                   size_t slots_bytes, const uint32_t* epoch, uint32_t site, int* err, zk_stream_t stream);
#ifdef ZK_EXPERIMENTS   /* measured: slower than the two launches (profiles/r04_negative_results.txt item 9) */
/* The two products of a feed-forward sub-layer on few rows (the decode step: func.py:327-338 at batch x beam rows) in one
 * launch: h = relu(x W1 + b1) (bf16 [M, F]), then -- behind a barrier among the launch's workgroups -- parts[z] = h[:, K_z]
 * W2[K_z, :] as zk_gemm_parts leaves them (fp32 [M, H] at parts + z M H; *nparts_out parts).  counter: a zeroed device
 * uint64 shared only by calls with the same (ceil(M/64), F).  Returns 2 (nothing launched) when the shape is not covered. */
int zk_ffn_pair(const void* x, const void* W1, const float* b1, void* h, const void* W2, float* parts, int M, int F, int H, int K1,
                int ldx, int ldw1, int ldw2, int splits, int* nparts_out, void* counter, int* err, zk_stream_t stream);
#endif
int zk_gemm_plan(int M, int N, int K, int out_f32, int plain);  /* gen | (bm/8)<<8 | (bn/8)<<16 | splits<<24 | producer waves<<28 chosen by impl=0 */
int zk_gemm_set_generation(int gen);   /* 1 = register-staged kernel, 2 = LDS-DMA ring kernel (default) */
/* K-segmented GEMM: C bf16 [M, ldc] = sum_s A_s [M, kseg] x B_s (+ bf16 residual, may alias C) in ONE launch --
   gradient contributions that the reference accumulates with add_n over the users of a tensor (the encoder
   output feeds the cross-attention K / V projections of every decoder layer, transformer.py:120-160).
   a_segs / b_segs: host arrays of nseg (<= 16) device pointers, all segments share lda / ldb;
   tb = 1: B_s is [N, ldb] with K contiguous, tb = 0: [kseg, ldb].  kseg % 64 == 0. */
int zk_gemm_kseg(const void* const* a_segs, const void* const* b_segs, int nseg, int kseg, void* C, int M, int N,
                 int lda, int ldb, int ldc, int tb, const void* residual, int ldr, zk_stream_t stream);
int zk_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
            int ta, int tb, int out_f32, float alpha, const float* bias, const void* residual, int ldr,
            int act, const void* aux, int ldaux, float aux_scale, float drop_p, const uint64_t* seed,
            uint32_t sid, int impl, void* workspace, size_t ws_bytes, zk_stream_t stream);

/* Grouped launch of independent GEMMs with the same ta/tb in one grid (the deferred weight
 * gradients of several layers; the cross-attention k_map / v_map projections of every decoder
 * layer, func.py:206-216).  descs: DEVICE array of nprob records
 *   { const void* A, B; void* C; const float* bias; const void* res /* bf16 residual or NULL,
 *     may alias C */; int M, N, K, lda, ldb, ldc, out_f32, tile_start, tiles_n, ldr;
 *     float* colsum /* NULL, or fp32 [N]: receives sum_k B[k][n] (tb = 0, tiles 5 / 6: the bias gradient beside a
 *     weight gradient, func.py:16,58-60) */; long pad; }   (96 bytes)
 * tile_start = running sum of ceil(M/T)*ceil(N/T) with T = 128 (tile=1) or 64 (tile=4). */
int zk_gemm_grouped(const void* descs, int nprob, int total_tiles, int ta, int tb, int tile,
                    zk_stream_t stream);

#ifdef ZK_EXPERIMENTS   /* measured slower than GEMM + zk_ce_fused (profiles/r02_fused_ce_256_tile.txt): make EXPERIMENTS=1 */
/* ---- transformer.py:182-216 + util.py:88-103 fused for training: logits = feat . E^T and the label-smoothed
 * cross entropy WITHOUT materialising the [T, V] logits.  fwd: ce fp32 [T] (may be NULL), lse fp32 [T]
 * (log-sum-exp of every row, kept for the backward); bwd: recomputes the logits tile by tile and writes
 * dlogits bf16 [T, ldd] = w_row * (softmax - soft labels) (columns >= V zero), the operand of the two
 * logits-gradient GEMMs.  feat bf16 [T, K] (ldf); E bf16 [>= V rows, K] (lde).  label smoothing as util.py:88-103
 * (p = 1 - eps on the gold id, q = eps / (V - 1) elsewhere, normaliser subtracted). */
size_t zk_logits_ce_workspace(int T, int V);
int zk_logits_ce_fwd(const void* feat, const void* E, const int* ids, float* ce, float* lse, int T, int V, int K,
                     int ldf, int lde, float label_smooth, void* workspace, size_t ws_bytes, zk_stream_t stream);
int zk_logits_ce_bwd(const void* feat, const void* E, const int* ids, const float* w, const float* lse,
                     void* dlogits, int T, int V, int K, int ldf, int lde, int ldd, float label_smooth,
                     zk_stream_t stream);
#endif /* ZK_EXPERIMENTS */

/* ---- func.py:218-256 dot_attention core (+ modules/rpr.py:10-75 relative positions).
 * q/k/v/out: [B*L, ld] bf16, head h at columns [h*d,(h+1)*d) (split/combine_heads,
 * func.py:68-104, folded into addressing).  kmask: fp32 [B,Lk] (1 valid / 0 pad) or NULL;
 * causal: mask keys j > q_pos0+i; masked logits get -mask_inf added (func.py:372-387,
 * finite 1e8).  lse: fp32 [B,nh,Lq] log-sum-exp saved for the backward (may be NULL).
 * rpr_k/rpr_v: bf16 [2*max_rel+1, d] tables or NULL.  bsq/bsk/bsv: batch strides in
 * elements (0 = L*ld); kv_group: k/v/kmask batch index = b / kv_group (decode: beam-tiled
 * queries attend to un-tiled per-sentence encoder keys, search.py:36-39 never materialised).
 * pos_dev (device int, may be NULL) carries the decode time step for hipGraph replay: pos_flags bit 0
 * -> q_pos0 = *pos_dev, bit 1 -> only keys 0..*pos_dev are valid (self-attention over the k/v cache,
 * func.py:199-205, with Lk = the allocated cache length).
 * Relative positions on the MFMA kernels (d = 64) are DECOMPOSED (modules/rpr.py:10-75): the caller
 * computes rpr_gq = Q_h . rpr_k^T (FP32 [B*Lq, rpr_ldg], entry (t, h, r) at t*rpr_ldg + h*rpr_nrp + r)
 * with a GEMM, the kernel gathers it into the scores and writes rpr_pb = sums of P over the keys of each
 * relative index (same layout, bf16); the caller finishes O += rpr_pb . rpr_v.  The backward takes
 * rpr_gq, rpr_gd = dO_h . rpr_v^T (fp32) and writes rpr_pb, rpr_dsb (bf16 bucket sums of P and dS): dQ += dsb . rpr_k,
 * d rpr_k = dsb^T Q, d rpr_v = pb^T dO are the caller's GEMMs.  Without rpr_gq the reference kernels
 * apply the tables directly (any head size).
 * zk_attn_fwd with impl | 256 and rpr_k / rpr_v but no rpr_gq: the relative-position terms are FOLDED into the MFMA tile
 * (d = 64, 2*max_rel+1 <= 64): both tables are staged in LDS, G = Q_h . rpr_k^T is one MFMA pass, the scores gather from it,
 * the bucket sums of P stay in LDS and O += PB . rpr_v follows the P.V product -- no table products through HBM, no extra
 * launches, single-query decode steps included; shapes it does not cover fall back to the reference kernel (impl 0). */
int zk_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int nh, int Lq,
                int Lk, int d, int ldq, int ldk, int ldv, int ldo, const float* kmask, int causal,
                int q_pos0, float scale, float mask_inf, const void* rpr_k, const void* rpr_v, int max_rel,
                float drop_p, const uint64_t* seed, uint32_t sid, long bsq, long bsk, long bsv, int kv_group,
                int impl, const int* pos_dev, int pos_flags, const void* rpr_gq, void* rpr_pb, int rpr_ldg,
                int rpr_nrp, zk_stream_t stream);
size_t zk_attn_bwd_workspace(int B, int nh, int Lq);
/* zk_attn_bwd with impl | 256, rpr_k / rpr_v / drpr_k / drpr_v but no decomposed products, Lq, Lk <= 64, d = 64 and a
 * workspace of zk_attn_bwd_rpr_workspace bytes: relative positions FOLDED into the single-tile backward kernel (tables,
 * G = Q.Rk^T, Gd = dO.Rv^T, bucket sums of dS / P and dQ += dsb.Rk in LDS; per-(sentence, head) table-gradient partials
 * in the workspace, summed into drpr_k / drpr_v by one reduction launch).
 * Table-gradient contract: WITH impl | 256 the call OVERWRITES drpr_k / drpr_v on every path (when the folded kernel
 * declines a shape -- alignment, ld % 8, Lq / Lk > 64, workspace -- the reference kernels run on tables the entry
 * point has cleared itself); WITHOUT the bit the reference kernels ADD to drpr_k / drpr_v with atomics and the
 * caller clears them beforehand.
 * impl | 512 (together with | 256): when the folded kernel runs, its per-(sentence, head) partials are LEFT in the
 * workspace -- fp32 [B*nh][2][64][64] behind the B*nh*Lq floats of D, (2*max_rel+1)*64 leading elements of each slab
 * valid -- and the call returns 1 instead of 0: the caller sums them (all attention layers of a step in one
 * zk_reduce_grouped launch).  When it does not run the bit is ignored and 0 is returned.
 * impl | 1024 (together with | 256): the folded kernel in its first form -- every tile of both phases resident, 151 KB of
 * LDS, one workgroup per CU -- instead of the 72-KB form (two workgroups per CU, identical results); for A/B runs and tests.
 * (oproj_dy, oproj_w) non-null: the gradient of the attention output is not read from `dout` (which may be null) but
 * computed inside the single-tile kernel as dY . W_o[h*64 .. h*64+63, :]^T -- dY bf16 [B*Lq, oproj_lddy], W_o = the
 * o_map weight [nh*64, oproj_ldw] row-major, oproj_n columns (a multiple of 128) -- i.e. the dgrad GEMM of the output
 * projection (func.py:226-240 `o = linear(o, ..., scope="o_map")` differentiated) folded into the attention backward:
 * one launch and one [B*Lq, nh*64] matrix in HBM less.  Only the single-tile kernel does it (Lq, Lk <= 64, d = 64, no
 * decomposed rpr products): when the call would run any other kernel NOTHING is launched and 2 is returned, and the caller
 * forms `dout` itself and calls again without the pair. */
size_t zk_attn_bwd_rpr_workspace(int B, int nh, int Lq);
int zk_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                const float* lse, void* dq, void* dk, void* dv, float* drpr_k, float* drpr_v, int B, int nh,
                int Lq, int Lk, int d, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk,
                int lddv, const float* kmask, int causal, int q_pos0, float scale, float mask_inf,
                const void* rpr_k, const void* rpr_v, int max_rel, float drop_p, const uint64_t* seed,
                uint32_t sid, int impl, void* workspace, size_t ws_bytes, const void* rpr_gq,
                const void* rpr_gd, void* rpr_pb, void* rpr_dsb, int rpr_ldg, int rpr_nrp, const void* oproj_dy,
                int oproj_lddy, const void* oproj_w, int oproj_ldw, int oproj_n, zk_stream_t stream);
/* out[i] (+)= sum over s of in[s*stride + i], i < n  (fp32; head-wise partial table gradients of the
   decomposed rpr path) */
int zk_sum_slices(float* out, const float* in, int nslices, size_t n, size_t stride, int accumulate,
                  zk_stream_t stream);

/* ---- transformer.py:16-33 / 88-119 embedding * sqrt(H) + shared bias + timing signal
 * (func.py:341-369; `timing` = host-precomputed fp32 [Lmax,H] table).  shift=1: decoder
 * training input (zero first step, transformer.py:108-110).  zero_flag: device int, !=0
 * zeroes the embedding term (transformer.py:113-115).  pos0_dev (device int, may be NULL)
 * overrides pos0 at run time, so a captured decode-step graph sees the current time step.  */
int zk_embed_fwd(const int* ids, const void* table, const float* bias, const float* timing, void* out, int B,
                 int L, int H, float scale, int shift, int pos0, const int* zero_flag, float drop_p,
                 const uint64_t* seed, uint32_t sid, const int* pos0_dev, zk_stream_t stream);
/* round 6: the encoder's and the decoder's input embeddings of a training step in ONE launch (side a: ids [B, La] as they
 * are; side b: ids [B, Lb] shifted right by one position, transformer.py:104-108); per side the arithmetic of zk_embed_fwd.
 * ln_epoch (may be NULL): the epoch word of the in-launch LayerNorm exchanges, advanced as zk_ln_epoch_bump advances it */
int zk_embed_fwd_pair(const int* ids_a, const void* table_a, void* out_a, int La, uint32_t sid_a, const int* ids_b,
                      const void* table_b, void* out_b, int Lb, uint32_t sid_b, const float* bias, const float* timing, int B,
                      int H, float scale, float drop_p, const uint64_t* seed, uint32_t* ln_epoch, zk_stream_t stream);
/* round 6: more small launches of the training step merged pairwise.  zk_embed_bwd_sorted_pair: the gradient scatters of
 * two DIFFERENT embedding tables (arguments per side as zk_embed_bwd_sorted) in one launch.  zk_colsum_pair: out = colsum(a)
 * + colsum(b) (the shared input bias of transformer.py:16-33 / 88-119; rows r % skip == 0 of a side left out when skip > 0)
 * in two launches instead of four; workspace >= zk_colsum_workspace(rows_a, N) + zk_colsum_workspace(rows_b, N). */
int zk_embed_bwd_sorted_pair(const int* rows_a, const int* seg_a, const int* uid_a, const int* n_a, int max_a, const void* dout_a,
                             float* dtable_a, int acc_a, uint32_t sid_a, const int* rows_b, const int* seg_b, const int* uid_b,
                             const int* n_b, int max_b, const void* dout_b, float* dtable_b, int acc_b, uint32_t sid_b, int H,
                             float scale, float drop_p, const uint64_t* seed, zk_stream_t stream);
int zk_colsum_pair(const void* a, int rows_a, int lda, int skip_a, uint32_t sid_a, const void* b, int rows_b, int ldb,
                   int skip_b, uint32_t sid_b, int N, float* out, float drop_p, const uint64_t* seed, void* workspace,
                   size_t ws_bytes, zk_stream_t stream);
/* dtable (fp32 [V,H]) and dbias (fp32 [H]) are ACCUMULATED into with atomics. */
int zk_embed_bwd(const int* ids, const void* dout, float* dtable, float* dbias, int B, int L, int H,
                 float scale, int shift, float drop_p, const uint64_t* seed, uint32_t sid, zk_stream_t stream);

/* Same gradient without atomics: the host (which owns the ids) sorts token rows by id; one wave
 * per distinct id sums its rows.  rows_sorted [n_used], seg [n_uniq+1], uid [n_uniq] int32 device
 * arrays, n_uniq_dev device int (<= max_uniq).  accumulate=0 overwrites the touched rows.      */
int zk_embed_bwd_sorted(const int* rows_sorted, const int* seg, const int* uid, const int* n_uniq_dev,
                        int max_uniq, const void* dout, float* dtable, int H, float scale, int accumulate,
                        float drop_p, const uint64_t* seed, uint32_t sid, zk_stream_t stream);

#ifdef ZK_EXPERIMENTS   /* measured: no gain over the LayerNorm launches (profiles/r04_negative_results.txt): make EXPERIMENTS=1 */
/* ---- round 4: residual + LayerNorm WITHOUT a launch of its own (func.py:289-303, 321-324 in the post-LN order of
 * transformer.py:57-58; the forward half of the 30 LayerNorm launches of a Transformer-base step).
 *   zk_gemm_ln   forward linear (func.py:14-65; A [M,K] x B [K,N], bf16 out) with the LayerNorm around it folded into
 *                the epilogue.  PRODUCER (stat_out != NULL): C = residual + dropout(A B + bias), the sub-layer's
 *                un-normalised sum, and stat_out [M][N/64][2] = {sum, M2} of every (row, 64-column group) of the
 *                stored values.  res_part != NULL: the residual operand is the previous sub-layer's un-normalised sum
 *                and is normalised on the fly (res_part [M][np][2], res_gamma / res_beta [N]).  CONSUMER
 *                (in_c != NULL): A is an un-normalised sum (statistics in_part [M][np][2], np = K/64), B the weight
 *                with gamma folded in, in_c / bias the vectors of zk_ln_fold:
 *                C = act(rstd (A B - mu in_c) + bias), then dropout.  act: 0 none, 1 ReLU.
 *   zk_ln_fold   per step, from the fp32 masters: Wf = bf16(gamma_k W_kn), c_n = sum_k Wf_kn, d_n = sum_k beta_k W_kn
 *                + b_n for every consumer weight; descs = DEVICE array of {W, gamma, beta, b, Wf, c, d pointers; int K, N,
 *                block_start (running sum of N/64), pad} (64 bytes), total_blocks = that sum.
 *   zk_add_ln_bwd_lazy   the backward of such a LayerNorm: statistics from `part`; y_out (optional) receives
 *                LN(sum) as zk_add_ln_fwd would have written it -- the X operand of the consumer's weight gradient. */
int zk_gemm_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const float* bias,
               const void* residual, int ldr, int act, float drop_p, const uint64_t* seed, uint32_t sid, float* stat_out,
               const float* in_part, const float* in_c, const float* res_part, const float* res_gamma,
               const float* res_beta, int np, float eps, zk_stream_t stream);
int zk_ln_fold(const void* descs, int nprob, int total_blocks, zk_stream_t stream);
int zk_add_ln_bwd_lazy(const void* dout, const void* sum, const float* part, const float* gamma, const float* beta,
                       void* y_out, void* dsum, void* dy, float* dgamma, float* dbeta, float* dbias_prev, int rows, int H,
                       float eps, float drop_p, const uint64_t* seed, uint32_t sid, void* workspace, size_t ws_bytes,
                       int defer_reduce, zk_stream_t stream);
#endif /* ZK_EXPERIMENTS */

/* ---- func.py:321-324 residual_fn + func.py:289-303 layer_norm (post-LN, eps inside
 * rsqrt): out = LN(x + dropout(y)).  sum_out/mean/rstd are saved for the backward.    */
int zk_add_ln_fwd(const void* x, const void* y, const float* gamma, const float* beta, void* out,
                  void* sum_out, float* mean, float* rstd, int rows, int H, float eps, float drop_p,
                  const uint64_t* seed, uint32_t sid, zk_stream_t stream);
size_t zk_add_ln_bwd_workspace(int rows, int H);
/* dsum = d(x + drop(y)); dy = dsum*dropmask (only written when drop_p>0); dgamma/dbeta and
 * dbias_prev (= column sum of dy: bias grad of the linear that produced y) are OVERWRITTEN. */
int zk_add_ln_bwd(const void* dout, const void* sum, const float* mean, const float* rstd, const float* gamma,
                  void* dsum, void* dy, float* dgamma, float* dbeta, float* dbias_prev, int rows, int H,
                  float drop_p, const uint64_t* seed, uint32_t sid, void* workspace, size_t ws_bytes,
                  int defer_reduce, zk_stream_t stream);
/* defer_reduce=1: only the per-block partial sums are written to `workspace` (which must then be
 * private to this call); finish them later, possibly on another stream, with: */
int zk_add_ln_bwd_reduce(const void* workspace, int rows, int H, float* dgamma, float* dbeta, float* dbias_prev,
                         zk_stream_t stream);

/* column sum of a bf16 [rows,N] matrix -> fp32 [N] (bias gradients, func.py:58-60) */
size_t zk_colsum_workspace(int rows, int N);
int zk_colsum(const void* a, int rows, int N, int lda, float* out, void* workspace, size_t ws_bytes,
              zk_stream_t stream);
/* skip_L>0: skip rows r with r%skip_L==0 (shifted decoder input); accumulate: out += ;
 * dropout mask index = r*N + c (the shared embedding bias gradient, transformer.py:27,102) */
int zk_colsum_ex(const void* a, int rows, int N, int lda, float* out, int skip_L, int accumulate, float drop_p,
                 const uint64_t* seed, uint32_t sid, void* workspace, size_t ws_bytes, zk_stream_t stream);

/* Grouped column reductions (all bias / LayerNorm-parameter gradients of a layer group in two
 * launches).  DEVICE descriptor arrays:
 *   colsum : { const void* a; float* partials; int rows, N, lda, gy, block_start, pad; }   (40 B)
 *            gy = zk_colsum_rowchunks(rows); blocks of problem = ceil(N/64)*gy
 *   reduce : { const float* partials; float* out[3]; int nblk, nq, H, block_start; }       (48 B)
 *            blocks of problem = nq*ceil(H/16); nblk = gy (colsum) or zk_ln_bwd_blocks(rows) */
int zk_colsum_grouped(const void* descs, int nprob, int total_blocks, zk_stream_t stream);
int zk_reduce_grouped(const void* descs, int nprob, int total_blocks, zk_stream_t stream);
int zk_ln_bwd_blocks(int rows);
int zk_colsum_rowchunks(int rows);

/* ---- util.py:88-103 label_smooth + transformer.py:198-207 cross entropy on fp32 logits.
 * ce_out[r] = -sum soft*log_softmax - normaliser; dlogits (bf16 [rows,ld], NULL to skip)
 * = w[r]*(softmax - soft).                                                            */
int zk_ce_fused(const float* logits, const int* ids, const float* w, float* ce_out, void* dlogits, int rows,
                int V, int ld, float label_smooth, zk_stream_t stream);
/* transformer.py:208-216: mask=(id!=0); w = loss_scale*mask/(len_b*B); per-sentence loss and mean */
int zk_target_stats(const int* ids, float* mask, float* w, int B, int L, float loss_scale, zk_stream_t stream);
int zk_loss_reduce(const float* ce, const int* ids, float* per_sample, float* loss, int B, int L,
                   zk_stream_t stream);
int zk_make_mask(const int* ids, float* mask, int n, zk_stream_t stream);
/* Everything the training step derives from the ids alone, in ONE launch at the head of the (captured) step
 * (zk_prep.hip): what TensorFlow does on the device for the reference -- the grouping of the token rows by id inside the
 * gradient of tf.nn.embedding_lookup (tf.IndexedSlices / unsorted_segment_sum, main.py:28), the padding masks
 * (func.py:372-387) and the loss weights (transformer.py:198-211).
 *   src_ids int32 [B, Ls], tgt_ids int32 [B, Lt] (NULL: source side only);
 *   *_rows [B*L], *_seg [B*L + 1], *_uid [B*L], *_n [1]: the inputs of zk_embed_bwd_sorted -- token rows sorted by
 *   (id, row), i.e. the stable grouping; target side: row (b, t) carries id[b, t-1], rows with t = 0 none
 *   (transformer.py:99-113).  A NULL *_rows pointer skips that side's sort;
 *   smask [B, Ls], tmask / tw [B, Lt] fp32 (each optional): as zk_make_mask / zk_target_stats;
 *   max_id: every id is below it (the larger vocabulary; 0 = unknown): lets the sort run on 32-bit keys (id << bits(rows)
 *   | row) when they fit, ~3x faster than the 64-bit network;
 *   scratch: zk_batch_prep_workspace(B*Ls) + zk_batch_prep_workspace(B*Lt) bytes (0 up to 16384 rows per side). */
size_t zk_batch_prep_workspace(int rows);
int zk_batch_prep(const int* src_ids, const int* tgt_ids, int B, int Ls, int Lt, int* src_rows, int* src_seg,
                  int* src_uid, int* src_n, int* tgt_rows, int* tgt_seg, int* tgt_uid, int* tgt_n, float* smask,
                  float* tmask, float* tw, float loss_scale, int max_id, void* scratch, size_t scratch_bytes,
                  zk_stream_t stream);
/* Up to 16 small device-to-device copies in one launch (host arrays of device pointers / byte counts, 4-byte granularity):
 * the id-dependent arrays of the next batch move from the staging buffers a side stream filled (upload + zk_batch_prep,
 * overlapping the previous step) into the static buffers of the captured step. */
int zk_copy_many(void* const* dsts, const void* const* srcs, const size_t* nbytes, int n, zk_stream_t stream);
int zk_all_equal(const int* ids, int n, int value, int* flag, zk_stream_t stream);

/* ---- transformer_aan.py:92-117,165-192 average attention network (train-time scan + gate) */
int zk_aan_fwd(const void* x, const float* mask, void* cat, int B, int L, int H, int use_mask,
               zk_stream_t stream);
/* use_mask bit 0: aan_mask (run.py:117); bit 1 (backward only): dcat[:, H:] is already folded into dyg
   (the use_ffn variant, transformer_aan.py:176-183) */
int zk_add_bf16(void* out, int ldo, const void* a, int lda, const void* b, int ldb, int rows, int cols,
                zk_stream_t stream);
int zk_aan_bwd(const void* dcat, const void* dxg, const void* dyg, const void* ds, const float* mask, void* dx,
               int B, int L, int H, int use_mask, zk_stream_t stream);
int zk_aan_gate_fwd(const void* z, const void* cat, void* out, int rows, int H, zk_stream_t stream);
int zk_aan_gate_bwd(const void* dg, const void* z, const void* cat, void* dz, void* dxg, void* dyg, int rows,
                    int H, zk_stream_t stream);

/* ---- utils/cycle.py:86-101 + tf.train.AdamOptimizer (main.py:178-181) on flat buffers.
 * hyper (device fp32[12]): lr_t, beta1, beta2, eps, grad_scale, clip_norm(0=off), gnorm(in), skipped(out),
 * EMA decay (zk_ema), gnorm upper bound of safe_nan (main.py:325-329; 0 = off), [10] STICKY count of updates that
 * were skipped or saw a non-finite gradient norm (never cleared by the kernels: the loop reads it before every
 * display / checkpoint, main.py:316-319), 1 reserved.
 * zk_l2norm: out[0] = scale*||x||_2; zk_adam's pnorm_out = ||p|| before the update.  The update (and the EMA)
 * is skipped, and hyper[7] set, when gnorm is not finite or exceeds the bound.
 * zk_adam_step: the whole update of a step.  norm_free = 1 (cycle.py:98-101: clip_grad_norm 0.0, and no safe_nan --
 * the update does not depend on the global norm, which is only reported): gradient norm -> hyper[6] (+ flags),
 * TF1 Adam, bf16 shadow refresh and parameter norm in ONE pass over the flat buffers (the reference fetches
 * train_op and gradient_norm together, main.py:309-312); norm_free = 0: hyper[6] must hold the norm (zk_l2norm).
 * seed (device uint64, may be NULL): the dropout step seed, advanced by one in the same launch.
 * skip_word (device int, may be NULL; round 5): non-zero = a launch of this step reported a fault of its own (the error word
 * of the in-launch LayerNorm exchange, zk_gemm_add_ln's `err`): the update is NOT applied, hyper[6] = NaN, hyper[7] = 1 and
 * the sticky hyper[10] is incremented -- the failure-detection rule of main.py:316-332 applied on the device, on the step
 * that failed.  The word itself is left as it is (the host reads and reports it).
 * zk_norm_flag: fold a norm that zk_l2norm wrote to hyper[6] AFTER per-bucket updates into hyper[7] / hyper[10]. */
size_t zk_norm_workspace(void);
int zk_l2norm(const float* x, size_t n, float scale, float* out, void* workspace, size_t ws_bytes,
              zk_stream_t stream);
int zk_adam(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float* hyper,
            float* pnorm_out, void* workspace, size_t ws_bytes, zk_stream_t stream);
size_t zk_adam_step_workspace(void);
int zk_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float* hyper,
                 float* pnorm_out, uint64_t* seed, int norm_free, const int* skip_word, void* workspace, size_t ws_bytes,
                 zk_stream_t stream);
#ifdef ZK_EXPERIMENTS   /* measured slower than gradient launch + Adam pass (profiles/r04_negative_results.txt): make EXPERIMENTS=1 */
/* ---- round 4: the optimiser update of the weight matrices INSIDE the launch that makes their gradients
 * (utils/cycle.py:94-101 norm-free form + main.py:178-181 TF1 Adam, fused into the autodiff mirror of func.py:14-65).
 *   zk_gemm_grouped_update   the grouped weight-gradient launch of 256 x 256 tiles (as zk_gemm_grouped with ta = 1,
 *       tb = 0, tile 8 | 256: bias column sums ride along); a descriptor with pad & 1 marks a problem whose output C is a
 *       whole variable inside the flat gradient buffer `grad_base`: its tiles do not store the gradient but run Adam on
 *       the accumulators against master / m / v / shadow at the same offset (hyper as zk_adam_step) and leave the wave's
 *       {sum g^2, sum theta^2} in sq [total_tiles][8][2] (zeros from the other tiles).
 *   zk_adam_step_segments    the norm-free update of everything else: segments [seg_lo[s], +len) of the flat buffers
 *       (DEVICE int64 arrays, elements, multiples of 4, ascending; prefix = running lengths, prefix[0] = 0, total =
 *       prefix[nseg]), then gradient / parameter norms over both parts (extra = sq above, n_extra = total_tiles * 8). */
int zk_gemm_grouped_update(const void* descs, int nprob, int total_tiles, float* master, float* m, float* v, void* shadow,
                           const float* grad_base, const float* hyper, float* sq, zk_stream_t stream);
int zk_adam_step_segments(float* p, const float* g, float* m, float* v, void* shadow, const long* seg_lo, const long* prefix,
                          int nseg, long total, float* hyper, float* pnorm_out, uint64_t* seed, const float* extra,
                          int n_extra, void* workspace, size_t ws_bytes, zk_stream_t stream);
#endif /* ZK_EXPERIMENTS */
#ifdef ZK_EXPERIMENTS   /* Adam beside the encoder backward: 5.01 vs 4.94 ms (DESIGN 6b): make EXPERIMENTS=1 */
/* the norm-free update in pieces: TF1 Adam on n elements writing its partial sums of squares into workspace slot
 * `slot` (< 16); zk_adam_finish sums nslots slots -> hyper[6] (+ flags), pnorm_out, seed += 1.  Lets the update of
 * the parameters whose gradients are final run beside the rest of the backward (and behind per-bucket all-reduces). */
size_t zk_adam_range_workspace(void);
int zk_adam_range(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float* hyper, int slot,
                  void* workspace, size_t ws_bytes, zk_stream_t stream);
int zk_adam_finish(float* hyper, float* pnorm_out, uint64_t* seed, int nslots, const void* workspace, size_t ws_bytes,
                   zk_stream_t stream);
#endif /* ZK_EXPERIMENTS */
int zk_norm_flag(float* hyper, zk_stream_t stream);
int zk_cast_f32_bf16(const float* x, void* y, size_t n, zk_stream_t stream);
int zk_cast_bf16_f32(const void* x, float* y, size_t n, zk_stream_t stream);
int zk_zero(void* p, size_t bytes, zk_stream_t stream);
/* Row-sparse exchange of an embedding-table gradient between data-parallel ranks (the reference moves it as
 * tf.IndexedSlices: values / indices concatenated across towers and de-duplicated, utils/parallel.py:142-181).
 * Payload of one rank = [ids int32 x R][rows x R x H] (fp32 or bf16 rows; unused slots carry id -1):
 *   zk_rows_pack        slot u < *n_uniq_dev: ids[u] = uid[u], rows[u] = dtable[uid[u]] (cast); clear_rows != 0 also
 *                       zeroes that table row, so that the table is rebuilt from the payloads of ALL ranks
 *   zk_rows_scatter_add dtable[ids[u]] += rows[u] for every valid slot of ONE payload (ids unique inside a payload: no
 *                       atomics; the payloads of the N ranks are added by N launches in rank order, so every rank ends
 *                       with a bit-identical table)
 * R, H multiples of 4. */
size_t zk_rows_payload_bytes(int R, int H, int bf16);
/* dst bf16 [cols, ld_dst] = transpose of src bf16 [rows, ld_src]: the operand layout of the fused decode kernels (a
 * projection weight with its input dimension contiguous), made once per weight version */
int zk_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, zk_stream_t stream);
int zk_rows_pack(float* dtable, const int* uid, const int* n_uniq_dev, void* out, int R, int H, int out_bf16,
                 int clear_rows, zk_stream_t stream);
int zk_rows_scatter_add(float* dtable, const void* payload, int R, int H, int in_bf16, int vocab_rows,
                        zk_stream_t stream);
/* host-only CRC32C of a byte range (seed crc = 0 for a fresh checksum): TensorFlow-bundle checkpoint
   tensors, utils/saver.py:75,131-170 */
uint32_t zk_crc32c(const void* data, size_t n, uint32_t crc);
int zk_spin(uint32_t usec, zk_stream_t stream);   /* measurement aid: occupy the stream for usec (<= 200 ms) */
int zk_tune(int key, int value);   /* A/B switches for measurements; key 0 = wide LayerNorm-backward kernel */
/* utils/cycle.py:113-119 (tf.train.ExponentialMovingAverage): ema -= (1 - hyper[8]) * (ema - p); skipped like
   the Adam update when hyper[6] (gradient norm) is not finite */
int zk_ema(float* ema, const float* p, const float* hyper, size_t n, zk_stream_t stream);
int zk_axpby_f32(float* y, const float* x, float a, float b, size_t n, zk_stream_t stream);

/* ---- utils/parallel.py:134-208 average_gradients -> RCCL sum collectives over xGMI on the caller's stream (the 1/N
 * of the tower mean is zk_adam_step's grad_scale).  The communicator is an opaque handle OWNED BY THE CALLER:
 * rank 0 draws a 128-byte id (zk_comm_unique_id) and ships it to the other ranks by its own means (here: the
 * torch.distributed store), every rank calls zk_comm_init (collective; the current HIP device is the rank's GPU) and
 * later zk_comm_destroy.  librccl is dlopen()ed on first use: zk_comm_available() = 0 on a box without it.
 * dtype: 0 fp32, 1 bf16 (all-gather also 2 = int32).  Errors: 1000 + ncclResult_t. */
int zk_comm_available(void);
int zk_comm_unique_id(void* id128);
int zk_comm_init(const void* id128, int nranks, int rank, void** comm_out);
int zk_comm_destroy(void* comm);
int zk_comm_size(const void* comm);
int zk_comm_allreduce(void* comm, void* buf, size_t count, int dtype, zk_stream_t stream);
/* n in-place all-reduces in ONE RCCL group; bufs / counts are HOST arrays */
int zk_comm_allreduce_multi(void* comm, void* const* bufs, const size_t* counts, int n, int dtype,
                            zk_stream_t stream);
/* recv[r*count .. (r+1)*count) <- send of rank r (row-sparse source-embedding gradient, parallel.py:142-181) */
int zk_comm_allgather(void* comm, const void* send, void* recv, size_t count, int dtype, zk_stream_t stream);
/* round 6: ordering a second stream behind the compute stream WITHOUT an event on the compute stream (an event recorded
 * there slows every dispatch of that stream: 40-65 us per training step, DESIGN.md 6e).  flag: an 8-byte aligned device
 * word that only ever grows.  zk_flag_add (on the producer's stream): flag += 1 with agent-scope release; zk_flag_wait (on
 * the consumer's stream): a one-thread kernel that ends once flag >= target (gives up after ~2 s and sets *err, may be
 * NULL).  zero_amd/utils/parallel.py RcclComm uses the pair per gradient bucket under ZERO_HIP_COMM_HANDOFF=flag. */
int zk_flag_add(void* flag, zk_stream_t stream);
int zk_flag_wait(const void* flag, unsigned long long target, int* err, zk_stream_t stream);

#ifdef ZK_EXPERIMENTS   /* measured 1.25x slower than launch-per-op (profiles/r02_layer_program_experiment.txt) */
/* ---- Layer program (zk_layer.hip): a run of dependent, sentence-local ops -- the linear / attention / residual +
 * LayerNorm chain of the encoder and decoder stacks (transformer.py:35-69, 121-181; func.py:194-338) -- executed by ONE
 * persistent launch instead of one launch per op.  The B sentences are dealt to the 8 XCDs; every XCD walks the op list
 * on its own sentences with a barrier among ITS workgroups between ops, activations staying in its L2.  The ops run
 * the same tile functions as the launch-per-op kernels: bit-identical results.
 * Recording: between zk_prog_begin(B) and zk_prog_end on one host thread, zk_gemm (untransposed A), zk_attn_fwd,
 * zk_attn_bwd (one 64x64 tile per sentence and head) and zk_add_ln_fwd append an op instead of launching; any other
 * variant makes the recording fail (zk_prog_end returns -2 and the caller issues ordinary launches).  Entry points
 * that are not listed here must not be called while recording.  A program is either a forward chain (gemm, attention
 * forward, residual + LayerNorm) or a backward chain (gemm, attention backward): zk_prog_end reports which in
 * *is_backward, to be passed to zk_prog_launch.  zk_prog_end copies the ops to a HOST buffer; the
 * caller uploads them and owns the device copy and the state buffer (zk_prog_state_bytes(), any content).
 * state after a launch (ints): [576] workgroups found on another XCD than their group's (handled: that group runs
 * placement-independent barriers), [577] != 0: a barrier timed out and the launch drained (results invalid). */
size_t zk_prog_op_bytes(void);
size_t zk_prog_state_bytes(void);
int zk_prog_begin(int sentences);
int zk_prog_end(void* ops_out, size_t cap_bytes, int* nops, int* is_backward);
int zk_prog_launch(const void* ops_dev, int nops, int sentences, int backward, void* state_dev, zk_stream_t stream);
#endif /* ZK_EXPERIMENTS */

/* dropout plumbing */
int zk_dropout_mask(float* out, size_t n, float drop_p, const uint64_t* seed, uint32_t sid, zk_stream_t stream);
int zk_seed_advance(uint64_t* seed, uint64_t inc, zk_stream_t stream);

/* ---- search.py:143-176 decode step tail (see zk_decode.hip) */
size_t zk_beam_topk_workspace(int B, int K, int k2);
int zk_beam_topk(const float* logits, const float* prev_log_probs, float* topk_scores, int* topk_index, int B,
                 int K, int V, int ld, int k2, float temperature, float length_penalty, int forbid_id,
                 float forbid_value, const int* scal_dev, void* workspace, size_t ws_bytes, zk_stream_t stream);
/* search.py:198-210 beam reordering: dst row r <- src row index[r] (NULL: r); bytes, multiples of 16 */
int zk_gather_rows(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                   size_t row_bytes, zk_stream_t stream);
/* period > 0: rows = n_tables * period stacked tables share one index of length period (the caches of every
   decoder layer reordered by one launch) */
int zk_gather_rows_ex(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                      size_t row_bytes, int period, zk_stream_t stream);
/* transformer_fuse (func.py:258-275) merged attention: the averaged v_map(query) term summed into the
   cross-attention heads.  train: out = att + cumavg_mask(vq) and its transpose; decode: cache += vq,
   att += cache/(t+1) (func.py:262-272) */
int zk_cumavg_add_fwd(const void* vq, const float* mask, const void* att, void* out, int B, int L, int H,
                      zk_stream_t stream);
int zk_cumavg_bwd(const void* dy, const float* mask, void* dvq, int B, int L, int H, zk_stream_t stream);
int zk_fuse_decode(const void* vq, float* cache, void* att, int rows, int H, float inv_count, const int* time_dev,
                   zk_stream_t stream);
/* Host side of a beam-search step in C (pure host code, no device work): the stop test of search.py:85-113
   and the alive / finished bookkeeping of search.py:168-228 on int32 [B, K, Tcap] sequence buffers. */
int zk_beam_host_should_stop(int B, int K, const float* log_probs, const float* fin_scores,
                             const unsigned char* fin_flags, const float* max_target_length, const int* mtl_i,
                             int time, float alpha);
int zk_beam_host_step(int B, int K, int V, int Tcap, int time, const float* topk_scores, const int* topk_idx,
                      int* seq, int* fin_seq, float* log_probs, float* scores, float* fin_scores,
                      unsigned char* fin_flags, const int* mtl_i, int eos_id, int pad_id, float penalty,
                      int* flat_idx, int* next_tok);
/* The same two statements with the search state RESIDENT ON THE DEVICE, as the first and the last node of
   the decode-step graph (no host round trip per step; replaces the per-step session.run boundary of
   search.py:60-275 / main.py:399-460).  ctrl int32[4]: [0] steps taken (= next time step), [1] stopped,
   [2] error (the step ran into the cache cap Tmax).  stepbuf int32[>=3]: [0] time, [1] length penalty of the
   step (fp32 bits), [2] banned symbol or -1 -- the per-step scalars the other kernels of the graph read.
   pen_table[t] = ((5 + t + 1) / 6)^alpha for t < Tcap and max_lp[b] = ((5 + max_target_length[b]) / 6)^alpha
   are computed by the caller in fp32.  prepare: stop test of the coming step; when it fires the state is
   frozen (advance becomes a no-op), otherwise it publishes the step scalars.  advance: alive / finished
   update from the step's [B, 2K] survivors; writes the next step's tokens, log-probs (prev) and flat beam
   indices.  seq / fin_seq: int32 [B, K, Tcap]; fin_flags: int32 [B, K].  K <= 16, 2*K*Tcap*4 <= 64 KiB. */
int zk_beam_dev_prepare(int* ctrl, int* stepbuf, const float* pen_table, const float* max_lp, const int* mtl_i,
                        const float* topk_scores, const int* topk_idx, int* seq, int* fin_seq, float* log_probs,
                        float* scores, float* fin_scores, int* fin_flags, int* flat_idx, int* next_tok,
                        float* prev, int B, int K, int V, int Tcap, int Tmax, int eos_id, int pad_id,
                        zk_stream_t stream);
int zk_beam_dev_advance(int* ctrl, int* stepbuf, const float* pen_table, const float* max_lp, const int* mtl_i,
                        const float* topk_scores, const int* topk_idx, int* seq, int* fin_seq, float* log_probs,
                        float* scores, float* fin_scores, int* fin_flags, int* flat_idx, int* next_tok,
                        float* prev, int B, int K, int V, int Tcap, int Tmax, int eos_id, int pad_id,
                        zk_stream_t stream);
/* zk_beam_topk on the step's logits (k2 = 2K, log-softmax, length penalty and EOS ban read from stepbuf[1..2]) followed by
 * zk_beam_dev_advance, with the merge of the chunked top-k and the bookkeeping in ONE launch (both are one block per
 * sentence; search.py:115-236).  Same results as the two calls.  Returns -2 with nothing launched when the shape needs
 * the unchunked top-k: call the two entry points then.  workspace: zk_beam_topk_workspace(B, K, 2K). */
int zk_beam_topk_advance(const float* logits, int ld, float temperature, float forbid_value, void* workspace,
                         size_t ws_bytes, int* ctrl, int* stepbuf, const float* pen_table, const float* max_lp,
                         const int* mtl_i, const float* topk_scores, const int* topk_idx, int* seq, int* fin_seq,
                         float* log_probs, float* scores, float* fin_scores, int* fin_flags, int* flat_idx, int* next_tok,
                         float* prev, int B, int K, int V, int Tcap, int Tmax, int eos_id, int pad_id, zk_stream_t stream);
/* search.py:143-145 (enable_noise_beam_search): logits += Gumbel noise -log(-log(u + eps) + eps), util.py:189-195 */
int zk_add_gumbel(float* logits, int rows, int V, int ld, float eps, const uint64_t* seed, uint32_t sid,
                  zk_stream_t stream);
/* k/v cache rows with the time step in device memory (hipGraph replay of func.py:199-205 and of the
   beam reorder search.py:206-209).  mode 0 (append): dst[r][*time_dev] <- src[r] (unit_bytes);
   mode 1 (reorder): dst[r][0 .. *time_dev) <- src[index[r]][0 .. *time_dev) in units of unit_bytes;
   max_units bounds the launch; period as in zk_gather_rows_ex. */
int zk_cache_rows(const void* src, size_t src_stride, const int* index, void* dst, size_t dst_stride, int rows,
                  size_t unit_bytes, int max_units, const int* time_dev, int mode, int period, zk_stream_t stream);
/* transformer_aan.py:110-112: cache += x; cat = [x | cache/(t+1)] */
/* Decode-step form of the residual + LayerNorm of the decoder with its row-local neighbours in the same launch
 * (transformer_aan.py:165-192, func.py:289-303): the sub-layer output y is ybuf (bf16 [rows, H]), or
 *   z/cat_in != NULL: the AAN gate sigma(z_i) x + sigma(z_f) y, computed first (into ybuf), or
 *   parts != NULL   : bf16(sum_p parts[p*part_stride + r*H + :] + bias) -- the output projection left as nparts fp32
 *                     partial products by zk_dec_cross / zk_dec_self (summed in the fixed order p = 0 .. nparts-1);
 * out = LayerNorm(x + y); cache/cat_out != NULL: the next layer's running sum and [x | average] follow (zk_aan_decode).
 * The row stays in registers from the loads to the output and is rounded to bf16 where the separate kernels stored bf16:
 * the first two forms equal zk_aan_gate_fwd + zk_add_ln_fwd + zk_aan_decode bit for bit.  ybuf is only read (first form). */
int zk_ln_decode(const void* x, void* ybuf, const float* gamma, const float* beta, void* out, int rows, int H, float eps,
                 const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                 float* cache, void* cat_out, float inv_count, const int* time_dev, zk_stream_t stream);
/* One attention sub-layer of a cached decode step in ONE launch (transformer.py:120-175 at Lq = 1; func.py:124-287):
 * grid B * nh, workgroup (sentence b, head h) owns the sentence's R <= 8 beam rows: [prologue: the previous sub-layer's
 * residual + LayerNorm, arguments x .. time_dev as zk_ln_decode; gamma == NULL: none, x is the block input; otherwise
 * xout receives the normalised rows] -> q_h = x Wq[:, h] + bq -> softmax(scale q_h K_h^T + mask) V_h -> the head's share
 * ctx_h Wo[h rows, :] of the output projection as fp32 out_parts[h][B*R][H].  The caller finishes with
 * zk_ln_decode(parts = out_parts, nparts = nh, part_stride = B*R*H, bias = o_map bias) or the next prologue.
 * The projection weights are passed TRANSPOSED (wqt = q_map^T [H, H], wot = o_map^T [H, H]: row = output channel, input
 * dimension contiguous, row strides ldwq / ldwo) so that a matrix-core fragment is one 16-byte load.
 * d = 64 per head, H = nh * 64 a power of two in 128 .. 2048.  Cross: keys / values of sentence b at k + b*bsk + j*ldk
 * (func.py:206-216 mk / mv), kmask [B, ldmask] (1 = valid) or NULL.  Values are rounded to bf16 where the launch-per-op
 * path stores bf16.
 * Relative positions (modules/rpr.py:10-75 with last = 1; round 4): rpr_k / rpr_v bf16 [2 max_rel + 1, 64] or NULL
 * (max_rel <= 31).  The query sits at position pos (*pos_dev when given; self-attention: its time step): key j adds
 * q . rpr_k[clip(pos - j) + max_rel] to its score and P_j rpr_v[clip(pos - j) + max_rel] to the context. */
int zk_dec_cross(const void* x, void* ybuf, const float* gamma, const float* beta, void* xout, int H, float eps,
                 const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                 float* cache, void* cat_out, float inv_count, const int* time_dev, const void* wqt, int ldwq,
                 const float* bq, const void* k, const void* v, int ldk, int ldv, long bsk, long bsv, const float* kmask,
                 int ldmask, const void* wot, int ldwo, float* out_parts, int B, int R, int nh, int Lk, float scale,
                 float mask_inf, const void* rpr_k, const void* rpr_v, int max_rel, int pos, const int* pos_dev,
                 zk_stream_t stream);
/* Beam rows per workgroup of zk_dec_cross / zk_dec_self (1 .. 16; 0 = built-in default); returns the previous setting.
 * A measurement knob (scripts/dec_attn_trace.py): results do not depend on it. */
int zk_dec_group(int n);
/* Self-attention over per-beam caches (func.py:199-205): wqkvt = qkv_map^T [3H, H] (q | k | v rows), bias [3H];
 * kcache / vcache bf16 [B*R, Tmax, H]: this step's key / value are written at slot time (*time_dev when given), keys
 * 0 .. time attended. */
int zk_dec_self(const void* x, void* ybuf, const float* gamma, const float* beta, void* xout, int H, float eps,
                const void* z, const void* cat_in, const float* parts, int nparts, long part_stride, const float* bias,
                float* cache, void* cat_out, float inv_count, const int* ln_time_dev, const void* wqkvt, int ldw,
                const float* bqkv, void* kcache, void* vcache, int Tmax, int time, const int* time_dev, const void* wot,
                int ldwo, float* out_parts, int B, int R, int nh, float scale, const void* rpr_k, const void* rpr_v,
                int max_rel, zk_stream_t stream);
int zk_aan_decode(const void* x, float* cache, void* cat, int rows, int H, float inv_count, const int* time_dev,
                  zk_stream_t stream);
/* dynamic LDS bytes a zk_dec_cross / zk_dec_self workgroup needs for hidden size H and Lk keys (a CU has 160 KiB):
 * lets the caller choose the launch-per-op path for shapes that do not fit, before the batch starts decoding.
 * max_rel < 0: without relative-position tables. */
size_t zk_dec_attn_lds(int H, int Lk, int max_rel);
/* Decoder input of one decode position in one launch (transformer.py:88-119; was zk_all_equal + zk_embed_fwd +
 * zk_aan_decode): every fed id == pad_id (first step) -> zero embedding, else table[id] * scale + bias; + timing[pos];
 * cache / cat != NULL: the first layer's average-attention update (transformer_aan.py:110-112).  *pos_dev overrides
 * pos0 and inv_count (= 1 / (pos + 1)).
 * gather_src != NULL: the beam reorder of the running sums of all nl layers rides along (search.py:206-209):
 * cache [nl, rows, H] <- gather_src[l][gather_idx[r]] (+ the new row for layer 0). */
int zk_dec_embed(const int* ids, int pad_id, const void* table, const float* bias, const float* timing, void* out, int rows,
                 int H, float scale, int pos0, const int* pos_dev, float* cache, void* cat, float inv_count,
                 const float* gather_src, const int* gather_idx, int nl, zk_stream_t stream);

/* The replay loop of the device-resident search in one call (no interpreter between two decode steps): the two parity
 * graphs of the step alternately (starting with `parity`), poll replays per group, the search's 16-byte control block
 * {time, stop, overflow, -} copied into one of two pinned 4-int slots behind every group and read one group later; ends
 * when the stop flag is up or more than max_launch replays went out, with the stream drained.  search.py:85-113 (the stop
 * test) decides on the device; this only carries the flag to the host. */
int zk_beam_dev_run(void* graph_even, void* graph_odd, int parity, const int* ctrl_dev, int* ctrl_pinned8, int max_launch,
                    int poll, zk_stream_t stream, int* launched_out, int* newest_slot_out);

/* ---- round 5: the fp32 decode path (hp.decode_dtype = "float32"; zero_amd/csrc/zk_f32.hip, zero_amd/models/_decode_f32.py).
 * The reference computes in float32 by default (utils/dtype.py:12-15, run.py `default_dtype`); this mode rounds where it
 * rounds -- fp32 MASTER weights, fp32 activations, fp32 accumulation -- so that greedy / beam hypotheses can be compared
 * token for token with the fp32 oracle (the bf16 path cannot be: two correct implementations that round at different points
 * part on ~3 % of the sentences).  Every matrix is plain fp32 row-major; all pointers are device pointers.
 *   zk_f32_gemm       func.py:14-65 linear / transformer.py:182-196 logits: C[M,N] = A[M,K] x (tb ? B[N,K]^T : B[K,N])
 *                     (+ bias[N]) (act 1: ReLU, func.py:332).  K % 4 == 0, lda % 4 == 0 (tb: ldb % 4 == 0), A (tb: B) 16-byte
 *                     aligned.  v_mfma_f32_32x32x2_f32 = an fp32 fmaf chain per output in a fixed k order.
 *   zk_f32_embed      transformer.py:16-33 / 88-119: out[r] = table[ids[r]] * scale + bias + timing[pos0 (or *pos_dev) + r % L];
 *                     all_pad (device flag from zk_all_equal, may be NULL): non-zero -> the embedding part is exact zeros
 *                     (transformer.py:113-115, the first decode step).
 *   zk_f32_add_ln     func.py:321-324 + 289-303: out = gamma (s - mean) / sqrt(var + eps) + beta, s = x + y (y may be NULL).
 *   zk_f32_attn       func.py:218-256 on q [B][Lq][..] / k, v [B / kv_group][Lk][..] (row strides ld*, sentence strides bs*
 *                     in elements; head h at columns h d ..): q * scale, + (1 - kmask) * (-mask_inf) (func.py:372-387;
 *                     kmask fp32 [B / kv_group, ldmask] may be NULL), softmax, x V.  nkeys_dev (may be NULL): only the
 *                     first *nkeys_dev + 1 keys exist (the self-attention cache of decode position *nkeys_dev).
 *                     rpr_k / rpr_v (may be NULL; modules/rpr.py:10-75): relative-position tables [2 max_rel + 1, d] --
 *                     logits += q . r_k[clip(i - j) + max_rel], o += sum_j p_j r_v[...], i = q_pos0 (or *q_pos_dev) + row.
 *   zk_f32_add_rows   out[r] = a[r] + b[r]: the merged attention's o + aan_o (func.py:258-275, transformer_fuse).
 *   zk_f32_aan_step   transformer_aan.py:110-112: cat[r] = [x[r] | (x[r] + cache[r]) / (t + 1)], cache[r] += x[r].
 *   zk_f32_gate       transformer_aan.py:186-189: g = sigmoid(z[:, :H]) cat[:, :H] + sigmoid(z[:, H:]) cat[:, H:].
 * (cache appends and beam reorders are byte moves: zk_cache_rows / zk_gather_rows_ex with 4-byte elements.) */
int zk_f32_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int tb,
                const float* bias, int act, zk_stream_t stream);
int zk_f32_embed(const int* ids, int rows, int L, const float* table, const float* bias, const float* timing, int timing_rows,
                 float* out, int H, float scale, int pos0, const int* pos_dev, const int* all_pad, zk_stream_t stream);
int zk_f32_add_ln(const float* x, const float* y, const float* gamma, const float* beta, float* out, int rows, int H, float eps,
                  zk_stream_t stream);
int zk_f32_attn(const float* q, const float* k, const float* v, float* out, int B, int nh, int Lq, int Lk, int d, int ldq,
                int ldk, int ldv, int ldo, long bsq, long bsk, long bsv, long bso, const float* kmask, int ldmask, int kv_group,
                float scale, float mask_inf, const int* nkeys_dev, const float* rpr_k, const float* rpr_v, int max_rel,
                int q_pos0, const int* q_pos_dev, zk_stream_t stream);
int zk_f32_aan_step(const float* x, float* cache, float* cat, int rows, int H, int time, const int* time_dev,
                    zk_stream_t stream);
int zk_f32_gate(const float* z, const float* cat, float* g, int rows, int H, zk_stream_t stream);
int zk_f32_add_rows(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int cols,
                    zk_stream_t stream);
/* round 6: row-local neighbours of the fp32 decode step's launches folded together (a step is a chain of ~5 us launches).
 *   zk_f32_ln_fused    zk_f32_add_ln with, optionally, the average-attention gate as the producer of y (z / cat_in given,
 *                      x = ybuf = NULL: y = sigmoid(z_i) x_c + sigmoid(z_f) y_c and the residual is x_c = cat_in[:, :H];
 *                      transformer_aan.py:186-192) and / or the NEXT layer's average attention from the normalised row
 *                      (cache / cat_out given: cache += out, cat_out = [out | cache / (t + 1)]; transformer_aan.py:110-112).
 *                      H % 4 == 0, H <= 2048, every pointer 16-byte aligned.
 *   zk_f32_embed_step  zk_f32_embed for one decode position with the all-pad test of transformer.py:113-115 inside the
 *                      launch (pad_id >= 0) and, optionally, the first layer's average attention (cache / cat_out).
 *   zk_f32_gemm_legacy A/B switch: 1 = the round-5 GEMM kernels for every shape; returns the old value (< 0: query). */
int zk_f32_ln_fused(const float* x, const float* ybuf, const float* z, const float* cat_in, const float* gamma,
                    const float* beta, float* out, int rows, int H, float eps, float* cache, float* cat_out, int time,
                    const int* time_dev, zk_stream_t stream);
int zk_f32_embed_step(const int* ids, int rows, const float* table, const float* bias, const float* timing, int timing_rows,
                      float* out, int H, float scale, int pos0, const int* pos_dev, int pad_id, float* cache, float* cat_out,
                      zk_stream_t stream);
int zk_f32_gemm_legacy(int on);

/* hipGraph plumbing: capture a sequence of the calls above once, replay per step */
int zk_graph_begin(zk_stream_t stream);
int zk_graph_end(zk_stream_t stream, void** exec_out);
int zk_graph_launch(void* exec, zk_stream_t stream);
/* rewrite the arguments of the ONE zk_copy_many launch a captured graph holds (arguments as for zk_copy_many, n >= 1): the
 * training step's graph starts with the copy of the prepared batch out of one of several staging sets -- the feed_dict of
 * main.py:286-294 -- and the set changes from step to step (zero_amd/main.py Trainer.step) */
int zk_graph_set_copy_many(void* exec, void* const* dsts, const void* const* srcs, const size_t* nbytes, int n);
/* the most (dst, src) pairs one zk_copy_many launch takes (16): a commit with more pairs makes several launches and cannot
 * be the rewritable first node of a step graph */
int zk_copy_many_max(void);
int zk_graph_destroy(void* exec);
/* number of nodes (= kernel launches) of the graph the last zk_graph_end instantiated: lets bench.py report the
 * launches per captured step without a profiler */
int zk_graph_last_nodes(void);

/* hardware-layout probes used by the GPU tests */
int zk_probe_mfma32(const void* A, const void* Bt, float* D, zk_stream_t stream);
int zk_probe_mfma16(const void* A, const void* Bt, float* D, zk_stream_t stream);
int zk_probe_tr16(void* out, zk_stream_t stream);
/* measurement aid: reads each of `bytes` (16-byte aligned, multiple of 16 KB) once with access pattern 0..4 (zk_probe.hip)
 * -- a known byte count per access shape for calibrating the FETCH_SIZE counter (scripts/fetch_calibrate.sh) */
int zk_probe_read(const void* src, size_t bytes, int pattern, float* sink, zk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ZERO_HIP_H_ */
