// zk_prog.h -- recording side of the layer program (zk_layer.hip): while a program is being recorded on the calling
// thread, the entry points that can be part of one (zk_gemm, zk_attn_fwd, zk_attn_bwd, zk_add_ln_fwd, zk_add_ln_bwd)
// append an op with exactly the arguments they would have launched with, instead of launching.
#pragma once
#include "zk_gemm.h"

struct AttnArgs;   // zk_attn_dev.h

#ifdef ZK_EXPERIMENTS
bool zk_prog_active();
// the call cannot be expressed as a program op: recording is marked failed (zk_prog_end reports it) and the caller
// returns this error
int zk_prog_reject(const char* why);
int zk_prog_record_gemm(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int ta, int tb,
                        int bm, int bn, const GemmEpi& e);
int zk_prog_record_attn_fwd(const AttnArgs& a, bf16_t* out, int ldo, float* lse, int nkt);
int zk_prog_record_attn_bwd64(const AttnArgs& a, const bf16_t* o, int ldo, const bf16_t* dout, int lddo,
                              const float* lse, bf16_t* dq, int lddq, bf16_t* dk, int lddk, bf16_t* dv, int lddv);
int zk_prog_record_add_ln_fwd(const bf16_t* x, const bf16_t* y, const float* gamma, const float* beta, bf16_t* out,
                              bf16_t* sum_out, float* mean, float* rstd, int rows, int H, float eps, uint32_t thr,
                              float inv_keep, const uint64_t* seed, uint32_t sid);
#else
// default build (no `make EXPERIMENTS=1`): the layer program (zk_layer.hip, measured 1.25x slower than launch-per-op,
// profiles/r02_layer_program_experiment.txt) is not compiled; the recording hooks fold to constants
inline bool zk_prog_active() { return false; }
inline int zk_prog_reject(const char*) { return -1; }
inline int zk_prog_record_gemm(const bf16_t*, const bf16_t*, int, int, int, int, int, int, int, int, int, const GemmEpi&) { return -1; }
inline int zk_prog_record_attn_fwd(const AttnArgs&, bf16_t*, int, float*, int) { return -1; }
inline int zk_prog_record_attn_bwd64(const AttnArgs&, const bf16_t*, int, const bf16_t*, int, const float*, bf16_t*, int,
                                     bf16_t*, int, bf16_t*, int) { return -1; }
inline int zk_prog_record_add_ln_fwd(const bf16_t*, const bf16_t*, const float*, const float*, bf16_t*, bf16_t*, float*,
                                     float*, int, int, float, uint32_t, float, const uint64_t*, uint32_t) { return -1; }
#endif
