// zk_probe.hip -- hardware-layout probes (test infrastructure for the kernels' assumptions).
// Each probe loads MFMA operands with the lane->element mapping the production kernels
// assume and writes D through the assumed C/D mapping; the GPU tests compare with A*B.
#include "zk_common.h"
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short v4s_t __attribute__((ext_vector_type(4)));

// A [32][16] row-major, Bt [32][16] (= B^T: n-major, k contiguous), D [32][32]
__global__ void k_probe_mfma32(const bf16_t* A, const bf16_t* Bt, float* D) {
  const int lane = threadIdx.x;
  const uint4 a = *reinterpret_cast<const uint4*>(A + (lane & 31) * 16 + (lane >> 5) * 8);
  const uint4 b = *reinterpret_cast<const uint4*>(Bt + (lane & 31) * 16 + (lane >> 5) * 8);
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    D[row * 32 + col] = acc[r];
  }
}
// A [16][32], Bt [16][32], D [16][16]
__global__ void k_probe_mfma16(const bf16_t* A, const bf16_t* Bt, float* D) {
  const int lane = threadIdx.x;
  const uint4 a = *reinterpret_cast<const uint4*>(A + (lane & 15) * 32 + (lane >> 4) * 8);
  const uint4 b = *reinterpret_cast<const uint4*>(Bt + (lane & 15) * 32 + (lane >> 4) * 8);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}
// ds_read_b64_tr_b16 semantics dump: lds[i] = i, lane l supplies byte address 8*l
__global__ void k_probe_tr16(short* out) {
  __shared__ short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) v4s_t*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

extern "C" {
int zk_probe_mfma32(const void* A, const void* Bt, float* D, hipStream_t s) {
  hipLaunchKernelGGL(k_probe_mfma32, dim3(1), dim3(64), 0, s, (const bf16_t*)A, (const bf16_t*)Bt, D);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_probe_mfma16(const void* A, const void* Bt, float* D, hipStream_t s) {
  hipLaunchKernelGGL(k_probe_mfma16, dim3(1), dim3(64), 0, s, (const bf16_t*)A, (const bf16_t*)Bt, D);
  ZK_LAUNCH_CHECK();
  return 0;
}
int zk_probe_tr16(void* out, hipStream_t s) {
  hipLaunchKernelGGL(k_probe_tr16, dim3(1), dim3(64), 0, s, (short*)out);
  ZK_LAUNCH_CHECK();
  return 0;
}
}
