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

// ---- read-pattern probe for calibrating the FETCH_SIZE counter (MI355X_MICROARCH.md, HBM section: only the wide coalesced
// 16 B / lane read is calibrated -- x2 on gfx950 -- "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Every pattern reads each of the `bytes` exactly once; the launch name carries the pattern so
// that scripts/pmc_traffic.py separates them:
//   0: 16 B per lane, a wave instruction covers 1 KB contiguous (the streaming kernels, LDS-DMA of the GEMM ring)
//   1: the MFMA-fragment pattern of the decode kernels (zk_decfuse.hip load_wq / load_wo): lane -> row (lane & 15), 16 B at
//      column (lane >> 4) * 16 B of a 1 KB row: 16 rows x 64 B per instruction, the next instruction the next 64 B of the rows
//   2: 4 B per lane (256 B contiguous per instruction), 3: 8 B per lane (512 B), 4: 2 B per lane (128 B)
template <int PATTERN>
__global__ void __launch_bounds__(256) k_probe_read(const unsigned char* __restrict__ src, size_t bytes, float* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  float acc = 0.f;
  if (PATTERN == 0) {
    for (size_t o = wave * 1024; o + 1024 <= bytes; o += nwaves * 1024) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + o + lane * 16);
      acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
    }
  } else if (PATTERN == 1) {
    // blocks of 16 rows x 1 KB: 16 instructions of 16 rows x 64 B
    for (size_t o = wave * 16384; o + 16384 <= bytes; o += nwaves * 16384) {
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + o + (size_t)(lane & 15) * 1024 + ks * 64 + (lane >> 4) * 16);
        acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
      }
    }
  } else {
    constexpr int W = PATTERN == 2 ? 4 : (PATTERN == 3 ? 8 : 2);
    for (size_t o = wave * 64 * W; o + 64 * W <= bytes; o += nwaves * 64 * W) {
      if (W == 4) acc += __uint_as_float(*reinterpret_cast<const uint32_t*>(src + o + lane * 4));
      else if (W == 8) { const uint2 v = *reinterpret_cast<const uint2*>(src + o + lane * 8); acc += __uint_as_float(v.x ^ v.y); }
      else acc += (float)*reinterpret_cast<const unsigned short*>(src + o + lane * 2);
    }
  }
  if (acc == 123.456f) sink[0] = acc;          // keeps the loads
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
int zk_probe_read(const void* src, size_t bytes, int pattern, float* sink, hipStream_t s) {
  ZK_CHECK_ARG(pattern >= 0 && pattern <= 4, "zk_probe_read: pattern 0..4");
  ZK_CHECK_ARG((((uintptr_t)src) & 15) == 0, "zk_probe_read: src must be 16-byte aligned");
  const dim3 g(2048), b(256);
  switch (pattern) {
    case 0: hipLaunchKernelGGL(k_probe_read<0>, g, b, 0, s, (const unsigned char*)src, bytes, sink); break;
    case 1: hipLaunchKernelGGL(k_probe_read<1>, g, b, 0, s, (const unsigned char*)src, bytes, sink); break;
    case 2: hipLaunchKernelGGL(k_probe_read<2>, g, b, 0, s, (const unsigned char*)src, bytes, sink); break;
    case 3: hipLaunchKernelGGL(k_probe_read<3>, g, b, 0, s, (const unsigned char*)src, bytes, sink); break;
    default: hipLaunchKernelGGL(k_probe_read<4>, g, b, 0, s, (const unsigned char*)src, bytes, sink); break;
  }
  ZK_LAUNCH_CHECK();
  return 0;
}
}
