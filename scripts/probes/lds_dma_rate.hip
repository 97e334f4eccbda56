// Issue rate of LDS-DMA (global_load_lds_dwordx4) per wave and per CU on gfx950, L2-resident source.
//   mode 0: s_mov m0 before every load (what zk_gemm2.hip does)   mode 1: M0 written once, same LDS piece
//   mode 2: M0 written once, instruction offset selects the piece (checks that the offset moves the LDS side)
//   mode 3: plain global_load_dwordx4 into VGPRs
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/lds_dma_rate scripts/probes/lds_dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k_rate(const unsigned char* __restrict__ src, unsigned long long* out, int iters,
                                               unsigned* check) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned lbase = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) unsigned char*)lds) + (wave % 8) * 8192;
  // 8 KiB per wave-iteration out of a 64-KiB window per workgroup (L2 resident after the warm-up launch)
  const unsigned char* p = src + ((size_t)blockIdx.x << 16) + lane * 16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  if (MODE == 1 || MODE == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lbase) : "memory");
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* q = p + (size_t)(((it * nw + wave) * 8192) & 0xffff);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(q + j * 1024), "s"(lbase + j * 1024) : "memory");
      else if (MODE == 1)
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(q + j * 1024) : "memory");
      else if (MODE == 2)
        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(q + (j & 3) * 1024 * 0), "n"(0) : "memory");
      else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(q + j * 1024));
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
    }
    if (MODE != 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // at most 2 iterations in flight
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[(blockIdx.x * nw + wave) * 2 + 0] = t1 - t0;
    out[(blockIdx.x * nw + wave) * 2 + 1] = t2 - t0;
  }
  __syncthreads();
  if (check && blockIdx.x == 0 && tid < 64) check[tid] = reinterpret_cast<unsigned*>(lds)[tid * 4] ^ acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int MODE>
static void run(const unsigned char* src, unsigned long long* dout, unsigned* dcheck, int waves, int iters) {
  const int grid = 256;
  hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(waves * 64), 0, 0, src, dout, iters, dcheck);   // warm L2
  hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(waves * 64), 0, 0, src, dout, iters, dcheck);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * waves * 2);
  (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < grid * waves; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  a /= grid * waves; b /= grid * waves;
  const double ninstr = iters * 8.0;
  printf("mode %d  waves/CU %2d : %6.1f cycles per instr per wave (issue), %6.1f incl. drain  ->  %5.1f B/clk/CU\n", MODE, waves,
         a / ninstr, b / ninstr, waves * ninstr * 1024.0 / b);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned char* src; unsigned long long* dout; unsigned* dcheck;
  if (hipMalloc(&src, (size_t)256 << 16) != hipSuccess || hipMemset(src, 1, (size_t)256 << 16) != hipSuccess ||
      hipMalloc(&dout, 256 * 16 * 2 * 8) != hipSuccess || hipMalloc(&dcheck, 256) != hipSuccess) return 1;
  for (int waves : {1, 2, 4, 8, 16}) {
    run<0>(src, dout, dcheck, waves, 64);
    run<1>(src, dout, dcheck, waves, 64);
    run<3>(src, dout, dcheck, waves, 64);
  }
  return 0;
}
