// Cost of a software barrier among co-resident workgroups on gfx950: all 256 (one per CU) on one counter, or
// eight independent groups of 32 (workgroup b joins group b % 8 -- the XCD it is observed to run on).
// Counter = monotonic arrival count, relaxed agent-scope atomics; the data handed over a real barrier would
// need device-scope accesses on top (split-K experiment).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/xcd_barrier scripts/probes/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k_barrier(int* cnt, unsigned long long* out, int iters, int groups, int* xcc_ids) {
  extern __shared__ int lds_pad[];                           // occupies LDS like a GEMM workgroup would
  if (threadIdx.x == 0) lds_pad[0] = 0;
  const int g = blockIdx.x % groups, gsize = gridDim.x / groups;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_ids[blockIdx.x] = (int)(xcc & 0xf);
  }
  __syncthreads();
  unsigned long long t0 = 0;
  for (int it = 0; it <= iters; ++it) {
    if (it == 1) t0 = __builtin_readcyclecounter();       // iteration 0 lines everybody up
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&cnt[g * 64], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (it + 1) * gsize;
      while (__hip_atomic_load(&cnt[g * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int* cnt; unsigned long long* out; int* xcc;
  if (hipMalloc(&cnt, 8 * 64 * 4) != hipSuccess || hipMalloc(&out, 512 * 8) != hipSuccess || hipMalloc(&xcc, 512 * 4) != hipSuccess) return 1;
  const int iters = 200;
  // (workgroups, threads, dynamic LDS bytes): one 64-thread workgroup per CU; two 512-thread, 64-KiB workgroups per CU
  const int cfg[2][3] = {{256, 64, 0}, {512, 512, 65536}};
  for (int c = 0; c < 2; ++c) {
    const int nwg = cfg[c][0];
    for (int groups : {1, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cnt, 0, 8 * 64 * 4);
        hipLaunchKernelGGL(k_barrier, dim3(nwg), dim3(cfg[c][1]), cfg[c][2], 0, cnt, out, iters, groups, xcc);
        (void)hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(nwg); std::vector<int> x(nwg);
      (void)hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(x.data(), xcc, nwg * 4, hipMemcpyDeviceToHost);
      double a = 0; for (auto v : h) a += v; a /= nwg;
      int match = 0; for (int b = 0; b < nwg; ++b) match += (x[b] == b % 8);
      printf("%d workgroups x %d threads, %d group(s) of %3d: %.0f cycles per barrier (%.2f us at 2.4 GHz); blockIdx %% 8 == XCC_ID for %d of %d\n",
             nwg, cfg[c][1], groups, nwg / groups, a / iters, a / iters / 2400.0, match, nwg);
    }
  }
  return 0;
}
