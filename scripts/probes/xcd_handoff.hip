// Can one workgroup hand a buffer to another INSIDE a kernel with plain stores / loads?  Producer: plain global
// stores, s_waitcnt vmcnt(0), relaxed agent-scope flag.  Consumer: spins on the flag, optionally invalidates its L1
// (buffer_inv sc1), reads with plain loads and counts wrong words.  Pair S: workgroups 0 and 8 (same XCD, b % 8
// equal); pair X: workgroups 1 and 2 (different XCDs).  The same 64 KiB are rewritten every round, so stale cache
// lines show up as mismatches.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/xcd_handoff scripts/probes/xcd_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define WORDS 16384
template <int INV>
__global__ void __launch_bounds__(256) k_handoff(unsigned* bufS, unsigned* bufX, int* flags, unsigned* errs,
                                                 unsigned long long* cycles, int rounds) {
  const int b = blockIdx.x, tid = threadIdx.x;
  int pair, role;                       // role 0 producer, 1 consumer
  if (b == 0) { pair = 0; role = 0; } else if (b == 8) { pair = 0; role = 1; }
  else if (b == 1) { pair = 1; role = 0; } else if (b == 2) { pair = 1; role = 1; }
  else return;
  unsigned* buf = pair == 0 ? bufS : bufX;
  int* ready = flags + pair * 64;
  int* ack = flags + pair * 64 + 32;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < rounds; ++it) {
    if (role == 0) {
      if (tid == 0) while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      for (int i = tid; i < WORDS; i += 256) buf[i] = (unsigned)it * 1000003u + (unsigned)i;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ready, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid == 0) while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it + 1) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      if (INV) asm volatile("buffer_inv sc1" ::: "memory");
      unsigned bad = 0;
      for (int i = tid; i < WORDS; i += 256) bad += (buf[i] != (unsigned)it * 1000003u + (unsigned)i);
      if (bad) atomicAdd(&errs[pair], bad);
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ack, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0 && role == 1) cycles[pair] = __builtin_readcyclecounter() - t0;
}

template <int INV>
static void run(unsigned* bufS, unsigned* bufX, int* flags, unsigned* errs, unsigned long long* cyc) {
  const int rounds = 300;
  (void)hipMemset(flags, 0, 128 * 4); (void)hipMemset(errs, 0, 8); (void)hipMemset(bufS, 0, WORDS * 4); (void)hipMemset(bufX, 0, WORDS * 4);
  hipLaunchKernelGGL(k_handoff<INV>, dim3(256), dim3(256), 0, 0, bufS, bufX, flags, errs, cyc, rounds);
  (void)hipDeviceSynchronize();
  unsigned e[2]; unsigned long long c[2];
  (void)hipMemcpy(e, errs, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("consumer L1 invalidate %s: same-XCD pair %u wrong words of %d, %.0f cycles per hand-off round trip | cross-XCD pair %u wrong words, %.0f cycles\n",
         INV ? "buffer_inv sc1" : "none          ", e[0], rounds * WORDS, (double)c[0] / rounds, e[1], (double)c[1] / rounds);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *bufS, *bufX, *errs; int* flags; unsigned long long* cyc;
  if (hipMalloc(&bufS, WORDS * 4) != hipSuccess || hipMalloc(&bufX, WORDS * 4) != hipSuccess || hipMalloc(&flags, 128 * 4) != hipSuccess ||
      hipMalloc(&errs, 8) != hipSuccess || hipMalloc(&cyc, 16) != hipSuccess) return 1;
  run<0>(bufS, bufX, flags, errs, cyc);
  run<1>(bufS, bufX, flags, errs, cyc);
  return 0;
}
