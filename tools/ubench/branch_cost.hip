// Lone-wave issue costs on gfx950: cycles per loop iteration for K independent v_fma + one taken backward branch,
// with F extra forward branches (taken / not taken) per iteration.  hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, int F, bool TAKEN>
__global__ void kern(unsigned long long* out, int iters, float seed, int flag) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < K / 8; ++k) {
      asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                   "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
      // forward branch over 8 instructions: taken when flag == 0 (TAKEN) / never taken (flag compared the other way)
      if (TAKEN) asm volatile("s_cmp_eq_u32 %1, 0\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n"
                              "v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n1:\n" : "+v"(a0) : "s"(flag) : "scc");
      else asm volatile("s_cmp_eq_u32 %1, 1\n s_cbranch_scc1 1f\n s_nop 0\n1:\n" : "+v"(a0) : "s"(flag) : "scc");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); }
}

template <int K, int F, bool TAKEN>
static double run(unsigned long long* d, int iters) {
  hipLaunchKernelGGL((kern<K, F, TAKEN>), dim3(1), dim3(64), 0, 0, d, iters, 1.0f, 0);
  hipLaunchKernelGGL((kern<K, F, TAKEN>), dim3(1), dim3(64), 0, 0, d, iters, 1.0f, 0);
  unsigned long long h[2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  return (double)h[0] / iters;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  const int it = 20000;
  printf("K v_fma + loop branch         : K=16 %.1f  K=64 %.1f  K=256 %.1f cycles/iter\n", run<16, 0, true>(d, it), run<64, 0, true>(d, it), run<256, 0, true>(d, it));
  printf("K=64 + F taken fwd branches   : F=0 %.1f  F=4 %.1f  F=16 %.1f\n", run<64, 0, true>(d, it), run<64, 4, true>(d, it), run<64, 16, true>(d, it));
  printf("K=64 + F not-taken branches   : F=0 %.1f  F=4 %.1f  F=16 %.1f\n", run<64, 0, false>(d, it), run<64, 4, false>(d, it), run<64, 16, false>(d, it));
  return 0;
}
