// Does global_load_lds_dwordx4 (gfx950) place lane l's 16 bytes at M0 base + 16 l?  Copies 4 KB through LDS and checks.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/glds_copy tools/ubench/glds_copy.hip && /tmp/glds_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const f4* __restrict__ src, f4* __restrict__ dst) {
  __shared__ f4 tile[2][256];
  const int wave = threadIdx.x >> 6;
  for (int b = 0; b < 2; ++b)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b * 256 + threadIdx.x),
                                     (__attribute__((address_space(3))) void*)(&tile[b][wave * 64]), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  dst[threadIdx.x] = tile[0][threadIdx.x];
  dst[256 + threadIdx.x] = tile[1][255 - threadIdx.x];
}
int main() {
  std::vector<float> h(2048), o(2048, -1.0f);
  for (int i = 0; i < 2048; i++) h[i] = (float)i;
  f4 *s, *d;
  hipMalloc(&s, 8192); hipMalloc(&d, 8192);
  hipMemcpy(s, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d);
  hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; i++) if (o[i] != h[i]) bad++;
  for (int t = 0; t < 256; t++) for (int c = 0; c < 4; c++) if (o[1024 + 4 * t + c] != h[1024 + 4 * (255 - t) + c]) bad++;
  printf("global_load_lds_dwordx4: %d mismatches\n", bad);
  return bad != 0;
}
