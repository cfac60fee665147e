// radix_bits.hip — how many bits per pass?  The yardstick for "three 11-bit passes instead of four 8-bit ones" (VERDICT r05, item 9):
// rocPRIM's onesweep radix sort — the same algorithm as csrc/kernels/radix.hpp, tuned by its authors — instantiated with 8, 10 and 11
// bits per pass on the index build's data (u32 minimizer-like hashes + u64 payload, 4 x 10^8 elements).  rocPRIM appears HERE ONLY; the
// product library does not link it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o radix_bits radix_bits.hip && ./radix_bits [n]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
template <int BITS, int BS, int IPT, rocprim::block_radix_rank_algorithm ALG>
float run(uint32_t *k, uint32_t *ko, uint64_t *v, uint64_t *vo, size_t n)
{
  using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;
  size_t tb = 0; void *tmp = nullptr;
  if (rocprim::radix_sort_pairs<cfg>(nullptr, tb, k, ko, v, vo, n, 0, 32, 0) != hipSuccess) return -1;
  hipMalloc(&tmp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(a);
    if (rocprim::radix_sort_pairs<cfg>(tmp, tb, k, ko, v, vo, n, 0, 32, 0) != hipSuccess) return -2;
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  hipFree(tmp);
  return best;
}
__global__ void k_make(uint32_t *k, uint64_t *v, uint64_t n)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t h = 0xffffffffu;
    for (int q = 0; q < 24; q++) { x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; const uint32_t w = (uint32_t)(x >> 16); h = w < h ? w : h; }
    k[i] = h; v[i] = i;
  }
}
int main(int argc, char **argv)
{
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 400000000ull;
  uint32_t *k, *ko; uint64_t *v, *vo;
  hipMalloc(&k, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&v, n * 8); hipMalloc(&vo, n * 8);
  hipLaunchKernelGGL(k_make, dim3(8192), dim3(256), 0, 0, k, v, n); hipDeviceSynchronize();
  using A = rocprim::block_radix_rank_algorithm;
  printf("rocPRIM onesweep, u32 keys + u64 values, n = %zu (minimizer-like hashes: minimum of 24 uniform values)\n", n);
  printf("  8 bits x 4 passes, 512 x 12, match      : %8.3f ms\n", run<8, 512, 12, A::match>(k, ko, v, vo, n));
  printf("  8 bits x 4 passes, 1024 x 6 (default-ish): %8.3f ms\n", run<8, 1024, 6, A::match>(k, ko, v, vo, n));
  printf("  11 bits x 3 passes, 512 x 12, match     : %8.3f ms\n", run<11, 512, 12, A::match>(k, ko, v, vo, n));
  printf("  11 bits x 3 passes, 256 x 24, match     : %8.3f ms\n", run<11, 256, 24, A::match>(k, ko, v, vo, n));
  printf("  10 bits (4 passes: 10+10+10+2), 512 x 12 : %8.3f ms\n", run<10, 512, 12, A::match>(k, ko, v, vo, n));
  return 0;
}
