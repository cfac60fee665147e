// Random 8-byte reads from a window of W bytes of a 4 GiB table: how many per second does MI355X serve from L2 (4 MiB per XCD),
// from the 256 MiB Infinity Cache and from HBM?  Decides whether the L1 probe should walk the hash table in hash-range passes.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/gather tools/ubench/gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MLP>
__global__ __launch_bounds__(256) void k_gather(const uint2 *__restrict__ t, uint64_t base, uint64_t mask, int rounds, uint32_t *out)
{
  uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9e3779b97f4a7c15ull + 12345;
  uint32_t acc = 0;
  for (int r = 0; r < rounds; r++) {
    uint2 v[MLP];
#pragma unroll
    for (int q = 0; q < MLP; q++) {
      x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
      v[q] = t[base + (x & mask)];
    }
#pragma unroll
    for (int q = 0; q < MLP; q++) acc += v[q].x ^ v[q].y;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// Random 8-byte WRITES (round 5): what a bucketed ("merge-join") seed probe would pay to put its (first, count) results back into
// sketch order — the scatter that follows the L2-resident table walk.
template <int MLP>
__global__ __launch_bounds__(256) void k_scatter(uint2 *__restrict__ t, uint64_t mask, int rounds)
{
  uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9e3779b97f4a7c15ull + 12345;
  for (int r = 0; r < rounds; r++) {
#pragma unroll
    for (int q = 0; q < MLP; q++) {
      x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
      t[x & mask] = make_uint2((uint32_t)x, (uint32_t)r);
    }
  }
}

int main()
{
  const uint64_t nEntries = 1ull << 29;       // 4 GiB of 8-byte entries
  uint2 *t; uint32_t *out;
  CK(hipMalloc(&t, nEntries * 8)); CK(hipMalloc(&out, 4));
  CK(hipMemset(t, 1, nEntries * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%10s %6s %14s %12s\n", "window", "MLP", "reads/s", "ms per 4e8");
  for (uint64_t wBytes = 8ull << 20; wBytes <= 4ull << 30; wBytes <<= 1) {
    for (int mlp : {1, 4}) {
      const int rounds = 16 / mlp * 4, blocks = 256 * 32;
      const double reads = (double)blocks * 256 * rounds * mlp;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        if (mlp == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, t, 0ull, wBytes / 8 - 1, rounds, out);
        else hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, t, 0ull, wBytes / 8 - 1, rounds, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%7llu MiB %6d %14.3e %12.2f\n", (unsigned long long)(wBytes >> 20), mlp, reads / (best * 1e-3), 4e8 / (reads / (best * 1e-3)) * 1e3);
    }
  }
  printf("\nrandom 8-byte writes\n%10s %14s %12s\n", "window", "writes/s", "ms per 4e8");
  for (uint64_t wBytes = 8ull << 20; wBytes <= 4ull << 30; wBytes <<= 3) {
    const int rounds = 16, blocks = 256 * 32;
    const double writes = (double)blocks * 256 * rounds * 4;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_scatter<4>, dim3(blocks), dim3(256), 0, 0, t, wBytes / 8 - 1, rounds);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%7llu MiB %14.3e %12.2f\n", (unsigned long long)(wBytes >> 20), writes / (best * 1e-3), 4e8 / (writes / (best * 1e-3)) * 1e3);
  }
  return 0;
}
