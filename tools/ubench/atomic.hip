// atomic.hip — what does a returning global atomicAdd on ONE hot address cost when every workgroup of a large grid performs one
// (the pool cursors of k_sketch_fused / k_l1: 1.67 M workgroups per launch)?   hipcc --offload-arch=gfx950 -O3 -o atomic atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_atomic(unsigned long long *ctr, int nCtr, unsigned long long *out, int work)
{
  __shared__ unsigned long long base;
  // some independent arithmetic first, like a real kernel (`work` dependent multiply-adds per thread)
  unsigned long long x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < work; i++) x = x * 0x9E3779B97F4A7C15ull + i;
  if (threadIdx.x == 0) base = nCtr ? atomicAdd(&ctr[(blockIdx.x % nCtr) * 16], 100ull) : (unsigned long long)blockIdx.x * 100ull;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = base + (x & 1);
}

int main()
{
  const int nBlocks = 1666000;
  unsigned long long *ctr, *out;
  hipMalloc(&ctr, 4096 * 16 * 8); hipMalloc(&out, (size_t)nBlocks * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int work : {0, 2000}) {
    for (int nCtr : {0, 1, 8, 64, 1024}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        hipMemset(ctr, 0, 4096 * 16 * 8);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_atomic, dim3(nBlocks), dim3(256), 0, 0, ctr, nCtr, out, work);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      printf("work %5d  counters %5d : %8.3f ms for %d workgroups (%.1f ns per workgroup)\n", work, nCtr, best, nBlocks, best * 1e6 / nBlocks);
    }
  }
  return 0;
}
