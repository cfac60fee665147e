// scan.hpp — device-wide exclusive prefix sum (int32 counts -> uint32 offsets), three small kernels, recursive on
// the block totals.  Used for tile record offsets and per-fragment candidate offsets.
#pragma once
#include "common.hpp"

namespace ani {

constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = kTPB * kScanPerThread;   // 2048

static __global__ __launch_bounds__(kTPB) void k_scan_blocks(const int32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                                                      int32_t *__restrict__ blockTotals)
{
  __shared__ int ws[16];
  const uint32_t first = blockIdx.x * kScanPerBlock + threadIdx.x * kScanPerThread;
  int loc[kScanPerThread]; int sum = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; j++) { uint32_t i = first + j; int x = i < n ? in[i] : 0; loc[j] = sum; sum += x; }
  int tot; int off = block_excl_scan(sum, ws, &tot);
#pragma unroll
  for (int j = 0; j < kScanPerThread; j++) { uint32_t i = first + j; if (i < n) out[i] = (uint32_t)(off + loc[j]); }
  if (threadIdx.x == 0) blockTotals[blockIdx.x] = tot;
}

static __global__ void k_scan_add(uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ blockOffsets)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += blockOffsets[i / kScanPerBlock];
}

}  // namespace ani
