// sort_device.hip — the one library call on the path: a stable LSD radix sort of (hash, seqId<<32|wpos) pairs that
// orders the reference minimizers by hash (≙ filling minimizerPosLookupIndex, src/map/include/winSketch.hpp:181-193).
// rocPRIM's device radix sort is used as plumbing; everything else on the path is hand-written.
#include <cstring>
#include <cstdint>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

extern "C" int ani_sort_pairs_u32_u64(const uint32_t *keysIn, uint32_t *keysOut, const uint64_t *valsIn, uint64_t *valsOut,
                                      size_t n, hipStream_t stream)
{
  if (n == 0) return 0;
  size_t tmpBytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmpBytes, keysIn, keysOut, valsIn, valsOut, n, 0, 32, stream);
  if (e != hipSuccess) return (int)e;
  void *tmp = nullptr;
  e = hipMalloc(&tmp, tmpBytes ? tmpBytes : 1);
  if (e != hipSuccess) return (int)e;
  e = rocprim::radix_sort_pairs(tmp, tmpBytes, keysIn, keysOut, valsIn, valsOut, n, 0, 32, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  return (int)(e != hipSuccess ? e : e2);
}

// u64 keys only: the (seqId<<32 | wpos) seed hits of one oversized fragment (L1 global-memory path)
extern "C" int ani_sort_keys_u64(const uint64_t *keysIn, uint64_t *keysOut, size_t n, hipStream_t stream)
{
  if (n == 0) return 0;
  size_t tmpBytes = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, tmpBytes, keysIn, keysOut, n, 0, 64, stream);
  if (e != hipSuccess) return (int)e;
  void *tmp = nullptr;
  e = hipMalloc(&tmp, tmpBytes ? tmpBytes : 1);
  if (e != hipSuccess) return (int)e;
  e = rocprim::radix_sort_keys(tmp, tmpBytes, keysIn, keysOut, n, 0, 64, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  return (int)(e != hipSuccess ? e : e2);
}
