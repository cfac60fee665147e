// sort_device.hip — the one library call on the path: a stable LSD radix sort of (hash, seqId<<32|wpos) pairs that
// orders the reference minimizers by hash (≙ filling minimizerPosLookupIndex, src/map/include/winSketch.hpp:181-193), and of
// the 64-bit seed hits of an oversized fragment.  rocPRIM's device radix sort is used as plumbing;
// everything else on the path is hand-written.  Temporary storage is supplied by the caller (two-phase API: tmp == nullptr
// returns the required size) so that it comes out of the caching allocator.
#include <cstring>
#include <cstdint>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

extern "C" int ani_sort_pairs_u32_u64(const uint32_t *keysIn, uint32_t *keysOut, const uint64_t *valsIn, uint64_t *valsOut,
                                      size_t n, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  if (n == 0) { if (!tmp) *tmpBytes = 0; return 0; }
  hipError_t e = rocprim::radix_sort_pairs(tmp, *tmpBytes, keysIn, keysOut, valsIn, valsOut, n, 0, 32, stream);
  if (e != hipSuccess || !tmp) return (int)e;
  return (int)hipStreamSynchronize(stream);
}

extern "C" int ani_sort_keys_u64(const uint64_t *keysIn, uint64_t *keysOut, size_t n, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  if (n == 0) { if (!tmp) *tmpBytes = 0; return 0; }
  hipError_t e = rocprim::radix_sort_keys(tmp, *tmpBytes, keysIn, keysOut, n, 0, 64, stream);
  if (e != hipSuccess || !tmp) return (int)e;
  return (int)hipStreamSynchronize(stream);
}

// the low `endBit` bits only (the batched L1 path packs (fragment, seqId, wpos) into as few bits as the index chunk needs); no
// host synchronisation: the caller's next launch is on the same stream
extern "C" int ani_sort_keys_u64_bits(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  if (n == 0) { if (!tmp) *tmpBytes = 0; return 0; }
  if (endBit < 1) endBit = 1;
  if (endBit > 64) endBit = 64;
  return (int)rocprim::radix_sort_keys(tmp, *tmpBytes, keysIn, keysOut, n, 0, (unsigned)endBit, stream);
}

// (querySeqId<<32 | refSeqId, record index): orders mapping records handed to ani_compute_cgi in an arbitrary order
extern "C" int ani_sort_pairs_u64_u32(const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                      size_t n, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  if (n == 0) { if (!tmp) *tmpBytes = 0; return 0; }
  hipError_t e = rocprim::radix_sort_pairs(tmp, *tmpBytes, keysIn, keysOut, valsIn, valsOut, n, 0, 64, stream);
  if (e != hipSuccess || !tmp) return (int)e;
  return (int)hipStreamSynchronize(stream);
}
