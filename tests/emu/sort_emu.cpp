/* TEST INFRASTRUCTURE ONLY — CPU stand-in for fastani_amd/csrc/sort_device.hip (rocPRIM radix sort) in the emu build. */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
#include <vector>
extern "C" int ani_sort_pairs_u32_u64(const uint32_t *keysIn, uint32_t *keysOut, const uint64_t *valsIn, uint64_t *valsOut,
                                      size_t n, void *tmp, size_t *tmpBytes, hipStream_t)
{
  if (!tmp) { *tmpBytes = 16; return 0; }
  std::vector<uint32_t> ord(n);
  std::iota(ord.begin(), ord.end(), 0u);
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return keysIn[a] < keysIn[b]; });
  for (size_t i = 0; i < n; i++) { keysOut[i] = keysIn[ord[i]]; valsOut[i] = valsIn[ord[i]]; }
  return 0;
}

extern "C" int ani_sort_keys_u64(const uint64_t *keysIn, uint64_t *keysOut, size_t n, void *tmp, size_t *tmpBytes, hipStream_t)
{
  if (!tmp) { *tmpBytes = 16; return 0; }
  std::vector<uint64_t> v(keysIn, keysIn + n);
  std::sort(v.begin(), v.end());
  for (size_t i = 0; i < n; i++) keysOut[i] = v[i];
  return 0;
}

extern "C" int ani_sort_keys_u64_bits(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t)
{
  if (!tmp) { *tmpBytes = 16; return 0; }
  const uint64_t mask = endBit >= 64 ? ~0ull : ((1ull << endBit) - 1);
  std::vector<uint64_t> v(keysIn, keysIn + n);
  std::stable_sort(v.begin(), v.end(), [mask](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
  for (size_t i = 0; i < n; i++) keysOut[i] = v[i];
  return 0;
}

extern "C" int ani_sort_pairs_u64_u32(const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                      size_t n, void *tmp, size_t *tmpBytes, hipStream_t)
{
  if (!tmp) { *tmpBytes = 16; return 0; }
  std::vector<uint32_t> ord(n);
  std::iota(ord.begin(), ord.end(), 0u);
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return keysIn[a] < keysIn[b]; });
  for (size_t i = 0; i < n; i++) { keysOut[i] = keysIn[ord[i]]; valsOut[i] = valsIn[ord[i]]; }
  return 0;
}
