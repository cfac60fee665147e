// lane_xor<M> (common.hpp: DPP / v_permlane*_swap forms of "the value of lane ^ M") checked lane by lane, and the register bitonic sort
// (block_sort) against std::sort for every size class.   hipcc --offload-arch=gfx950 -O3 -I../../fastani_amd/csrc/kernels lanexor.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>
#include "common.hpp"
using namespace ani;
__global__ void k(uint32_t *out)
{
  uint32_t x = threadIdx.x * 3 + 7;
  out[0 * 64 + threadIdx.x] = lane_xor<1>(x);
  out[1 * 64 + threadIdx.x] = lane_xor<2>(x);
  out[2 * 64 + threadIdx.x] = lane_xor<4>(x);
  out[3 * 64 + threadIdx.x] = lane_xor<8>(x);
  out[4 * 64 + threadIdx.x] = lane_xor<16>(x);
  out[5 * 64 + threadIdx.x] = lane_xor<32>(x);
}
// wave_incl_scan_dpp (row_shr / row_bcast DPP adds) against the shuffle form, four waves with different data
__global__ void k_scan(int *out)
{
  const int v = (int)((threadIdx.x * 2654435761u) >> 24) + (threadIdx.x & 3);
  out[threadIdx.x] = wave_incl_scan_dpp(v);
  out[256 + threadIdx.x] = wave_incl_scan(v);
}
template <class K> __global__ __launch_bounds__(kTPB) void k_sort(K *data, const int *n, int cap)
{
  __shared__ K a[4096];
  const int m = n[blockIdx.x];
  for (int i = threadIdx.x; i < m; i += kTPB) a[i] = data[(size_t)blockIdx.x * cap + i];
  block_sort<K>(a, m);
  for (int i = threadIdx.x; i < m; i += kTPB) data[(size_t)blockIdx.x * cap + i] = a[i];
}
template <class K> int check_sort()
{
  const int sizes[] = {0, 1, 2, 63, 64, 65, 100, 128, 129, 255, 256, 257, 400, 512, 513, 600, 1000, 1024, 1025, 2000, 2048, 2049, 3000, 4095, 4096};
  const int nb = sizeof sizes / sizeof sizes[0], cap = 4096;
  std::vector<K> h((size_t)nb * cap), ref;
  uint64_t x = 88172645463325252ull;
  for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (K)(x >> (sizeof(K) == 4 ? 40 : 9)); }     // duplicates happen in the 32-bit case
  ref = h;
  K *d; int *dn; (void)hipMalloc(&d, h.size() * sizeof(K)); (void)hipMalloc(&dn, sizeof sizes);
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(K), hipMemcpyHostToDevice); (void)hipMemcpy(dn, sizes, sizeof sizes, hipMemcpyHostToDevice);
  k_sort<K><<<nb, kTPB>>>(d, dn, cap);
  (void)hipMemcpy(h.data(), d, h.size() * sizeof(K), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; b++) {
    std::sort(ref.begin() + (size_t)b * cap, ref.begin() + (size_t)b * cap + sizes[b]);
    if (!std::equal(ref.begin() + (size_t)b * cap, ref.begin() + (size_t)(b + 1) * cap, h.begin() + (size_t)b * cap)) { printf("sort<%zu bytes> n=%d WRONG\n", sizeof(K), sizes[b]); bad++; }
  }
  return bad;
}
int main()
{
  uint32_t *d; (void)hipMalloc(&d, 6 * 64 * 4); k<<<1, 64>>>(d); uint32_t h[6 * 64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 6; m++) for (int l = 0; l < 64; l++) { uint32_t e = (uint32_t)((l ^ (1 << m)) * 3 + 7); if (h[m * 64 + l] != e) { if (bad < 10) printf("m=%d lane %d got %u want %u\n", 1 << m, l, h[m * 64 + l], e); bad++; } }
  printf("lane_xor: bad=%d\n", bad);
  { int *ds; (void)hipMalloc(&ds, 512 * 4); k_scan<<<1, 256>>>(ds); int hs[512]; (void)hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost);
    int sb = 0; for (int i = 0; i < 256; i++) if (hs[i] != hs[256 + i]) { if (sb < 10) printf("scan lane %d: dpp %d shuffle %d\n", i, hs[i], hs[256 + i]); sb++; }
    printf("wave_incl_scan_dpp: bad=%d\n", sb); bad += sb; }
  const int b32 = check_sort<uint32_t>(), b64 = check_sort<uint64_t>();
  printf("block_sort: wrong size classes u32=%d u64=%d\n", b32, b64);
  return (bad | b32 | b64) != 0;
}
