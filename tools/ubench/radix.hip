// radix.hip — the hand-written index sort (csrc/kernels/radix.hpp through csrc/sort_device.hip) against rocPRIM's radix_sort_pairs
// on the data it sorts in the benchmark: n minimizer records (hash = minimum of w = 24 uniform 32-bit values: the distribution of
// winnowed minimizer hashes; seqId / wpos increasing).  Checks the two outputs for equality (both sorts are stable) and times them.
// rocPRIM appears HERE ONLY, as the yardstick: the product library does not link it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../fastani_amd/csrc -o radix radix.hip && ./radix [n]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../fastani_amd/csrc/sort_device.hip"

__global__ void k_make_records(uint32_t *rec, uint64_t n, uint32_t perContig)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t h = 0xffffffffu;
    for (int k = 0; k < 24; k++) { x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; const uint32_t v = (uint32_t)(x >> 16); h = v < h ? v : h; }
    rec[3 * i] = h; rec[3 * i + 1] = (uint32_t)(i / perContig); rec[3 * i + 2] = (uint32_t)(i % perContig) * 12u;
  }
}
__global__ void k_split(const uint32_t *rec, uint64_t n, uint32_t *k, uint64_t *v)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { k[i] = rec[3 * i]; v[i] = ((uint64_t)rec[3 * i + 1] << 32) | rec[3 * i + 2]; }
}
__global__ void k_compare(const uint32_t *a, const uint32_t *b, const uint64_t *va, const uint64_t *vb, uint64_t n, unsigned long long *bad)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (a[i] != b[i] || va[i] != vb[i]) atomicAdd(bad, 1ull);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 400000000ull;
  uint32_t *rec, *mHash, *tmpK, *sHash, *rK, *rKo; int32_t *mSeq, *mWpos; uint64_t *tmpV, *sSW, *rV, *rVo; unsigned long long *bad;
  CK(hipMalloc(&rec, n * 12)); CK(hipMalloc(&mHash, n * 4)); CK(hipMalloc(&mSeq, n * 4)); CK(hipMalloc(&mWpos, n * 4));
  CK(hipMalloc(&tmpK, n * 4)); CK(hipMalloc(&tmpV, n * 8)); CK(hipMalloc(&sHash, n * 4)); CK(hipMalloc(&sSW, n * 8));
  CK(hipMalloc(&rK, n * 4)); CK(hipMalloc(&rV, n * 8)); CK(hipMalloc(&rKo, n * 4)); CK(hipMalloc(&rVo, n * 8)); CK(hipMalloc(&bad, 8));
  hipLaunchKernelGGL(k_make_records, dim3(8192), dim3(256), 0, 0, rec, n, 400000u);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  // ---- ours: three buffers of uneven size, like an index chunk built from several record parts
  const size_t cut1 = n / 3 + 777, cut2 = 2 * (n / 3) + 5;
  const void *pieces[3] = {rec, rec + 3 * cut1, rec + 3 * cut2}; const size_t cnt[3] = {cut1, cut2 - cut1, (size_t)n - cut2};
  size_t tb = 0; void *tmp = nullptr;
  int rc = ani_sort_index(pieces, cnt, 3, 0, n, mHash, mSeq, mWpos, tmpK, tmpV, sHash, sSW, nullptr, &tb, 0, nullptr, nullptr);
  CK(hipMalloc(&tmp, tb + 256));
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(a);
    rc = ani_sort_index(pieces, cnt, 3, 0, n, mHash, mSeq, mWpos, tmpK, tmpV, sHash, sSW, tmp, &tb, 0, nullptr, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    if (rc) { printf("ani_sort_index failed: %d\n", rc); return 1; }
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  printf("ani_sort_index   n = %llu : %8.3f ms  (workspace %.1f MB; histogram + SoA write + 4 passes; %.2f TB/s on 4 + 12 + 4 x 24 bytes per record)\n",
         (unsigned long long)n, best, tb / 1048576.0, (double)n * (16 + 96) / best / 1e9);
  // ---- rocPRIM on the split arrays (what round 3 did: k_index_split + radix_sort_pairs)
  size_t rtb = 0; void *rtmp = nullptr;
  CK(rocprim::radix_sort_pairs(nullptr, rtb, rK, rKo, rV, rVo, n, 0, 32, 0));
  CK(hipMalloc(&rtmp, rtb));
  float bestSplit = 1e9f, bestSort = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_split, dim3(8192), dim3(256), 0, 0, rec, n, rK, rV);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); bestSplit = ms < bestSplit ? ms : bestSplit;
    hipEventRecord(a);
    CK(rocprim::radix_sort_pairs(rtmp, rtb, rK, rKo, rV, rVo, n, 0, 32, 0));
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); bestSort = ms < bestSort ? ms : bestSort;
  }
  printf("rocPRIM pairs    n = %llu : %8.3f ms  (+ %.3f ms for the split into key / value arrays, without the SoA write)\n", (unsigned long long)n, bestSort, bestSplit);
  CK(hipMemset(bad, 0, 8));
  hipLaunchKernelGGL(k_compare, dim3(8192), dim3(256), 0, 0, sHash, rKo, sSW, rVo, n, bad);
  unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("outputs %s (%llu differing entries)\n", hb ? "DIFFER" : "identical", hb);
  // the SoA arrays
  std::vector<uint32_t> h0(8), r0(24); std::vector<int32_t> s0(8), w0(8);
  CK(hipMemcpy(h0.data(), mHash + cut1 - 4, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(s0.data(), mSeq + cut1 - 4, 32, hipMemcpyDeviceToHost));
  CK(hipMemcpy(w0.data(), mWpos + cut1 - 4, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r0.data(), rec + 3 * (cut1 - 4), 96, hipMemcpyDeviceToHost));
  bool soaOk = true;
  for (int i = 0; i < 8; i++) soaOk = soaOk && h0[i] == r0[3 * i] && (uint32_t)s0[i] == r0[3 * i + 1] && (uint32_t)w0[i] == r0[3 * i + 2];
  printf("SoA arrays across a buffer border: %s\n", soaOk ? "ok" : "WRONG");
  // ---- the small sorts
  for (uint64_t m : {1666000ull, 100000000ull}) {
    if (m > n) continue;
    uint64_t *kin = (uint64_t *)rV, *kout = (uint64_t *)rVo;
    size_t t2 = 0; void *tmp2 = nullptr;
    for (int bits : {11, 40}) {
      rc = ani_sort_keys_u64_bits(kin, kout, m, bits, nullptr, &t2, 0);
      CK(hipMalloc(&tmp2, t2 + 256));
      hipEventRecord(a);
      rc = ani_sort_keys_u64_bits(kin, kout, m, bits, tmp2, &t2, 0);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      size_t r2 = 0; void *rt2 = nullptr;
      CK(rocprim::radix_sort_keys(nullptr, r2, kin, (uint64_t *)sSW, m, 0, bits, 0));
      CK(hipMalloc(&rt2, r2));
      hipEventRecord(a);
      CK(rocprim::radix_sort_keys(rt2, r2, kin, (uint64_t *)sSW, m, 0, bits, 0));
      hipEventRecord(b); hipEventSynchronize(b);
      float ms2; hipEventElapsedTime(&ms2, a, b);
      CK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(k_compare, dim3(4096), dim3(256), 0, 0, (const uint32_t *)kout, (const uint32_t *)sSW, kout, (const uint64_t *)sSW, m, bad);
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("keys u64, %2d bits, n = %9llu : ours %7.3f ms (rc %d), rocPRIM %7.3f ms, %s\n", bits, (unsigned long long)m, ms, rc, ms2, hb ? "DIFFER" : "identical");
      hipFree(tmp2); hipFree(rt2);
    }
  }
  return 0;
}
