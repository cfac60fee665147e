// synth.hpp — deterministic synthetic genomes, generated directly into 2-bit packed device memory.
// Clustered population after SURVEY.md §8d: genome g belongs to cluster g/20; member m = g%20 is the cluster's
// ancestor with i.i.d. substitutions at rate kRatePermille[m]/1000.  Counter-based (splitmix64 finaliser), so any
// base of any genome is a pure function of (seed, genome, position); oracle/ani_oracle.c:orc_synth_genome computes
// the same function on the CPU (tests compare them byte for byte).  `variant` re-draws the substitutions of every
// member while keeping the cluster ancestors: variant v of genome g is an independent descendant with the same rate.
#pragma once
#include "common.hpp"

namespace ani {

__host__ __device__ __forceinline__ uint64_t sm64_fin(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t sm64_stream(uint64_t key, uint64_t p) { return sm64_fin(key + (p + 1) * 0x9E3779B97F4A7C15ULL); }

__host__ __device__ __forceinline__ uint32_t synth_rate_permille(int m)
{
  // 0, .5%, 1% ... 25%  
  return m == 0 ? 0 : m == 1 ? 5 : m <= 16 ? (uint32_t)(m - 1) * 10 : m == 17 ? 170 : m == 18 ? 200 : 250;
}

__host__ __device__ __forceinline__ uint32_t synth_base(uint64_t keyAnc, uint64_t keyMut, uint32_t thr, uint64_t p)
{
  uint32_t b = (uint32_t)(sm64_stream(keyAnc, p) >> 62);
  const uint64_t u = sm64_stream(keyMut, p);
  if ((uint32_t)(u >> 40) < thr) {
    const uint32_t sub = 1 + (uint32_t)((((u >> 8) & 0xFFFF) * 3) >> 16);
    b = (b + sub) & 3;
  }
  return b;
}

// member m of a cluster of `clusterSize` genomes -> index into the 20 divergence rates: the first 20 members as in SURVEY.md section 8d;
// larger clusters (a species-dense database: --cluster-size of bench.py) cycle through the 19 non-zero rates, so that no two
// members are identical copies
__host__ __device__ __forceinline__ int synth_rate_index(int m) { return m < 20 ? m : 1 + (m - 20) % 19; }

static __global__ void k_synth_packed(uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, int32_t clusterSize,
    uint32_t *__restrict__ out)
{
  const int32_t wordsPerGenome = (genomeLen + 15) >> 4;
  const long long total = (long long)nGenomes * wordsPerGenome;
  const uint64_t root = sm64_fin(seed);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int32_t gi = (int32_t)(i / wordsPerGenome), wi = (int32_t)(i % wordsPerGenome);
    const int32_t g = firstGenomeId + gi;
    const uint64_t keyAnc = sm64_fin(root + 2 * (uint64_t)(g / clusterSize));
    const uint64_t keyMut = sm64_fin(root + 2 * (uint64_t)g + 1) ^ sm64_fin(variant);   // variant 0 leaves the key unchanged
    const uint32_t thr = (uint32_t)(((uint64_t)synth_rate_permille(synth_rate_index(g % clusterSize)) * 16777216ULL + 500ULL) / 1000ULL);
    uint32_t word = 0;
    for (int j = 0; j < 16; j++) {
      const int32_t p = wi * 16 + j;
      if (p < genomeLen) word |= synth_base(keyAnc, keyMut, thr, (uint64_t)p) << (2 * j);
    }
    out[i] = word;
  }
}

}  // namespace ani
