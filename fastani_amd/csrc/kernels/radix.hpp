// radix.hpp — stable LSD radix sort for gfx950, hand-written ("onesweep": one histogram read of the keys for all passes, then one
// read + one write of the data per 8-bit pass, tile prefixes by decoupled look-back).
//
// What it sorts on this path:
//   * the reference index: minimizer records (hash, seqId, wpos) in position order -> (hash, seqId<<32|wpos) in hash order, every
//     hash's occurrences still in (seqId, wpos) order — the flat form of filling minimizerPosLookupIndex
//     (src/map/include/winSketch.hpp:181-193: `minimizerPosLookupIndex[e.hash].push_back(...)` in position order).  The first pass
//     reads the 12-byte records where they lie (RecordSrc) and the histogram pass writes the position-ordered SoA arrays on the way,
//     so there is no separate split kernel and no (key, value) copy of the input;
//   * the seed hits of the batched L1 path, all fragments of a group at once (src/map/include/computeMap.hpp:320 `std::sort`);
//   * small orderings (fragment processing order, foreign mapping lists for cgi::computeCGI).
//
// A pass, per tile of 6144 keys (512 threads x 12 keys, wave-striped so that memory order = (wave, round, lane) order):
//   1. digit of every key; its rank among the EARLIER keys of the tile with the same digit (stability): inside a wave by a
//      match-any over 8 ballots (lanes with the same digit form a group; rank = group members in lower lanes), across rounds and
//      waves by per-wave digit counters in LDS;
//   2. thread d owns digit d: tile count, exclusive prefix over the earlier tiles by decoupled look-back (status word per (tile,
//      digit): flag | count; a tile publishes its aggregate at once and its inclusive prefix as soon as it knows it);
//   3. keys, then values, are put into tile order in LDS and written out so that consecutive threads write consecutive addresses
//      of one digit's output run.
// Tile ids are handed out by an atomic counter, so a tile only ever waits for tiles that already run (forward progress); a spin
// bound turns a scheduling surprise into an error flag instead of a hang.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace ani {

constexpr int kRadixBits = 8;
constexpr int kRadixDigits = 1 << kRadixBits;
#ifndef ANI_RADIX_TPB
#define ANI_RADIX_TPB 512
#define ANI_RADIX_KPT 12
#endif
constexpr int kRadixTPB = ANI_RADIX_TPB;                 // threads per workgroup of a pass
constexpr int kRadixKPT = ANI_RADIX_KPT;                 // keys per thread
constexpr int kRadixTile = kRadixTPB * kRadixKPT;        // 6144 keys per tile
constexpr int kRadixMaxPasses = 8;
static_assert(kRadixDigits == kTPB, "thread d owns digit d");

constexpr unsigned long long kRadixFlagAgg = 1ull << 62, kRadixFlagPrefix = 2ull << 62, kRadixValueMask = (1ull << 62) - 1;
constexpr unsigned kRadixSpinLimit = 1u << 24;          // ~seconds; a healthy look-back needs a handful of polls

struct RadixNoVal {};

// digit of a key for the pass that starts at bit `shift`; bits at and above endBit are not part of the sort key
template <class KeyT> __host__ __device__ __forceinline__ uint32_t radix_digit(KeyT k, int shift, int endBit)
{
  const int bits = endBit - shift < kRadixBits ? endBit - shift : kRadixBits;
  return (uint32_t)(k >> shift) & ((1u << bits) - 1u);
}

// ---- sources of pass 0 ----
// plain arrays (every pass after the first reads these)
template <class KeyT, class ValT> struct ArraySrc {
  const KeyT *keys; const ValT *vals;
  __device__ __forceinline__ KeyT key(uint64_t i) const { return keys[i]; }
  __device__ __forceinline__ ValT val(uint64_t i) const { return vals[i]; }
};
template <class KeyT> struct ArraySrc<KeyT, RadixNoVal> {
  const KeyT *keys; const RadixNoVal *vals;
  __device__ __forceinline__ KeyT key(uint64_t i) const { return keys[i]; }
  __device__ __forceinline__ RadixNoVal val(uint64_t) const { return RadixNoVal(); }
};
// the position-ordered SoA arrays of an index chunk (written by the histogram read of the records): key = hash, value = seqId << 32 | wpos
struct SoaSrc {
  const uint32_t *mHash; const int32_t *mSeq, *mWpos;
  __device__ __forceinline__ uint32_t key(uint64_t i) const { return mHash[i]; }
  __device__ __forceinline__ uint64_t val(uint64_t i) const { return ((uint64_t)(uint32_t)mSeq[i] << 32) | (uint32_t)mWpos[i]; }
};
// 12-byte minimizer records (skch::MinimizerInfo, base_types.hpp:22-53) with set-global seqIds; the index chunk's seqIds are local
struct RecordSrc {
  const uint32_t *rec; uint32_t seqBase;
  __device__ __forceinline__ uint32_t key(uint64_t i) const { return rec[3 * i]; }
  __device__ __forceinline__ uint64_t val(uint64_t i) const { return ((uint64_t)(rec[3 * i + 1] - seqBase) << 32) | rec[3 * i + 2]; }
};

// number of set bits of the 64-bit mask (lo, hi) in the lanes below this one
__device__ __forceinline__ uint32_t radix_mbcnt(uint32_t lo, uint32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
#else
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long m = (((unsigned long long)hi << 32) | lo) & ((1ull << lane) - 1ull);
  return (uint32_t)__popcll(m);
#endif
}

// ---- agent-scope loads / stores of the look-back status words (cross-XCD visibility: these go past the per-XCD L2).  RELAXED: a
// status word carries everything the reader needs (flag and count in one 64-bit value), nothing else is published through it — and
// acquire / release at agent scope would invalidate / write back the XCD's caches on every poll (measured: 605 ms instead of ~12 for
// 4 x 10^8 records, profiles/r04a_ubench_radix.txt) ----
__device__ __forceinline__ unsigned long long radix_status_load(const unsigned long long *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *(const volatile unsigned long long *)p;
#endif
}
__device__ __forceinline__ void radix_status_store(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *(volatile unsigned long long *)p = v;
#endif
}

// Digit histograms of every pass in ONE read of the keys: hist[pass * 256 + digit] (64-bit, zeroed by the host).  The index build
// passes `soa` = true: the same read writes the position-ordered SoA arrays of the chunk (mHash / mSeq / mWpos at `soaBase + i`).
template <class KeyT, class Src>
static __global__ __launch_bounds__(kTPB) void k_radix_histogram(Src src, uint64_t n, int beginBit, int endBit, int nPasses,
    unsigned long long *__restrict__ hist,
                                                          uint32_t *__restrict__ mHash, int32_t *__restrict__ mSeq, int32_t *__restrict__ mWpos)
{
  // one copy of the counters per wave: the top digit of minimizer hashes takes a few dozen values, and 64 lanes adding to a handful of
  // LDS addresses serialise
  __shared__ unsigned int h[(kTPB / kWave) * kRadixMaxPasses * kRadixDigits];
  const int wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (kTPB / kWave) * nPasses * kRadixDigits; i += kTPB) h[i] = 0;
  block_barrier();
  unsigned int *mine = h + wv * nPasses * kRadixDigits;
  for (uint64_t i = (uint64_t)blockIdx.x * kTPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kTPB) {
    const KeyT k = src.key(i);
    if (mHash) {
      if constexpr (std::is_same<Src, RecordSrc>::value) {
        const uint64_t v = src.val(i);
        mHash[i] = (uint32_t)k; mSeq[i] = (int32_t)(v >> 32); mWpos[i] = (int32_t)(uint32_t)v;
      }
    }
    // pass p sorts the bits from beginBit + 8 p
    for (int p = 0; p < nPasses; p++) atomicAdd(&mine[p * kRadixDigits + (int)radix_digit(k, beginBit + p * kRadixBits, endBit)], 1u);
  }
  block_barrier();
  for (int i = threadIdx.x; i < nPasses * kRadixDigits; i += kTPB) {
    unsigned int c = 0;
    for (int w = 0; w < kTPB / kWave; w++) c += h[w * nPasses * kRadixDigits + i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}

// exclusive scan of every pass's 256 digit counts, in place (one workgroup, thread d = digit d)
static __global__ __launch_bounds__(kTPB) void k_radix_scan(unsigned long long *__restrict__ hist, int nPasses)
{
  __shared__ unsigned long long part[kTPB / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  for (int p = 0; p < nPasses; p++) {
    const unsigned long long v = hist[p * kRadixDigits + threadIdx.x];
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) { const unsigned long long o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    block_barrier();
    if (lane == kWave - 1) part[wv] = incl;
    block_barrier();
    unsigned long long base = 0;
    for (int i = 0; i < wv; i++) base += part[i];
    hist[p * kRadixDigits + threadIdx.x] = base + incl - v;
  }
}

// One pass over `n` elements that form the tiles [tileBase, tileBase + ceil(n / kRadixTile)) of the whole sequence (kRadixTile = 6144 keys) (the index chunk's
// records come as several buffers: one launch per buffer, the look-back runs across them).
//   digitBase  [256] exclusive digit offsets of this pass (k_radix_scan)
//   status     [(all tiles) * 256], zeroed;  tileCounter zeroed per launch;  errFlag: set when a look-back gave up
// 512 threads x 12 keys: the tile's life is a string of latencies (key loads, one LDS round trip per ranked key, the look-back's
// round trips past the L2, value loads) and what hides them is waves — 8 per workgroup; LDS per workgroup = the staging tile (6144 x max(key, value)
// bytes: 48 KiB for the index's u32 / u64 pass, 24 KiB for u32-only) + 8 KiB of per-wave digit counters + 1 KiB of run bases ≈ 57 KiB for
// the index pass, i.e. 2 workgroups = 16 waves per CU of 160 KiB — not instruction-level tricks: 256 x 16 (the same tile, 105 VGPRs, 16 waves per CU) measured
// 4.7 ms per pass of 4 x 10^8
// records against 2.7 ms for rocPRIM's onesweep (profiles/r04c_radix_kernel_stats.csv).
template <class KeyT, class ValT, class Src>
static __global__ __launch_bounds__(kRadixTPB) void k_radix_pass(Src src, KeyT *__restrict__ keysOut, ValT *__restrict__ valsOut, uint64_t n, int shift,
    int endBit,
                                                          const unsigned long long *__restrict__ digitBase, unsigned long long *__restrict__ status,
                                                          unsigned int *__restrict__ tileCounter, uint32_t tileBase, unsigned int *__restrict__ errFlag)
{
  constexpr bool kHasVal = !std::is_same<ValT, RadixNoVal>::value;
  constexpr int kValBytes = kHasVal ? (int)sizeof(ValT) : 0;
  constexpr int kStageBytes = kRadixTile * ((int)sizeof(KeyT) > kValBytes ? (int)sizeof(KeyT) : kValBytes);
  constexpr int kWaves = kRadixTPB / kWave;
  __shared__ __attribute__((aligned(16))) unsigned char stage[kStageBytes];
  __shared__ unsigned int wc[kWaves * kRadixDigits];                // per-wave digit counters, then each wave's base inside the tile
  // global position of local position 0 of each digit's run, minus that local position (mod 2^32: n < 2^32)
  __shared__ uint32_t gbase[kRadixDigits];
  __shared__ int ws[32];
  __shared__ unsigned int sTile, sErr;
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = t >> 6;
  if (t == 0) { sTile = atomicAdd(tileCounter, 1u); sErr = 0; }
#pragma unroll
  for (int x = t; x < kWaves * kRadixDigits; x += kRadixTPB) wc[x] = 0;
  block_barrier();
  const uint32_t tile = sTile;
  const uint64_t base = (uint64_t)tile * kRadixTile;
  if (base >= n) return;                                             // (grid padded beyond the tiles: workgroup-uniform)
  unsigned int *myWc = wc + wv * kRadixDigits;
  KeyT key[kRadixKPT]; uint32_t rank[kRadixKPT];
  // ---- 1. digits and ranks inside the tile ----
  // (32-bit element numbers relative to the tile: the 64-bit part of an address is the tile's, once)
  const uint32_t nHere = (uint32_t)(n - base < (uint64_t)kRadixTile ? n - base : (uint64_t)kRadixTile);     // keys of this tile
  const uint32_t digitMask = (1u << (endBit - shift < kRadixBits ? endBit - shift : kRadixBits)) - 1u;
  const uint32_t e0 = (uint32_t)(wv * kRadixKPT) * kWave + lane;     // this thread's element j is e0 + j * 64
#pragma unroll
  for (int j = 0; j < kRadixKPT; j++) key[j] = e0 + j * kWave < nHere ? src.key(base + e0 + j * kWave) : (KeyT)0;
#pragma unroll
  for (int j = 0; j < kRadixKPT; j++) {
    const bool valid = e0 + j * kWave < nHere;
    const uint32_t d = (uint32_t)(key[j] >> shift) & digitMask;
    // match-any: the lanes of the wave that hold the same digit.  Per digit bit one vote; a lane's copy of the vote is XOR-ed with
    // its own bit (all ones / zero), and what is left set in the OR over the eight bits are the lanes that differ somewhere.
    // (The vector ALU is half of this kernel's time — 142 wave instructions per key in the first form, SQ counters
    // profiles/r04n_pmcradix_SQ_WAVES_summary.txt — so the votes are spelled in 32-bit halves.)
    const unsigned long long vmask = __ballot(valid);
    uint32_t dl = 0, dh = 0;
#pragma unroll
    for (int b = 0; b < kRadixBits; b++) {
      const uint32_t B = 0u - ((d >> b) & 1u);
      const unsigned long long m = __ballot(B != 0u);
      dl |= (uint32_t)m ^ B; dh |= (uint32_t)(m >> 32) ^ B;
    }
    const uint32_t pl = ~dl & (uint32_t)vmask, ph = ~dh & (uint32_t)(vmask >> 32);
    const uint32_t below = radix_mbcnt(pl, ph), cnt = (uint32_t)__popc(pl) + (uint32_t)__popc(ph);
    // the group's lowest lane moves the wave's counter on (one returning LDS add per group) and hands the old value to the others
    uint32_t pre = 0;
    if (valid && below == 0) pre = atomicAdd(&myWc[d], cnt);
    const int leader = pl ? __ffs(pl) - 1 : (ph ? __ffs(ph) + 31 : lane);
    pre = (uint32_t)__shfl((int)pre, leader);
    rank[j] = pre + below;
  }
  block_barrier();
  // ---- 2. thread d < 256: digit d's count in the tile, wave bases, local offset, look-back ----
  uint32_t total = 0;
  if (t < kRadixDigits) {
#pragma unroll
    for (int w = 0; w < kWaves; w++) { const uint32_t c = wc[w * kRadixDigits + t]; wc[w * kRadixDigits + t] = total; total += c; }
  }
  const uint64_t gt = (uint64_t)tileBase + tile;                     // tile number in the whole sequence
  unsigned long long *st = status + gt * kRadixDigits + (t & (kRadixDigits - 1));
  if (t < kRadixDigits) radix_status_store(st, (gt == 0 ? kRadixFlagPrefix : kRadixFlagAgg) | total);      // as early as it is known: later tiles wait for it
  // exclusive scan of the 256 digit counts (waves 0..3 hold them)
  int localOff, tileCount;
  {
    const int incl = wave_incl_scan((int)total);
    if (lane == kWave - 1) ws[wv] = incl;
    block_barrier();
    int b = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kRadixDigits / kWave; w++) { const int x = ws[w]; if (w < wv) b += x; tot += x; }
    localOff = b + incl - (int)total; tileCount = tot;
  }
  if (t < kRadixDigits) {
    unsigned long long excl = 0;
    if (gt != 0) {
#if defined(ANI_RADIX_FAKE_LOOKBACK)
      excl = gt * (unsigned long long)total;                        // timing experiment only (tools/ubench): plausible positions without waiting for anyone
      if (t == kRadixDigits - 1) excl = 0; else { const unsigned long long room = digitBase[t + 1] - digitBase[t] - total; if (excl > room) excl = room; }
#else
      const unsigned long long *pp = st - kRadixDigits;
      for (uint64_t p = gt; p-- > 0; pp -= kRadixDigits) {
        unsigned long long v = radix_status_load(pp);
        unsigned spins = 0;
        while ((v >> 62) == 0) {
          if (++spins > kRadixSpinLimit || ((spins & 1023u) == 0 && *(volatile unsigned int *)errFlag)) { sErr = 1; atomicOr(errFlag, 1u); break; }
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_s_sleep(1);
#endif
          v = radix_status_load(pp);
        }
        if ((v >> 62) == 0) break;
        excl += v & kRadixValueMask;
        if (v & kRadixFlagPrefix) break;
      }
#endif
      radix_status_store(st, kRadixFlagPrefix | (excl + total));
    }
    gbase[t] = (uint32_t)(digitBase[t] + excl) - (uint32_t)localOff;
    // local position of a key = keys of smaller digits (localOff) + same-digit keys of earlier waves (the wave's base) + its rank inside
    // the wave: the wave bases take the digit's local offset in place
#pragma unroll
    for (int w = 0; w < kWaves; w++) wc[w * kRadixDigits + t] += (uint32_t)localOff;
  }
  block_barrier();
  if (sErr) return;                                                  // a look-back gave up: write nothing (the host fails the call)
  // ---- 3. keys into tile order through LDS, out in runs; then the values along the same positions ----
  KeyT *sk = (KeyT *)stage;
  uint32_t lpos[kRadixKPT];
#pragma unroll
  for (int j = 0; j < kRadixKPT; j++) {
    const uint32_t d = (uint32_t)(key[j] >> shift) & digitMask;
    lpos[j] = myWc[d] + rank[j];
    if (e0 + j * kWave < nHere) sk[lpos[j]] = key[j];
  }
  [[maybe_unused]] ValT val[kHasVal ? kRadixKPT : 1];
  if constexpr (kHasVal) {                                           // the values' round trip to memory runs under the keys' way out
#pragma unroll
    for (int j = 0; j < kRadixKPT; j++) if (e0 + j * kWave < nHere) val[j] = src.val(base + e0 + j * kWave);
  }
  block_barrier();
  uint32_t opos[kRadixKPT];                                          // n < 2^32 (checked by the host)
#pragma unroll
  for (int j = 0; j < kRadixKPT; j++) {
    const int x = t + j * kRadixTPB;
    if (x < tileCount) {
      const KeyT k = sk[x];
      const uint32_t d = (uint32_t)(k >> shift) & digitMask;
      opos[j] = gbase[d] + (uint32_t)x;
      keysOut[opos[j]] = k;
    }
  }
  if constexpr (kHasVal) {
    ValT *sv = (ValT *)stage;
    block_barrier();                                                 // every key has left the stage
#pragma unroll
    for (int j = 0; j < kRadixKPT; j++) if (e0 + j * kWave < nHere) sv[lpos[j]] = val[j];
    block_barrier();
#pragma unroll
    for (int j = 0; j < kRadixKPT; j++) {
      const int x = t + j * kRadixTPB;
      if (x < tileCount) valsOut[opos[j]] = sv[x];
    }
  }
}

}  // namespace ani
