// l1.hpp — L1 stage: seed lookup and candidate regions, one workgroup per query fragment.
//
// Restates skch::Map::doL1Mapping (src/map/include/computeMap.hpp:252-304: probe every unique fragment hash in the
// lookup index and collect all (seqId,wpos) occurrences) and computeL1CandidateRegions (:313-354: sort the hits,
// slide a run of `minimumHits` consecutive hits, keep runs that stay on one contig and span < fragLen, merge
// overlapping candidates).  Stateless form, SURVEY.md App. A.3.
//
// Two passes.  k_l1_probe: one lane per sketch hash, the index chunk's probe table (index.hpp), no LDS.  k_l1<HLO,HCAP>: one workgroup per fragment with HLO <
// H <= HCAP seed hits,
// gather the hit runs into LDS as 64-bit (seqId<<32 | wpos) keys, bitonic sort in registers (common.hpp: block_sort), flag valid runs, compact, flag
// group heads by neighbour comparison (run starts/ends are non-decreasing, so "overlaps the previous candidate" only needs
// the previous valid run), scan, emit.  Two LDS classes (<= 2048 hits: 6 workgroups per CU; <= 4096: 3), the larger driven by a fragment list.  Fragments
// with more hits take the batched global-memory path: k_l1_big_offsets / _gather -> device radix sort -> k_l1_big_unpack / _candidates.
#pragma once
#include "common.hpp"
#include "index.hpp"

namespace ani {

constexpr int kL1MaxS = 2048;           // sketch hashes per fragment the LDS classes accept
// ... class S (its 4 KiB of scratch hold the per-hash hit offsets); a fragment with more goes to class M whatever its hit count
constexpr int kL1SmallMaxS = 1024;
constexpr int kFragHashCapL1 = 4096;    // = kFragHashCap (sketch.hpp): the most sketch hashes a fragment can have
// The LDS classes.  A class's hit array is a power of two long (block_sort pads to one); its last kL1LdsSpare entries are never hits —
// the class accepts 16 hits fewer — and hold the workgroup's scan scratch and two scalars: with them inside the array a class-S workgroup
// is 16 + 4 KiB and a class-M workgroup 32 + 8 KiB of LDS EXACTLY, i.e. 160 KiB / 40 KiB = four class-M workgroups per CU instead of
// three.  (The sort overwrites the spare entries with its padding when it runs on the full array; none of them is live across it.)
constexpr int kL1LdsSpare = 16;
constexpr int kL1HitCapSmall = 2048 - kL1LdsSpare;
// class S: 16 KiB hits + 4 KiB scratch, <= 72 registers -> 7 workgroups per CU.  The kernel's time follows the number of resident
// workgroups (a workgroup's life is a chain of latencies — probe results, hit runs, LDS round trips of the sort — and what hides them
// is other workgroups): 5 per CU 35.8 ms, 6 per CU 30.0 ms, 7 per CU 27.1 ms per benchmark step (profiles/r06f_l1_occupancy_ab.txt)
constexpr int kL1HitCapMid = 4096 - kL1LdsSpare;      // class M: 32 KiB + 8 KiB -> 4 workgroups per CU
constexpr int kL1HitCapMax = kL1HitCapMid;   // beyond: the batched global-memory path.  (A class L of 8192 hits — 96 KiB of LDS, one workgroup
                                             // per CU — existed until round 3: 0.43 us per fragment where the batched path takes 0.33, measured at
                                             // 493 k such fragments per step of the cluster-size-100 benchmark.)
static_assert(kL1HitCapMax <= kBlockSortMax, "block_sort (common.hpp) sorts at most kBlockSortMax keys");
constexpr int kL1FilterMinHits = 300;   // below this the sort is cheaper than the noise filter
// log2 of the occupancy counters per tiling: four bit arrays in the class's scratch (S: 4 x 1 KiB, M: 4 x 2 KiB)
template <int HCAP> __host__ __device__ constexpr int kL1FilterBits() { return HCAP == kL1HitCapSmall ? 13 : 14; }

// What the gather kernels need to know about a fragment, in ONE 16-byte load at the fragment's place in the processing order (written by
// k_l1_probe): the workgroup kernels are bound by the latencies of their short lives (kL1HitCapSmall above), and order -> fragment ->
// (sketch size, hits) -> (pool offset, minimumHits) were three dependent round trips to memory in front of the first probe result.
struct __attribute__((aligned(16))) L1FragDesc { int32_t f; uint32_t sm; int32_t H; uint32_t off; };      // sm = sketch size | minimumHits << 16
__device__ __forceinline__ L1FragDesc l1_frag_desc(const L1FragDesc *p)
{
  const uint4 v = *(const uint4 *)p;
  L1FragDesc d; d.f = (int32_t)v.x; d.sm = v.y; d.H = (int32_t)v.z; d.off = v.w;
  return d;
}

struct L1Args {
  const uint32_t *qPool; const uint32_t *fragOff; const int32_t *fragS; int32_t nFrag;
  const TableSlot *table; uint32_t tableSlots; const uint64_t *sSW; int bucketW; uint32_t nIndex;
  const int32_t *minHitsLUT; int32_t lutMaxS;
  int L;
  // striped pool (common.hpp: pool_take): capacity per stripe, cursors
  int32_t *candFrag, *candSeq, *candStart, *candEnd; uint32_t candCap; unsigned long long *candCount;
  uint32_t *fragCandOff; int32_t *fragCandCnt; int32_t *fragHits;
  uint32_t *probeFirst, *probeCnt;      // per sketch hash (aligned with qPool): occurrence run in the hash-sorted index
  int32_t *midList; unsigned int *midCount;       // fragments with kL1HitCapSmall < H <= kL1HitCapMid
  int32_t *bigList; unsigned int *bigCount;       // fragments beyond the LDS classes (s > kL1MaxS or H > kL1HitCapMax)
  unsigned int *overflowCount;                    // fragments with >= 2^31 seed hits (fragHits = -1): the call fails
  unsigned long long hitLimit;                    // ... 2^31 - 16 (lowered by tests: ANI_TEST_L1_HIT_LIMIT)
  unsigned long long *sumHits, *tinyCount, *smallCount;
  int filterShift;                      // log2 of the tile width of the noise filter: smallest power of two >= 2 * L
  const int32_t *fragOrder;             // processing order of the fragments (nullptr: ascending), see map_stage
  L1FragDesc *fragDesc;                 // [nFrag] by position in the processing order (k_l1_probe writes, the gather kernels read)
  int filterMinHits;                    // kL1FilterMinHits (ANI_TEST_L1_FILTER_MIN: test knob; 0 = always filter, large = never)
  int ldsHitCap;                        // fragments with more seed hits take the batched global-memory path (<= kL1HitCapMax; ANI_TEST_L1_LDS_MAX)
  int tinyPath;                         // fragments with <= 64 seed hits are finished by one wave (l1_tiny; ANI_TEST_L1_TINY=0 switches it off: A/B and tests)
};

// Counter indices of a hit's two tiles for the noise filter (k_l1): tiles of width 2^shift, the second tiling offset by half a tile.
// The xor-shift between the two multiplications matters: without it the index is (seqId * K) >> n for hits in tile 0 of consecutive
// contigs — a draft assembly's thousands of short contigs — and that sequence covers only half of the counters (2300 contigs:
// 1172 distinct indices instead of ~2140; a collision only keeps a hit that could have been dropped, but it kept 97 % of them).
template <int BITS>
__device__ __forceinline__ void l1_filter_tiles(uint64_t hv, int shift, uint32_t &ia, uint32_t &ib)
{
  const uint32_t wpos = (uint32_t)hv, sq = (uint32_t)(hv >> 32);
  uint32_t ha = sq * 0x9E3779B1u + (wpos >> shift), hb = sq * 0xC2B2AE3Du + ((wpos + (1u << (shift - 1))) >> shift);
  ha ^= ha >> 15; hb ^= hb >> 15;
  ia = (ha * 0x85EBCA77u) >> (32 - BITS); ib = (hb * 0x27D4EB2Fu) >> (32 - BITS);
}

// occurrences of hash h in the hash-ordered payload array: [first, first+cnt)
__device__ __forceinline__ void l1_probe(const L1Args &a, uint32_t h, uint32_t &first, uint32_t &cnt) { table_probe(a.table, a.bucketW, a.tableSlots, h,
    first, cnt); }

__device__ __forceinline__ int32_t hit_seq(uint64_t h) { return (int32_t)(h >> 32); }
__device__ __forceinline__ int32_t hit_wpos(uint64_t h) { return (int32_t)(uint32_t)h; }

// run starting at sorted hit a is a candidate (computeMap.hpp:332)
__device__ __forceinline__ bool l1_valid(const uint64_t *hits, int a, int m, int L)
{
  const uint64_t x = hits[a], y = hits[a + m - 1];
  return hit_seq(x) == hit_seq(y) && hit_wpos(y) - hit_wpos(x) < L;
}
// j-th valid run opens a new candidate (computeMap.hpp:342-350, negated)
template <class VT>
__device__ __forceinline__ bool l1_head(const uint64_t *hits, const VT *V, int j, int m, int L)
{
  if (j == 0) return true;
  const uint64_t x = hits[V[j]], px = hits[V[j - 1]];
  int32_t start = hit_wpos(hits[V[j] + m - 1]) - L + 1; if (start < 0) start = 0;
  return hit_seq(x) != hit_seq(px) || hit_wpos(px) < start;
}

// Sorted hits -> candidate regions of one fragment (computeMap.hpp:313-354).  hits/V may live in LDS or in global memory.
// VT: type of the run indices in V (uint16_t in class S: its hits number <= 2048)
template <class VT>
__device__ inline void l1_emit_candidates(const L1Args &a, int f, int s, int H, int m /* minimumHits of s, :301 */, const uint64_t *hits, VT *V,
                                          int *ws, unsigned long long *sBasePtr)
{
  const int t = threadIdx.x;
  if (m < 1) m = 1;                                                     // :316
  const int nA = H - m + 1;
  int nG = 0;
  if (nA > 0) {
    // valid runs, compacted in order
    int per = (nA + kTPB - 1) / kTPB;
    int lo = t * per, hi = lo + per < nA ? lo + per : nA;
    int c = 0;
    for (int x = lo; x < hi; x++) c += l1_valid(hits, x, m, a.L);
    int nv; int r = block_excl_scan(c, ws, &nv);
    for (int x = lo; x < hi; x++) if (l1_valid(hits, x, m, a.L)) V[r++] = (VT)x;
    block_barrier_mem();
    // candidate heads
    per = (nv + kTPB - 1) / kTPB;
    lo = t * per; hi = lo + per < nv ? lo + per : nv;
    c = 0;
    for (int j = lo; j < hi; j++) c += l1_head(hits, V, j, m, a.L);
    int g = block_excl_scan(c, ws, &nG);
    if (t == 0) *sBasePtr = pool_take(a.candCount, a.candCap, (unsigned long long)nG);
    block_barrier_mem();
    const unsigned long long base = *sBasePtr & ~kPoolOverflowBit;
    if (!(*sBasePtr & kPoolOverflowBit)) {
      for (int j = lo; j < hi; j++) {
        const bool head = l1_head(hits, V, j, m, a.L);
        if (head) g++;
        const unsigned long long slot = base + (unsigned long long)(g - 1);     // group of run j
        if (head) {
          int32_t start = hit_wpos(hits[V[j] + m - 1]) - a.L + 1; if (start < 0) start = 0;   // :335
          a.candFrag[slot] = f; a.candSeq[slot] = hit_seq(hits[V[j]]); a.candStart[slot] = start;
        }
        if (j == nv - 1 || l1_head(hits, V, j + 1, m, a.L)) a.candEnd[slot] = hit_wpos(hits[V[j]]);   // :336,:347
      }
    }
    if (t == 0) a.fragCandOff[f] = (uint32_t)base;
  } else if (t == 0) a.fragCandOff[f] = 0;
  if (t == 0) a.fragCandCnt[f] = nG;
}

// The same over hits that lie in GLOBAL memory (the batched path of fragments beyond the LDS classes: 10^4..10^6 hits per
// fragment).  l1_emit_candidates gives every thread a contiguous slice of the list, which is right for LDS and ruinous for global
// memory (the 64 lanes of a wave touch 64 different cache lines per step: k_l1_big_candidates took 2.9 of the 5.3 seconds of the
// cluster-size-500 benchmark step that way).  Here the list is walked in chunks of 256 consecutive entries, one per thread —
// coalesced — with one workgroup scan per chunk; three walks: valid runs (compacted into V in order), candidate heads (counted),
// emission.
__device__ inline void l1_emit_candidates_stream(const L1Args &a, int f, int H, int m, const uint64_t *__restrict__ hits, int *__restrict__ V,
                                                 int *ws, unsigned long long *sBasePtr)
{
  const int t = threadIdx.x;
  if (m < 1) m = 1;                                                     // :316
  const int nA = H - m + 1;
  int nG = 0;
  if (nA > 0) {
    int nv = 0;
    for (int base = 0; base < nA; base += kTPB) {                       // valid runs, in order
      const int x = base + t;
      const int c = x < nA ? (int)l1_valid(hits, x, m, a.L) : 0;
      int tot; const int r = block_excl_scan(c, ws, &tot);
      if (c) V[nv + r] = x;
      nv += tot;
    }
    block_barrier_mem();
    for (int base = 0; base < nv; base += kTPB) {                       // candidate heads
      const int j = base + t;
      const int c = j < nv ? (int)l1_head(hits, V, j, m, a.L) : 0;
      int tot; (void)block_excl_scan(c, ws, &tot);
      nG += tot;
    }
    if (t == 0) *sBasePtr = pool_take(a.candCount, a.candCap, (unsigned long long)nG);
    block_barrier_mem();
    const unsigned long long base0 = *sBasePtr & ~kPoolOverflowBit;
    if (!(*sBasePtr & kPoolOverflowBit)) {
      int gBefore = 0;                                                  // heads in the chunks before this one
      for (int base = 0; base < nv; base += kTPB) {
        const int j = base + t;
        const bool in = j < nv;
        const bool head = in && l1_head(hits, V, j, m, a.L);
        int tot; const int g = gBefore + block_excl_scan((int)head, ws, &tot) + (head ? 1 : 0);    // group of run j, 1-based
        if (in) {
          const unsigned long long slot = base0 + (unsigned long long)(g - 1);
          if (head) {
            int32_t start = hit_wpos(hits[V[j] + m - 1]) - a.L + 1; if (start < 0) start = 0;   // :335
            a.candFrag[slot] = f; a.candSeq[slot] = hit_seq(hits[V[j]]); a.candStart[slot] = start;
          }
          if (j == nv - 1 || l1_head(hits, V, j + 1, m, a.L)) a.candEnd[slot] = hit_wpos(hits[V[j]]);   // :336,:347
        }
        gBefore += tot;
      }
    }
    if (t == 0) a.fragCandOff[f] = (uint32_t)base0;
  } else if (t == 0) a.fragCandOff[f] = 0;
  if (t == 0) a.fragCandCnt[f] = nG;
}

// Pass 1: probes only.  No LDS, one lane per query hash.  A probe is a chain of dependent loads (sketch hash -> table slot), and
// the kernel is bound by that latency, not by bandwidth: measured, one cache line per probe instead of three bought nothing as long
// as a lane had a single probe in flight.  So a workgroup takes kL1ProbeFrags fragments at once and every lane runs that many
// independent chains side by side.  Writes (first, cnt) per sketch hash and H per fragment.
constexpr int kL1ProbeFrags = 4;      // (2 and 8 measure the same 10.3-11.0 ms per step: with 65 GB of table lines fetched per step the kernel runs at 6 TB/s)
static __global__ __launch_bounds__(kTPB) void k_l1_probe(L1Args a)
{
  __shared__ unsigned long long wsum[kTPB / kWave];
  const int i0 = xcd_item(blockIdx.x, gridDim.x) * kL1ProbeFrags;
  if (i0 >= a.nFrag) return;
  int fq[kL1ProbeFrags], s[kL1ProbeFrags]; uint32_t off[kL1ProbeFrags]; unsigned long long c[kL1ProbeFrags];
  int smax = 0;
#pragma unroll
  for (int q = 0; q < kL1ProbeFrags; q++) {
    fq[q] = i0 + q < a.nFrag ? (a.fragOrder ? a.fragOrder[i0 + q] : i0 + q) : -1;
    s[q] = fq[q] >= 0 ? a.fragS[fq[q]] : 0; if (s[q] < 0) s[q] = 0;
    off[q] = fq[q] >= 0 ? a.fragOff[fq[q]] : 0u;
    c[q] = 0;
    smax = s[q] > smax ? s[q] : smax;
  }
  for (int i = threadIdx.x; i < smax; i += kTPB) {
    uint32_t h[kL1ProbeFrags], fi[kL1ProbeFrags], cn[kL1ProbeFrags];
#pragma unroll
    for (int q = 0; q < kL1ProbeFrags; q++) h[q] = i < s[q] ? a.qPool[off[q] + i] : 0u;
#pragma unroll
    for (int q = 0; q < kL1ProbeFrags; q++) { fi[q] = 0; cn[q] = 0; if (i < s[q]) l1_probe(a, h[q], fi[q], cn[q]); }
#pragma unroll
    for (int q = 0; q < kL1ProbeFrags; q++)
      if (i < s[q]) { a.probeFirst[off[q] + i] = fi[q]; a.probeCnt[off[q] + i] = cn[q]; c[q] += cn[q]; }
  }
#pragma unroll
  for (int q = 0; q < kL1ProbeFrags; q++) {
    const int f = fq[q];
    if (f < 0) break;                               // workgroup-uniform
    // the fragment's seed hits: summed in 64 bits (a hash of a repetitive reference alone can have 10^9 occurrences); hit counts and
    // offsets are 32-bit from here on, so a fragment with >= 2^31 hits is marked (-1) and the call fails with ANI_ERR_LIMIT
    unsigned long long H64 = c[q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) H64 += __shfl_down(H64, d);
    block_barrier();
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = H64;
    block_barrier();
    if (threadIdx.x == 0) {
      H64 = 0;
      for (int w = 0; w < kTPB / kWave; w++) H64 += wsum[w];
      const bool tooMany = H64 > a.hitLimit;
      int H = tooMany ? -1 : (int)H64;
      // Fewer seed hits than a candidate region needs (minimumHits, computeMap.hpp:316-336) find nothing: such a fragment — a third
      // of the visits of a fragment set to a foreign reference shard or index chunk, which collect two or three chance hits — counts
      // as one without hits from here on.  (Any fragment with s <= kL1MaxS, whatever path its hit count would have sent it to: below
      // minimumHits hits there is no candidate on any path.  sumHits / seedHits still counts the dropped hits: they were probed.)
      int m = (s[q] > 0 && s[q] <= a.lutMaxS) ? a.minHitsLUT[s[q]] : 1;
      if (H > 0 && s[q] <= kL1MaxS && H < (m < 1 ? 1 : m)) H = 0;
      a.fragHits[f] = H;
      { L1FragDesc d; d.f = f; d.sm = (uint32_t)s[q] | ((uint32_t)(m < 0 ? 0 : (m > 0xffff ? 0xffff : m)) << 16); d.H = H; d.off = off[q];
        a.fragDesc[i0 + q] = d; }
      if (H64) atomicAdd(stat_slot(a.sumHits), H64);
      if (tooMany) atomicAdd(a.overflowCount, 1u);
      else if (s[q] > 0) {
        // fragments of the larger LDS classes are listed so that their 48 / 96 KiB workgroups are only launched for them
        if (s[q] <= kL1MaxS && H <= a.ldsHitCap) {
          // (a long sketch with a handful of hits: the wave kernel)
          // (positions in the processing order: fragDesc)
          if (H > kL1HitCapSmall || (s[q] > kL1SmallMaxS && !(H <= 4 * kWave && a.tinyPath))) a.midList[atomicAdd(a.midCount, 1u)] = i0 + q;
          // (striped statistics counters: the host launches k_l1_tiny if there are any,
          else if (H > 0 && H <= 4 * kWave && a.tinyPath) atomicAdd(stat_slot(a.tinyCount), 1ull);
          // and k_l1<0, 2048> over a list instead of over every fragment if there are few)
          else if (H > 0) atomicAdd(stat_slot(a.smallCount), 1ull);
        } else a.bigList[atomicAdd(a.bigCount, 1u)] = f;         // beyond every LDS class: global-memory path
      }
    }
  }
}

// Fragments with at most 256 seed hits: ONE WAVE per fragment, four fragments per workgroup, 3 KiB of LDS each — no workgroup
// barrier, no scan over LDS, one key per lane.  This is what a fragment looks like against an index that holds no relative of its
// genome — every visit of a fragment to a foreign reference shard of a multi-GPU run, to a foreign index chunk of a large database:
// ~240 probes, a few chance hits (minimizer hashes crowd the low end of the 32-bit range), rarely a candidate.  Such a fragment is a
// chain of five dependent round trips to memory and nothing else, so what counts is how many of them a CU has in flight: 32 here,
// 6 in the workgroup kernel (24 KiB of LDS per fragment), which took 17.5 ms for the 1.67 M fragment visits of one rank of an 8-GPU
// job, 1.46 M of them of this kind (profiles/r04g_bench_sim8.json.log).
//   gather: lane l takes the sketch hashes l, l + 64, ...; the order of the hits does not matter (they are sorted next), so a wave
//           scan of the lanes' hit counts places them;  sort: 1, 2 or 4 keys per lane in registers over the lane-exchange network
//           (common.hpp: wave_sort_regs);  runs / heads / emission: computeMap.hpp:313-354 in rounds of 64 runs, ballots instead of scans.
constexpr int kL1HitCapTiny = 4 * kWave;           // 256 hits: four keys per lane
constexpr int kL1TinyFrags = kTPB / kWave;
// sort the H <= 64 * KPT hits in hits[] (LDS of this wave; padded with ~0 up to 64 * KPT) through registers
template <int KPT> __device__ __forceinline__ void l1_tiny_sort(uint64_t *hits, int H)
{
  const int lane = threadIdx.x & (kWave - 1);
  uint64_t k[KPT];
#pragma unroll
  // any assignment of the unordered keys will do: the conflict-free one
  for (int r = 0; r < KPT; r++) { const int x = r * kWave + lane; k[r] = x < H ? hits[x] : ~0ull; }
  ANI_WAVE_SYNC();
  wave_sort_regs<uint64_t, KPT>(k);                  // :320
#pragma unroll
  for (int r = 0; r < KPT; r++) hits[lane * KPT + r] = k[r];
  ANI_WAVE_SYNC();
}
__device__ __forceinline__ void l1_tiny(const L1Args &a, int f, int s, int H, uint32_t off, int m, uint64_t *hits, int *V)
{
  const int lane = threadIdx.x & (kWave - 1);
  if (m < 1) m = 1;                                  // :316
  int mine = 0;
  for (int i = lane; i < s; i += kWave) mine += (int)a.probeCnt[off + i];
  int o = wave_incl_scan(mine) - mine;
  for (int i = lane; i < s; i += kWave) {
    const int c = (int)a.probeCnt[off + i];
    if (c) { const uint32_t fi = a.probeFirst[off + i]; for (int x = 0; x < c; x++) hits[o + x] = a.sSW[fi + x]; o += c; }
  }
  ANI_WAVE_SYNC();
  if (H <= kWave) l1_tiny_sort<1>(hits, H);          // wave-uniform
  else if (H <= 2 * kWave) l1_tiny_sort<2>(hits, H);
  else l1_tiny_sort<4>(hits, H);
  // runs of m hits on one contig within < L (computeMap.hpp:326-336), positions x = r * 64 + lane, compacted in order
  const int nA = H - m + 1;
  int nG = 0;
  if (nA > 0) {
    int nv = 0;
    for (int r = 0; r * kWave < nA; r++) {
      const int x = r * kWave + lane;
      const bool valid = x < nA && l1_valid(hits, x, m, a.L);
      const unsigned long long vm = __ballot(valid);
      if (valid) V[nv + __popcll(vm & ((1ull << lane) - 1ull))] = x;
      nv += __popcll(vm);
    }
    ANI_WAVE_SYNC();
    // candidate heads (:342-350): counted, then emitted with the group number of every run
    for (int r = 0; r * kWave < nv; r++) {
      const int j = r * kWave + lane;
      nG += __popcll(__ballot(j < nv && l1_head(hits, V, j, m, a.L)));
    }
    unsigned long long take = 0;
    if (lane == 0) take = pool_take(a.candCount, a.candCap, (unsigned long long)nG);
    take = __shfl(take, 0);
    const unsigned long long base = take & ~kPoolOverflowBit;
    if (!(take & kPoolOverflowBit)) {
      int gBefore = 0;                                                            // heads in the rounds before this one
      for (int r = 0; r * kWave < nv; r++) {
        const int j = r * kWave + lane;
        const bool in = j < nv;
        const bool head = in && l1_head(hits, V, j, m, a.L);
        const unsigned long long hm = __ballot(head);
        if (in) {
          const int g = gBefore + __popcll(hm & ((2ull << lane) - 1ull));         // group of run j, 1-based
          const unsigned long long slot = base + (unsigned long long)(g - 1);
          if (head) {
            int32_t start = hit_wpos(hits[V[j] + m - 1]) - a.L + 1; if (start < 0) start = 0;   // :335
            a.candFrag[slot] = f; a.candSeq[slot] = hit_seq(hits[V[j]]); a.candStart[slot] = start;
          }
          if (j == nv - 1 || l1_head(hits, V, j + 1, m, a.L)) a.candEnd[slot] = hit_wpos(hits[V[j]]);   // :336,:347
        }
        gBefore += __popcll(hm);
      }
    }
    if (lane == 0) a.fragCandOff[f] = (uint32_t)base;
  } else if (lane == 0) a.fragCandOff[f] = 0;
  if (lane == 0) a.fragCandCnt[f] = nG;
}
static __global__ __launch_bounds__(kTPB) void k_l1_tiny(L1Args a)
{
  __shared__ uint64_t hits[kL1TinyFrags][kL1HitCapTiny];
  __shared__ int V[kL1TinyFrags][kL1HitCapTiny];
  const int wv = wave_uniform((int32_t)(threadIdx.x >> 6));                  // the wave's fragment, its counts and offsets: scalar registers
  const int i = xcd_item(blockIdx.x, gridDim.x) * kL1TinyFrags + wv;
  if (i >= a.nFrag) return;
  const L1FragDesc d = l1_frag_desc(a.fragDesc + i);
  const int f = wave_uniform(d.f), s = wave_uniform((int32_t)(d.sm & 0xffffu)), H = wave_uniform(d.H);
  // the workgroup classes / the batched path (same class predicate as k_l1_probe: H > ldsHitCap is bigList's) / nothing to do (k_l1<0, 2048> or k_l1_list
  // writes the zero counts)
  if (s <= 0 || s > kL1MaxS || H <= 0 || H > kL1HitCapTiny || H > a.ldsHitCap) return;
  l1_tiny(a, f, s, H, wave_uniform(d.off), wave_uniform((int32_t)(d.sm >> 16)), hits[wv], V[wv]);
}

// The fragments of class S (256 < H <= 2048) as a list, for batches in which they are the exception: a fragment set that meets a
// foreign reference shard or index chunk has no seed hit at all for most of its fragments, and a workgroup that finds nothing to do
// still costs its dispatch — ~9 ns each with 24 KiB of LDS to reserve, 13 of the 17 ms k_l1<0, 2048> took for the 1.67 M fragment
// visits of one rank of an 8-GPU job (profiles/r04h_bench_sim8.json.log).  One thread per fragment: fragments without hits get their
// zero counts here, class-S fragments are appended in order (one atomic per workgroup).
static __global__ __launch_bounds__(kTPB) void k_l1_list(L1Args a, int32_t *__restrict__ list, unsigned int *__restrict__ cursor)
{
  __shared__ unsigned int wcount[kTPB / kWave], sBase;
  const int i = blockIdx.x * kTPB + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  int f = -1; bool small = false;
  if (i < a.nFrag) {
    f = a.fragOrder ? a.fragOrder[i] : i;
    const int s = a.fragS[f], H = a.fragHits[f];
    if (s <= 0 || H <= 0) { a.fragCandCnt[f] = 0; a.fragCandOff[f] = 0; }        // (H < 0: overflow marker of k_l1_probe, the host fails the call)
    else small = s <= kL1SmallMaxS && H <= a.ldsHitCap && H <= kL1HitCapSmall && !(H <= kL1HitCapTiny && a.tinyPath);
  }
  const unsigned long long m = __ballot(small);
  if (lane == 0) wcount[wv] = (unsigned int)__popcll(m);
  block_barrier();
  if (threadIdx.x == 0) {
    unsigned int tot = 0;
    for (int w = 0; w < kTPB / kWave; w++) { const unsigned int c = wcount[w]; wcount[w] = tot; tot += c; }
    sBase = tot ? atomicAdd(cursor, tot) : 0u;
  }
  block_barrier();
  if (small) list[sBase + wcount[wv] + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = i;      // (position in the processing order: fragDesc)
}

// Pass 2: gather + sort + candidate regions for the fragments whose hit count is in (HLO, HCAP]; everything in LDS.
// (Measured and removed in round 3: class M "in two walks" — the hit runs read twice, first only to feed the filter's counters,
// then to stage the survivors in the 24 KiB of class S instead of 48 KiB.  1000 references: class M 4.8 -> 4.3 ms per launch; 10 000
// references, where class M is mostly noise and the gather is all there is: L1 stage 83 -> 87 ms per 300 queries.  One read wins.
// Also measured and removed: 32-bit sort keys (rank of the contig among the fragment's distinct contigs << posBits | wpos: 3 instead of
// 7 instructions per key and lane-exchange stage).  The ranks need a hash set of the contig ids in LDS, the distinct ids sorted, a
// look-up per hit — five more barrier-separated phases and 20 more registers (5 instead of 6 workgroups per CU): k_l1<0,2048> went
// from 30.7 to 41.2 ms per step, bit-exact.)
template <int HLO, int HCAP>
static __global__ __launch_bounds__(kTPB, (HLO == 0 ? 7 : 1)) void k_l1(L1Args a, const int32_t *__restrict__ list)
{
  // scratch V: the per-hash hit offsets during the gather (pOff), four bit arrays during the noise filter, the indices of the valid runs
  // (16 bit) during emission.  Class S: 4 KiB (<= 1024 sketch hashes, 13-bit filter); class M: 8 KiB (<= 2048 sketch hashes, 14-bit filter).
  using VT = uint16_t;
  constexpr int kArr = HCAP + kL1LdsSpare;             // 2048 / 4096
  constexpr int kMaxS = HLO == 0 ? kL1SmallMaxS : kL1MaxS;
  constexpr int NBW = (1 << kL1FilterBits<HCAP>()) / 32;          // words per bit array; V holds {seenA, twiceA, seenB, twiceB}
  constexpr int kVWords = 4 * NBW > kMaxS ? 4 * NBW : kMaxS;
  static_assert((kArr & (kArr - 1)) == 0 && kVWords * 4 >= kArr * (int)sizeof(VT) && kVWords >= kMaxS, "the scratch holds the run indices and the hit offsets");
  __shared__ uint64_t hits[kArr];
  __shared__ __attribute__((aligned(16))) uint32_t Vraw[kVWords];
  VT *V = (VT *)Vraw;
  int *ws = (int *)&hits[HCAP];                                   // 16 ints of scan scratch ...
  unsigned long long &sBase = *(unsigned long long *)&hits[HCAP + 8];     // ... and two scalars in the spare entries of the hit array
  int &sKeep = *(int *)&hits[HCAP + 9];
  int pos;                                           // the fragment's position in the processing order
  if (list) pos = list[blockIdx.x];
  else {
    pos = xcd_item(blockIdx.x, gridDim.x);
    if (pos >= a.nFrag) return;
  }
  const int t = threadIdx.x;
  const L1FragDesc d = l1_frag_desc(a.fragDesc + pos);                // fragment, sketch size | minimumHits, seed hits, pool offset: one load
  const int f = d.f, s = (int)(d.sm & 0xffffu), H = d.H;
  if (HLO == 0 && (s <= 0 || H <= 0)) {              // (H < 0: overflow marker of k_l1_probe, the host fails the call)
    if (t == 0) { a.fragCandCnt[f] = 0; a.fragCandOff[f] = 0; }
    return;
  }
  if (HLO == 0 && (s > kL1MaxS || H > a.ldsHitCap)) return;        // beyond the LDS classes: k_l1_big_* below
  if (HLO == 0 && H <= kL1HitCapTiny && a.tinyPath) return;        // a handful of hits: k_l1_tiny has them (one wave per fragment)
  if (H > HCAP || s <= 0 || s > kL1MaxS) return;                 // another class handles it ...
  // ... class S takes sketches of <= 1024 hashes, class M the longer ones as well (k_l1_probe lists them)
  if (HLO == 0 ? s > kMaxS : (H <= HLO && s <= kL1SmallMaxS)) return;
  const uint32_t off = d.off;
  const int m = (int)(d.sm >> 16);                  // minimumHits of s (computeMap.hpp:301)
  int *pOff = (int *)Vraw;                          // hit offsets per probe alias V (V is only written after the gather)
  constexpr int kPerS = kMaxS / kTPB;                // sketch hashes per thread at most
  uint32_t pFirst[kPerS];                           // the runs' starts travel with the counts: one round trip to memory instead of two
#pragma unroll
  for (int j = 0; j < kPerS; j++) {
    const int i = t + j * kTPB;
    if (i < s) { pOff[i] = (int)a.probeCnt[off + i]; pFirst[j] = a.probeFirst[off + i]; }
  }
  block_barrier();
  block_array_excl_scan(pOff, s, ws);
  // gather (computeMap.hpp:283-299).  (Tried and measured equal: four loads in flight per lane; one lane per hit instead of per
  // sketch hash — again in round 6, with the bisection over the offsets in LDS and a lane's eight loads independent of each other:
  // 27.1 ms against 26.8, although the walk below is 43 % of a workgroup's life by the kernel's own clock, profiles/r06o_l1_phase_clock.txt:
  // the lines arrive at the memory system's pace for random 128-byte lines, however the requests are issued.)  Measured by compiling the later phases out
  // (1000 x 1000 x 5 Mbp, ms per step of this kernel): gather 13.4,
  // noise filter +2.2, sort +10.3 (in registers; the LDS network it replaced: +14.8), candidate emission +4.0.  The gather reads one
  // short run (~5 entries = 40 bytes) per sketch hash from a random place of the hash-ordered payload: ~45 KB of 128-byte lines per
  // fragment for 10 KB of hits, 77 GB per step by the FETCH_SIZE counter — it runs at the memory system's pace for such lines.
#pragma unroll
  for (int j = 0; j < kPerS; j++) {
    const int i = t + j * kTPB;
    if (i < s) {
      const int o = pOff[i], e = (i + 1 < s) ? pOff[i + 1] : H;
      const uint32_t fi = pFirst[j];
      for (int c = 0; c < e - o; c++) hits[o + c] = a.sSW[fi + c];
    }
  }
  // Noise filter.  Minimizer hashes are minima over w k-mers, so they crowd the low end of the 32-bit range and most seed hits of
  // a fragment against a large reference set are chance collisions: isolated hits (measured on 1000 x 5 Mbp: ~1300 hits per
  // fragment, ~700 of them alone in their 4-kb neighbourhood).  With minimumHits m >= 2 a hit without another hit of the same
  // contig within < fragLen is in no valid run (computeMap.hpp:326-336), and a run that becomes consecutive once it is gone
  // spans it, i.e. would have been such a neighbour: dropping those hits leaves every candidate unchanged and shortens the sort.
  // "Has a neighbour" is tested conservatively with two offset tilings of width W >= 2 fragLen (two points closer than W/2 share a
  // tile in one of them) hashed into 2-bit occupancy counters: a collision only keeps a hit that could have been dropped.
  int n = H;
  if (m >= 2 && H > a.filterMinHits) {
    uint32_t *bits = Vraw;
    block_barrier();                                 // the gather is complete
    for (int i = t; i < 4 * NBW; i += kTPB) bits[i] = 0u;
    if (t == 0) sKeep = 0;
    block_barrier();
    constexpr int PER = kArr / kTPB;
    uint64_t hv[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int x = t + j * kTPB;
      if (x < H) {
        hv[j] = hits[x];
        uint32_t ia, ib;
        l1_filter_tiles<kL1FilterBits<HCAP>()>(hv[j], a.filterShift, ia, ib);
        const uint32_t ba = 1u << (ia & 31), bb = 1u << (ib & 31);
        if (atomicOr(&bits[ia >> 5], ba) & ba) atomicOr(&bits[NBW + (ia >> 5)], ba);
        if (atomicOr(&bits[2 * NBW + (ib >> 5)], bb) & bb) atomicOr(&bits[3 * NBW + (ib >> 5)], bb);
      }
    }
    block_barrier();
    // (the counter indices are computed again rather than kept across the barrier: sixteen registers, which decide how many workgroups a CU holds)
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int x = t + j * kTPB;
      if (x < H) {
        uint32_t ia, ib;
        l1_filter_tiles<kL1FilterBits<HCAP>()>(hv[j], a.filterShift, ia, ib);
        const bool keep = ((bits[NBW + (ia >> 5)] >> (ia & 31)) | (bits[3 * NBW + (ib >> 5)] >> (ib & 31))) & 1u;
        if (keep) hits[atomicAdd(&sKeep, 1)] = hv[j];      // every lane has read its hits: compaction in place, order irrelevant
      }
    }
    block_barrier();
    n = wave_uniform(sKeep);
    // sKeep lives in the hit array's spare entries, which the sort's padding overwrites: every thread has read it
    block_barrier();
  }
  block_sort<uint64_t>(hits, n);                    // :320 (starts with a barrier: the gather is complete)

  l1_emit_candidates(a, f, s, n, m, hits, V, ws, &sBase);
}

// ------------------------------------------------------------------------------------------------
// Fragments beyond the LDS classes — thousands of near-identical references (a species-dense database sends EVERY fragment here), or
// low-complexity references with 10^5..10^6 occurrences of one hash — take the same algorithm over global memory, BATCHED: the big
// fragments of a sub-batch are cut into groups (a few 10^8 hits each), and a group costs five launches whatever its size:
//   k_l1_big_offsets     per fragment: exclusive scan of its probes' run lengths (the start of every hash's hits in the fragment's list)
//   k_l1_big_gather      one workgroup per tile of 16384 hits: hit x of fragment i -> key (i, seqId, wpos) in ONE 64-bit word
//                        (field widths from the chunk's contig count and longest contig), read from the hash-ordered payload
//   device radix sort    over the used key bits only: orders by fragment, then (seqId, wpos) — computeMap.hpp:320 for all fragments at once
//   k_l1_big_unpack      keys back to (seqId << 32 | wpos)
//   k_l1_big_candidates  one workgroup per fragment over its sorted slice (l1_emit_candidates_stream: the algorithm of the LDS classes,
//                        walked in coalesced chunks of 256 entries)
// ------------------------------------------------------------------------------------------------
constexpr int kL1BigTileHits = 16384;
struct L1BigArgs {
  const int32_t *frag;          // [n] fragment ids of the group
  const uint32_t *sOff;         // [n+1] prefix of the sketch sizes (into hashOff)
  const uint64_t *hitOff;       // [n+1] prefix of the hit counts (into keys / V)
  const uint32_t *tileFirst;    // [n+1] first tile of every fragment
  int32_t *hashOff;             // [sum s] start of every hash's hits inside its fragment's list
  uint64_t *keys;               // [sum H]
  int n; int shiftSeq, shiftRank;       // key = rank << shiftRank | seqId << shiftSeq | wpos
};

static __global__ void k_l1_big_info(const int32_t *__restrict__ list, uint32_t n, const int32_t *__restrict__ fragS, const int32_t *__restrict__ fragHits,
    int32_t *__restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int f = list[i]; out[2 * i] = fragS[f]; out[2 * i + 1] = fragHits[f]; }
}

static __global__ __launch_bounds__(kTPB) void k_l1_big_offsets(L1Args a, L1BigArgs g)
{
  __shared__ int ws[16];
  const int i = blockIdx.x;
  const int f = g.frag[i];
  const int s = a.fragS[f];
  const uint32_t off = a.fragOff[f];
  int32_t *ho = g.hashOff + g.sOff[i];
  for (int j = threadIdx.x; j < s; j += kTPB) ho[j] = (int)a.probeCnt[off + j];
  block_barrier_mem();
  block_array_excl_scan(ho, s, ws);
}

static __global__ __launch_bounds__(kTPB) void k_l1_big_gather(L1Args a, L1BigArgs g)
{
  __shared__ int sho[kFragHashCapL1];
  __shared__ int sFrag;
  // tile -> fragment: last i with tileFirst[i] <= tile
  if (threadIdx.x == 0) {
    int lo = 0, hi = g.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.tileFirst[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    sFrag = lo;
  }
  block_barrier();
  const int i = sFrag;
  const int f = g.frag[i];
  const int s = a.fragS[f], H = a.fragHits[f];
  const uint32_t off = a.fragOff[f];
  const int32_t *ho = g.hashOff + g.sOff[i];
  for (int j = threadIdx.x; j < s; j += kTPB) sho[j] = ho[j];
  block_barrier();
  const int x0 = (int)(blockIdx.x - g.tileFirst[i]) * kL1BigTileHits;
  const int x1 = x0 + kL1BigTileHits < H ? x0 + kL1BigTileHits : H;
  uint64_t *out = g.keys + g.hitOff[i];
  // a lane's hits are 256 apart: consecutive lanes read consecutive entries of (mostly) one hash's run
  int j = 0;
  for (int x = x0 + (int)threadIdx.x; x < x1; x += kTPB) {
    // hash whose run holds hit x: last j with sho[j] <= x (runs may be empty); the previous answer is a lower bound
    int lo = j, hi = s;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sho[mid] <= x) lo = mid; else hi = mid; }
    j = lo;
    const uint64_t hit = a.sSW[a.probeFirst[off + j] + (uint32_t)(x - sho[j])];
    out[x] = ((uint64_t)(uint32_t)i << g.shiftRank) | ((hit >> 32) << g.shiftSeq) | (uint64_t)(uint32_t)hit;
  }
}

static __global__ void k_l1_big_unpack(uint64_t *__restrict__ keys, uint64_t n, int shiftSeq, int shiftRank)
{
  const uint64_t mSeq = (1ull << (shiftRank - shiftSeq)) - 1, mPos = (1ull << shiftSeq) - 1;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t k = keys[x];
    keys[x] = (((k >> shiftSeq) & mSeq) << 32) | (k & mPos);
  }
}

static __global__ __launch_bounds__(kTPB) void k_l1_big_candidates(L1Args a, L1BigArgs g, int *__restrict__ V)
{
  __shared__ int ws[16];
  __shared__ unsigned long long sBase;
  const int i = blockIdx.x;
  const int f = g.frag[i];
  const int s = a.fragS[f];
  l1_emit_candidates_stream(a, f, a.fragHits[f], s <= a.lutMaxS ? a.minHitsLUT[s] : 1, g.keys + g.hitOff[i], V + g.hitOff[i], ws, &sBase);
}

// Reorder candidates into the reference's callback order — fragment ascending, then (seqId, start) as produced — or, for a batch
// of several genomes, into the fragments' processing order (`order`; the candidates of one fragment stay together either way).
// fragCandCnt and orderedOff are indexed by position in that order.
static __global__ void k_l1_order(const uint32_t *__restrict__ fragCandOff, const int32_t *__restrict__ fragCandCnt,
                           const uint32_t *__restrict__ orderedOff, const int32_t *__restrict__ order, int32_t nFrag,
                           const int32_t *__restrict__ inSeq, const int32_t *__restrict__ inStart, const int32_t *__restrict__ inEnd,
                           int32_t *__restrict__ outFrag, int32_t *__restrict__ outSeq, int32_t *__restrict__ outStart,
                           int32_t *__restrict__ outEnd)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nFrag) return;
  const int f = order ? order[i] : i;
  const int n = fragCandCnt[i];
  const uint32_t src = fragCandOff[f], dst = orderedOff[i];
  for (int j = 0; j < n; j++) {
    outFrag[dst + j] = f; outSeq[dst + j] = inSeq[src + j]; outStart[dst + j] = inStart[src + j]; outEnd[dst + j] = inEnd[src + j];
  }
}

}  // namespace ani
