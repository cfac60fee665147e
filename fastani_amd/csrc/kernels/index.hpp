// index.hpp — reference index over the position-ordered minimizer records.
//
// Replaces skch::Sketch::index (src/map/include/winSketch.hpp:181-193: unordered_map<hash, vector<{seqId,wpos}>>)
// and skch::Sketch::searchIndex (:259-270) by flat device arrays:
//   position order (≙ minimizerIndex :93)      mHash[n], mSeq[n], mWpos[n], mWin[n] (window links of the L2 event stream) + contigFirstMin[nContigs+1]
//   hash order    (≙ minimizerPosLookupIndex)  sHash[n] (sorted), sSW[n] = seqId<<32|wpos carried through the stable sort, so
//                                              every hash's occurrence list is one contiguous run in (seqId,wpos) order (:186-190)
//   probe table   table[2 x distinct]          order-preserving open-addressing table {hash, first, count} over the distinct hashes (below)
//   same-hash links (for the L2 set semantics) DupLinks: for the few entries that have one, the neighbouring NEAR occurrence of the same hash in
//                                              position order (one that can share a super-window) — a sorted list + one bit per entry
#pragma once
#include "common.hpp"

namespace ani {

constexpr uint32_t kWinMask = 0x3fffu, kWinMoreBit = 1u << 30, kWinDupBit = 1u << 31;
constexpr int kWinShiftA = 14;

// (records -> position-ordered SoA arrays: written by the histogram read of the index sort, radix.hpp: k_radix_histogram)

// Same-hash links.  The sliding map of the reference is a SET (slidingMap.hpp:150-154, :178): an entry whose hash already sits in the
// super-window changes nothing when it enters, and one whose hash stays behind changes nothing when it leaves.  What L2 needs for
// that is, per entry, the neighbouring occurrence of the same hash in position order IF it is near enough to share a super-window —
// which almost no entry has (two of ~240 32-bit hashes colliding inside 3 kb, or a tandem repeat).  Round 3 kept two dense int32
// arrays (8 of the index's 45 bytes per minimizer); now: one bit per entry ("has a link") and a sorted list of 64-bit half-records
//     entry << 32 | kind << 31 | other entry          kind 0 = `other` is the previous near occurrence, 1 = the next one
// searched by bisection — only behind the bit (general kernel) or the event's nearDup flag (k_l2_sim).
struct DupLinks { const uint64_t *list; uint32_t n; const uint32_t *bits; };
__host__ __device__ __forceinline__ uint32_t dup_lower_bound(const DupLinks &d, uint32_t j)
{
  uint32_t lo = 0, hi = d.n;
  const uint64_t key = (uint64_t)j << 32;
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (d.list[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
__host__ __device__ __forceinline__ bool dup_flag(const DupLinks &d, uint32_t j) { return d.n != 0 && ((d.bits[j >> 5] >> (j & 31u)) & 1u); }
// previous / next near occurrence of entry j's hash, -1 if none (call behind dup_flag or the nearDup flag of the window links)
__host__ __device__ __forceinline__ int32_t dup_prev(const DupLinks &d, uint32_t j)
{
  const uint32_t x = dup_lower_bound(d, j);
  if (x < d.n) { const uint64_t v = d.list[x]; if ((uint32_t)(v >> 32) == j && !((v >> 31) & 1ull)) return (int32_t)(v & 0x7fffffffull); }
  return -1;
}
__host__ __device__ __forceinline__ int32_t dup_next(const DupLinks &d, uint32_t j)
{
  uint32_t x = dup_lower_bound(d, j);
  for (int k = 0; k < 2 && x < d.n; k++, x++) { const uint64_t v = d.list[x]; if ((uint32_t)(v >> 32) != j) break;
    if ((v >> 31) & 1ull) return (int32_t)(v & 0x7fffffffull); }
  return -1;
}

static __global__ void k_index_join(const uint32_t *__restrict__ mHash, const int32_t *__restrict__ mSeq, const int32_t *__restrict__ mWpos,
                             uint32_t n, uint32_t seqBase, uint32_t *__restrict__ records)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    records[3 * (size_t)i] = mHash[i]; records[3 * (size_t)i + 1] = (uint32_t)mSeq[i] + seqBase; records[3 * (size_t)i + 2] = (uint32_t)mWpos[i];
  }
}

// After the stable sort by hash (sHash[r], sSW[r] = seqId<<32|wpos; equal hashes stay in position order): unique-hash count
// and, for NEAR duplicates only, the same-hash links: two half-records per pair appended to `pairs` (unordered; the host sorts them),
// the "has a link" bit of both entries, and the nearDup flag in their window links (bit 31 of mWin, if the chunk has window links).
//   nearDup: two same-hash entries j' < j of one contig can share a super-window.  All entries of a window except its first
//   lie within cmw = countMinimizerWindows positions; the first entry (MIIteratorL2 keeps the minimizer that is active at the
//   window start) may trail by less than the gap to its successor: possible iff wpos[j] - wpos[j'] <= cmw + (wpos[j'+1] - wpos[j']).
// Entries without a link behave as plain set members in L2 (links of non-near pairs would only ever be compared against the bounds of
// one window).  `nPairs` counts every pair; a pair beyond `pairCap` is not stored (the host reruns the kernel with room for all).
// one same-hash pair (r - 1, r) of the sorted index: the link records if the two are near duplicates
__device__ __forceinline__ void index_link_pair(uint32_t r, const uint64_t *__restrict__ sSW, const int32_t *__restrict__ mWpos,
    const int32_t *__restrict__ contigFirstMin,
                                                int32_t cmw, uint64_t *__restrict__ pairs, uint32_t pairCap, unsigned int *__restrict__ nPairs,
                                                uint32_t *__restrict__ dupBits, uint32_t *__restrict__ mWin)
{
  const uint64_t a = sSW[r - 1], b = sSW[r];                   // a is positionally before b
  if ((a >> 32) != (b >> 32)) return;                          // different contigs never share a window
  const int32_t seq = (int32_t)(a >> 32), wa = (int32_t)(uint32_t)a, wb = (int32_t)(uint32_t)b;
  // positional indices: binary search on wpos inside the contig's slice
  int32_t lo = contigFirstMin[seq], hi = contigFirstMin[seq + 1];
  const int32_t cHi = hi;
  while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (mWpos[mid] < wa) lo = mid + 1; else hi = mid; }
  const int32_t ia = lo;
  const int32_t gap = (ia + 1 < cHi) ? mWpos[ia + 1] - wa : 0;
  if (wb - wa > cmw + gap) return;
  hi = cHi;
  while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (mWpos[mid] < wb) lo = mid + 1; else hi = mid; }
  const int32_t ib = lo;
  const unsigned int slot = atomicAdd(nPairs, 1u);
  if (slot < pairCap) {
    pairs[2 * (size_t)slot] = ((uint64_t)(uint32_t)ia << 32) | (1ull << 31) | (uint32_t)ib;         // ia: its next near occurrence is ib
    pairs[2 * (size_t)slot + 1] = ((uint64_t)(uint32_t)ib << 32) | (uint32_t)ia;                    // ib: its previous one is ia
  }
  atomicOr(&dupBits[ia >> 5], 1u << (ia & 31)); atomicOr(&dupBits[ib >> 5], 1u << (ib & 31));
  if (mWin) { atomicOr(&mWin[ia], kWinDupBit); atomicOr(&mWin[ib], kWinDupBit); }
}

// (round 5: four consecutive entries per thread from one 16-byte load — one entry per thread and iteration was a chain of ~190
//  dependent 4-byte loads per thread, 2.0 ms for the 1.6 GB of hashes.)
static __global__ void k_index_links(const uint32_t *__restrict__ sHash, const uint64_t *__restrict__ sSW, uint32_t n,
                              const int32_t *__restrict__ mWpos, const int32_t *__restrict__ contigFirstMin, int32_t cmw,
                              uint64_t *__restrict__ pairs, uint32_t pairCap, unsigned int *__restrict__ nPairs,
                              uint32_t *__restrict__ dupBits, uint32_t *__restrict__ mWin, unsigned long long *__restrict__ nUnique)
{
  unsigned long long uniq = 0;
  const uint32_t nQuads = (n + 3u) >> 2;
  const bool vec = ((uintptr_t)sHash & 15u) == 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nQuads; t += gridDim.x * blockDim.x) {
    const uint32_t r0 = t << 2;
    uint32_t h[5];
    h[0] = r0 ? sHash[r0 - 1] : 0u;
    if (vec && r0 + 4u <= n) { const uint4 v = *(const uint4 *)(sHash + r0); h[1] = v.x; h[2] = v.y; h[3] = v.z; h[4] = v.w; }
    else {
#pragma unroll
      for (int k = 0; k < 4; k++) h[1 + k] = r0 + k < n ? sHash[r0 + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t r = r0 + k;
      if (r >= n) break;
      const bool samePrev = r > 0 && h[k] == h[k + 1];
      uniq += !samePrev;
      if (samePrev) index_link_pair(r, sSW, mWpos, contigFirstMin, cmw, pairs, pairCap, nPairs, dupBits, mWin);
    }
  }
  // one atomic per workgroup (the grid is small: a grid-stride loop covers the index)
  __shared__ unsigned long long part[8];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) uniq += __shfl_down(uniq, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = uniq;
  block_barrier();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (unsigned w = 0; w < (blockDim.x >> 6); w++) t += part[w];
    if (t) atomicAdd(nUnique, t);
  }
}

// The nearDup flag of the window links (bit 31 of mWin) for every entry that has a same-hash link: set from the sorted half-records
// once both the links (main stream) and the window links (side stream) are there (round 5: k_index_links used to OR the bit into
// mWin itself, which tied the two kernels into one order).
static __global__ void k_index_mark_dups(const uint64_t *__restrict__ dupList, uint32_t nDup, uint32_t *__restrict__ mWin)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nDup) atomicOr(&mWin[(uint32_t)(dupList[i] >> 32)], kWinDupBit);
}

// Bucket key of a hash.  Minimizer hashes are minima over w k-mer hashes, so their density falls off like w (1 - v)^(w-1) over
// the 32-bit range v = h / 2^32: buckets cut from the top bits of h would hold dozens of entries at the low end and none at the
// high end.  The key is the CDF instead, ~ 2^32 (1 - (1 - v)^w) in 32-bit fixed point (square-and-multiply on the high halves
// of the products): non-decreasing in h — every step is a floor of a product of non-decreasing factors — so a bucket is still
// a contiguous run of the hash-sorted index, and close to uniformly filled.
__host__ __device__ __forceinline__ uint32_t bucket_key(uint32_t h, int w)
{
  uint32_t p = ~h, r = 0xffffffffu;
  for (int e = w; e; e >>= 1) {
    if (e & 1) r = (uint32_t)(((uint64_t)r * p) >> 32);
    p = (uint32_t)(((uint64_t)p * p) >> 32);
  }
  return ~r;
}

// ------------------------------------------------------------------------------------------------
// Probe table (≙ minimizerPosLookupIndex.find, winSketch.hpp:181-193): an order-preserving open-addressing table of the distinct
// hashes.  Slot of a hash = its (monotone) bucket key scaled to the table size; the distinct hashes are inserted in sorted order
// with linear probing, so a cluster stays sorted and the i-th distinct hash lands at
//     p_i = max(slot_i, p_(i-1) + 1) = i + max_(j <= i) (slot_j - j),
// a running maximum — no atomics, no collisions to resolve.  A slot is 12 bytes {hash, first, cnt}: the hash's run in sSW (a
// sentinel {~0, n | 2^31, 0} closes the table).  A probe reads the slot of the hash and walks forward while the entries are
// smaller: one dependent load in most cases.  (With 8-byte slots the run length was the next occupied slot's `first` minus this
// one's — at load 0.25 that was four more dependent loads per probe, and the probe kernel does nothing but wait for its loads.)
// ------------------------------------------------------------------------------------------------
struct TableSlot { uint32_t hash, first, cnt; };
constexpr uint32_t kSlotEmpty = 0xffffffffu;
constexpr int kTableBlock = kTPB * 8;
__host__ __device__ __forceinline__ uint32_t table_slot(uint32_t h, int w, uint32_t nSlots) { return (uint32_t)(((uint64_t)bucket_key(h, w) * nSlots) >> 32); }

// a thread's eight consecutive hashes of the sorted index from r0 (a multiple of 8; 0 beyond n) and which of them start a run
// (round 5: two 16-byte loads and one 4-byte load instead of sixteen 4-byte loads at a 32-byte lane stride)
__device__ __forceinline__ int table_load8(const uint32_t *__restrict__ sHash, uint32_t n, uint32_t r0, uint32_t (&h)[8], bool (&head)[8])
{
  uint32_t prev = (r0 > 0 && r0 <= n) ? sHash[r0 - 1] : 0u;
  if (r0 + 8u <= n && ((uintptr_t)sHash & 15u) == 0) {
    const uint4 a = *(const uint4 *)(sHash + r0), b = *(const uint4 *)(sHash + r0 + 4);
    h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = r0 + k < n ? sHash[r0 + k] : 0u;
  }
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t r = r0 + k;
    head[k] = r < n && (r == 0 || prev != h[k]);
    prev = h[k];
    cnt += head[k];
  }
  return cnt;
}

// pass 1: per block of kTableBlock entries of the hash-sorted index: number of distinct hashes that start in it, and
// max(slot - index inside the block) over them (INT_MIN if none)
// The table size follows the number of distinct hashes, which k_index_links has just counted on the device: the kernel derives
// nSlots from that counter (table_slots) so the host reads the counters and the block totals in ONE round trip.
__host__ __device__ __forceinline__ uint32_t table_slots(unsigned long long nUnique)
{
  const unsigned long long v = nUnique * 2ull;
  return (uint32_t)(v < 1024ull ? 1024ull : v > 0x7ffffff0ull ? 0x7ffffff0ull : v);
}
static __global__ __launch_bounds__(kTPB) void k_table_block_totals(const uint32_t *__restrict__ sHash, uint32_t n, int w,
    const unsigned long long *__restrict__ nUnique,
                                                             int32_t *__restrict__ blockCnt, int32_t *__restrict__ blockBest)
{
  __shared__ int ws[16];
  const uint32_t nSlots = table_slots(*nUnique);
  const uint32_t r0 = blockIdx.x * (uint32_t)kTableBlock + threadIdx.x * 8u;
  uint32_t h[8]; bool head[8];
  const int cnt = table_load8(sHash, n, r0, h, head);
  int total; int idx = block_excl_scan(cnt, ws, &total);
  int best = INT32_MIN;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (head[k]) { const int v = (int)table_slot(h[k], w, nSlots) - idx; best = v > best ? v : best; idx++; }
  const int inc = block_incl_maxscan(best, ws);
  if (threadIdx.x == kTPB - 1) { blockCnt[blockIdx.x] = total; blockBest[blockIdx.x] = inc; }
}

// pass 2: the distinct hashes into their slots.  carryCnt[b] = distinct hashes before block b, carryBest[b] = max(slot - global
// index) over them (both finished on the host from pass 1)
static __global__ __launch_bounds__(kTPB) void k_table_scatter(const uint32_t *__restrict__ sHash, uint32_t n, int w, uint32_t nSlots,
                                                        const int32_t *__restrict__ carryCnt, const int32_t *__restrict__ carryBest,
                                                        TableSlot *__restrict__ table)
{
  __shared__ int ws[kTPB + 16];
  const uint32_t r0 = blockIdx.x * (uint32_t)kTableBlock + threadIdx.x * 8u;
  uint32_t h[8]; bool head[8];
  const int cnt = table_load8(sHash, n, r0, h, head);
  int total; const int excl = block_excl_scan(cnt, ws, &total);
  const int g0 = carryCnt[blockIdx.x] + excl;       // global index of my first distinct hash
  int slot[8]; int best = INT32_MIN, g = g0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (head[k]) { slot[k] = (int)table_slot(h[k], w, nSlots); const int v = slot[k] - g; best = v > best ? v : best; g++; }
  const int inc = block_incl_maxscan(best, ws);
  block_barrier();
  ws[8 + threadIdx.x] = inc;
  block_barrier();
  int run = carryBest[blockIdx.x];
  if (threadIdx.x > 0) { const int pv = ws[8 + threadIdx.x - 1]; run = pv > run ? pv : run; }
  g = g0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (head[k]) {
      const int v = slot[k] - g; run = v > run ? v : run;
      TableSlot e; e.hash = h[k]; e.first = r0 + k;
      // run length: the thread's own eight entries are in registers (h[] is 0 beyond n, and a hash of 0 cannot follow a larger one
      // in sorted order unless the run is over); only a run that reaches the end of the eight continues in memory
      uint32_t c = 1; bool open = true;
#pragma unroll
      for (int j = 1; j < 8; j++) if (k + j < 8) { open = open && r0 + k + j < n && h[k + j] == h[k]; c += open ? 1u : 0u; }
      if (open) { uint32_t r = r0 + 8; while (r < n && sHash[r] == h[k]) { c++; r++; } }
      e.cnt = c;
      table[(uint32_t)(g + run)] = e;
      g++;
    }
}

// occurrences of hash h in the hash-ordered payload array: [first, first + cnt)
__device__ __forceinline__ void table_probe(const TableSlot *__restrict__ table, int w, uint32_t nSlots, uint32_t h, uint32_t &first, uint32_t &cnt)
{
  uint32_t s = table_slot(h, w, nSlots);
  TableSlot e = table[s];
  while (e.first != kSlotEmpty && e.hash < h) e = table[++s];        // the sentinel {~0, n | 2^31} stops the walk at the latest
  first = 0; cnt = 0;
  if ((e.first & 0x80000000u) || e.hash != h) return;                 // empty slot, the sentinel, or a larger hash: not in the index
  first = e.first; cnt = e.cnt;
}

// contigFirstMin[c] = first position-ordered entry with seqId >= c, for c = 0..nContigs
static __global__ void k_index_contig_first(const int32_t *__restrict__ mSeq, uint32_t n, int32_t nContigs,
                                     int32_t *__restrict__ contigFirstMin)
{
  for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= nContigs; c += gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      uint32_t mid = lo + ((hi - lo) >> 1);
      if (mSeq[mid] < c) lo = mid + 1; else hi = mid;
    }
    contigFirstMin[c] = (int32_t)lo;
  }
}

// Window links for the L2 event stream (l2.hpp).  A candidate's sliding super-window (MIIteratorL2.hpp:74-96) deletes entry j at
// position D_j = wpos[j+1] and inserts entry e at I_e = wpos[e] - cmw + 1; the merged order of those events (a delete before an
// insert at the same position, as the reference applies them) does not depend on the candidate or the query, only on the
// reference positions, so it is computed once per index, one 32-bit word per entry:
//   bits 14..27  A[j] = (first e with I_e >= D_j) - j            inserts that precede the delete of j: entries below j + A[j]
//   bits  0..13  B[e] = e - (first x with wpos[x] > wpos[e] - cmw + 1)   deletes that precede the insert of e: entries up to e - B[e] - 2
//   bit 30       the insert at the same position follows the delete of j (I_e == D_j): no evaluation in between
//   bit 31       nearDup (OR-ed in afterwards by k_index_links, which needs the hash-sorted index; this kernel runs beside the sort)
// Both offsets count entries inside one super-window (<= cmw + 1 < 2^14, checked by the host).  One thread per entry, two binary
// searches over at most cmw entries; the lanes of a wave search neighbouring ranges with the same step pattern, so the loads
// coalesce.
// a global load the optimiser cannot merge with an LDS load of the other branch into one flat load of a selected pointer
#if defined(__HIP_DEVICE_COMPILE__)
#define ANI_LOAD_APART(p) __builtin_nontemporal_load(p)
#else
#define ANI_LOAD_APART(p) (*(p))
#endif
constexpr int kWinHalo = 768;             // positions staged in LDS on either side of a workgroup's entries (~3 typical spans)
constexpr int kWinBlock = 1024;           // entries per workgroup (four per thread: the halo is read once per 1024 entries, 2.5 x the data instead of 7 x)
static __global__ __launch_bounds__(256) void k_index_window_links(const int32_t *__restrict__ mSeq, const int32_t *__restrict__ mWpos,
                                                            const int32_t *__restrict__ contigFirstMin, uint32_t n,
                                                            int32_t cmw1, int32_t expect /* entries per cmw positions */, uint32_t *__restrict__ mWin)
{
  // The searches of neighbouring entries touch the same few hundred positions on either side: staged once (coalesced), searched
  // in LDS — the kernel was nothing but chains of dependent global loads.  A search that leaves the staged range (unusually sparse
  // stretches) reads global memory as before.
  // (32-bit index arithmetic: a chunk holds < 2^31 entries.  The staged read is an LDS read of a clamped offset, the rare miss a
  //  separate global load behind a branch: written as one conditional expression the compiler selected between the two POINTERS and
  //  issued flat loads — 64-bit compares and address selects, ~500 instructions executed per entry (SQ counters in
  //  profiles/r05af_pmc_SQ_WAVES_summary.txt); 5.09 -> 4.86 ms beside the build's main stream, profiles/r05ag_index_timeline.txt.)
  __shared__ int32_t sw[kWinBlock + 2 * kWinHalo];
  const int32_t j0 = (int32_t)(blockIdx.x * (uint32_t)kWinBlock);
  const int32_t w0 = j0 > kWinHalo ? j0 - kWinHalo : 0;
  const uint32_t span = (uint32_t)(((uint32_t)j0 + (uint32_t)(kWinBlock + kWinHalo) < n ? (uint32_t)j0 + (uint32_t)(kWinBlock + kWinHalo) : n) - (uint32_t)w0);
  for (uint32_t x = threadIdx.x; x < span; x += 256) sw[x] = mWpos[(uint32_t)w0 + x];
  block_barrier();
  // (a macro, not a lambda: captured by reference the __shared__ array becomes a generic pointer and every read a flat load)
#define wpos_at(x_) ([&](int32_t xx_, int32_t staged_) -> int32_t { return (uint32_t)(xx_ - w0) < span ? staged_ : ANI_LOAD_APART(mWpos + xx_); }((x_), \
    sw[(uint32_t)((x_) - w0) < span ? (uint32_t)((x_) - w0) : 0u]))
  // Four CONSECUTIVE entries per thread (round 5; it was four entries 256 apart): both answers are non-decreasing in j inside a
  // contig — the thresholds grow with wpos[j] and the range bounds with j — so only the thread's first entry (and the first of a
  // contig) runs the two searches; the next ones advance the previous answers by the one or two entries the window has moved
  // (up to kWinLinear steps, then the search on what is left).  Measured beside the build's main stream: 5.1 ms per 4 x 10^8
  // entries before and after (profiles/r05s, r05t_index_timeline.txt) — there the kernel is paced by what the other queue leaves it, not by its searches.
  constexpr int kPer = kWinBlock / 256, kWinLinear = 6;
  const uint32_t jj0 = (uint32_t)j0 + threadIdx.x * (uint32_t)kPer;
  if (jj0 >= n) return;
  int32_t sqv[kPer]; uint32_t out[kPer];
  const bool full = jj0 + kPer <= n, vec = full && (((uintptr_t)mSeq | (uintptr_t)mWin) & 15u) == 0;
  if (vec) { const uint4 v = *(const uint4 *)(mSeq + jj0); sqv[0] = (int32_t)v.x; sqv[1] = (int32_t)v.y; sqv[2] = (int32_t)v.z; sqv[3] = (int32_t)v.w; }
  else {
#pragma unroll
    for (int q = 0; q < kPer; q++) sqv[q] = jj0 + q < n ? mSeq[jj0 + q] : -1;
  }
  static_assert(kPer == 4, "k_index_window_links: four entries per thread (one 16-byte load / store)");
  int32_t prevSq = -1, prevB = 0, prevA = 0, cLo = 0, cHi = 0;
  bool prevHasA = false;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    if (jj0 + q >= n) break;
    const int32_t j = (int32_t)(jj0 + q);
    const int32_t sq = sqv[q], wj = wpos_at(j);
    const bool cont = q > 0 && sq == prevSq;
    if (!cont) { cLo = contigFirstMin[sq]; cHi = contigFirstMin[sq + 1]; }
    // u = first x in [max(cLo, j - cmw1), j] with wpos[x] > wj - cmw1   (wpos is strictly increasing: at most one entry per position)
    int32_t lo = j - cmw1 > cLo ? j - cmw1 : cLo, hi = j;
    bool search = true;
    if (cont) {
      lo = prevB > lo ? prevB : lo;
      search = false;
#pragma unroll 1
      for (int st = 0; lo < hi && wpos_at(lo) <= wj - cmw1; lo++) if (++st > kWinLinear) { search = true; break; }
    } else {
      // Both answers lie about `expect` entries away; a 64-entry bracket around that guess is tried first (two loads + 6 steps
      // instead of 12 steps over the whole super-window), the full range only where the local density is unusual.
      const int32_t g0 = j - expect - 32, g1 = j - expect + 32;
      if (g0 >= lo && wpos_at(g0) <= wj - cmw1) lo = g0 + 1;
      if (g1 >= lo && g1 < hi && wpos_at(g1) > wj - cmw1) hi = g1;
    }
    if (search) while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (wpos_at(mid) <= wj - cmw1) lo = mid + 1; else hi = mid; }
    prevB = lo;
    const uint32_t b = (uint32_t)(j - lo);
    uint32_t a = 0, more = 0;
    const bool hasA = j + 1 < cHi;
    if (hasA) {
      const int32_t tgt = wpos_at(j + 1) + cmw1;
      lo = j + 1; hi = j + 2 + cmw1 < cHi ? j + 2 + cmw1 : cHi;
      search = true;
      if (cont && prevHasA) {
        lo = prevA > lo ? prevA : lo;           // (prevA <= the previous hi <= hi)
        search = false;
#pragma unroll 1
        for (int st = 0; lo < hi && wpos_at(lo) < tgt; lo++) if (++st > kWinLinear) { search = true; break; }
      } else {
        const int32_t g0 = j + expect - 32, g1 = j + expect + 32;
        if (g0 >= lo && g0 < hi && wpos_at(g0) < tgt) lo = g0 + 1;
        if (g1 >= lo && g1 < hi && wpos_at(g1) >= tgt) hi = g1;
      }
      if (search) while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (wpos_at(mid) < tgt) lo = mid + 1; else hi = mid; }
      prevA = lo;
      a = (uint32_t)(lo - j);
      more = (lo < cHi && wpos_at(lo) == tgt) ? kWinMoreBit : 0u;
    }
    prevHasA = hasA; prevSq = sq;
    out[q] = (a > kWinMask ? kWinMask : a) << kWinShiftA | (b > kWinMask ? kWinMask : b) | more;
  }
  if (vec) { uint4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3]; *(uint4 *)(mWin + jj0) = v; }
  else {
#pragma unroll
    for (int q = 0; q < kPer; q++) if (jj0 + q < n) mWin[jj0 + q] = out[q];
  }
}

#undef wpos_at

// 12-byte records in position order with GLOBAL seqIds: out[c] = first record with seqId >= seqIdBase + c, c = 0..nContigs.
// Used to cut a record stream into index chunks at genome borders.
static __global__ void k_records_contig_first(const uint32_t *__restrict__ records, uint64_t n, int32_t seqIdBase, int32_t nContigs,
                                       uint64_t *__restrict__ out)
{
  for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= nContigs; c += gridDim.x * blockDim.x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      if ((int32_t)records[3 * mid + 1] < seqIdBase + c) lo = mid + 1; else hi = mid;
    }
    out[c] = lo;
  }
}

// records read from a sketch file for a genome range that does not start at contig 0: seqIds relative to the range
static __global__ void k_records_rebase(uint32_t *__restrict__ records, uint64_t n, int32_t c0)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) records[3 * i + 1] -= (uint32_t)c0;
}

// Exact number of distinct hashes over several index chunks (Sketch::sanityCheck needs it, winSketch.hpp:298-318): every
// distinct hash of chunk C (its first entry in hash order) is looked up in one earlier chunk E; seen[r] is set when found.
static __global__ void k_index_mark_shared(const uint32_t *__restrict__ sHashC, uint32_t nC, const TableSlot *__restrict__ tableE, uint32_t nSlotsE, int w,
                                    uint8_t *__restrict__ seen)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nC; r += gridDim.x * blockDim.x) {
    const uint32_t h = sHashC[r];
    if (r > 0 && sHashC[r - 1] == h) continue;
    if (seen[r]) continue;
    uint32_t first, cnt;
    table_probe(tableE, w, nSlotsE, h, first, cnt);
    if (cnt) seen[r] = 1;
  }
}
static __global__ void k_count_flags(const uint8_t *__restrict__ flags, uint32_t n, unsigned long long *__restrict__ total)
{
  unsigned long long c = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) c += flags[r] != 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}

// Sampled position index for the three searchIndex() calls of every L2 candidate (computeMap.hpp:424-436): posSample[posBase[c] + b]
// = first position-ordered entry of contig c with wpos >= b << kPosSampleShift.  A search then is one table read plus a binary
// search inside one 256-position bin (~20 entries = one or two cache lines; k_l2_ranges is HBM-bound on exactly these lines)
// instead of over the whole contig.
constexpr int kPosSampleShift = 8;
static __global__ void k_index_pos_sample(const int32_t *__restrict__ mWpos, const int32_t *__restrict__ contigFirstMin,
                                   const uint32_t *__restrict__ posBase, int32_t nContigs, uint32_t totalBins, uint32_t n,
                                   uint32_t *__restrict__ posSample)
{
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= totalBins; g += gridDim.x * blockDim.x) {
    if (g == totalBins) { posSample[g] = n; continue; }
    int32_t lo = 0, hi = nContigs - 1;                              // last contig with posBase[c] <= g
    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (posBase[mid] <= g) lo = mid; else hi = mid - 1; }
    const int32_t c = lo;
    const int32_t pos = (int32_t)((g - posBase[c]) << kPosSampleShift);
    int32_t a = contigFirstMin[c], b = contigFirstMin[c + 1];
    while (a < b) { const int32_t mid = a + ((b - a) >> 1); if (mWpos[mid] < pos) a = mid + 1; else b = mid; }
    posSample[g] = (uint32_t)a;
  }
}

}  // namespace ani
