// l2.hpp — L2 stage: best MinHash/Jaccard placement of a fragment inside one L1 candidate region.
//
// Restates skch::Map::computeL2MappedRegions (src/map/include/computeMap.hpp:418-497), skch::SlideMapper
// (src/map/include/slidingMap.hpp:112-284) and skch::MIIteratorL2::next (src/map/include/MIIteratorL2.hpp:74-96).
//
// The reference keeps an ordered map (query sketch ∪ reference minimizers of the current super-window) with a
// pivot on its s-th smallest key and maintains sharedSketchElements = |top-s(Q ∪ R) ∩ Q ∩ R| incrementally.
// Here the same quantity is maintained WITHOUT an ordered container.  With Q = q_1 < ... < q_s:
//   every reference hash h is either some q_i (rank i, 1-based) or falls in gap g = #{q < h}  (0..s);
//   b[i]  = 1 iff q_i occurs in the window;        n[g] = number of DISTINCT non-query hashes of gap g in the window;
//   G(i)  = i + sum_{g<i} n[g]  is the rank of q_i inside Q ∪ R;  iStar = max{i : G(i) <= s}  (0 if none);
//   shared = sum_{i<=iStar} b[i].
// One window event changes G by exactly one at the affected ranks, so iStar moves by at most one step per event
// and every update is O(1).  Hashes that occur several times inside one window behave as a set in the reference
// (slidingMap.hpp:150-154: REV status; :178: NOOP when a later occurrence re-tagged wposR); with the same-hash
// links over the position-ordered index (index.hpp: DupLinks) this is: an insertion of entry j is effective iff its
// previous near occurrence is < beg, a deletion of entry j is effective iff its next near occurrence is not inside
// the inserted range [.., endIns).
//
// Parallelisation: one lane per candidate (tens of millions of candidates per many-to-many run give the
// parallelism); state words live in a lane-interleaved scratch array so that lane l touches
// scratch[word * laneStride + l].
#pragma once
#include <type_traits>
#include "common.hpp"
#include "index.hpp"

namespace ani {

struct L2Result {
  int32_t best;        // sharedSketchSize of the best placement (0 if the loop never ran)
  int32_t firstPos;    // beginOptimalPos (computeMap.hpp:468)
  int32_t lastPos;     // lastOptimalPos  (:469,:474)
  int32_t entries;     // m_c = reference minimizers in [beg0, last)
  int32_t steps;       // super-window placements evaluated
};

// state word g (0..s):  bit 0 = b[g] (q_g present in window; unused for g = 0), bits 1.. = n[g]
struct L2State {
  uint32_t *w;          // base pointer of this lane's word 0
  size_t stride;        // distance between consecutive words of one lane
  __host__ __device__ __forceinline__ uint32_t get(int g) const { return w[(size_t)g * stride]; }
  __host__ __device__ __forceinline__ void set(int g, uint32_t v) const { w[(size_t)g * stride] = v; }
};

// first index in [lo, hi) of the position-ordered wpos array with wpos >= pos (winSketch.hpp:259-270 restricted
// to the contig's own slice [lo, hi))
__host__ __device__ __forceinline__ int32_t lower_bound_wpos(const int32_t *mWpos, int32_t lo, int32_t hi, int32_t pos)
{
  while (lo < hi) {
    int32_t mid = lo + ((hi - lo) >> 1);
    if (mWpos[mid] < pos) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// rank lookup: returns (idx << 1) | isQ where idx = #{q < h}; if isQ then h == q[idx] (0-based) i.e. rank idx+1
__host__ __device__ __forceinline__ uint32_t q_rank(const uint32_t *q, int s, uint32_t h)
{
  int lo = 0, hi = s;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (q[mid] < h) lo = mid + 1; else hi = mid;
  }
  return ((uint32_t)lo << 1) | (uint32_t)(lo < s && q[lo] == h);
}

struct L2Sim {
  int s, iStar, cStar, shared;
  L2State st;

  __host__ __device__ __forceinline__ void init(int s_, L2State st_)
  {
    s = s_; st = st_; iStar = s_; cStar = 0; shared = 0;
    for (int g = 0; g <= s_; g++) st.set(g, 0u);
  }
  // reference minimizer with hash h enters the window (slidingMap.hpp:137-161 + :231-254), effective insert only
  __host__ __device__ __forceinline__ void insert(uint32_t code)
  {
    const int idx = (int)(code >> 1);
    if (code & 1u) {                       // CPLD: query hash q_{idx+1} gains its reference partner
      const int i = idx + 1;
      st.set(i, st.get(i) | 1u);
      if (i <= iStar) shared++;
    } else {                               // UNIQ: new non-query hash in gap idx
      const int g = idx;
      st.set(g, st.get(g) + 2u);
      if (g < iStar) {
        cStar++;
        if (iStar + cStar > s) {           // q_iStar is pushed out of the s smallest
          shared -= (int)(st.get(iStar) & 1u);
          iStar--;
          cStar -= (int)(st.get(iStar) >> 1);
        }
      }
    }
  }
  // reference minimizer leaves the window (slidingMap.hpp:167-211 + :261-284), effective delete only
  __host__ __device__ __forceinline__ void erase(uint32_t code)
  {
    const int idx = (int)(code >> 1);
    if (code & 1u) {                       // UPD
      const int i = idx + 1;
      st.set(i, st.get(i) & ~1u);
      if (i <= iStar) shared--;
    } else {                               // DEL
      const int g = idx;
      st.set(g, st.get(g) - 2u);
      if (g < iStar) cStar--;
      if (iStar < s) {
        const int ng = (int)(st.get(iStar) >> 1);
        if (iStar + 1 + cStar + ng <= s) { // q_{iStar+1} moves into the s smallest
          cStar += ng;
          iStar++;
          shared += (int)(st.get(iStar) & 1u);
        }
      }
    }
  }
};

// One candidate, sequentially.  [cLo, cHi) = slice of the position-ordered index that belongs to the candidate's contig.
__host__ __device__ inline L2Result l2_candidate(const uint32_t *q, int s,
                                                 const uint32_t *mHash, const int32_t *mWpos,
                                                 const DupLinks &dup,
                                                 int32_t cLo, int32_t cHi, int32_t rangeStart, int32_t rangeEnd,
                                                 int L, int w, int k, L2State st)
{
  L2Result r; r.best = 0; r.firstPos = 0; r.lastPos = 0; r.steps = 0;
  const int32_t cmw = L - (w - 1) - (k - 1);                                     // computeMap.hpp:428
  int32_t beg = lower_bound_wpos(mWpos, cLo, cHi, rangeStart);                   // :424
  int32_t end = lower_bound_wpos(mWpos, cLo, cHi, mWpos[beg] + cmw);             // :431
  const int32_t last = lower_bound_wpos(mWpos, cLo, cHi, rangeEnd + L);          // :435
  r.entries = last - beg;
  L2Sim sim; sim.init(s, st);
  for (int32_t j = beg; j < end; j++)                                            // :448 first super-window
    if (!dup_flag(dup, (uint32_t)j) || dup_prev(dup, (uint32_t)j) < beg) sim.insert(q_rank(q, s, mHash[j]));
  int32_t pb = beg, pe = end;
  int32_t pos = mWpos[beg];
  while (end < last) {                                                           // :455
    if (pb != beg) {                                                             // :461 delete_ref(prev_beg)
      const int32_t nx = dup_flag(dup, (uint32_t)pb) ? dup_next(dup, (uint32_t)pb) : -1;
      if (!(nx >= 0 && nx < pe)) sim.erase(q_rank(q, s, mHash[pb]));
    }
    if (pe != end) {                                                             // :465 insert_ref(prev_end)
      if (!dup_flag(dup, (uint32_t)pe) || dup_prev(dup, (uint32_t)pe) < beg) sim.insert(q_rank(q, s, mHash[pe]));
    }
    const int32_t wb = mWpos[beg];
    if (sim.shared > r.best) { r.best = sim.shared; r.firstPos = wb; r.lastPos = wb; }   // :468-476
    else if (sim.shared == r.best) r.lastPos = wb;
    r.steps++;
    pb = beg; pe = end;
    // MIIteratorL2::next (MIIteratorL2.hpp:74-96)
    const int32_t d1 = mWpos[beg + 1] - pos;
    const int32_t d2 = mWpos[end] - (pos + cmw - 1);
    const int32_t adv = d1 < d2 ? d1 : d2;
    pos += adv;
    if (adv == d1) beg++;
    if (adv == d2) end++;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// Kernels.
//
// General kernel (k_l2): one lane per candidate, everything on the fly (rank lookups by binary search in global
// memory, state words in a lane-interleaved global scratch array).  Any sketch size, any region length.  It is the
// safety net: candidates are routed here through `list`.
//
// Fast path (k_l2_ranges -> k_l2_codes -> k_l2_sim), taken when s <= kL2FastMaxS and the candidate's index range is
// <= kL2FastMaxEntries long:
//   k_l2_ranges  one lane per candidate: the three searchIndex() calls of computeMap.hpp:424-436 and the length of the
//                candidate's event stream
//   k_l2_codes   one workgroup per fragment: the fragment sketch sits in LDS; every reference minimizer of every
//                candidate range is ranked against it once and written as its (at most) two window events — it enters the
//                super-window, it leaves it — at their places in the candidate's MERGED event stream.  The merged order of the
//                events depends on the reference positions only and is precomputed per index entry (index.hpp:
//                k_index_window_links), so a place is a few integer operations.  One event = 16 bits:
//                  bit 0 = hash is a query hash, bits 1..9 = gap (or rank-1), bit 10 = a same-hash neighbour may share a
//                  super-window with this entry (nearDup, precomputed at index build), bit 11 = insert (else delete),
//                  bit 12 = do not evaluate the window after this event (first super-window still filling / the insert of
//                  the same position follows)
//   k_l2_sim     one lane per candidate: applies its events in order, one per loop pass, 8 events per 16-byte load, no position
//                arithmetic and no second cursor.  State in LDS: one byte per sketch rank (7-bit gap counter + presence bit),
//                byte-interleaved over the wave.  Entries flagged nearDup consult the same-hash links (exact set semantics,
//                slidingMap.hpp:150-154,:178).  A gap counter that would pass 127 sends the candidate to k_l2.
// ------------------------------------------------------------------------------------------------
struct L2Args {
  // candidates (SoA)
  const int32_t *candFrag, *candSeq, *candStart, *candEnd;
  int32_t nCand;
  // fragment sketches
  const uint32_t *qPool; const uint32_t *fragOff; const int32_t *fragS;
  // reference index, position order
  const uint32_t *mHash; const int32_t *mWpos; DupLinks dup;       // same-hash links of near duplicates (index.hpp)
  const uint32_t *mWin;            // window links + flags per entry (index.hpp: k_index_window_links)
  const int32_t *contigFirstMin;   // [nContigs+1]
  const uint32_t *posBase, *posSample;   // sampled position index (index.hpp: k_index_pos_sample)
  int rankShift;                   // k_l2_codes rank table: 2048 linear buckets of 2^rankShift hashes from 0; minimizer hashes are minima of w
                                   // k-mer hashes, ~96 % of them lie below 2^32 * 3 / w, so the table covers [0, 2^(32 - floor(log2 w) + 1))
  int L, w, k;
  // lane-interleaved scratch of the general kernel: (maxS+1) words per lane
  uint32_t *scratch; size_t laneStride;
  // outputs
  int32_t *outBest, *outFirst, *outLast;
  unsigned long long *sumEntries, *sumSteps, *sumQ;
};

static __global__ __launch_bounds__(kTPB) void k_l2(L2Args a, const int32_t *__restrict__ list, int32_t listCount, int32_t listBase)
{
  const int32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t li = listBase + lane;
  unsigned long long e = 0, st = 0, sq = 0;
  if (li < listCount && (size_t)lane < a.laneStride) {
    const int32_t c = list ? list[li] : li;
    const int32_t f = a.candFrag[c];
    const int32_t seq = a.candSeq[c];
    L2State state; state.w = a.scratch + lane; state.stride = a.laneStride;
    L2Result r = l2_candidate(a.qPool + a.fragOff[f], a.fragS[f], a.mHash, a.mWpos, a.dup,
                              a.contigFirstMin[seq], a.contigFirstMin[seq + 1], a.candStart[c], a.candEnd[c],
                              a.L, a.w, a.k, state);
    a.outBest[c] = r.best; a.outFirst[c] = r.firstPos; a.outLast[c] = r.lastPos;
    e = (unsigned long long)r.entries; st = (unsigned long long)r.steps; sq = (unsigned long long)a.fragS[f];
  }
  // counters for the algorithmic-byte figure: one atomic per wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { e += __shfl_down(e, d); st += __shfl_down(st, d); sq += __shfl_down(sq, d); }
  if ((threadIdx.x & 63) == 0 && (e | st | sq)) { atomicAdd(stat_slot(a.sumEntries), e); atomicAdd(stat_slot(a.sumSteps), st);
    atomicAdd(stat_slot(a.sumQ), sq); }
}

// ---------------------------------------------------------------- fast path
constexpr int kL2FastMaxS = 319;
constexpr int kL2FastMaxEntries = 16384;
// LDS state of the simulation kernel: one 8-bit field per index g = 0..MAXS,
//     field[g] = n[g] << 1 | b[g+1]        (7-bit gap counter + presence bit of the query hash that closes the gap)
// so an event with code index idx — query hash of rank idx+1 or non-query hash of gap idx — touches exactly field[idx], and the
// pivot logic, which needs n[j] together with b[j+1], reads exactly field[j].  Fields are byte-interleaved over the wave
// (field g of lane l at byte g*64 + l, see l2_field_off; the conflict-free word-interleaved alternative costs three more
// address instructions per access and measured the same kernel time).
//   class A  s <= 255: 64 words = 256 B per lane -> 16 KiB per wave, 10 waves per CU
//   class B  s <= 319: 80 words = 320 B per lane -> 20 KiB per wave,  8 waves per CU
// A counter that would pass 127 sends the candidate to the general kernel.
template <int MAXS>
struct L2Geom {
  static constexpr int kMaxS = MAXS;
  static constexpr int kStateWords = (MAXS + 1 + 3) / 4;
  static constexpr int kWords = kStateWords;
};
using L2GeomA = L2Geom<255>;
using L2GeomB = L2Geom<319>;
constexpr int kL2SimTPB = 64;

struct L2Range { int32_t beg0, end0, last, nEvents; };    // nEvents = length of the candidate's event stream

struct L2FastArgs {
  L2Args g;
  int32_t c0, c1;                  // candidate chunk [c0, c1)
  L2Range *ranges;                 // [c1-c0]
  int32_t *codeCount;              // [c1-c0] 16-bit events per candidate rounded up to 8, 0 = not on the fast path (or nothing to simulate)
  const uint32_t *codeOff;         // [c1-c0] exclusive scan of codeCount (in entries)
  uint32_t *codes;
  int32_t *slowFlag;               // [c1-c0] 1 = take the general kernel
  const uint32_t *fragCandOff;     // ordered candidate offset per position in the fragments' processing order [nFrag]
  const int32_t *fragOrder;        // processing order (nullptr: fragment ascending)
  int32_t nFrag, fragBase, nFragChunk;   // first fragment of the chunk, fragments that own its candidates
  int32_t allowFast;               // test knob ANI_TEST_L2_PATH: 0 = everything to the general kernel, 2 = everything to class B
};

// lower_bound_wpos over a contig's slice [.., cHi) through the sampled position index: the answer lies inside the target's bin
__device__ __forceinline__ int32_t l2_search_pos(const L2Args &g, uint32_t pb, int32_t nb, int32_t cHi, int32_t pos)
{
  int32_t bin = (pos < 0 ? 0 : pos) >> kPosSampleShift; bin = bin < nb ? bin : nb - 1;
  const int32_t lo = (int32_t)g.posSample[pb + (uint32_t)bin];
  const int32_t hi = bin + 1 < nb ? (int32_t)g.posSample[pb + (uint32_t)bin + 1] : cHi;
  return lower_bound_wpos(g.mWpos, lo, hi, pos);
}

static __global__ __launch_bounds__(kTPB) void k_l2_ranges(L2FastArgs a)
{
  const int32_t c = a.c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.c1) return;
  const int32_t i = c - a.c0;
  const int32_t seq = a.g.candSeq[c];
  const int32_t cLo = a.g.contigFirstMin[seq], cHi = a.g.contigFirstMin[seq + 1];
  const int32_t cmw = a.g.L - (a.g.w - 1) - (a.g.k - 1);
  L2Range r;
  // lower_bound_wpos over the contig's slice through the sampled position index: the answer lies inside the target's bin
  const uint32_t pb = a.g.posBase[seq]; const int32_t nb = (int32_t)(a.g.posBase[seq + 1] - pb);
  auto search = [&](int32_t pos) -> int32_t { return l2_search_pos(a.g, pb, nb, cHi, pos); };
  r.beg0 = search(a.g.candStart[c]);
  const int32_t wposBeg0 = a.g.mWpos[r.beg0];
  r.end0 = search(wposBeg0 + cmw);
  r.last = search(a.g.candEnd[c] + a.g.L);
  const int32_t m = r.last - r.beg0;
  const int32_t s = a.g.fragS[a.g.candFrag[c]];
  const bool fast = a.allowFast && s >= 1 && s <= kL2FastMaxS && m >= 1 && m <= kL2FastMaxEntries;
  // event stream: the first super-window's entries [beg0, end0), then one insert per entry of [end0, last-1) and one delete per
  // entry up to the last one that leaves before entry last-1 would enter (which ends the loop, computeMap.hpp:455)
  int32_t nEv = 0;
  if (fast && r.end0 < r.last) {
    const int32_t lastDel = (r.last - 1) - (int32_t)(a.g.mWin[r.last - 1] & kWinMask) - 2;     // deletes with D_j <= I_(last-1)
    const int32_t nDel = lastDel >= r.beg0 ? lastDel - r.beg0 + 1 : 0;
    nEv = (r.end0 - r.beg0) + (r.last - 1 - r.end0) + nDel;
  }
  r.nEvents = nEv;
  a.ranges[i] = r;
  if (fast && nEv == 0) {                            // the window never fits: the loop of :455 does not run, nothing is evaluated
    a.g.outBest[c] = 0; a.g.outFirst[c] = 0; a.g.outLast[c] = 0;
    a.codeCount[i] = 0; a.slowFlag[i] = 16;          // done
    return;
  }
  a.codeCount[i] = fast ? ((nEv + 8) & ~7) : 0;      // 16-bit events, padded to 16-byte blocks with at least one pad slot (k_l2_codes)
  a.slowFlag[i] = fast ? ((s <= L2GeomA::kMaxS && a.allowFast != 2) ? 0 : 4) : 1;
}

constexpr uint32_t kL2DupBit = 1u << 10, kL2InsBit = 1u << 11, kL2NoEvalBit = 1u << 12;
constexpr int kL2DeltaShift = 13;         // bits 13..15: the event's signed change of its field (+2 / +1 insert, -1 / -2 delete; query hash = 1)
static_assert((kWinDupBit >> 21) == kL2DupBit && (kWinMoreBit >> 18) == kL2NoEvalBit, "flag bits of the window links shift into the event code");
constexpr int kL2RankBuckets = 2048;
// rank-table bucket of a hash: linear buckets over the low end of the range, where minimizer hashes live (see L2Args::rankShift)
__device__ __forceinline__ int l2_rank_bucket(uint32_t h, int sh) { const uint32_t b = h >> sh;
  return (int)(b < (uint32_t)(kL2RankBuckets - 1) ? b : (uint32_t)(kL2RankBuckets - 1)); }

// The fragment sketch and its rank table in LDS (k_l2_codes).  st2[b] = #{q : bucket(q) < b} | entries of bucket b
// << 16.  Counted and scanned (one LDS atomic per sketch hash on 16-bit counters packed in pairs, eight buckets per thread, one
// workgroup scan) — the first form walked, per sketch hash, the buckets up to the next hash: a loop as long as the longest gap among
// the 64 hashes of a wave.  cnt2: kL2RankBuckets / 2 dwords of scratch (free again when the function returns).  Ends with a barrier.
__device__ __forceinline__ void l2_rank_table(const uint32_t *__restrict__ q, int s, int sh, uint32_t *qs, uint32_t *st2, uint32_t *cnt2, int *scanScratch)
{
  static_assert(kL2FastMaxS <= 2 * kTPB && kL2RankBuckets == 8 * kTPB, "two sketch hashes and eight buckets per thread");
  { uint4 z; z.x = z.y = z.z = z.w = 0u; ((uint4 *)cnt2)[threadIdx.x] = z; }
  const int i0 = (int)threadIdx.x, i1 = (int)threadIdx.x + kTPB;
  const uint32_t q0 = i0 < s ? q[i0] : 0xffffffffu, q1 = i1 < s ? q[i1] : 0xffffffffu;
  if (i0 < s) qs[i0] = q0;
  if (i1 < s) qs[i1] = q1;
  if (threadIdx.x < 2) qs[s + threadIdx.x] = 0xffffffffu;             // the two-entry probe may read one or two slots past the sketch
  block_barrier();
  if (i0 < s) { const int b = l2_rank_bucket(q0, sh); atomicAdd(&cnt2[b >> 1], 1u << ((b & 1) << 4)); }
  if (i1 < s) { const int b = l2_rank_bucket(q1, sh); atomicAdd(&cnt2[b >> 1], 1u << ((b & 1) << 4)); }
  block_barrier();
  {
    const uint4 c4 = ((const uint4 *)cnt2)[threadIdx.x];               // buckets 8 t .. 8 t + 7
    const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
    uint32_t n[8], loc[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { n[j] = (j & 1) ? (cw[j >> 1] >> 16) : (cw[j >> 1] & 0xffffu); loc[j] = sum; sum += n[j]; }
    int tot;
    const uint32_t base = (uint32_t)block_excl_scan((int)sum, scanScratch, &tot);
    uint4 o0, o1;
    o0.x = (base + loc[0]) | (n[0] << 16); o0.y = (base + loc[1]) | (n[1] << 16); o0.z = (base + loc[2]) | (n[2] << 16); o0.w = (base + loc[3]) | (n[3] << 16);
    o1.x = (base + loc[4]) | (n[4] << 16); o1.y = (base + loc[5]) | (n[5] << 16); o1.z = (base + loc[6]) | (n[6] << 16); o1.w = (base + loc[7]) | (n[7] << 16);
    ((uint4 *)st2)[2 * threadIdx.x] = o0; ((uint4 *)st2)[2 * threadIdx.x + 1] = o1;
  }
  block_barrier();
}
// Rank of a reference hash in the sketch: (entries of the sketch below h) << 1 | (h is a sketch hash).  The two sketch entries at the
// bucket's start decide it unless the bucket holds more than two and both are below h (`deep`: the caller's wave vote sends those to
// l2_rank_deep); entries behind the bucket's own belong to later buckets, i.e. are larger than any hash of this bucket, so they need
// no "is it in the bucket" test (the sketch is followed by two 0xffffffff sentinels).
__device__ __forceinline__ uint32_t l2_rank_fast(const uint32_t *qs, const uint32_t *st2, int sh, uint32_t h, uint32_t &deep)
{
  const uint32_t sp = st2[l2_rank_bucket(h, sh)];                   // first sketch entry of the bucket | entries in it << 16
  const uint32_t lo = sp & 0xffffu;
  const uint32_t q0 = qs[lo], q1 = qs[lo + 1];
  deep |= (uint32_t)(q1 < h) & (uint32_t)(sp > 0x2ffffu);
  return ((lo + (uint32_t)(q0 < h) + (uint32_t)(q1 < h)) << 1) | (uint32_t)((q0 == h) | (q1 == h));
}
__device__ __forceinline__ uint32_t l2_rank_deep(const uint32_t *qs, const uint32_t *st2, int sh, int s, uint32_t h, uint32_t rk)
{
  const uint32_t sp = st2[l2_rank_bucket(h, sh)];
  int lo = (int)(sp & 0xffffu), hi = lo + (int)(sp >> 16);
  if (hi - lo > 2) {
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (qs[mid] < h) lo = mid + 1; else hi = mid; }
    rk = ((uint32_t)lo << 1) | (uint32_t)(lo < s && qs[lo] == h);    // == q_rank(qs, s, h)
  }
  return rk;
}

// One workgroup per fragment: the fragment sketch and its rank table sit in LDS; each of the four waves takes every fourth
// candidate of the fragment and works on its own from there (no workgroup barrier after the set-up).
// What bounds it (SQ counters, profiles/r02s_sq_wait.txt): the two events of an entry land hundreds of slots apart, and a scattered
// 2-byte global store costs the memory pipeline about one cycle per lane — 8 such stores per lane and pass were the kernel's time,
// whatever its instruction count or the bytes moved.  So a wave assembles its candidate's stream in a private LDS window (scattered
// LDS writes cost a few cycles per wave) and writes it out in 16-byte pieces, 1 KiB per wave instruction; only streams beyond the
// window (ranges of thousands of entries) are stored directly.
// A lane ranks four entries per pass, 64 apart (every load is one contiguous 256-byte run per wave); the loads of the next pass or
// candidate are in flight while the current one is ranked.  The rank lookup has no loop: 2048 buckets for ~240 sketch hashes, so a
// bucket holds at most two of them except in rare cases, which a wave vote sends to a binary search.
constexpr int kL2StageEvents = 2048;        // events per wave window (4 KiB)
constexpr int kL2CandBatch = 40;            // candidates whose descriptors are fetched at once
static __global__ __launch_bounds__(kTPB) void k_l2_codes(L2FastArgs a)
{
  __shared__ uint32_t qs[kL2FastMaxS + 2];
  __shared__ __attribute__((aligned(16))) uint32_t st2[kL2RankBuckets];
  __shared__ __attribute__((aligned(16))) uint16_t stageAll[(kTPB / kWave) * kL2StageEvents];
  __shared__ L2Range cRange[kL2CandBatch];         // the candidates' descriptors, fetched side by side (nEvents < 0: no stream)
  __shared__ uint64_t cOff[kL2CandBatch];
  __shared__ int cRangeScan[8];                    // scratch of the workgroup scan
#ifdef ANI_L2C_PAD_WORDS
  // occupancy experiment (profiles/r06h_l2_codes_occupancy_ab.txt): dead LDS, five workgroups per CU instead of six
  __shared__ volatile uint32_t padLds[ANI_L2C_PAD_WORDS];
  if (threadIdx.x == 0) padLds[blockIdx.x & (ANI_L2C_PAD_WORDS - 1)] = 1u;
#endif
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Consecutive fragments of a query map to overlapping
  // reference ranges, so XCD x takes a contiguous eighth of the chunk's fragments: neighbours share their reference reads in L2.
  const int32_t per = (int32_t)(gridDim.x >> 3);
  const int32_t fl = (int32_t)(blockIdx.x & 7) * per + (int32_t)(blockIdx.x >> 3);
  if (fl >= a.nFragChunk) return;
  const int32_t fi = a.fragBase + fl;              // position in the processing order
  const int32_t f = a.fragOrder ? a.fragOrder[fi] : fi;
  const int32_t s = a.g.fragS[f];
  int32_t cA = (int32_t)a.fragCandOff[fi], cB = (fi + 1 < a.nFrag) ? (int32_t)a.fragCandOff[fi + 1] : a.g.nCand;
  if (cA < a.c0) cA = a.c0;
  if (cB > a.c1) cB = a.c1;
  if (cA >= cB || s < 1 || s > kL2FastMaxS) return;
  const uint32_t *q = a.g.qPool + a.g.fragOff[f];
  // A candidate's descriptor is three dependent memory round trips away from its first hash (count -> range -> hashes); walked
  // candidate by candidate that chain was most of a workgroup's life.  All descriptors of the fragment are fetched here, together
  // with the sketch.
  auto fetch_cands = [&](int32_t base) {
    const int32_t c = base + (int32_t)threadIdx.x;
    if ((int)threadIdx.x < kL2CandBatch && c < cB) {
      const int32_t i = c - a.c0;
      L2Range r = a.ranges[i];
      if (a.codeCount[i] == 0) r.nEvents = -1;
      cRange[threadIdx.x] = r; cOff[threadIdx.x] = a.codeOff[i];
    }
  };
  fetch_cands(cA);
  const int sh = a.g.rankShift;
  l2_rank_table(q, s, sh, qs, st2, (uint32_t *)stageAll /* kL2RankBuckets / 2 dwords of the windows, which are free until the waves start */, cRangeScan);
  // from here on the waves go their own ways; the counters are dead, the windows are free

  const int lane = threadIdx.x & (kWave - 1), wv = wave_uniform((int32_t)(threadIdx.x >> 6));
  uint16_t *stage = stageAll + wv * kL2StageEvents;
  // Work items of this wave = (candidate, pass of 4 * 64 entries), in order.  Everything in an item is wave-uniform and kept in
  // scalar registers (wave_uniform): the item loop then branches on scalars and the per-entry bounds tests take scalar operands.
  struct Item {
    int32_t c; uint32_t jb;                          // candidate, first entry of the pass
    char *ob; const char *hb, *wb;                   // wave-uniform bases: event stream, hashes, window links (32-bit byte offsets per lane)
    uint32_t m, nInit, nInsAll, nDel, dump; bool staged, lastPass;
  };
  constexpr uint32_t kPass = 4u * kWave;
  int32_t batch0 = cA, batch1 = cA + kL2CandBatch < cB ? cA + kL2CandBatch : cB;      // candidates whose descriptors are in LDS
  auto open_cand = [&](int32_t c, Item &it) -> bool {          // first candidate >= c of this wave (within the batch) that has a stream
    for (; c < batch1; c += kTPB / kWave) {
      L2Range r = cRange[c - batch0];
      r.nEvents = wave_uniform(r.nEvents);
      if (r.nEvents < 0) continue;
      r.beg0 = wave_uniform(r.beg0); r.end0 = wave_uniform(r.end0); r.last = wave_uniform(r.last);
      it.c = c; it.jb = 0;
      it.ob = (char *)((uint16_t *)a.codes + wave_uniform(cOff[c - batch0]));
      it.hb = (const char *)(a.g.mHash + r.beg0); it.wb = (const char *)(a.g.mWin + r.beg0);
      it.m = (uint32_t)(r.last - r.beg0);
      it.nInit = (uint32_t)(r.end0 - r.beg0); it.nInsAll = it.m - 1;          // inserts (first window included) are the entries [0, m-1)
      it.nDel = (uint32_t)r.nEvents - it.nInsAll;                             // deletes are the entries [0, nDel)
      it.dump = (uint32_t)r.nEvents;                                          // pad slot
      it.staged = (uint32_t)r.nEvents < (uint32_t)kL2StageEvents;
      it.lastPass = kPass >= it.m;
      return true;
    }
    return false;
  };
  auto next_item = [&](const Item &cur, Item &nx) -> bool {
    if (!cur.lastPass) { nx = cur; nx.jb = cur.jb + kPass; nx.lastPass = nx.jb + kPass >= nx.m; return true; }
    return open_cand(cur.c + kTPB / kWave, nx);
  };
  auto load_item = [&](const Item &it, uint32_t (&h)[4], uint32_t (&wl)[4]) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t x = it.jb + lane + e * kWave;
      const uint32_t off = (x < it.m ? x : it.m - 1) * 4u;
      h[e] = *(const uint32_t *)(it.hb + off); wl[e] = *(const uint32_t *)(it.wb + off);
    }
  };
  Item cur, nxt;
  for (;;) {
  bool have = open_cand(batch0 + wv, cur);
  uint32_t h[4], wl[4], hn[4], wn[4];
  if (have) load_item(cur, h, wl);
  while (have) {
    const bool haveNext = next_item(cur, nxt);
    if (haveNext) load_item(nxt, hn, wn);
    // A pass ranks up to four entries per lane; the last pass of a candidate takes only as many as are left (a candidate of the
    // benchmark has ~600 entries: 256 + 256 + 92 — at four per lane whatever is left, a fifth of all the work was done on entries
    // beyond the range).  E is wave-uniform: a scalar branch picks the variant.
    auto pass_body = [&](auto eTag) {
      constexpr int E = decltype(eTag)::value;
      // Rank = entries of the sketch below h.  The two sketch entries at the bucket's start decide it unless the bucket holds more
      // than two and both are below h: entries behind the bucket's own belong to later buckets, i.e. are larger than any hash of
      // this bucket, so they need no "is it in the bucket" test (the sketch is followed by two 0xffffffff sentinels).
      uint32_t rk[4];
      uint32_t deep = 0;
  #pragma unroll
      for (int e = 0; e < E; e++) rk[e] = l2_rank_fast(qs, st2, sh, h[e], deep);
      if (__any(deep != 0)) {
  #pragma unroll
        for (int e = 0; e < E; e++) rk[e] = l2_rank_deep(qs, st2, sh, s, h[e], rk[e]);
      }
      // Two events per entry, stored without control flow: an event that does not exist (entries beyond the range, the never-inserted
      // last entry, entries that never leave) goes to the pad slot behind the stream (k_l2_ranges reserves one).
      uint32_t pi[4], pd[4]; uint16_t ci[4], cdl[4];
      const uint32_t dumpSlot = cur.dump;
  #pragma unroll
      for (int e = 0; e < E; e++) {
        const uint32_t x = cur.jb + lane + e * kWave;
        const uint32_t cd = rk[e] | ((wl[e] >> 21) & kL2DupBit);                          // kWinDupBit (bit 31) -> kL2DupBit (bit 10)
        // field change of the event, 3-bit two's complement: insert +2 (010), of a query hash +1 (001); delete -2 (110) / -1 (111)
        const uint32_t mq = 0u - (rk[e] & 1u);                                            // all ones for a query hash
        // insert of entry x: after the inserts of the entries before it and the deletes of the entries up to x - B - 2
        const int32_t db = (int32_t)x - (int32_t)(wl[e] & kWinMask) - 1;                 // deletes that precede it
        pi[e] = x < cur.nInsAll ? x + (uint32_t)(db < 0 ? 0 : db) : dumpSlot;
        ci[e] = (uint16_t)((cd | kL2InsBit | (2u << kL2DeltaShift) | ((int32_t)x < (int32_t)cur.nInit - 1 ? kL2NoEvalBit : 0u)) ^ (mq & (3u << kL2DeltaShift)));
        // delete of entry x: after the deletes of the entries before it and the inserts of the entries below x + A (at least the first
        // window's)
        const uint32_t ib = x + ((wl[e] >> kWinShiftA) & kWinMask);
        pd[e] = x < cur.nDel ? x + (ib < cur.nInit ? cur.nInit : ib) : dumpSlot;
        // kWinMoreBit (bit 30) -> kL2NoEvalBit (bit 12)
        cdl[e] = (uint16_t)(cd | ((wl[e] >> 18) & kL2NoEvalBit) | (6u << kL2DeltaShift) | (mq & (1u << kL2DeltaShift)));
      }
      if (cur.staged) {                                                                   // wave-uniform choice
  #pragma unroll
        for (int e = 0; e < E; e++) { stage[pi[e]] = ci[e]; stage[pd[e]] = cdl[e]; }
      } else {
  #pragma unroll
        for (int e = 0; e < E; e++) { *(uint16_t *)(cur.ob + pi[e] * 2u) = ci[e]; *(uint16_t *)(cur.ob + pd[e] * 2u) = cdl[e]; }
      }
    };
    {
      const uint32_t left = cur.m - cur.jb;          // entries from the first of this pass to the end of the range (>= 1)
      if (left > 3u * kWave) pass_body(std::integral_constant<int, 4>());
      else if (left > 2u * kWave) pass_body(std::integral_constant<int, 3>());
      else if (left > (uint32_t)kWave) pass_body(std::integral_constant<int, 2>());
      else pass_body(std::integral_constant<int, 1>());
    }
    if (cur.staged && cur.lastPass) {                // the candidate's stream is complete in the window: write it out
      ANI_WAVE_SYNC();
      const uint32_t nOut = (cur.dump + 8u) & ~7u;   // = codeCount: the stream and its pad, whole 16-byte pieces
      for (uint32_t o = (uint32_t)lane * 8u; o < nOut; o += kWave * 8u) *(uint4 *)(cur.ob + o * 2u) = *(const uint4 *)(stage + o);
      ANI_WAVE_SYNC();                               // the window is reused by the next candidate
    }
    have = haveNext;
    if (have) {
      cur = nxt;
#pragma unroll
      for (int e = 0; e < 4; e++) { h[e] = hn[e]; wl[e] = wn[e]; }
    }
  }
  if (batch1 >= cB) break;                         // workgroup-uniform; fragments with more candidates than a batch are rare
  block_barrier();
  fetch_cands(batch1);
  batch0 = batch1; batch1 = batch0 + kL2CandBatch < cB ? batch0 + kL2CandBatch : cB;
  block_barrier();
  }
}

// One window event, written without control flow (selects only).  INS: the entry enters the window (slidingMap.hpp:137-161
// + :231-254), otherwise it leaves (:167-211 + :261-284).  Both LDS reads (the entry's own field and the field next to the
// pivot) are issued up front; `on` = false turns the event into a no-op.  Field g of this lane sits at F[g << 6]
// (byte-interleaved over the wave: one shift to address it; the only bank conflicts are between the four lanes of a dword
// column whose g differ by a multiple of 4 — the LDS pipe has an order of magnitude of slack under the VALU work of a step).
struct L2Regs { int s, iStar, tot, shared; uint32_t ovfAcc; };   // tot = iStar + (non-query hashes below the pivot) = rank of q_iStar in the union
constexpr bool kL2ByteInterleave = true;
__device__ __forceinline__ int l2_field_off(int g) { return kL2ByteInterleave ? (g << 6) : (((g >> 2) << 8) + (g & 3)); }

// The kernel is VALU-bound, so the event is applied with as few vector instructions as the arithmetic allows: the field change comes
// ready-made in the event code; a counter that passes 127 is not intercepted but noticed afterwards (every new field value is OR-ed
// into ovfAcc; bit 8 = a byte overflowed — the lane then computes garbage, harmlessly: iStar stays in [0, s] by construction, and the
// candidate is redone by the general kernel); tot is carried instead of being re-added from two registers.
__device__ __forceinline__ void l2_apply(uint8_t *F, L2Regs &r, uint32_t code, bool INS, bool on)
{
  const bool isQ = (code & 1u) != 0;
  const int idx = (int)((code >> 1) & 0x1ffu);
  const int d0 = ((int)(code << (32 - kL2DeltaShift - 3))) >> 29;     // sign-extended 3-bit field
  const int d = on ? d0 : 0;
  const int sg = INS ? 1 : -1;
  int j = r.iStar - (INS ? 1 : 0); j = j < 0 ? 0 : j;               // pivot-adjacent field: n[j], b[j+1]
  uint8_t *pOwn = F + l2_field_off(idx);
  const int own = *pOwn;
  int fj = F[l2_field_off(j)];
  const int nw = own + d;                                            // counter lives in bits 1..7, presence in bit 0
  r.ovfAcc |= (uint32_t)nw;
  *pOwn = (uint8_t)nw;
  fj += (idx == j) ? d : 0;                                          // the event itself changed field[j]
  const int c1 = (fj >> 1) + 1;
  const bool lt = idx < r.iStar;                                     // rank idx+1 <= iStar  <=>  gap idx < iStar
  const int t = (on && lt) ? sg : 0;
  r.shared += isQ ? t : 0;
  r.tot += isQ ? 0 : t;
  // insert: q_iStar leaves the s smallest iff rank idx+1 <= iStar and iStar + cStar > s;
  // delete: q_{iStar+1} joins them iff iStar < s and iStar + cStar + 1 + n[iStar] <= s.  One comparison against s serves both.
  const int reach = INS ? r.tot : r.tot + c1;
  const bool over = reach > r.s;
  // insert: lt; delete: iStar < s (two selects and one comparison: cheaper than selecting between two flags)
  const bool room = (INS ? idx : r.iStar) < (INS ? r.iStar : r.s);
  const bool mv = on && !isQ && room && (over == INS);
  const int mvm = -(int)mv;                                          // all ones if the pivot moves (masks, not selects: no branch around the rest)
  const int mone = sg & mvm;                                         // insert: -1 on everything, delete: +1
  r.iStar -= mone;
  r.shared -= mone & -(fj & 1);
  r.tot -= (INS ? c1 : -c1) & mvm;                                   // the pivot moved over q and the n[] non-query hashes next to it
}

// slowFlag protocol: 0 = class A (pending or done), 4 = class B pending (s in 256..319), 8 = class B done, 16 = done without a
//                    simulation (the window never fits the range), 1 = outside every fast-path limit,
//                    3 = a gap counter overflowed -> general kernel
// `list` (optional): candidate ids to run, densely packed so that the few class-B candidates fill whole waves instead of
// leaving one busy lane in every wave; *listCount is read on the device (no host round trip).
//
// The kernel is instruction-issue-bound, so the loop carries nothing but the events: 8 events arrive per 16-byte load (the next
// block is in flight while the current one is applied), an event is a register field — no ring, no position arithmetic, no
// choice between two cursors — and control flow is wave-uniform: a lane whose stream has ended idles (`on` = false) until the
// longest stream of the wave is done (lanes are ordered by stream length), the rare same-hash-neighbour look-ups sit behind a
// wave vote.  Positions are not tracked at all: the window start of an evaluation is "entry number delCount", and the two
// positions the result needs are read from the index when the stream is done.
// eight event codes in four words.  (A vector value, not HIP's uint4 struct: a conditional copy of a whole uint4 made the compiler
// keep the block in scratch memory and store it there once per block — 29 GB of writes per benchmark step that nothing ever read
// back — and member-wise assignment split the 16-byte load into four.)
typedef uint32_t L2Block __attribute__((vector_size(16)));
__device__ __forceinline__ L2Block l2_block_zero() { const L2Block b = {0u, 0u, 0u, 0u}; return b; }
__device__ __forceinline__ L2Block l2_block_load(bool on, const uint4 *p, L2Block otherwise)
{
  if (on) otherwise = *(const L2Block *)p;
  return otherwise;
}
template <class G>
static __global__ __launch_bounds__(kL2SimTPB) void k_l2_sim(L2FastArgs a, const int32_t *__restrict__ list, const unsigned int *__restrict__ listCount)
{
  __shared__ uint32_t lds[(kL2SimTPB / kWave) * G::kWords * kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  uint32_t *W = lds + wv * (G::kWords * kWave);                       // this wave's LDS
  uint8_t *F = (uint8_t *)W + (kL2ByteInterleave ? lane : 4 * lane);   // field g of this lane: F[l2_field_off(g)]
  uint32_t *S = W + lane;                                             // dword x of this lane: S[x * kWave]
  const int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t c = a.c0 + slot;
  if (list) {
    const unsigned int n = *listCount;
    if ((unsigned int)(blockIdx.x * blockDim.x) >= n) return;         // whole workgroup beyond the list
    c = (unsigned int)slot < n ? list[slot] : a.c1;
  }
  // four lanes share each state dword, so the whole wave clears the state, whether or not a lane has a candidate
#pragma unroll
  for (int x = 0; x < G::kStateWords; x++) S[x * kWave] = 0u;
  unsigned long long cntE = 0, cntS = 0, cntQ = 0;
  const int myFlag = c < a.c1 ? a.slowFlag[c - a.c0] : 1;
  const bool mine = (G::kMaxS == 255) ? (myFlag == 0) : (myFlag == 4);
  const int32_t i = mine ? c - a.c0 : 0;
  L2Range r; r.beg0 = 0; r.end0 = 0; r.last = 0; r.nEvents = 0;
  if (mine) r = a.ranges[i];
  const int n = r.nEvents;                                            // 0 for a lane without a candidate: it idles
  L2Regs R; R.s = mine ? a.g.fragS[a.g.candFrag[c]] : 1; R.iStar = R.s; R.tot = R.s; R.shared = 0; R.ovfAcc = 0;
  const uint4 *p = (const uint4 *)((const uint16_t *)a.codes + (mine ? a.codeOff[i] : 0u));      // idle lanes read the head of the buffer
  const int nBlk = (n + 7) >> 3;
  int best = 0, begAtBest = -1, begAtLast = -1, steps = 0, delCount = 0;
  // A lane without a candidate of this class (class-B and general-path candidates sit in the same length-ordered list: about every
  // tenth lane) reads no stream: its words stay zero — event 0 is "delete a non-query hash from field 0 and evaluate", harmless on
  // its own cleared state — so that it counts as PLAIN below instead of sending its whole wave through the general form for good.
  L2Block cur = l2_block_load(mine, p, l2_block_zero());
  // one block = 8 events.  PLAIN: every lane of the wave has all 8 events and none of them is flagged nearDup (two wave votes per
  // block decide) — then an event needs no bounds test and no look-up path; otherwise the block takes the general form.
  auto run_block = [&](auto plainTag, int blk, const L2Block &blkWords) {
    constexpr bool PLAIN = decltype(plainTag)::value;
    const uint32_t wd[4] = {blkWords[0], blkWords[1], blkWords[2], blkWords[3]};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const uint32_t code = (e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffffu);
      const int ev = 8 * blk + e;                                     // events applied so far
      const bool on = PLAIN ? true : ev < n;
      const bool INS = (code & kL2InsBit) != 0;
      bool eff = on;
      if (!PLAIN) {
        if (__any(on && (code & kL2DupBit) != 0)) {
          if (on && (code & kL2DupBit)) {
            const int insCount = ev - delCount;                       // entries [delCount, insCount) are in the window
            if (INS) eff = dup_prev(a.g.dup, (uint32_t)(r.beg0 + insCount)) < r.beg0 + delCount;            // new iff no same-hash entry in [beg, end)
            // stays iff a later same-hash entry is in the window
            else { const int32_t nx = dup_next(a.g.dup, (uint32_t)(r.beg0 + delCount)); eff = !(nx >= 0 && nx < r.beg0 + insCount); }
          }
        }
      }
      l2_apply(F, R, code, INS, eff);
      delCount += (on && !INS) ? 1 : 0;
      // evaluate (computeMap.hpp:468-476) with the window starting at entry number delCount
      const bool evl = on && (code & kL2NoEvalBit) == 0;
      const int se = evl ? R.shared : -1;                             // best >= 0: an event without evaluation never compares
      const bool better = se > best, tieOrBetter = se >= best;
      best = better ? R.shared : best;
      begAtBest = better ? delCount : begAtBest;
      begAtLast = tieOrBetter ? delCount : begAtLast;
      steps += evl ? 1 : 0;
    }
  };
  // Phase A: the first super-window fills.  Its inserts are never evaluated (kL2NoEvalBit: computeMap.hpp:448 inserts [beg, end)
  // before the loop of :455 looks at anything), so they need no pivot — an event is its field update, 9 instructions instead of
  // ~55 — and the pivot is found once afterwards: the state is a function of the window's multiset, not of the order of the events,
  //     iStar = max{i : G(i) <= s},  G(i) = i + sum_{g<i} n[g]  (strictly increasing),  tot = G(iStar),  shared = sum_{i<=iStar} b[i].
  // A quarter of all events are such inserts (~240 of ~960 per candidate).  The phase lasts while every lane's block holds eight of
  // them; from then on the general loop below takes over, mid-window if need be.
  // What it buys is bounded by the LDS round trip of the byte read-modify-write chain (eight dependent ones per block, 2.5 waves
  // per SIMD to hide them): k_l2_sim 47.7 -> 43.7 ms per step.  Measured and dropped: the updates as LDS atomics on the shared
  // dwords, returning (43.9 ms) or fire-and-forget with a sum check against carries between the lanes' bytes (46.7 ms); four events
  // at a time with their fields read together and same-field events added up in registers (44.5 ms).
  int blk = 0;
  {
    constexpr uint32_t kFillPair = (kL2InsBit | kL2NoEvalBit) * 0x10001u, kDupPair = kL2DupBit * 0x10001u;
    for (; __any(mine); blk++) {
      const uint32_t andw = cur[0] & cur[1] & cur[2] & cur[3], orw = cur[0] | cur[1] | cur[2] | cur[3];
      const bool fill = 8 * blk + 8 <= n && (andw & kFillPair) == kFillPair;
      if (!__all(fill || !mine)) break;
      const int nb = blk + 1 < nBlk ? blk + 1 : blk;
      const L2Block nxt = l2_block_load(mine, p + nb, cur);
      const bool anyDup = __any(mine && (orw & kDupPair) != 0);         // rare: an entry with a same-hash neighbour nearby
      if (mine) {
        const uint32_t wd[4] = {cur[0], cur[1], cur[2], cur[3]};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const uint32_t code = (e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffffu);
          int d = (int)((code >> kL2DeltaShift) & 7u);                   // an insert's change is +1 or +2
          // the window starts at the candidate's first entry and nothing has left it: entry number 8 blk + e is new unless a
          // same-hash entry lies in [beg0, it) (slidingMap.hpp:150-154)
          if (anyDup && (code & kL2DupBit) && dup_prev(a.g.dup, (uint32_t)(r.beg0 + 8 * blk + e)) >= r.beg0) d = 0;
          uint8_t *pOwn = F + l2_field_off((int)((code >> 1) & 0x1ffu));
          const int nw = (int)*pOwn + d;
          R.ovfAcc |= (uint32_t)nw;
          *pOwn = (uint8_t)nw;
        }
      }
      cur = nxt;
    }
    if (blk > 0) {                                                      // wave-uniform
      int iS = 0, tot = 0, sh = 0, acc = 0; bool open = mine;
      for (int i = 1; __any(open && i <= R.s); i++) {
        const int f = F[l2_field_off(i - 1)];                           // n[i-1] << 1 | b[i]
        acc += f >> 1;
        open = open && i <= R.s && i + acc <= R.s;
        iS = open ? i : iS; tot = open ? i + acc : tot; sh += open ? (f & 1) : 0;
      }
      R.iStar = mine ? iS : R.iStar; R.tot = mine ? tot : R.tot; R.shared = mine ? sh : R.shared;
    }
  }
  for (; __any(blk < nBlk); blk++) {
    const int nb = blk + 1 < nBlk ? blk + 1 : blk;                     // never beyond the lane's own stream
    const L2Block nxt = l2_block_load(mine, p + nb, cur);
    const uint32_t orw = cur[0] | cur[1] | cur[2] | cur[3];
    const bool plain = !mine || (8 * blk + 8 <= n && (orw & (kL2DupBit | (kL2DupBit << 16))) == 0);
    if (__all(plain)) run_block(std::true_type(), blk, cur);
    else run_block(std::false_type(), blk, cur);
    cur = nxt;
  }
  if (mine) {
    if (R.ovfAcc & 0x100u) a.slowFlag[i] = 3;
    else {
      a.slowFlag[i] = (G::kMaxS == 255) ? 0 : 8;     // class B runs concurrently with class A: its "done" must not read as class A
      a.g.outBest[c] = best;
      a.g.outFirst[c] = begAtBest >= 0 ? a.g.mWpos[r.beg0 + begAtBest] : 0;
      a.g.outLast[c] = begAtLast >= 0 ? a.g.mWpos[r.beg0 + begAtLast] : 0;
      cntE = (unsigned long long)(r.last - r.beg0); cntS = (unsigned long long)steps; cntQ = (unsigned long long)R.s;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { cntE += __shfl_down(cntE, d); cntS += __shfl_down(cntS, d); cntQ += __shfl_down(cntQ, d); }
  if (lane == 0 && (cntE | cntS | cntQ)) { atomicAdd(stat_slot(a.g.sumEntries), cntE); atomicAdd(stat_slot(a.g.sumSteps), cntS);
    atomicAdd(stat_slot(a.g.sumQ), cntQ); }
}

// (Rounds 4 - 5 kept a second simulation kernel here, k_l2_sim_pair: two candidates per lane with the state in packed 16-bit halves —
// 38.7 vector instructions per event instead of 54.4, bit-exact, and slower (65.7 vs 44.1 ms per step: twice the LDS per wave, half
// the waves per SIMD; profiles/r04ad_pair_ab.txt, docs/history.md section 2.5).  Removed in round 5; the last commit that has it: 7bf79f3.)


// candidates of the chunk that must take the general kernel -> list (order irrelevant)
// after both simulation classes ran: slowFlag 1 = outside the fast-path limits, 3 = gap counter overflow, 0 / 8 = done
static __global__ void k_l2_collect_slow(int32_t c0, int32_t n, const int32_t *__restrict__ slowFlag, int32_t *__restrict__ list,
                                  unsigned int *__restrict__ count, unsigned long long *__restrict__ reasons /* [4] */)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (slowFlag[i] & 3)) { list[atomicAdd(count, 1u)] = c0 + i; atomicAdd(&reasons[slowFlag[i] & 3], 1ull); }
}

// Candidates of a chunk ordered by the length of their code stream (counting sort on codeCount / 16): the 64 lanes of a wave
// then run about the same number of steps instead of waiting for the longest of 64 random candidates.
constexpr int kL2LenBuckets = 1024;      // codeCount <= 2 * 16384 -> bucket = codeCount >> 5
static __global__ __launch_bounds__(kTPB) void k_l2_len_hist(const int32_t *__restrict__ codeCount, int32_t n, unsigned int *__restrict__ hist)
{
  __shared__ unsigned int h[kL2LenBuckets];
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) h[i] = 0;
  block_barrier();
  for (int i = blockIdx.x * kTPB + threadIdx.x; i < n; i += gridDim.x * kTPB) {
    int b = codeCount[i] >> 5; b = b >= kL2LenBuckets ? kL2LenBuckets - 1 : b;
    atomicAdd(&h[kL2LenBuckets - 1 - b], 1u);                  // longest first
  }
  block_barrier();
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) if (h[i]) atomicAdd(&hist[i], h[i]);
}
static __global__ __launch_bounds__(kTPB) void k_l2_len_scan(unsigned int *__restrict__ hist)     // one workgroup: exclusive scan in place
{
  __shared__ int ws[16];
  block_array_excl_scan((int *)hist, kL2LenBuckets, ws);
}
static __global__ __launch_bounds__(kTPB) void k_l2_len_scatter(const int32_t *__restrict__ codeCount, int32_t c0, int32_t n,
                                                         unsigned int *__restrict__ cursor, int32_t *__restrict__ order)
{
  // one global atomic per (workgroup, bucket): ranks inside the workgroup come from LDS atomics
  __shared__ unsigned int cnt[kL2LenBuckets], base[kL2LenBuckets];
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) cnt[i] = 0;
  block_barrier();
  const int i = blockIdx.x * kTPB + threadIdx.x;
  int b = 0; unsigned int r = 0;
  if (i < n) {
    b = codeCount[i] >> 5; b = b >= kL2LenBuckets ? kL2LenBuckets - 1 : b;
    b = kL2LenBuckets - 1 - b;
    r = atomicAdd(&cnt[b], 1u);
  }
  block_barrier();
  for (int j = threadIdx.x; j < kL2LenBuckets; j += kTPB) if (cnt[j]) base[j] = atomicAdd(&cursor[j], cnt[j]);
  block_barrier();
  if (i < n) order[base[b] + r] = c0 + i;
}

// class-B candidates of the chunk (slowFlag == 4) -> dense list
static __global__ void k_l2_collect_class(int32_t c0, int32_t n, const int32_t *__restrict__ slowFlag, int32_t flag, int32_t *__restrict__ list,
                                   unsigned int *__restrict__ count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && slowFlag[i] == flag) list[atomicAdd(count, 1u)] = c0 + i;
}

}  // namespace ani
