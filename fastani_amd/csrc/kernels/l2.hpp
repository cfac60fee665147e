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
// (slidingMap.hpp:150-154: REV status; :178: NOOP when a later occurrence re-tagged wposR); with prevSame/nextSame
// links over the position-ordered index this is: an insertion of entry j is effective iff prevSame[j] < beg, a
// deletion of entry j is effective iff nextSame[j] is not inside the inserted range [.., endIns).
//
// Parallelisation: one lane per candidate (tens of millions of candidates per many-to-many run give the
// parallelism); state words live in a lane-interleaved scratch array so that lane l touches
// scratch[word * laneStride + l].
#pragma once
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
                                                 const int32_t *prevSame, const int32_t *nextSame,
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
    if (prevSame[j] < beg) sim.insert(q_rank(q, s, mHash[j]));
  int32_t pb = beg, pe = end;
  int32_t pos = mWpos[beg];
  while (end < last) {                                                           // :455
    if (pb != beg) {                                                             // :461 delete_ref(prev_beg)
      const int32_t nx = nextSame[pb];
      if (!(nx >= 0 && nx < pe)) sim.erase(q_rank(q, s, mHash[pb]));
    }
    if (pe != end) {                                                             // :465 insert_ref(prev_end)
      if (prevSame[pe] < beg) sim.insert(q_rank(q, s, mHash[pe]));
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
//   k_l2_ranges  one lane per candidate: the three searchIndex() calls of computeMap.hpp:424-436
//   k_l2_codes   one workgroup per fragment: the fragment sketch sits in LDS; every reference minimizer of every
//                candidate range is ranked against it once and stored as one 16-bit entry
//                  bit 0 = hash is a query hash, bits 1..9 = gap (or rank-1), bit 10 = a same-hash neighbour may share
//                  a super-window with this entry (nearDup, precomputed at index build), bits 11..15 = wpos - previous
//                  wpos (31 = escape: read the index)
//   k_l2_sim     one lane per candidate: the sliding simulation over those entries, one window event per loop pass.  State in
//                LDS: one byte per sketch rank (7-bit gap counter + presence bit), byte-interleaved over the wave.  Entries are
//                streamed by two monotone cursors through 16-entry LDS rings (+ 8 entries prefetched in registers per cursor);
//                refills are issued at wave-uniform points every 8 passes so that their latency never sits on an event's
//                critical path.  Entries flagged nearDup consult prevSame/nextSame (exact set semantics,
//                slidingMap.hpp:150-154,:178).  A gap counter that would pass 127 sends the candidate to k_l2.
// ------------------------------------------------------------------------------------------------
struct L2Args {
  // candidates (SoA)
  const int32_t *candFrag, *candSeq, *candStart, *candEnd;
  int32_t nCand;
  // fragment sketches
  const uint32_t *qPool; const uint32_t *fragOff; const int32_t *fragS;
  // reference index, position order
  const uint32_t *mHash; const int32_t *mWpos; const int32_t *prevSame; const int32_t *nextSame;
  const uint8_t *mDelta;           // min(wpos - previous wpos, 31) | nearDup << 5
  const int32_t *contigFirstMin;   // [nContigs+1]
  const uint32_t *posBase, *posSample;   // sampled position index (index.hpp: k_index_pos_sample)
  int rankShift;                   // k_l2_codes rank table: 512 linear buckets of 2^rankShift hashes from 0; minimizer hashes are minima of w
                                   // k-mer hashes, ~96 % of them lie below 2^32 * 3 / w, so the table covers [0, 2^(32 - floor(log2 w) + 1))
  int L, w, k;
  // lane-interleaved scratch of the general kernel: (maxS+1) words per lane
  uint32_t *scratch; size_t laneStride;
  // outputs
  int32_t *outBest, *outFirst, *outLast;
  unsigned long long *sumEntries, *sumSteps, *sumQ;
};

__global__ __launch_bounds__(kTPB) void k_l2(L2Args a, const int32_t *__restrict__ list, int32_t listCount, int32_t listBase)
{
  const int32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t li = listBase + lane;
  unsigned long long e = 0, st = 0, sq = 0;
  if (li < listCount && (size_t)lane < a.laneStride) {
    const int32_t c = list ? list[li] : li;
    const int32_t f = a.candFrag[c];
    const int32_t seq = a.candSeq[c];
    L2State state; state.w = a.scratch + lane; state.stride = a.laneStride;
    L2Result r = l2_candidate(a.qPool + a.fragOff[f], a.fragS[f], a.mHash, a.mWpos, a.prevSame, a.nextSame,
                              a.contigFirstMin[seq], a.contigFirstMin[seq + 1], a.candStart[c], a.candEnd[c],
                              a.L, a.w, a.k, state);
    a.outBest[c] = r.best; a.outFirst[c] = r.firstPos; a.outLast[c] = r.lastPos;
    e = (unsigned long long)r.entries; st = (unsigned long long)r.steps; sq = (unsigned long long)a.fragS[f];
  }
  // counters for the algorithmic-byte figure: one atomic per wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { e += __shfl_down(e, d); st += __shfl_down(st, d); sq += __shfl_down(sq, d); }
  if ((threadIdx.x & 63) == 0 && (e | st | sq)) { atomicAdd(stat_slot(a.sumEntries), e); atomicAdd(stat_slot(a.sumSteps), st); atomicAdd(stat_slot(a.sumQ), sq); }
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
//   class A  s <= 255: 64 words + 16-entry cursor rings (16 words) = 320 B per lane -> 20 KiB per wave, 8 waves per CU
//   class B  s <= 319: 80 words + rings                            = 384 B per lane -> 24 KiB per wave, 6 waves per CU
// A counter that would pass 127 sends the candidate to the general kernel.
template <int MAXS>
struct L2Geom {
  static constexpr int kMaxS = MAXS;
  static constexpr int kStateWords = (MAXS + 1 + 3) / 4;
  static constexpr int kRingWords = 16;                     // 2 cursors x 16 entries x 16 bit
  static constexpr int kWords = kStateWords + kRingWords;
};
using L2GeomA = L2Geom<255>;
using L2GeomB = L2Geom<319>;
constexpr int kL2SimTPB = 64;

struct L2Range { int32_t beg0, end0, last, wposBeg0; };

struct L2FastArgs {
  L2Args g;
  int32_t c0, c1;                  // candidate chunk [c0, c1)
  L2Range *ranges;                 // [c1-c0]
  int32_t *codeCount;              // [c1-c0] 16-bit entries per candidate rounded up to 8, 0 = not on the fast path
  const uint32_t *codeOff;         // [c1-c0] exclusive scan of codeCount (in entries)
  uint32_t *codes;
  int32_t *slowFlag;               // [c1-c0] 1 = take the general kernel
  const uint32_t *fragCandOff;     // ordered candidate offset per fragment [nFrag]
  int32_t nFrag, fragBase;         // first fragment of the chunk
  int32_t allowFast;               // test knob ANI_L2_PATH: 0 = everything to the general kernel, 2 = everything to class B
};

__global__ __launch_bounds__(kTPB) void k_l2_ranges(L2FastArgs a)
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
  auto search = [&](int32_t pos) -> int32_t {
    int32_t bin = (pos < 0 ? 0 : pos) >> kPosSampleShift; bin = bin < nb ? bin : nb - 1;
    const int32_t lo = (int32_t)a.g.posSample[pb + (uint32_t)bin];
    const int32_t hi = bin + 1 < nb ? (int32_t)a.g.posSample[pb + (uint32_t)bin + 1] : cHi;
    return lower_bound_wpos(a.g.mWpos, lo, hi, pos);
  };
  r.beg0 = search(a.g.candStart[c]);
  r.wposBeg0 = a.g.mWpos[r.beg0];
  r.end0 = search(r.wposBeg0 + cmw);
  r.last = search(a.g.candEnd[c] + a.g.L);
  a.ranges[i] = r;
  const int32_t m = r.last - r.beg0;
  const int32_t s = a.g.fragS[a.g.candFrag[c]];
  const bool fast = a.allowFast && s >= 1 && s <= kL2FastMaxS && m >= 1 && m <= kL2FastMaxEntries;
  a.codeCount[i] = fast ? ((m + 7) & ~7) : 0;      // 16-bit entries, padded to 16-byte blocks
  a.slowFlag[i] = fast ? ((s <= L2GeomA::kMaxS && a.allowFast != 2) ? 0 : 4) : 1;
}

constexpr uint32_t kL2DupBit = 1u << 10;
constexpr int kL2RankBuckets = 512;
// rank-table bucket of a hash: linear buckets over the low end of the range, where minimizer hashes live (see L2Args::rankShift)
__device__ __forceinline__ int l2_rank_bucket(uint32_t h, int sh) { const uint32_t b = h >> sh; return (int)(b < (uint32_t)(kL2RankBuckets - 1) ? b : (uint32_t)(kL2RankBuckets - 1)); }
constexpr uint32_t kL2DwEscape = 31u;

__global__ __launch_bounds__(kTPB) void k_l2_codes(L2FastArgs a)
{
  __shared__ uint32_t qs[kL2FastMaxS + 1];
  __shared__ uint16_t st[kL2RankBuckets + 2];
  const int32_t f = a.fragBase + blockIdx.x;
  const int32_t s = a.g.fragS[f];
  int32_t cA = (int32_t)a.fragCandOff[f], cB = (f + 1 < a.nFrag) ? (int32_t)a.fragCandOff[f + 1] : a.g.nCand;
  if (cA < a.c0) cA = a.c0;
  if (cB > a.c1) cB = a.c1;
  if (cA >= cB || s < 1 || s > kL2FastMaxS) return;
  const uint32_t *q = a.g.qPool + a.g.fragOff[f];
  for (int i = threadIdx.x; i < s; i += kTPB) qs[i] = q[i];
  // rank table: st[b] = #{q : bucket(q) < b}; a lookup then searches the one or two sketch entries of the hash's bucket
  const int sh = a.g.rankShift;
  for (int i = threadIdx.x; i <= s; i += kTPB) {
    const int b0 = i > 0 ? l2_rank_bucket(q[i - 1], sh) + 1 : 0;
    const int b1 = i < s ? l2_rank_bucket(q[i], sh) : kL2RankBuckets;
    for (int b = b0; b <= b1; b++) st[b] = (uint16_t)i;
  }
  __syncthreads();
  for (int32_t c = cA; c < cB; c++) {
    const int32_t i = c - a.c0;
    if (a.codeCount[i] == 0) continue;
    const L2Range r = a.ranges[i];
    uint16_t *out = (uint16_t *)a.codes + a.codeOff[i];
    const uint8_t *__restrict__ dlt = a.g.mDelta + r.beg0;            // unsigned 32-bit offsets from per-candidate bases:
    const uint32_t *__restrict__ hsh = a.g.mHash + r.beg0;            // scalar base + vector offset addressing, no 64-bit index math
    const uint32_t m = (uint32_t)(r.last - r.beg0);
    for (uint32_t j = threadIdx.x; j < m; j += kTPB) {
      const uint32_t dl = dlt[j];
      const uint32_t dw = j > 0 ? (dl & 31u) : 0u;
      const uint32_t h = hsh[j];
      const int rb = l2_rank_bucket(h, sh);
      int lo = st[rb], hi = st[rb + 1];
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (qs[mid] < h) lo = mid + 1; else hi = mid; }
      const uint32_t rk = ((uint32_t)lo << 1) | (uint32_t)((lo < s) & (qs[lo < s ? lo : s - 1] == h));   // == q_rank(qs, s, h), no branch
      out[j] = (uint16_t)(rk | ((dl & 32u) ? kL2DupBit : 0u) | (dw << 11));
    }
  }
}

// One monotone cursor over a candidate's 16-bit entries.  Entries [8b, 8b+16) sit in a 16-entry LDS ring (entry j at ring
// slot j & 15, lane-interleaved dwords => conflict-free ds_read_u16), block b+2 is in flight in registers.  sync() is called
// at wave-uniform points at most 8 consumed entries apart; its global load is unconditional and outside divergent control
// flow on purpose, so that it stays in flight until the next sync() instead of being waited for at a branch join.
struct L2Stream {
  const uint4 *p; int b; uint4 B;
  uint32_t *ring;                // this lane's dword 0 of the cursor's ring: dword d at ring[d * kWave]
  __device__ __forceinline__ void put(int blk, const uint4 v)
  {
    uint32_t *r = ring + ((blk & 1) * 4) * kWave;
    r[0] = v.x; r[kWave] = v.y; r[2 * kWave] = v.z; r[3 * kWave] = v.w;
  }
  __device__ __forceinline__ void init(const uint4 *p_, uint32_t *ring_)
  {
    p = p_; ring = ring_; b = 0;
    put(0, p[0]); put(1, p[1]); B = p[2];
  }
  __device__ __forceinline__ void sync(int j)
  {
    if (j - 8 * b >= 8) { put(b, B); b++; }      // block b is consumed: block b+2 takes its ring slot
    B = p[b + 2];
  }
  __device__ __forceinline__ uint32_t get(int j) const
  {
    const int e = j & 15;
    const uint16_t *h = (const uint16_t *)(ring + (e >> 1) * kWave);
    return h[e & 1];
  }
};

// One window event, written without control flow (selects only).  INS: the entry enters the window (slidingMap.hpp:137-161
// + :231-254), otherwise it leaves (:167-211 + :261-284).  Both LDS reads (the entry's own field and the field next to the
// pivot) are issued up front; `on` = false turns the event into a no-op.  Field g of this lane sits at F[g << 6]
// (byte-interleaved over the wave: one shift to address it; the only bank conflicts are between the four lanes of a dword
// column whose g differ by a multiple of 4 — the LDS pipe has an order of magnitude of slack under the VALU work of a step).
struct L2Regs { int s, iStar, cStar, shared; bool ovf; };
constexpr bool kL2ByteInterleave = true;
__device__ __forceinline__ int l2_field_off(int g) { return kL2ByteInterleave ? (g << 6) : (((g >> 2) << 8) + (g & 3)); }

__device__ __forceinline__ void l2_apply(uint8_t *F, L2Regs &r, uint32_t code, bool INS, bool on)
{
  const bool isQ = (code & 1u) != 0;
  const int idx = (int)((code >> 1) & 0x1ffu);
  const int sg = INS ? 1 : -1;
  int j = r.iStar - (INS ? 1 : 0); j = j < 0 ? 0 : j;               // pivot-adjacent field: n[j], b[j+1]
  uint8_t *pOwn = F + l2_field_off(idx);
  const int own = *pOwn;
  int fj = F[l2_field_off(j)];
  const int delta = isQ ? 1 : 2;                                     // counter lives in bits 1..7, presence in bit 0
  const bool full = INS && !isQ && own >= 254;
  r.ovf = r.ovf || (on && full);
  const bool act = on && !full;
  const int d = act ? (INS ? delta : -delta) : 0;
  *pOwn = (uint8_t)(own + d);
  fj += (idx == j) ? d : 0;                                          // the event itself changed field[j]
  const int cntj = fj >> 1;
  const bool lt = idx < r.iStar;                                     // rank idx+1 <= iStar  <=>  gap idx < iStar
  const int t = (act && lt) ? sg : 0;
  r.shared += isQ ? t : 0;
  r.cStar += isQ ? 0 : t;
  // insert: q_iStar leaves the s smallest;  delete: q_{iStar+1} joins them
  const int tot = r.iStar + r.cStar;
  const bool condI = lt & (tot > r.s), condD = (r.iStar < r.s) & (tot + 1 + cntj <= r.s);      // both sides, no control flow
  const bool cond = INS ? condI : condD;
  const bool mv = act && !isQ && cond;
  const int mone = mv ? sg : 0;                                      // insert: -1 on everything, delete: +1
  const int cm = mv ? cntj : 0;
  r.iStar -= mone;
  r.shared -= (fj & 1) ? mone : 0;
  r.cStar -= INS ? cm : -cm;
}

// slowFlag protocol: 0 = class A (pending or done), 4 = class B pending (s in 256..319), 8 = class B done,
//                    1 = outside every fast-path limit, 3 = a gap counter overflowed -> general kernel
// `list` (optional): candidate ids to run, densely packed so that the few class-B candidates fill whole waves instead of
// leaving one busy lane in every wave; *listCount is read on the device (no host round trip).
template <class G>
__global__ __launch_bounds__(kL2SimTPB) void k_l2_sim(L2FastArgs a, const int32_t *__restrict__ list, const unsigned int *__restrict__ listCount)
{
  __shared__ uint32_t lds[(kL2SimTPB / kWave) * G::kWords * kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  uint32_t *W = lds + wv * (G::kWords * kWave);                       // this wave's LDS
  uint8_t *F = (uint8_t *)W + (kL2ByteInterleave ? lane : 4 * lane);   // field g of this lane: F[l2_field_off(g)]
  uint32_t *S = W + lane;                                             // dword x of this lane: S[x * kWave] (cursor rings)
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
  if (mine) {
    const int32_t i = c - a.c0;
    const L2Range r = a.ranges[i];
    const int32_t f = a.g.candFrag[c];
    const int32_t cmw = a.g.L - (a.g.w - 1) - (a.g.k - 1);
    const int m = r.last - r.beg0;
    L2Regs R; R.s = a.g.fragS[f]; R.iStar = R.s; R.cStar = 0; R.shared = 0; R.ovf = false;
    // wpos of entry j given the previous entry's wpos and the entry's 5-bit delta (31 = look it up)
    auto next_wpos = [&](int32_t prev, uint32_t code, int j) -> int32_t {
      const uint32_t dw = code >> 11;
      int32_t wp = prev + (int32_t)dw;
      if (dw == kL2DwEscape) { wp = a.g.mWpos[r.beg0 + j]; ANI_CONSUME(wp); }     // rare; the load is waited for here, not in the loop body
      return wp;
    };
    const uint4 *base = (const uint4 *)((const uint16_t *)a.codes + a.codeOff[i]);
    L2Stream cb, ce;
    cb.init(base, S + G::kStateWords * kWave);
    ce.init(base, S + (G::kStateWords + 8) * kWave);
    // first super-window: entries [0, end0-beg0)  (computeMap.hpp:448)
    int end = r.end0 - r.beg0, beg = 0;
    int32_t wEnd = r.wposBeg0;                       // becomes wpos of entry `end`
    for (int j = 0; j < end; j++) {
      if ((j & 7) == 0) ce.sync(j);
      const uint32_t cd = ce.get(j);
      if (j > 0) wEnd = next_wpos(wEnd, cd, j);
      bool eff = true;
      if (cd & kL2DupBit) eff = a.g.prevSame[r.beg0 + j] < r.beg0;
      l2_apply(F, R, cd, true, eff);
    }
    uint32_t codeEnd = 0;
    if (end < m) { codeEnd = ce.get(end); wEnd = next_wpos(wEnd, codeEnd, end); }
    int32_t wBeg = r.wposBeg0;
    uint32_t codeBeg = cb.get(0);
    uint32_t codeBegNext = (1 < m) ? cb.get(1) : 0u;
    int32_t wBegNext = (1 < m) ? next_wpos(wBeg, codeBegNext, 1) : wBeg;
    int best = 0; int32_t firstPos = 0, lastPos = 0; int steps = 0;
    // Event-driven form of the loop at computeMap.hpp:455-481.  A step of the reference advances MIIteratorL2 to the nearer of
    //   pB = wpos[beg+1]            (the first entry leaves:  delete_ref(beg),  beg++)
    //   pE = wpos[end] - cmw + 1    (entry `end` fits:        insert_ref(end),  end++)
    // applies the delete and/or the insert and evaluates the window; the loop ends, unevaluated, with the insert that takes
    // `end` to the end of the range.  Nearly every step carries one event only, so one pass of this loop is ONE event: apply it,
    // advance its cursor, evaluate unless the insert of the same position (pB == pE: delete first, as the reference does) is
    // still to come.  wpos is strictly increasing inside a contig, so after such a delete the next event is that insert.
    if (end < m) {
      best = R.shared; firstPos = R.shared > 0 ? wBeg : 0; lastPos = wBeg; steps = 1;     // first pass of :455, no events
      const int32_t cmw1 = cmw - 1;
      int b1 = 1;                                    // beg + 1
      const uint32_t *ringB = S + G::kStateWords * kWave;
      for (int it = 0; !R.ovf; it++) {
        if ((it & 7) == 0) { cb.sync(b1); ce.sync(end); }
        const int32_t pE = wEnd - cmw1;
        const bool del = wBegNext <= pE;
        if (!del && end + 1 >= m) break;
        const bool more = del && wBegNext == pE;     // the insert of this step follows
        const uint32_t code = del ? codeBeg : codeEnd;
        bool eff = true;
        if (code & kL2DupBit) {
          if (del) {                                 // stays iff a later same-hash entry is already in the window
            const int32_t nx = a.g.nextSame[r.beg0 + b1 - 1];
            eff = !(nx >= 0 && nx < r.beg0 + end);
          } else eff = a.g.prevSame[r.beg0 + end] < r.beg0 + b1 - 1;       // new iff no same-hash entry in [beg, end)
        }
        // the cursor of this event advances: fetch entry beg+2 resp. end+1 (both inside the range, see above).  Read before the
        // state update so that its LDS latency overlaps the field reads of l2_apply instead of following its byte store.
        const int jf = (del ? b1 : end) + 1;
        const uint16_t *h = (const uint16_t *)(ringB + (del ? 0 : 8 * kWave) + ((jf & 15) >> 1) * kWave);
        const uint32_t cf = h[jf & 1];
        l2_apply(F, R, code, !del, eff);
        const int32_t wf = next_wpos(del ? wBegNext : wEnd, cf, jf);
        codeBeg = del ? codeBegNext : codeBeg; wBeg = del ? wBegNext : wBeg;
        codeBegNext = del ? cf : codeBegNext; wBegNext = del ? wf : wBegNext;
        codeEnd = del ? codeEnd : cf; wEnd = del ? wEnd : wf;
        b1 += del ? 1 : 0; end += del ? 0 : 1;
        // evaluate (:468-476)
        const bool better = !more && R.shared > best, tieOrBetter = !more && R.shared >= best;
        best = better ? R.shared : best;
        firstPos = better ? wBeg : firstPos;
        lastPos = tieOrBetter ? wBeg : lastPos;
        steps += more ? 0 : 1;
      }
    }
    if (R.ovf) a.slowFlag[i] = 3;
    else {
      a.slowFlag[i] = (G::kMaxS == 255) ? 0 : 8;     // class B runs concurrently with class A: its "done" must not read as class A
      a.g.outBest[c] = best; a.g.outFirst[c] = firstPos; a.g.outLast[c] = lastPos;
      cntE = (unsigned long long)m; cntS = (unsigned long long)steps; cntQ = (unsigned long long)R.s;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { cntE += __shfl_down(cntE, d); cntS += __shfl_down(cntS, d); cntQ += __shfl_down(cntQ, d); }
  if (lane == 0 && (cntE | cntS | cntQ)) { atomicAdd(stat_slot(a.g.sumEntries), cntE); atomicAdd(stat_slot(a.g.sumSteps), cntS); atomicAdd(stat_slot(a.g.sumQ), cntQ); }
}

// candidates of the chunk that must take the general kernel -> list (order irrelevant)
// after both simulation classes ran: slowFlag 1 = outside the fast-path limits, 3 = gap counter overflow, 0 / 8 = done
__global__ void k_l2_collect_slow(int32_t c0, int32_t n, const int32_t *__restrict__ slowFlag, int32_t *__restrict__ list,
                                  unsigned int *__restrict__ count, unsigned long long *__restrict__ reasons /* [4] */)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (slowFlag[i] & 3)) { list[atomicAdd(count, 1u)] = c0 + i; atomicAdd(&reasons[slowFlag[i] & 3], 1ull); }
}

// Candidates of a chunk ordered by the length of their code stream (counting sort on codeCount / 16): the 64 lanes of a wave
// then run about the same number of steps instead of waiting for the longest of 64 random candidates.
constexpr int kL2LenBuckets = 1024;      // codeCount <= 16384 -> bucket = codeCount >> 4
__global__ __launch_bounds__(kTPB) void k_l2_len_hist(const int32_t *__restrict__ codeCount, int32_t n, unsigned int *__restrict__ hist)
{
  __shared__ unsigned int h[kL2LenBuckets];
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) h[i] = 0;
  __syncthreads();
  for (int i = blockIdx.x * kTPB + threadIdx.x; i < n; i += gridDim.x * kTPB) {
    int b = codeCount[i] >> 4; b = b >= kL2LenBuckets ? kL2LenBuckets - 1 : b;
    atomicAdd(&h[kL2LenBuckets - 1 - b], 1u);                  // longest first
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) if (h[i]) atomicAdd(&hist[i], h[i]);
}
__global__ __launch_bounds__(kTPB) void k_l2_len_scan(unsigned int *__restrict__ hist)     // one workgroup: exclusive scan in place
{
  __shared__ int ws[16];
  block_array_excl_scan((int *)hist, kL2LenBuckets, ws);
}
__global__ __launch_bounds__(kTPB) void k_l2_len_scatter(const int32_t *__restrict__ codeCount, int32_t c0, int32_t n,
                                                         unsigned int *__restrict__ cursor, int32_t *__restrict__ order)
{
  // one global atomic per (workgroup, bucket): ranks inside the workgroup come from LDS atomics
  __shared__ unsigned int cnt[kL2LenBuckets], base[kL2LenBuckets];
  for (int i = threadIdx.x; i < kL2LenBuckets; i += kTPB) cnt[i] = 0;
  __syncthreads();
  const int i = blockIdx.x * kTPB + threadIdx.x;
  int b = 0; unsigned int r = 0;
  if (i < n) {
    b = codeCount[i] >> 4; b = b >= kL2LenBuckets ? kL2LenBuckets - 1 : b;
    b = kL2LenBuckets - 1 - b;
    r = atomicAdd(&cnt[b], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < kL2LenBuckets; j += kTPB) if (cnt[j]) base[j] = atomicAdd(&cursor[j], cnt[j]);
  __syncthreads();
  if (i < n) order[base[b] + r] = c0 + i;
}

// class-B candidates of the chunk (slowFlag == 4) -> dense list
__global__ void k_l2_collect_class(int32_t c0, int32_t n, const int32_t *__restrict__ slowFlag, int32_t flag, int32_t *__restrict__ list,
                                   unsigned int *__restrict__ count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && slowFlag[i] == flag) list[atomicAdd(count, 1u)] = c0 + i;
}

}  // namespace ani
