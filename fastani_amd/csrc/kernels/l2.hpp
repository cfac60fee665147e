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
// Kernel: one lane per candidate.
// ------------------------------------------------------------------------------------------------
struct L2Args {
  // candidates (SoA)
  const int32_t *candFrag, *candSeq, *candStart, *candEnd;
  int32_t nCand;
  // fragment sketches
  const uint32_t *qPool; const uint32_t *fragOff; const int32_t *fragS;
  // reference index, position order
  const uint32_t *mHash; const int32_t *mWpos; const int32_t *prevSame; const int32_t *nextSame;
  const int32_t *contigFirstMin;   // [nContigs+1]
  int L, w, k;
  // lane-interleaved scratch: (maxS+1) words per lane
  uint32_t *scratch; size_t laneStride;
  // outputs
  int32_t *outBest, *outFirst, *outLast;
  unsigned long long *sumEntries, *sumSteps, *sumQ;
};

__global__ __launch_bounds__(kTPB) void k_l2(L2Args a, int32_t candBase)
{
  const int32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t c = candBase + lane;
  unsigned long long e = 0, st = 0, sq = 0;
  if (c < a.nCand && (size_t)lane < a.laneStride) {
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
  if ((threadIdx.x & 63) == 0 && (e | st | sq)) { atomicAdd(a.sumEntries, e); atomicAdd(a.sumSteps, st); atomicAdd(a.sumQ, sq); }
}

}  // namespace ani
