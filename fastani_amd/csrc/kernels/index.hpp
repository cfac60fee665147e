// index.hpp — reference index over the position-ordered minimizer records.
//
// Replaces skch::Sketch::index (src/map/include/winSketch.hpp:181-193: unordered_map<hash, vector<{seqId,wpos}>>)
// and skch::Sketch::searchIndex (:259-270) by flat device arrays:
//   position order (≙ minimizerIndex :93)      mHash[n], mSeq[n], mWpos[n]      + contigFirstMin[nContigs+1]
//   hash order    (≙ minimizerPosLookupIndex)  sHash[n] (sorted), sIdx[n] (rank in position order; stable, so every
//                                              hash's occurrence list stays in (seqId,wpos) order like :186-190)
//   bucket table  bucketStart[2^bits + 1]      lower bounds of the top `bits` hash bits inside sHash
//   same-hash links (for the L2 set semantics) prevSame[n], nextSame[n]: neighbouring occurrence of the same hash in
//                                              position order, -1 if none
#pragma once
#include "common.hpp"

namespace ani {

// records: 12-byte (hash, seqId, wpos) triples in position order
__global__ void k_index_split(const uint32_t *__restrict__ records, uint32_t n,
                              uint32_t *__restrict__ mHash, int32_t *__restrict__ mSeq, int32_t *__restrict__ mWpos,
                              uint32_t *__restrict__ keyOut, uint32_t *__restrict__ valOut)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t h = records[3 * (size_t)i];
    mHash[i] = h; mSeq[i] = (int32_t)records[3 * (size_t)i + 1]; mWpos[i] = (int32_t)records[3 * (size_t)i + 2];
    keyOut[i] = h; valOut[i] = i;
  }
}

__global__ void k_index_join(const uint32_t *__restrict__ mHash, const int32_t *__restrict__ mSeq, const int32_t *__restrict__ mWpos,
                             uint32_t n, uint32_t *__restrict__ records)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    records[3 * (size_t)i] = mHash[i]; records[3 * (size_t)i + 1] = (uint32_t)mSeq[i]; records[3 * (size_t)i + 2] = (uint32_t)mWpos[i];
  }
}

// after the stable sort by hash: same-hash neighbours and the unique-hash count
// also: mWposF[idx] = wpos | nearDup << 31.  nearDup(j) = some same-hash entry j' of the same contig can share a super-window
// with j.  All entries of a window except its first lie within cmw = countMinimizerWindows positions; the first entry
// (MIIteratorL2 keeps the minimizer that is active at the window start) may trail by less than the gap to its successor.
// So for j' < j: possible iff wpos[j] - wpos[j'] <= cmw + (wpos[j'+1] - wpos[j']).  Entries without the flag behave as a
// plain set member in the L2 fast path; flagged ones consult prevSame/nextSame there.
__global__ void k_index_links(const uint32_t *__restrict__ sHash, const uint32_t *__restrict__ sIdx, uint32_t n,
                              int32_t *__restrict__ prevSame, int32_t *__restrict__ nextSame,
                              const int32_t *__restrict__ mSeq, const int32_t *__restrict__ mWpos, int32_t cmw,
                              uint32_t *__restrict__ mWposF,
                              unsigned long long *__restrict__ nUnique)
{
  unsigned long long uniq = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const uint32_t h = sHash[r], idx = sIdx[r];
    const bool samePrev = r > 0 && sHash[r - 1] == h;
    const bool sameNext = r + 1 < n && sHash[r + 1] == h;
    prevSame[idx] = samePrev ? (int32_t)sIdx[r - 1] : -1;
    nextSame[idx] = sameNext ? (int32_t)sIdx[r + 1] : -1;
    bool near = false;
    if (samePrev) {
      const uint32_t p = sIdx[r - 1];                    // p < idx (stable sort)
      near |= mSeq[p] == mSeq[idx] && mWpos[idx] - mWpos[p] <= cmw + (mWpos[p + 1] - mWpos[p]);
    }
    if (sameNext) {
      const uint32_t q = sIdx[r + 1];                    // q > idx
      near |= mSeq[q] == mSeq[idx] && mWpos[q] - mWpos[idx] <= cmw + (mWpos[idx + 1] - mWpos[idx]);
    }
    mWposF[idx] = (uint32_t)mWpos[idx] | (near ? 0x80000000u : 0u);
    uniq += !samePrev;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) uniq += __shfl_down(uniq, d);
  if ((threadIdx.x & 63) == 0 && uniq) atomicAdd(nUnique, uniq);
}

// hash-ordered payload for the L1 gather: sSW[r] = (seqId << 32) | wpos of the r-th entry in hash order, so that the
// occurrence list of one hash is one contiguous run of 8-byte words
__global__ void k_index_payload(const uint32_t *__restrict__ sIdx, const int32_t *__restrict__ mSeq, const int32_t *__restrict__ mWpos,
                                uint32_t n, uint64_t *__restrict__ sSW)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const uint32_t idx = sIdx[r];
    sSW[r] = ((uint64_t)(uint32_t)mSeq[idx] << 32) | (uint32_t)mWpos[idx];
  }
}

// bucketStart[b] = first r with (sHash[r] >> shift) >= b, for b = 0..nBuckets (inclusive)
__global__ void k_index_buckets(const uint32_t *__restrict__ sHash, uint32_t n, int shift, uint32_t nBuckets,
                                uint32_t *__restrict__ bucketStart)
{
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= nBuckets; b += gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = n;
    if (b == nBuckets) lo = n;
    else
      while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if ((sHash[mid] >> shift) < b) lo = mid + 1; else hi = mid;
      }
    bucketStart[b] = lo;
  }
}

// contigFirstMin[c] = first position-ordered entry with seqId >= c, for c = 0..nContigs
__global__ void k_index_contig_first(const int32_t *__restrict__ mSeq, uint32_t n, int32_t nContigs,
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

}  // namespace ani
