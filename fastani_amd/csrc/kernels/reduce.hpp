// reduce.hpp — identity estimate + filter per candidate (skch::Map::doL2Mapping, src/map/include/computeMap.hpp:363-410)
// and the ANI reducer (cgi::computeCGI, src/cgi/include/computeCoreIdentity.hpp:166-298).
//
// Identity and the 90 % CI filter are pure functions of (shared, s); the host evaluates the reference's float
// expressions once (host/stats.hpp) and uploads
//   idLUT[lutOff(s) + shared] = nucIdentity (float bits),   minShared[s] = smallest `shared` that passes :384.
//
// Reducer without sorting.  Candidates arrive ordered by (fragment, seqId, start), so all mappings of one
// (fragment, reference genome) pair are adjacent:
//   1-way (computeCoreIdentity.hpp:214-231): the group's maximum under (nucIdentity, refSeqId, refStartPos);
//   2-way (:237-254): atomicMax of the identity bits into a dense bin table [query genome][contig bin], bin =
//          refStartPos / (fragLen-20) (:194) — identities are non-negative floats, so their bit patterns order like uints;
//   mean  (:267-297): one lane per (query genome, reference genome) walks that genome's bins in (contig, bin) order
//          and adds the non-empty ones sequentially in float — the same order the reference sums in.
#pragma once
#include "common.hpp"

namespace ani {

__host__ __device__ __forceinline__ uint32_t lut_off(int s) { return (uint32_t)(((int64_t)(s - 1) * (s + 2)) / 2); }

struct FinishArgs {
  int32_t nCand;
  const int32_t *candFrag, *candSeq;
  const int32_t *best, *firstPos, *lastPos;
  const int32_t *fragS;
  const uint32_t *idLUT; const int32_t *minShared; int32_t lutMaxS;
  // outputs per candidate
  int32_t *refStart;        // meanOptimalPos (computeMap.hpp:496)
  uint32_t *idBits;         // nucIdentity bits, 0 if filtered out
};

static __global__ void k_finish_candidates(FinishArgs a)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nCand) return;
  const int s = a.fragS[a.candFrag[c]];
  const int b = a.best[c];
  a.refStart[c] = (a.firstPos[c] + a.lastPos[c]) / 2;
  uint32_t bits = 0;
  if (b > 0 && s <= a.lutMaxS && b >= a.minShared[s]) bits = a.idLUT[lut_off(s) + b];
  a.idBits[c] = bits;
}

struct OneWayArgs {
  int32_t nCand;
  const int32_t *candFrag, *candSeq, *refStart; const uint32_t *idBits;
  const int32_t *fragGenome;        // fragment -> query genome index inside the set the fragments were cut from
  int32_t genomeBase;               // ... minus this = query index inside the sub-batch
  const int32_t *contigGenome;      // reference contig -> reference genome
  const uint32_t *contigBinBase;    // reference contig -> first bin
  int32_t binWidth;                 // fragLen - 20
  uint32_t *bins; size_t binsPerQuery;
};

// one lane per candidate; the first candidate of each (fragment, genome) group resolves the group
static __global__ void k_oneway_bins(OneWayArgs a)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.nCand) return;
  const int f = a.candFrag[c];
  const int g = a.contigGenome[a.candSeq[c]];
  if (c > 0 && a.candFrag[c - 1] == f && a.contigGenome[a.candSeq[c - 1]] == g) return;   // not a group head
  uint32_t bBits = 0; int32_t bSeq = -1, bPos = -1;
  for (int x = c; x < a.nCand && a.candFrag[x] == f && a.contigGenome[a.candSeq[x]] == g; x++) {
    const uint32_t bits = a.idBits[x];
    if (!bits) continue;
    const int32_t sq = a.candSeq[x], ps = a.refStart[x];
    // cgid_types.hpp:35-36 order, keep the last = the maximum
    if (bits > bBits || (bits == bBits && (sq > bSeq || (sq == bSeq && ps > bPos)))) { bBits = bits; bSeq = sq; bPos = ps; }
  }
  if (bBits) {
    const size_t bin = (size_t)(a.fragGenome[f] - a.genomeBase) * a.binsPerQuery + a.contigBinBase[bSeq] + (uint32_t)(bPos / a.binWidth);
    atomicMax(&a.bins[bin], bBits);
  }
}

struct PairArgs {
  int32_t nQuery, nRefGenomes;
  const uint32_t *bins; size_t binsPerQuery;
  const uint32_t *genomeBinStart;      // [nRefGenomes+1] first bin of each reference genome
  // dense output [nQuery][outStride]: (countSeq, identity bits); countSeq == 0 = no row.  Dense and atomics-free, so the host
  // reads the rows back already in (query, reference) order.  A reference set that is split into several index chunks
  // fills one column block per chunk: this chunk's genome g goes to column outCol0 + g of the outStride genomes.
  uint32_t *pairCount; uint32_t *pairIdentity; int32_t outStride, outCol0;
};

// One wave per (query genome, reference genome) pair.  The genome's bins are read 64 at a time (coalesced); the float sum has
// to follow bin order, so the non-empty lanes of each 64-bin slab are added one after the other through v_readlane — unrelated
// pairs (nearly all of them) have no or a handful of non-empty bins and cost 27 loads + ballots per 5 Mbp reference.
static __global__ __launch_bounds__(kTPB) void k_pair_reduce(PairArgs a)
{
  const long long p = ((long long)blockIdx.x * kTPB + threadIdx.x) >> 6;      // wave-uniform
  if (p >= (long long)a.nQuery * a.nRefGenomes) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int qi = (int)(p / a.nRefGenomes), g = (int)(p % a.nRefGenomes);
  const uint32_t *b = a.bins + (size_t)qi * a.binsPerQuery;
  const uint32_t x1 = a.genomeBinStart[g + 1];
  float sum = 0.0f; int cnt = 0;
  for (uint32_t x0 = a.genomeBinStart[g]; x0 < x1; x0 += kWave) {
    const uint32_t bits = (x0 + lane < x1) ? b[x0 + lane] : 0u;
    unsigned long long m = __ballot(bits != 0);
    cnt += __popcll(m);
    while (m) {
      const int l = __ffsll(m) - 1;
      sum += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)bits, l));
      m &= m - 1;
    }
  }
  if (lane == 0) {
    const size_t o = (size_t)qi * (size_t)a.outStride + (size_t)(a.outCol0 + g);
    a.pairCount[o] = (uint32_t)cnt;
    a.pairIdentity[o] = cnt ? __float_as_uint(sum / cnt) : 0u;
  }
}

// mapping export for ani_map_query: compacted, candidate order preserved by a prior scan of the keep flags
static __global__ void k_emit_mappings(int32_t nCand, const int32_t *__restrict__ candFrag, const int32_t *__restrict__ candSeq,
                                const int32_t *__restrict__ refStart, const uint32_t *__restrict__ idBits,
                                const int32_t *__restrict__ best, const int32_t *__restrict__ fragS,
                                const int32_t *__restrict__ fragQuerySeqId, const uint32_t *__restrict__ outOff, int L,
                                int32_t seqBase /* first contig of the index chunk: refSeqId is global */,
                                uint32_t *__restrict__ out /* 11 words per mapping; upper bound filled on the host */)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nCand || !idBits[c]) return;
  uint32_t *m = out + 11 * (size_t)outOff[c];
  const int f = candFrag[c];
  m[0] = (uint32_t)L; m[1] = (uint32_t)refStart[c]; m[2] = (uint32_t)(refStart[c] + L - 1); m[3] = 0; m[4] = (uint32_t)(L - 1);
  m[5] = (uint32_t)(candSeq[c] + seqBase); m[6] = (uint32_t)fragQuerySeqId[f]; m[7] = idBits[c]; m[8] = 0;
  m[9] = (uint32_t)fragS[f]; m[10] = (uint32_t)best[c];
}

static __global__ void k_keep_flags(int32_t n, const uint32_t *__restrict__ idBits, int32_t *__restrict__ flags)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) flags[c] = idBits[c] != 0;
}

// out[i] = candidates of the i-th fragment in processing order (order == nullptr: fragment i)
static __global__ void k_clamp_counts(int32_t n, const int32_t *__restrict__ in, const int32_t *__restrict__ order, int32_t *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = in[order ? order[i] : i];
  out[i] = v > 0 ? v : 0;
}

// Processing order of the fragments of a multi-genome batch (see map_stage): sort key = running fragment id inside the genome
static __global__ void k_frag_order_keys(const int32_t *__restrict__ fragQSeq, int32_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n) { key[f] = (uint64_t)(uint32_t)fragQSeq[f]; idx[f] = (uint32_t)f; }
}

}  // namespace ani
