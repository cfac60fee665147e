// engine_map.hip — mapping and reduction: L1 seed lookup + candidate regions, L2 sliding MinHash, identity filter
// (≙ skch::Map, src/map/include/computeMap.hpp:112-545) and the ANI reducer (≙ cgi::computeCGI,
// src/cgi/include/computeCoreIdentity.hpp:166-298), for one query genome or fused for whole batches / kept fragment sets.
#include <atomic>
#include <thread>
#include "host/engine.hpp"
#include "kernels/l1.hpp"
#include "kernels/l2.hpp"
#include "kernels/reduce.hpp"

namespace anih {
using namespace ani;

// engine_l2.hip (a unit with its own compiler flags): the codes and simulation kernels of the L2 stage
void launch_l2_codes(unsigned grid, hipStream_t s, const L2FastArgs &fa);
void launch_l2_sim_a(unsigned grid, hipStream_t s, const L2FastArgs &fa, const int32_t *list, const unsigned int *listCount);
void launch_l2_sim_b(unsigned grid, hipStream_t s, const L2FastArgs &fa, const int32_t *list, const unsigned int *listCount);
static_assert(kL1FilterMinHits == 300 && kL1HitCapMax == 4080, "defaults of ani_ctx::l1FilterMin / l1LdsMax (host/engine.hpp)");

// L1 + L2 + identity for the fragments of `fs` against ONE index chunk; candidates and their results stay in the context's
// buffers (ocFrag/ocSeq [chunk-local seqIds]/refStart/idBits/l2Best) for the reducer or the mapping export.
int map_stage(ani_ctx *ctx, ani_sketch *set, IndexChunk *sk, const FragSet &fs, int32_t *nCandOut)
{
  if (ctx->timerPending.size() > 4096) flush_timers(ctx);
  const ani_params_t &p = set->params;
  const int k = p.kmerSize, w = p.windowSize, L = p.fragLen;
  const size_t nF = (size_t)fs.nFrag;
  const int maxS = fs.maxS;
  *nCandOut = 0;
  if (nF == 0) return ANI_OK;
  unsigned long long host[CNT_N];
  host[CNT_QPOOL] = fs.nHashes;
  ctx->counters.l1Probes += fs.nHashes;

  // ---- processing order ----
  // Fragments are numbered genome by genome.  The genomes of a batch are often close relatives (an all-vs-all run over a species, a
  // database sorted by taxonomy): fragment k of genome A and fragment k of its neighbour B then probe the same index entries, get the
  // same candidates and re-read the same reference ranges.  Processed in fragment order they are a whole genome (1666 workgroups,
  // 200 MB of traffic) apart; processed "k-th fragments of all genomes, then the (k+1)-th" they are neighbours and meet in one L2
  // (kernels map an XCD to a contiguous run of the order, xcd_item).  Results do not depend on the order: every kernel writes through
  // the fragment id, the candidates of a fragment stay contiguous, and single-genome calls (ani_map_query, whose mappings are
  // returned in callback order) keep the ascending order.
  const int32_t *fragOrder = nullptr;
  { const char *ev = getenv("ANI_TEST_FRAG_ORDER");
    if (fs.genomeFragments.size() >= 2 && fs.fragQSeq && !(ev && !strcmp(ev, "plain"))) {
      TRY(ctx->fragOrder.ensure(nF * 4)); TRY(ctx->fragOrderTmp.ensure(nF * 20));
      uint64_t *keyIn = ctx->fragOrderTmp.as<uint64_t>(), *keyOut = keyIn + nF; uint32_t *idxIn = (uint32_t *)(keyOut + nF);
      hipLaunchKernelGGL(k_frag_order_keys, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, fs.fragQSeq, (int32_t)nF, keyIn, idxIn);
      size_t tb = 0;
      int keyBits = 1;                              // keys are running fragment ids inside a genome: 11 bits for 5 Mbp genomes, two 8-bit passes
      { int32_t mx = 0; for (int32_t v : fs.genomeFragments) mx = std::max(mx, v); while (keyBits < 32 && (1ll << keyBits) <= (long long)mx) keyBits++; }
      int rc = ani_sort_pairs_u64_u32(keyIn, keyOut, idxIn, ctx->fragOrder.as<uint32_t>(), nF, keyBits, nullptr, &tb, ctx->stream);
      if (rc == 0) { TRY(ctx->sortTmp.ensure(tb + 256));
        rc = ani_sort_pairs_u64_u32(keyIn, keyOut, idxIn, ctx->fragOrder.as<uint32_t>(), nF, keyBits, ctx->sortTmp.p, &tb, ctx->stream); }
      if (rc != 0) return fail(ANI_ERR_DEVICE, "radix sort of the fragment order failed (%d)", rc);
      fragOrder = ctx->fragOrder.as<int32_t>();
    } }

  // ---- L1 ----
  TRY(ctx->fragCandOff.ensure(nF * 4)); TRY(ctx->fragCandCnt.ensure(nF * 4)); TRY(ctx->fragCandCntClamped.ensure(nF * 4));
  TRY(ctx->fragHits.ensure(nF * 4)); TRY(ctx->fragOrdOff.ensure((nF + 1) * 4));
  TRY(ctx->probeFirst.ensure((fs.poolSize + 1) * 4)); TRY(ctx->probeCnt.ensure((fs.poolSize + 1) * 4));      // indexed like the sketch pool
  // First guess of the candidate pool from the context's running estimate, never beyond what 32-bit candidate ids allow (the limit
  // is an error only when the batch really needs more: the retry below).
  const uint64_t kCandLimit = 0x7fffff00ull;      // whole stripes below 2^31
  // at least 4096 per stripe (4 MB): a few heavy fragments fit without a retry
  uint64_t ccap = std::min<uint64_t>(std::max<uint64_t>((uint64_t)((double)nF * ctx->candPerFrag) + ctx->candPoolMin, ctx->candPoolMin * kPoolStripes),
      kCandLimit);
  const bool smallBatch = nF < 16 * (size_t)kPoolStripes;
  if (smallBatch) ccap = std::min<uint64_t>(std::max<uint64_t>(ccap, ctx->smallBatchCandCap), kCandLimit);
  TRY(ctx->l1MidList.ensure(nF * 4)); TRY(ctx->l1BigList.ensure(nF * 4)); TRY(ctx->l1FragDesc.ensure(nF * sizeof(L1FragDesc)));
  unsigned nMid = 0, nBig = 0; unsigned long long nTiny = 0, nSmall = 0;
  std::vector<int32_t> bigFrags, bigInfo;          // fragments beyond the LDS classes and their (sketch size, seed hits)
  unsigned long long hitsTotal = 0;
  uint32_t probeOverflow = 0;                       // fragments k_l1_probe marked with >= 2^31 seed hits: latched after attempt 0 (the probe runs once)
  for (int attempt = 0;; attempt++) {
    ccap = (uint64_t)stripe_cap(ccap) * kPoolStripes;
    TRY(ctx->candFrag.ensure(ccap * 4)); TRY(ctx->candSeq.ensure(ccap * 4)); TRY(ctx->candStart.ensure(ccap * 4)); TRY(ctx->candEnd.ensure(ccap * 4));
    if (attempt == 0) TRY(zero_counters(ctx));
    // only the candidate pool is redone: k_l1_probe's results (CNT_HITS, CNT_NEG = its overflow marker, the class lists) stay
    else TRY(zero_cursors(ctx, POOL_CAND));
    L1Args a;
    a.qPool = fs.qPool; a.fragOff = fs.fragOff; a.fragS = fs.fragS; a.nFrag = (int32_t)nF;
    a.table = sk->table; a.tableSlots = sk->tableSlots; a.sSW = sk->sSW; a.bucketW = w; a.nIndex = sk->n;
    a.minHitsLUT = set->dMinHits; a.lutMaxS = set->dLutMaxS; a.L = L;
    a.candFrag = ctx->candFrag.as<int32_t>(); a.candSeq = ctx->candSeq.as<int32_t>(); a.candStart = ctx->candStart.as<int32_t>();
    a.candEnd = ctx->candEnd.as<int32_t>();
    a.candCap = stripe_cap(ccap); a.candCount = cur_ptr(ctx, POOL_CAND);
    a.fragCandOff = ctx->fragCandOff.as<uint32_t>(); a.fragCandCnt = ctx->fragCandCnt.as<int32_t>(); a.fragHits = ctx->fragHits.as<int32_t>();
    a.sumHits = cnt_ptr(ctx, CNT_HITS); a.tinyCount = cnt_ptr(ctx, CNT_TINY); a.smallCount = cnt_ptr(ctx, CNT_SMALL);
    a.filterShift = 1; while (a.filterShift < 30 && (1 << a.filterShift) < 2 * L) a.filterShift++;
    a.fragOrder = fragOrder; a.fragDesc = ctx->l1FragDesc.as<L1FragDesc>();
    a.filterMinHits = ctx->l1FilterMin; a.ldsHitCap = ctx->l1LdsMax; a.tinyPath = ctx->l1Tiny ? 1 : 0;
    a.probeFirst = ctx->probeFirst.as<uint32_t>(); a.probeCnt = ctx->probeCnt.as<uint32_t>();
    a.midList = ctx->l1MidList.as<int32_t>(); a.midCount = (unsigned int *)cnt_ptr(ctx, CNT_LISTM);
    a.bigList = ctx->l1BigList.as<int32_t>(); a.bigCount = (unsigned int *)cnt_ptr(ctx, CNT_LISTBIG);
    a.overflowCount = (unsigned int *)cnt_ptr(ctx, CNT_NEG); a.hitLimit = ctx->l1HitLimit;
    {
      StageTimer tm(ctx, &ctx->counters.msL1);
      if (attempt == 0) {
        { StageTimer tk(ctx, &ctx->counters.msL1Probe, 1);
          hipLaunchKernelGGL(k_l1_probe, dim3(pad8((nF + kL1ProbeFrags - 1) / kL1ProbeFrags)), dim3(kTPB), 0, ctx->stream, a); }
        // the probe routed the fragments to their classes: how many of each decides what is launched
        unsigned long long cls[CNT_N];
        TRY(read_counters(ctx, cls));
        nMid = (unsigned)cls[CNT_LISTM]; nBig = (unsigned)cls[CNT_LISTBIG]; nTiny = cls[CNT_TINY]; nSmall = cls[CNT_SMALL];
        ctx->counters.l1TinyFragments += nTiny;
      }
      if (nTiny && ctx->l1Tiny) { StageTimer tk(ctx, &ctx->counters.msL1Tiny, 1);
        hipLaunchKernelGGL(k_l1_tiny, dim3(pad8((nF + kL1TinyFrags - 1) / kL1TinyFrags)), dim3(kTPB), 0, ctx->stream, a); }
      {
        StageTimer tk(ctx, &ctx->counters.msL1Main, 1);
        // the usual case: (nearly) every fragment is of class S, one workgroup per fragment
        if (2 * nSmall >= nF || getenv("ANI_TEST_L1_DENSE"))
          hipLaunchKernelGGL((k_l1<0, kL1HitCapSmall>), dim3(pad8(nF)), dim3(kTPB), 0, ctx->stream, a, (const int32_t *)nullptr);
        else {                                                            // a fragment set against a foreign shard / chunk: class S is the exception, listed
          TRY(ctx->l1SmallList.ensure((nSmall + 1) * 4));
          HIP_TRY(hipMemsetAsync(cnt_ptr(ctx, CNT_LISTL), 0, 8, ctx->stream));
          hipLaunchKernelGGL(k_l1_list, dim3(grid_for(nF, kTPB)), dim3(kTPB), 0, ctx->stream, a, ctx->l1SmallList.as<int32_t>(), (unsigned int *)cnt_ptr(ctx,
              CNT_LISTL));
          if (nSmall) hipLaunchKernelGGL((k_l1<0, kL1HitCapSmall>), dim3((unsigned)nSmall), dim3(kTPB), 0, ctx->stream, a,
              (const int32_t *)ctx->l1SmallList.as<int32_t>());
        }
      }
      if (attempt == 0) {
        if (nBig) {
          bigFrags.resize(nBig); bigInfo.resize(2 * (size_t)nBig);
          TRY(ctx->l1BigV.ensure((size_t)nBig * 8));
          hipLaunchKernelGGL(k_l1_big_info, dim3(grid_for(nBig)), dim3(256), 0, ctx->stream, (const int32_t *)ctx->l1BigList.as<int32_t>(), nBig, fs.fragS,
                             (const int32_t *)ctx->fragHits.as<int32_t>(), ctx->l1BigV.as<int32_t>());
          HIP_TRY(hipMemcpyAsync(bigFrags.data(), ctx->l1BigList.p, (size_t)nBig * 4, hipMemcpyDeviceToHost, ctx->stream));
          HIP_TRY(hipMemcpyAsync(bigInfo.data(), ctx->l1BigV.p, (size_t)nBig * 8, hipMemcpyDeviceToHost, ctx->stream));
          HIP_TRY(hipStreamSynchronize(ctx->stream));
          ctx->counters.l1BigFragments += nBig;
        }
      }
      if (nMid) hipLaunchKernelGGL((k_l1<kL1HitCapSmall, kL1HitCapMid>), dim3(nMid), dim3(kTPB), 0, ctx->stream, a, (const int32_t *)a.midList);
      if (nBig) {
        // Fragments beyond every LDS class, batched (l1.hpp): groups of fragments whose hits fit the key buffers; one 64-bit key per
        // hit = (fragment rank in the group, seqId, wpos), field widths from this chunk's contig count and longest contig.
        StageTimer tb(ctx, &ctx->counters.msL1Big, 1);
        int bitsPos = 1, bitsSeq = 1;
        while (bitsPos < 31 && (1ll << bitsPos) <= (long long)sk->maxContigLen) bitsPos++;
        while (bitsSeq < 31 && (1ll << bitsSeq) <= (long long)sk->nContigs) bitsSeq++;
        const int shiftSeq = bitsPos, shiftRank = bitsPos + bitsSeq;
        const uint64_t maxFrags = std::min<uint64_t>(1ull << std::min(20, 64 - shiftRank), ctx->l1BigGroupFrags);
        const uint64_t budget = ctx->l1BigGroupHits;
        for (size_t b0 = 0; b0 < nBig;) {
          size_t b1 = b0; uint64_t hits = 0, hashes = 0, tiles = 0;
          std::vector<uint32_t> sOff{0}, tileFirst{0}; std::vector<uint64_t> hitOff{0};
          while (b1 < nBig && b1 - b0 < maxFrags && (b1 == b0 || hits + (uint64_t)bigInfo[2 * b1 + 1] <= budget)) {
            const uint64_t sz = (uint64_t)std::max(bigInfo[2 * b1], 0), H = (uint64_t)std::max(bigInfo[2 * b1 + 1], 0);
            hashes += sz; hits += H; tiles += (H + kL1BigTileHits - 1) / kL1BigTileHits;
            sOff.push_back((uint32_t)hashes); hitOff.push_back(hits); tileFirst.push_back((uint32_t)tiles);
            b1++;
          }
          const size_t n = b1 - b0;
          if (tiles > 0x7ffffff0ull || hashes > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "group of oversized fragments with %llu seed hits",
              (unsigned long long)hits);
          // group tables: [frag n][sOff n+1][tileFirst n+1] as 32-bit words, then hitOff (64-bit) — one upload
          const size_t w32 = n + 2 * (n + 1), tblBytes = ((w32 * 4 + 7) / 8) * 8 + (n + 1) * 8;
          TRY(ctx->l1BigTbl.ensure(tblBytes)); TRY(ctx->l1BigHash.ensure((hashes ? hashes : 1) * 4));
          TRY(ctx->l1BigHitsA.ensure((hits ? hits : 1) * 8)); TRY(ctx->l1BigHitsB.ensure((hits ? hits : 1) * 8));
          TRY(ctx->l1BigV.ensure((hits ? hits : 1) * 4 + (size_t)nBig * 8));
          uint8_t *hostTbl = nullptr;
          TRY(pinned_buffer(ctx, 2, tblBytes, (void **)&hostTbl));
          uint32_t *h32 = (uint32_t *)hostTbl;
          memcpy(h32, bigFrags.data() + b0, n * 4); memcpy(h32 + n, sOff.data(), (n + 1) * 4); memcpy(h32 + n + (n + 1), tileFirst.data(), (n + 1) * 4);
          memcpy(hostTbl + ((w32 * 4 + 7) / 8) * 8, hitOff.data(), (n + 1) * 8);
          HIP_TRY(hipMemcpyAsync(ctx->l1BigTbl.p, hostTbl, tblBytes, hipMemcpyHostToDevice, ctx->stream));
          L1BigArgs g;
          const uint32_t *d32 = ctx->l1BigTbl.as<uint32_t>();
          g.frag = (const int32_t *)d32; g.sOff = d32 + n; g.tileFirst = d32 + n + (n + 1);
          g.hitOff = (const uint64_t *)(ctx->l1BigTbl.as<uint8_t>() + ((w32 * 4 + 7) / 8) * 8);
          g.hashOff = ctx->l1BigHash.as<int32_t>(); g.keys = ctx->l1BigHitsA.as<uint64_t>(); g.n = (int)n; g.shiftSeq = shiftSeq; g.shiftRank = shiftRank;
          hipLaunchKernelGGL(k_l1_big_offsets, dim3((unsigned)n), dim3(kTPB), 0, ctx->stream, a, g);
          if (tiles) {
            hipLaunchKernelGGL(k_l1_big_gather, dim3((unsigned)tiles), dim3(kTPB), 0, ctx->stream, a, g);
            int rankBits = 1; while (rankBits < 64 - shiftRank && (1ull << rankBits) < n) rankBits++;
            size_t tbytes = 0;
            int rc = ani_sort_keys_u64_bits(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), (size_t)hits, shiftRank + rankBits, nullptr,
                &tbytes, ctx->stream);
            if (rc == 0) { TRY(ctx->sortTmp.ensure(tbytes + 256));
              rc = ani_sort_keys_u64_bits(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), (size_t)hits, shiftRank + rankBits, ctx->sortTmp.p,
                &tbytes, ctx->stream); }
            if (rc != 0) return fail(ANI_ERR_DEVICE, "radix sort of seed hits failed (%d)", rc);
            hipLaunchKernelGGL(k_l1_big_unpack, dim3(grid_for((size_t)hits, 256, 65535)), dim3(256), 0, ctx->stream, ctx->l1BigHitsB.as<uint64_t>(),
                (uint64_t)hits, shiftSeq, shiftRank);
          }
          g.keys = ctx->l1BigHitsB.as<uint64_t>();
          hipLaunchKernelGGL(k_l1_big_candidates, dim3((unsigned)n), dim3(kTPB), 0, ctx->stream, a, g, ctx->l1BigV.as<int>() + 2 * (size_t)nBig);
          HIP_TRY(hipGetLastError());
          HIP_TRY(hipStreamSynchronize(ctx->stream));      // the pinned table and the group buffers are reused by the next group
          b0 = b1;
        }
      }
      {
      hipLaunchKernelGGL(k_clamp_counts, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, (int32_t)nF, ctx->fragCandCnt.as<int32_t>(), fragOrder,
                         ctx->fragCandCntClamped.as<int32_t>());
      }
    }
    HIP_TRY(hipGetLastError());
    TRY(read_counters(ctx, host));
    if (attempt == 0) { hitsTotal = host[CNT_HITS]; probeOverflow = (uint32_t)host[CNT_NEG]; ctx->counters.l1MidFragments += nMid; }
    if (ctx->poolMaxStripe[POOL_CAND] <= stripe_cap(ccap)) break;
    if (attempt > 2) return fail(ANI_ERR_INTERNAL, "candidate pool did not converge");
    if (ccap >= kCandLimit) return fail(ANI_ERR_LIMIT, "more than 2^31 L1 candidates in one query batch");
    ccap = std::min<uint64_t>((uint64_t)(1.25 * (double)ctx->poolMaxStripe[POOL_CAND] * kPoolStripes) + ctx->candPoolMin, kCandLimit);
  }
  // Size the pool right next time (by the fullest stripe).  Only a batch that fills the stripes evenly says anything about the next
  // one: a handful of fragments sit in a handful of stripes, and "fullest stripe x 64 / fragments" of a one-fragment batch with
  // 1000 candidates would ask for 80 000 candidates per fragment of the next, million-fragment batch (a bogus 2^31 limit error
  // after 34 GB of pool; seen in the parity suite under ANI_TEST_POOL_POISON).  The estimate follows the batches down as well as up.
  if (!smallBatch) {                             // (a one-to-many query of 1666 fragments counts: without its update every call ran the L1 kernels twice)
    const double seen = 1.25 * (double)ctx->poolMaxStripe[POOL_CAND] * kPoolStripes / (double)nF;
    ctx->candSeen[ctx->candSeenAt++ & 31] = seen;
    double mx = 12.0;
    for (double v : ctx->candSeen) mx = std::max(mx, v);
    ctx->candPerFrag = mx;
  // the next small batch starts from what this one needed (bounded: 1 GB of pool)
  } else ctx->smallBatchCandCap = std::min<uint64_t>(grown_cap(ctx->poolMaxStripe[POOL_CAND]), (uint64_t)1 << 26);
  if (probeOverflow != 0)                      // k_l1_probe: hit counts and offsets are 32-bit per fragment
    return fail(ANI_ERR_LIMIT, "%u query fragment(s) have 2^31 or more seed hits in one index chunk (a hash with ~10^9 occurrences: low-complexity / "
                               "repetitive references); use the reference's -s sanity check or a smaller ANI_MAX_INDEX_MINIMIZERS", probeOverflow);
  ctx->counters.seedHits += hitsTotal;
  uint64_t nCand = 0;
  {
    StageTimer tm(ctx, &ctx->counters.msL1);
    TRY(device_scan(ctx, ctx->fragCandCntClamped.as<int32_t>(), ctx->fragOrdOff.as<uint32_t>(), (uint32_t)nF, &nCand));
    if (nCand) {
      TRY(ctx->ocFrag.ensure(nCand * 4)); TRY(ctx->ocSeq.ensure(nCand * 4)); TRY(ctx->ocStart.ensure(nCand * 4)); TRY(ctx->ocEnd.ensure(nCand * 4));
      hipLaunchKernelGGL(k_l1_order, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, ctx->fragCandOff.as<uint32_t>(), ctx->fragCandCntClamped.as<int32_t>(),
                         ctx->fragOrdOff.as<uint32_t>(), fragOrder, (int32_t)nF, ctx->candSeq.as<int32_t>(), ctx->candStart.as<int32_t>(),
                             ctx->candEnd.as<int32_t>(),
                         ctx->ocFrag.as<int32_t>(), ctx->ocSeq.as<int32_t>(), ctx->ocStart.as<int32_t>(), ctx->ocEnd.as<int32_t>());
      HIP_TRY(hipGetLastError());
    }
  }
  *nCandOut = (int32_t)nCand;
  ctx->counters.l1Candidates += nCand;
  if (nCand == 0) return ANI_OK;

  // ---- L2 ----
  TRY(ctx->l2Best.ensure(nCand * 4)); TRY(ctx->l2First.ensure(nCand * 4)); TRY(ctx->l2Last.ensure(nCand * 4));
  TRY(ctx->refStart.ensure(nCand * 4)); TRY(ctx->idBits.ensure(nCand * 4));
  {
    StageTimer tm(ctx, &ctx->counters.msL2);
    TRY(zero_counters(ctx));
    L2Args a;
    a.candFrag = ctx->ocFrag.as<int32_t>(); a.candSeq = ctx->ocSeq.as<int32_t>(); a.candStart = ctx->ocStart.as<int32_t>();
    a.candEnd = ctx->ocEnd.as<int32_t>();
    a.nCand = (int32_t)nCand; a.qPool = fs.qPool; a.fragOff = fs.fragOff; a.fragS = fs.fragS;
    a.mHash = sk->mHash; a.mWpos = sk->mWpos; a.dup = DupLinks{sk->dupList, sk->nDup, sk->dupBits}; a.mWin = sk->mWin; a.posBase = sk->posBase;
    a.posSample = sk->posSample;
    a.contigFirstMin = sk->contigFirstMin;
    { int lg = 0; while ((2 << lg) <= w) lg++; a.rankShift = 21 - std::max(0, lg - 1); }   // w = 24: 2048 buckets over [0, 2^29)
    a.L = L; a.w = w; a.k = k; a.scratch = nullptr; a.laneStride = 0;
    a.outBest = ctx->l2Best.as<int32_t>(); a.outFirst = ctx->l2First.as<int32_t>(); a.outLast = ctx->l2Last.as<int32_t>();
    a.sumEntries = cnt_ptr(ctx, CNT_ENTRIES); a.sumSteps = cnt_ptr(ctx, CNT_STEPS); a.sumQ = cnt_ptr(ctx, CNT_SUMQ);

    // ordered candidate offset per fragment on the host: chunk [c0,c1) -> fragment range
    uint32_t *ordOff = nullptr;
    TRY(pinned_buffer(ctx, 0, nF * 4, (void **)&ordOff));
    HIP_TRY(hipMemcpyAsync(ordOff, ctx->fragOrdOff.p, nF * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));

    // Chunks of 2^21 candidates, two buffer sets.  Main stream per chunk: ranges -> scan -> length ordering -> codes -> class-A
    // simulation.  Side stream, after the chunk's class-A launch: the few class-B candidates (s in 256..319; their launch is
    // bound by the serial length of one lane, not by throughput) and the collection of the leftovers — they run underneath the
    // next chunk's ranges/codes kernels, which leave the LDS free, instead of extending every chunk by a latency-bound tail.
    const size_t CH = ctx->l2ChunkCandidates;   // candidates per chunk, 2^21 (a chunk whose code entries exceed 2^32 is rejected by the scan)
    for (int p = 0; p < 2; p++) {
      TRY(ctx->l2Ranges[p].ensure(CH * sizeof(L2Range))); TRY(ctx->l2CodeCount[p].ensure(CH * 4)); TRY(ctx->l2CodeOff[p].ensure(CH * 4));
      TRY(ctx->l2SlowFlag[p].ensure(CH * 4)); TRY(ctx->l2ClassList[p].ensure(CH * 4));
      TRY(ctx->l2Order[p].ensure(CH * 4)); TRY(ctx->l2LenHist[p].ensure((kL2LenBuckets + 4) * 4));
    }
    TRY(ctx->l2SlowList.ensure(nCand * 4));
    HIP_TRY(hipMemsetAsync(cnt_ptr(ctx, CNT_NEG), 0, 8, ctx->stream));
    struct SideJoin { ani_ctx *c; ~SideJoin() { (void)hipStreamSynchronize(c->stream2); } } sideJoin{ctx};   // also on the error paths below
    HIP_TRY(hipEventRecord(ctx->evSetDone[0], ctx->stream)); HIP_TRY(hipEventRecord(ctx->evSetDone[1], ctx->stream));   // both sets free, counters zeroed
    HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->evSetDone[1], 0));
    // (measured, round 5: a small job — a rank's shard of a sharded all-vs-all, two chunks — cut into four so that more of the codes
    //  are written beside a simulation: L2 stage 12.9 -> 13.4 ms, profiles/r05y_ab_l2_chunk_auto_sim8.txt; not kept)
    size_t chunk = CH;
    int nChunk = 0;
    for (size_t c0 = 0; c0 < nCand;) {
      const size_t c1 = std::min<size_t>(nCand, c0 + chunk);
      const size_t n = c1 - c0;
      const int p = nChunk & 1;
      HIP_TRY(hipEventSynchronize(ctx->evSetDone[p]));                 // host: set p may be reallocated; (long done: two chunks ago)
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->evSetDone[p], 0));
      L2FastArgs fa;
      fa.g = a; fa.c0 = (int32_t)c0; fa.c1 = (int32_t)c1;
      fa.ranges = ctx->l2Ranges[p].as<L2Range>(); fa.codeCount = ctx->l2CodeCount[p].as<int32_t>(); fa.codeOff = ctx->l2CodeOff[p].as<uint32_t>();
      fa.codes = nullptr; fa.slowFlag = ctx->l2SlowFlag[p].as<int32_t>(); fa.fragCandOff = ctx->fragOrdOff.as<uint32_t>(); fa.fragOrder = fragOrder;
      fa.nFrag = (int32_t)nF;
      // fragments that own candidates c0 and c1-1 (ordOff is non-decreasing; fragments without candidates repeat a value)
      const int32_t fA = (int32_t)(std::upper_bound(ordOff, ordOff + nF, (uint32_t)c0) - ordOff) - 1;
      const int32_t fB = (int32_t)(std::upper_bound(ordOff, ordOff + nF, (uint32_t)(c1 - 1)) - ordOff) - 1;
      fa.fragBase = fA;
      { const char *ev = getenv("ANI_TEST_L2_PATH"); fa.allowFast = (ev && !strcmp(ev, "general")) ? 0 : (ev && !strcmp(ev, "classB")) ? 2 : 1; }
      // the 14-bit window links need cmw + 2 < 2^14 (and a window at all)
      if (L - (w - 1) - (k - 1) + 2 > (int)kWinMask || L - (w - 1) - (k - 1) < 1) fa.allowFast = 0;
      {
        StageTimer tk(ctx, &ctx->counters.msL2Ranges, 1);
        hipLaunchKernelGGL(k_l2_ranges, dim3(grid_for(n)), dim3(kTPB), 0, ctx->stream, fa);
      }
      fa.nFragChunk = fB - fA + 1;
      uint64_t nCodes = 0;
      {
        const int rc = device_scan(ctx, fa.codeCount, ctx->l2CodeOff[p].as<uint32_t>(), (uint32_t)n, &nCodes, ctx->l2CodeLimit);
        // very long candidate ranges: smaller chunk, same candidates again
        if (rc == ANI_ERR_LIMIT && n > 1) { chunk = (n + 1) / 2; ctx->counters.l2ChunkHalvings++; continue; }
        TRY(rc);
      }
      TRY(ctx->l2Codes[p].ensure((nCodes + 64) * 2));
      fa.codes = ctx->l2Codes[p].as<uint32_t>();
      if (nCodes) {
        {
          // order the chunk's candidates by code-stream length (longest first) for the simulation
          StageTimer tk(ctx, &ctx->counters.msL2Ranges, 1);
          HIP_TRY(hipMemsetAsync(ctx->l2LenHist[p].p, 0, kL2LenBuckets * 4, ctx->stream));
          // list length for the simulation launch
          HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ctx->l2LenHist[p].as<unsigned int>() + kL2LenBuckets), (int)n, 1, ctx->stream));
          hipLaunchKernelGGL(k_l2_len_hist, dim3(grid_for(n, kTPB, 2048)), dim3(kTPB), 0, ctx->stream, (const int32_t *)fa.codeCount, (int32_t)n,
              ctx->l2LenHist[p].as<unsigned int>());
          hipLaunchKernelGGL(k_l2_len_scan, dim3(1), dim3(kTPB), 0, ctx->stream, ctx->l2LenHist[p].as<unsigned int>());
          hipLaunchKernelGGL(k_l2_len_scatter, dim3(grid_for(n, kTPB)), dim3(kTPB), 0, ctx->stream, (const int32_t *)fa.codeCount, (int32_t)c0, (int32_t)n,
                             ctx->l2LenHist[p].as<unsigned int>(), ctx->l2Order[p].as<int32_t>());
        }
        {
          StageTimer tk(ctx, &ctx->counters.msL2Codes, 1);
          launch_l2_codes((unsigned)((fa.nFragChunk + 7) / 8 * 8), ctx->stream, fa);
        }
        // The simulation (VALU-bound) runs on the side stream so that the next chunk's ranges / codes kernels (memory- and latency-
        // bound) start beside it; ANI_TEST_L2_OVERLAP=0 keeps everything on the main stream.  Measured (1000 x 1000; round 3:
        // profiles/r03f_bench_overlap.json.log, round 4 A/B/A/B on one box: profiles/r04p_overlap_ab.txt): the L2 stage 91.4 -> 87.8 ms,
        // the step 211.5 -> 207.9 ms — all of it from the tails: a codes workgroup (25 KiB of LDS) does not fit the 16 KiB slot a
        // retiring simulation workgroup frees.  Default since round 4.  With it the per-kernel times (bench line, rocprofv3) are those
        // of kernels that share the machine for part of their run; the stage time (msL2, bracketed on the main stream, which waits for
        // both) is what the headline roofline fraction is computed from.
        const bool overlap = ctx->l2Overlap;
        hipStream_t simStream = ctx->stream;
        if (overlap) {
          HIP_TRY(hipEventRecord(ctx->evSimA[p], ctx->stream));            // codes of this chunk are written
          HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->evSimA[p], 0));
          simStream = ctx->stream2;
        }
        {
          StageTimer tk(ctx, &ctx->counters.msL2Kernel, 1, simStream);
          launch_l2_sim_a(grid_for(n, kL2SimTPB), simStream, fa, (const int32_t *)ctx->l2Order[p].as<int32_t>(),
              (const unsigned int *)ctx->l2LenHist[p].as<unsigned int>() + kL2LenBuckets);
        }
        ctx->counters.l2Launches++;
      }
      HIP_TRY(hipEventRecord(ctx->evSimA[p], ctx->stream));
      HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->evSimA[p], 0));
      if (nCodes) {
        StageTimer tk(ctx, &ctx->counters.msL2SimB, 1, ctx->stream2);
        // class-B candidates are compacted first so that they fill whole waves
        HIP_TRY(hipMemsetAsync(cnt_ptr(ctx, CNT_CLASSB), 0, 8, ctx->stream2));
        hipLaunchKernelGGL(k_l2_collect_class, dim3(grid_for(n)), dim3(256), 0, ctx->stream2, (int32_t)c0, (int32_t)n, (const int32_t *)fa.slowFlag, 4,
                           ctx->l2ClassList[p].as<int32_t>(), (unsigned int *)cnt_ptr(ctx, CNT_CLASSB));
        L2FastArgs fb = fa;       // class B accumulates its algorithmic-byte counters separately
        fb.g.sumEntries = cnt_ptr(ctx, CNT_ENTRIES_B); fb.g.sumQ = cnt_ptr(ctx, CNT_SUMQ_B); fb.g.sumSteps = cnt_ptr(ctx, CNT_STEPS_B);
        launch_l2_sim_b(grid_for(n, kL2SimTPB), ctx->stream2, fb, (const int32_t *)ctx->l2ClassList[p].as<int32_t>(), (const unsigned int *)cnt_ptr(ctx,
            CNT_CLASSB));
      }
      // whatever did not qualify (or overflowed a gap counter) is appended to the sub-batch's list for the general kernel
      hipLaunchKernelGGL(k_l2_collect_slow, dim3(grid_for(n)), dim3(256), 0, ctx->stream2, (int32_t)c0, (int32_t)n, (const int32_t *)fa.slowFlag,
                         ctx->l2SlowList.as<int32_t>(), (unsigned int *)cnt_ptr(ctx, CNT_NEG), cnt_ptr(ctx, CNT_REASON));
      HIP_TRY(hipEventRecord(ctx->evSetDone[p], ctx->stream2));
      HIP_TRY(hipGetLastError());
      c0 = c1; nChunk++;
    }
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->evSetDone[0], 0)); HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->evSetDone[1], 0));
    unsigned long long nSlow64 = 0;
    HIP_TRY(hipMemcpyAsync(&nSlow64, cnt_ptr(ctx, CNT_NEG), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nSlowTotal = (size_t)(uint32_t)nSlow64;
    if (nSlowTotal) {
      size_t lanes = (std::min<size_t>(nSlowTotal, (size_t)1 << 17) + kTPB - 1) / kTPB * kTPB;
      const size_t wordsPerLane = (size_t)maxS + 1;
      while (lanes > kTPB && lanes * wordsPerLane * 4 > ((size_t)2 << 30)) lanes = (lanes / 2 + kTPB - 1) / kTPB * kTPB;
      TRY(ctx->l2Scratch.ensure(lanes * wordsPerLane * 4));
      L2Args sa = a; sa.scratch = ctx->l2Scratch.as<uint32_t>(); sa.laneStride = lanes;
      StageTimer tk(ctx, &ctx->counters.msL2Slow, 1);
      for (size_t base = 0; base < nSlowTotal; base += lanes)
        hipLaunchKernelGGL(k_l2, dim3((unsigned)(lanes / kTPB)), dim3(kTPB), 0, ctx->stream, sa, (const int32_t *)ctx->l2SlowList.as<int32_t>(),
            (int32_t)nSlowTotal, (int32_t)base);
      HIP_TRY(hipGetLastError());
    }
    ctx->counters.l2SlowCandidates += nSlowTotal; ctx->counters.l2FastCandidates += nCand - nSlowTotal;
    FinishArgs fa;
    fa.nCand = (int32_t)nCand; fa.candFrag = a.candFrag; fa.candSeq = a.candSeq; fa.best = a.outBest; fa.firstPos = a.outFirst; fa.lastPos = a.outLast;
    fa.fragS = a.fragS; fa.idLUT = set->dIdLUT; fa.minShared = set->dMinShared; fa.lutMaxS = set->dLutMaxS;
    fa.refStart = ctx->refStart.as<int32_t>(); fa.idBits = ctx->idBits.as<uint32_t>();
    hipLaunchKernelGGL(k_finish_candidates, dim3(grid_for(nCand)), dim3(256), 0, ctx->stream, fa);
    HIP_TRY(hipGetLastError());
  }
  TRY(read_counters(ctx, host));
  ctx->counters.l2WindowEntries += host[CNT_ENTRIES] + host[CNT_ENTRIES_B]; ctx->counters.l2Steps += host[CNT_STEPS] + host[CNT_STEPS_B];
  ctx->counters.l2QueryHashes += host[CNT_SUMQ] + host[CNT_SUMQ_B];
  ctx->counters.l2WindowEntriesB += host[CNT_ENTRIES_B]; ctx->counters.l2QueryHashesB += host[CNT_SUMQ_B];
  ctx->counters.l2SlowLimit += host[CNT_REASON + 1]; ctx->counters.l2SlowDup += host[CNT_REASON + 2]; ctx->counters.l2SlowOverflow += host[CNT_REASON + 3];
  return ANI_OK;
}

// 1-way / 2-way / mean (computeCoreIdentity.hpp:214-297) for the candidates map_stage left in the context's buffers, against one
// index chunk; the (count, identity) results go to the chunk's column block of the dense [nQuery][nRefGenomes] table in ctx->rows.
// `compact`: the table holds this chunk's genomes only ([nQuery][chunk genomes]; a streamed set is reduced and read back chunk by chunk)
int reduce_stage(ani_ctx *ctx, ani_sketch *set, IndexChunk *sk, const FragSet &fs, int32_t nCand, int32_t nQuery, bool compact = false)
{
  if (nQuery == 0 || sk->nGenomes == 0) return ANI_OK;
  const size_t binsPerQuery = sk->totalBins;
  const size_t nBins = binsPerQuery * (size_t)nQuery;
  TRY(ctx->bins.ensure((nBins ? nBins : 1) * 4));
  const size_t nPairsAll = (size_t)nQuery * (size_t)(compact ? sk->nGenomes : set->nGenomes);
  TRY(ctx->rows.ensure((nPairsAll ? nPairsAll : 1) * 8));
  hipError_t e1;
  {
  StageTimer tm(ctx, &ctx->counters.msReduce);
  e1 = hipMemsetAsync(ctx->bins.p, 0, (nBins ? nBins : 1) * 4, ctx->stream);
  if (nCand) {
    OneWayArgs a;
    a.nCand = nCand; a.candFrag = ctx->ocFrag.as<int32_t>(); a.candSeq = ctx->ocSeq.as<int32_t>(); a.refStart = ctx->refStart.as<int32_t>();
    a.idBits = ctx->idBits.as<uint32_t>(); a.fragGenome = fs.fragGenome; a.genomeBase = fs.genomeBase; a.contigGenome = sk->contigGenome;
    a.contigBinBase = sk->contigBinBase; a.binWidth = set->params.fragLen - 20; a.bins = ctx->bins.as<uint32_t>(); a.binsPerQuery = binsPerQuery;
    hipLaunchKernelGGL(k_oneway_bins, dim3(grid_for((size_t)nCand)), dim3(256), 0, ctx->stream, a);
  }
  PairArgs pa;
  pa.nQuery = nQuery; pa.nRefGenomes = sk->nGenomes; pa.bins = ctx->bins.as<uint32_t>(); pa.binsPerQuery = binsPerQuery;
  pa.genomeBinStart = sk->genomeBinStart; pa.pairCount = ctx->rows.as<uint32_t>(); pa.pairIdentity = ctx->rows.as<uint32_t>() + nPairsAll;
  pa.outStride = compact ? sk->nGenomes : set->nGenomes; pa.outCol0 = compact ? 0 : sk->g0;
  const size_t nPairs = (size_t)nQuery * (size_t)sk->nGenomes;
  hipLaunchKernelGGL(k_pair_reduce, dim3((unsigned)((nPairs + 3) / 4)), dim3(256), 0, ctx->stream, pa);   // one wave per pair
  }
  HIP_TRY(e1); HIP_TRY(hipGetLastError());
  return ANI_OK;
}

// the dense table of a sub-batch (all chunks reduced) -> rows in (query, reference genome) order, appended to `rows`
// (`block`: the table is the compact one of that chunk — reference genome = block->g0 + column)
int collect_rows(ani_ctx *ctx, ani_sketch *set, const FragSet &fs, int32_t nQuery, int32_t firstQueryId, RowBuf *rows, const IndexChunk *block = nullptr,
    const int32_t *queryIds = nullptr)
{
  const int32_t nCols = block ? block->nGenomes : set->nGenomes, col0 = (block ? block->g0 : 0) + set->refIdBase;
  const size_t nPairs = (size_t)nQuery * (size_t)nCols;
  if (nPairs == 0) return ANI_OK;
  uint32_t *dense = nullptr;
  TRY(pinned_buffer(ctx, 1, nPairs * 8 + 8, (void **)&dense));
  HIP_TRY(hipMemcpyAsync(dense, ctx->rows.p, nPairs * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // rows per query, then every query's rows at their place: both loops on the host pool (round 5: the serial walk over the table was
  // 1.3 ms per sub-batch of the 1000 x 1000 job — the device idle meanwhile — and seconds of the 9 x 10^8 pairs of 90 000 x 10 000)
  std::vector<size_t> qOff((size_t)nQuery + 1, 0);
  parallel_for((size_t)nQuery, (uint64_t)nPairs * 8, [&](size_t qi) {
    const uint32_t *cnt = dense + qi * (size_t)nCols;
    size_t c = 0;
    for (int32_t g = 0; g < nCols; g++) c += cnt[g] != 0;
    qOff[qi + 1] = c;
  });
  for (int32_t qi = 0; qi < nQuery; qi++) qOff[qi + 1] += qOff[qi];
  const size_t m = qOff[nQuery];
  ani_cgi_t *out0 = rows->grow(m);
  if (!out0) return fail(ANI_ERR_NOMEM, "host allocation of %zu result rows failed", m);
  rows->n += m;
  parallel_for((size_t)nQuery, (uint64_t)nPairs * 8, [&](size_t qi) {                       // query ascending, reference ascending
    const uint32_t *cnt = dense + qi * (size_t)nCols, *idb = cnt + nPairs;
    ani_cgi_t *out = out0 + qOff[qi];
    for (int32_t g = 0; g < nCols; g++) {
      if (!cnt[g]) continue;
      ani_cgi_t r; r.refGenomeId = col0 + g; r.qryGenomeId = firstQueryId + (queryIds ? queryIds[qi] : (int32_t)qi); r.countSeq = (int32_t)cnt[g];
      r.totalQueryFragments = fs.genomeFragments[qi];
      memcpy(&r.identity, &idb[g], 4);
      *out++ = r;
    }
  });
  ctx->counters.cgiRows += m;
  return ANI_OK;
}

// Map + reduce for the fragments of `fs` (a whole set or a slice of one) against every index chunk (all resident); rows appended
int map_fragset(ani_ctx *ctx, ani_sketch *sk, const FragSet &fs, int32_t nQuery, int32_t firstQueryId, RowBuf *rows, const int32_t *queryIds = nullptr)
{
  TRY(upload_luts(sk, fs.maxS));
  for (IndexChunk *ch : sk->chunks) {
    int32_t nCand = 0;
    TRY(map_stage(ctx, sk, ch, fs, &nCand));
    TRY(reduce_stage(ctx, sk, ch, fs, nCand, nQuery));
  }
  return collect_rows(ctx, sk, fs, nQuery, firstQueryId, rows, nullptr, queryIds);
}

// Sub-batches of kept fragment sets, mapped against a whole reference set.  A resident set is walked sub-batch by sub-batch (every
// chunk per sub-batch, one dense result table per sub-batch).  A streamed set is walked CHUNK by chunk — build the chunk's index,
// map every sub-batch of every set against it, drop it — so that each chunk is built once per call however many query genomes
// there are (the reference's own loop has the same shape: per reference split, all queries; core_genome_identity.cpp:55-106);
// the rows of a sub-batch then come chunk by chunk and are put back into (query, reference) order at the end.
// queryIds: per genome, relative to firstQueryId (merged sets), else consecutive
struct SubBatch { const ani_fragset *set; int32_t g0, g1, firstQueryId; FragSet v; const int32_t *queryIds; };
int map_fragsets(ani_ctx *ctx, ani_sketch *sk, const std::vector<const ani_fragset *> &sets, const std::vector<int32_t> &firstQueryIds, RowBuf *rows)
{
  std::vector<SubBatch> sub;
  int maxS = 0;
  const uint64_t subFrags = ctx->subBatchFragments;
  for (size_t si = 0; si < sets.size(); si++) {
    const ani_fragset *f = sets[si];
    if (f->device != ctx->device) return fail(ANI_ERR_ARG, "fragment set lives on device %d, context on %d", f->device, ctx->device);
    if (f->params.kmerSize != sk->params.kmerSize || f->params.windowSize != sk->params.windowSize || f->params.fragLen != sk->params.fragLen)
      return fail(ANI_ERR_ARG, "fragment set and sketch were built with different parameters");
    maxS = std::max(maxS, f->fs.maxS);
    const int32_t nG = (int32_t)f->fs.genomeFragments.size();
    int32_t g0 = 0;
    while (g0 < nG) {
      // sub-batches bounded by fragments (2^20), by the bin table of the largest index chunk (8 GiB) and by the dense result table
      int32_t g1 = g0;
      const uint64_t maxQ = std::max<uint64_t>(1, std::min<uint64_t>((ctx->subBatchBinBytes) / (4ull * std::max<uint32_t>(sk->maxChunkBins, 1)),
                                                                 ((uint64_t)2 << 30) / (8ull * (uint64_t)std::max<int32_t>(sk->nGenomes, 1))));
      while (g1 < nG && (g1 == g0 || ((uint64_t)(f->genomeFragStart[g1] - f->genomeFragStart[g0]) < subFrags && (uint64_t)(g1 - g0) < maxQ))) g1++;
      const int64_t fA = f->genomeFragStart[g0], fB = f->genomeFragStart[g1];
      SubBatch sb; sb.set = f; sb.g0 = g0; sb.g1 = g1;
      sb.queryIds = f->genomeQueryId.empty() ? nullptr : f->genomeQueryId.data() + g0;
      sb.firstQueryId = firstQueryIds[si] + (sb.queryIds ? 0 : g0);
      FragSet &v = sb.v;                                       // slice [g0, g1) of the kept set
      v.nFrag = (int32_t)(fB - fA); v.maxS = f->fs.maxS; v.poolSize = f->fs.poolSize;
      v.genomeFragments.assign(f->fs.genomeFragments.begin() + g0, f->fs.genomeFragments.begin() + g1);
      v.qPool = f->fs.qPool; v.genomeBase = g0;
      if (v.nFrag) { v.fragOff = f->fs.fragOff + fA; v.fragS = f->fs.fragS + fA; v.fragGenome = f->fs.fragGenome + fA; v.fragQSeq = f->fs.fragQSeq + fA; }
      v.nHashes = f->fs.nFrag ? (uint64_t)((double)f->fs.nHashes * (double)v.nFrag / (double)f->fs.nFrag) : 0;      // statistics only (l1Probes)
      sub.push_back(std::move(sb));
      g0 = g1;
    }
  }
  if (!sk->streaming) {
    for (const SubBatch &sb : sub) TRY(map_fragset(ctx, sk, sb.v, sb.g1 - sb.g0, sb.firstQueryId, rows, sb.queryIds));
    return ANI_OK;
  }
  TRY(upload_luts(sk, maxS));
  std::vector<RowBuf> part(sub.size());
  for (size_t c = 0; c < sk->chunks.size(); c++) {
    TRY(ensure_chunk(sk, c));
    IndexChunk *ch = sk->chunks[c];
    for (size_t i = 0; i < sub.size(); i++) {
      const SubBatch &sb = sub[i];
      int32_t nCand = 0;
      TRY(map_stage(ctx, sk, ch, sb.v, &nCand));
      TRY(reduce_stage(ctx, sk, ch, sb.v, nCand, sb.g1 - sb.g0, true));
      TRY(collect_rows(ctx, sk, sb.v, sb.g1 - sb.g0, sb.firstQueryId, &part[i], ch, sb.queryIds));
    }
  }
  // a sub-batch's rows are (chunk, query, reference)-ordered and a chunk's references all precede the next chunk's: a stable
  // distribution by query — one counting pass, one scatter; the 7.8e7 rows of a 10 000 x 10 000 run are not comparison-sorted —
  // restores (query, reference) order.  The sub-batches are independent of each other (each owns a run of the output), so they go
  // through the host pool side by side (round 6: done one after the other on one thread this was 0.5 of the 5.2 s of a 10 000 x 10 000 step
  // and most of the minute of host time in the mapping calls of 90 000 x 90 000).
  {
    std::vector<size_t> partOff(sub.size() + 1, 0);
    for (size_t i = 0; i < sub.size(); i++) partOff[i + 1] = partOff[i] + part[i].n;
    const size_t total = partOff[sub.size()];
    if (total) {
      ani_cgi_t *out0 = rows->grow(total);
      if (!out0) return fail(ANI_ERR_NOMEM, "host allocation of %zu result rows failed", total);
      parallel_for(sub.size(), (uint64_t)total * sizeof(ani_cgi_t), [&](size_t i) {
        RowBuf &pb = part[i];
        if (!pb.n) return;
        ani_cgi_t *out = out0 + partOff[i];
        const int32_t q0 = sub[i].firstQueryId, nq = sub[i].g1 - sub[i].g0;
        const int32_t *ids = sub[i].queryIds;          // a merged set: the genomes' query ids (ascending), relative to q0; else consecutive
        auto slot_of = [&](int32_t id) -> size_t { return ids ? (size_t)(std::lower_bound(ids, ids + nq, id - q0) - ids) : (size_t)(id - q0); };
        std::vector<size_t> start((size_t)nq + 1, 0);
        for (size_t r = 0; r < pb.n; r++) start[slot_of(pb.p[r].qryGenomeId) + 1]++;
        for (int32_t q = 0; q < nq; q++) start[(size_t)q + 1] += start[(size_t)q];
        for (size_t r = 0; r < pb.n; r++) out[start[slot_of(pb.p[r].qryGenomeId)]++] = pb.p[r];
        free(pb.p); pb.p = nullptr; pb.n = pb.cap = 0;                // host memory of a big run: each part goes back as soon as it is merged
      });
      rows->n += total;
    }
  }
  return ANI_OK;
}
}  // namespace anih

extern "C" {

int ani_map_cgi_fragset(ani_ctx *ctx, const ani_sketch *skc, const ani_fragset *f, int32_t firstQueryId, ani_cgi_t **out, size_t *m)
{
  return ani_map_cgi_fragsets(ctx, skc, 1, &f, &firstQueryId, out, m);
}

int ani_map_cgi_fragsets(ani_ctx *ctx, const ani_sketch *skc, int32_t nSets, const ani_fragset *const *frags, const int32_t *firstQueryIds, ani_cgi_t **out,
    size_t *m)
{
  if (!ctx || !skc || nSets < 0 || (nSets && (!frags || !firstQueryIds)) || !out || !m) return fail(ANI_ERR_ARG, "null argument");
  for (int32_t i = 0; i < nSets; i++) if (!frags[i]) return fail(ANI_ERR_ARG, "null fragment set %d", i);
  ani_sketch *sk = const_cast<ani_sketch *>(skc);
  HIP_TRY(hipSetDevice(ctx->device));
  RowBuf rows;
  std::vector<const ani_fragset *> sets(frags, frags + nSets);
  std::vector<int32_t> firsts(firstQueryIds, firstQueryIds + nSets);
  TRY(map_fragsets(ctx, sk, sets, firsts, &rows));
  *m = rows.n; *out = rows.release();
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  return ANI_OK;
}

int ani_map_query(ani_ctx *ctx, const ani_sketch *skc, const ani_seq_batch_t *query, ani_mapping_t **out, size_t *n, uint64_t *totalQueryFragments)
{
  if (!ctx || !skc || !out || !n) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_batch(query));
  if (query->nGenomes != 1) return fail(ANI_ERR_ARG, "ani_map_query maps exactly one query genome (Map::Map takes one queryno, computeMap.hpp:93)");
  ani_sketch *sk = const_cast<ani_sketch *>(skc);
  HIP_TRY(hipSetDevice(ctx->device));
  DeviceBatch db; FragSet fs;
  TRY(upload_batch(ctx, query, 0, 1, &db));
  TRY(fragment_stage(ctx, sk->params, db, &fs));
  TRY(upload_luts(sk, fs.maxS));
  if (totalQueryFragments) *totalQueryFragments = (uint64_t)fs.genomeFragments[0];
  std::vector<ani_mapping_t> maps;
  for (size_t ci = 0; ci < sk->chunks.size(); ci++) {
    IndexChunk *ch = sk->chunks[ci];
    TRY(ensure_chunk(sk, ci));
    int32_t nCand = 0;
    TRY(map_stage(ctx, sk, ch, fs, &nCand));
    if (!nCand) continue;
    const size_t nC = (size_t)nCand;
    TRY(ctx->keepFlags.ensure(nC * 4)); TRY(ctx->keepOff.ensure((nC + 1) * 4));
    hipLaunchKernelGGL(ani::k_keep_flags, dim3(grid_for(nC)), dim3(256), 0, ctx->stream, (int32_t)nC, ctx->idBits.as<uint32_t>(), ctx->keepFlags.as<int32_t>());
    uint64_t nKeep = 0;
    TRY(device_scan(ctx, ctx->keepFlags.as<int32_t>(), ctx->keepOff.as<uint32_t>(), (uint32_t)nC, &nKeep));
    if (!nKeep) continue;
    TRY(ctx->mapOut.ensure(nKeep * 44));
    hipLaunchKernelGGL(ani::k_emit_mappings, dim3(grid_for(nC)), dim3(256), 0, ctx->stream, (int32_t)nC, ctx->ocFrag.as<int32_t>(), ctx->ocSeq.as<int32_t>(),
                       ctx->refStart.as<int32_t>(), ctx->idBits.as<uint32_t>(), ctx->l2Best.as<int32_t>(), fs.fragS,
                       fs.fragQSeq, ctx->keepOff.as<uint32_t>(), sk->params.fragLen, ch->c0, ctx->mapOut.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    const size_t o = maps.size();
    maps.resize(o + nKeep);
    HIP_TRY(hipMemcpyAsync(maps.data() + o, ctx->mapOut.p, nKeep * 44, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  // callback order (fragment, then reference position): every chunk's block is in that order and chunk c's contigs all precede
  // chunk c+1's, so a stable sort by fragment merges the blocks
  if (sk->chunks.size() > 1)
    std::stable_sort(maps.begin(), maps.end(), [](const ani_mapping_t &a, const ani_mapping_t &b) { return a.querySeqId < b.querySeqId; });
  // nucIdentityUpperBound (computeMap.hpp:378-381) is a host scalar per (sketchSize, shared); memoised
  std::unordered_map<uint64_t, float> memo;
  for (auto &m : maps) {
    const uint64_t key = ((uint64_t)(uint32_t)m.sketchSize << 32) | (uint32_t)m.conservedSketches;
    auto it = memo.find(key);
    if (it == memo.end()) {
      float id, ub; ani::stat::identity(m.conservedSketches, m.sketchSize, sk->params.kmerSize, &id, &ub);
      it = memo.emplace(key, ub).first;
    }
    m.nucIdentityUpperBound = it->second;
  }
  ctx->counters.mappings += maps.size();
  return to_host_malloc(maps, out, n);
}

int ani_compute_cgi(ani_ctx *ctx, const ani_sketch *skc, const ani_mapping_t *mappings, size_t n, uint64_t totalQueryFragments, int32_t queryFileNo,
                    ani_cgi_t **out, size_t *m)
{
  if (!ctx || !skc || !out || !m || (n && !mappings)) return fail(ANI_ERR_ARG, "null argument");
  ani_sketch *sk = const_cast<ani_sketch *>(skc);
  HIP_TRY(hipSetDevice(ctx->device));
  if (n > 0x7ffffff0ull) return fail(ANI_ERR_LIMIT, "too many mappings");
  // The device reducer expects the mappings of one (fragment, reference genome) next to each other, which is how Map reports
  // them (fragment, then reference position); only input that is not in that order is sorted (on the device, by
  // (querySeqId, refSeqId) with the record index as payload).
  bool ordered = true;
  for (size_t i = 0; i < n; i++) {
    const ani_mapping_t &a = mappings[i];
    if (a.refSeqId < 0 || a.refSeqId >= sk->nContigs) return fail(ANI_ERR_ARG, "mapping %zu refers to contig %d outside the sketch", i, a.refSeqId);
    if (a.refStartPos < 0 || a.refStartPos > sk->contigLen[a.refSeqId]) return fail(ANI_ERR_ARG, "mapping %zu has refStartPos outside its contig", i);
    if (a.nucIdentity <= 0.0f) return fail(ANI_ERR_ARG, "mapping %zu has non-positive identity", i);
    if (a.querySeqId < 0) return fail(ANI_ERR_ARG, "mapping %zu has a negative querySeqId", i);
    if (i && (mappings[i - 1].querySeqId > a.querySeqId || (mappings[i - 1].querySeqId == a.querySeqId
        && mappings[i - 1].refSeqId > a.refSeqId))) ordered = false;
  }
  std::vector<uint32_t> ord;
  if (!ordered) {
    ord.resize(n);
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; i++) { keys[i] = ((uint64_t)(uint32_t)mappings[i].querySeqId << 32) | (uint32_t)mappings[i].refSeqId; ord[i] = (uint32_t)i; }
    TRY(ctx->l1BigHitsA.ensure(n * 8)); TRY(ctx->l1BigHitsB.ensure(n * 8)); TRY(ctx->keepFlags.ensure(n * 4)); TRY(ctx->keepOff.ensure(n * 4));
    HIP_TRY(hipMemcpyAsync(ctx->l1BigHitsA.p, keys.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->keepFlags.p, ord.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    size_t tb = 0;
    int rc = ani_sort_pairs_u64_u32(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), ctx->keepFlags.as<uint32_t>(),
        ctx->keepOff.as<uint32_t>(), n, 64, nullptr, &tb, ctx->stream);
    if (rc == 0) { TRY(ctx->sortTmp.ensure(tb + 256));
      rc = ani_sort_pairs_u64_u32(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), ctx->keepFlags.as<uint32_t>(), ctx->keepOff.as<uint32_t>(),
        n, 64, ctx->sortTmp.p, &tb, ctx->stream); }
    if (rc != 0) return fail(ANI_ERR_DEVICE, "radix sort of mappings failed (%d)", rc);
    HIP_TRY(hipMemcpyAsync(ord.data(), ctx->keepOff.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  // compress querySeqId -> dense fragment index (all of one genome); split the candidate list by index chunk
  FragSet fs; fs.genomeFragments.assign(1, (int32_t)totalQueryFragments);
  const size_t nCh = sk->chunks.size();
  std::vector<std::vector<int32_t>> cFrag(nCh), cSeq(nCh), cStart(nCh); std::vector<std::vector<uint32_t>> cBits(nCh);
  int32_t lastQ = -1, f = -1;
  size_t chunkOfSeqHint = 0;
  for (size_t i = 0; i < n; i++) {
    const ani_mapping_t &a = mappings[ordered ? i : ord[i]];
    if (a.querySeqId != lastQ || f < 0) { f++; lastQ = a.querySeqId; }
    size_t c = chunkOfSeqHint;
    while (c + 1 < nCh && a.refSeqId >= sk->chunks[c]->c0 + sk->chunks[c]->nContigs) c++;
    while (c > 0 && a.refSeqId < sk->chunks[c]->c0) c--;
    chunkOfSeqHint = c;
    uint32_t bits; memcpy(&bits, &a.nucIdentity, 4);
    cFrag[c].push_back(f); cSeq[c].push_back(a.refSeqId - sk->chunks[c]->c0); cStart[c].push_back(a.refStartPos); cBits[c].push_back(bits);
  }
  fs.nFrag = f + 1;
  const size_t nF = (size_t)(f + 1);
  TRY(ctx->fragGenome.ensure((nF ? nF : 1) * 4));
  if (nF) HIP_TRY(hipMemsetAsync(ctx->fragGenome.p, 0, nF * 4, ctx->stream));
  fs.fragGenome = ctx->fragGenome.as<int32_t>(); fs.genomeBase = 0;
  for (size_t c = 0; c < nCh; c++) {
    const size_t nc = cFrag[c].size();
    if (nc) {
      TRY(ctx->ocFrag.ensure(nc * 4)); TRY(ctx->ocSeq.ensure(nc * 4)); TRY(ctx->refStart.ensure(nc * 4)); TRY(ctx->idBits.ensure(nc * 4));
      HIP_TRY(hipMemcpyAsync(ctx->ocFrag.p, cFrag[c].data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->ocSeq.p, cSeq[c].data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->refStart.p, cStart[c].data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->idBits.p, cBits[c].data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    TRY(reduce_stage(ctx, sk, sk->chunks[c], fs, (int32_t)nc, 1));
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // the next chunk reuses the candidate buffers
  }
  RowBuf rows;
  TRY(collect_rows(ctx, sk, fs, 1, queryFileNo, &rows));
  *m = rows.n; *out = rows.release();
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  return ANI_OK;
}

int ani_map_cgi_batch(ani_ctx *ctx, const ani_sketch *skc, const ani_seq_batch_t *queries, int32_t firstQueryId, ani_cgi_t **out, size_t *m)
{
  if (!ctx || !skc || !out || !m) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_batch(queries));
  ani_sketch *sk = const_cast<ani_sketch *>(skc);
  HIP_TRY(hipSetDevice(ctx->device));
  if (sk->streaming) {
    // a streamed reference set is walked chunk by chunk: the fragment sketches of ALL the queries are made first (1.6 MB per 5 Mbp
    // genome) so that every chunk is built once
    ani_fragset *f = nullptr;
    TRY(ani_fragset_build(ctx, &sk->params, queries, &f));
    const int rc = ani_map_cgi_fragset(ctx, sk, f, firstQueryId, out, m);
    ani_fragset_free(f);
    return rc;
  }
  RowBuf rows;
  // sub-batches bounded by fragments (2^20), by the bin table of the largest index chunk (8 GiB) and by the dense result table
  const int L = sk->params.fragLen;
  int32_t g0 = 0;
  while (g0 < queries->nGenomes) {
    int32_t g1 = g0; uint64_t fr = 0;
    const uint64_t maxQ = std::max<uint64_t>(1, std::min<uint64_t>((ctx->subBatchBinBytes) / (4ull * std::max<uint32_t>(sk->maxChunkBins, 1)),
                                                               ((uint64_t)2 << 30) / (8ull * (uint64_t)std::max<int32_t>(sk->nGenomes, 1))));
    while (g1 < queries->nGenomes && (g1 == g0 || (fr < ctx->subBatchFragments && (uint64_t)(g1 - g0) < maxQ))) {
      for (int32_t c = queries->genomeContigStart[g1]; c < queries->genomeContigStart[g1 + 1]; c++) fr += (uint64_t)(queries->contigLen[c] / L);
      g1++;
    }
    DeviceBatch db; FragSet fs;
    TRY(upload_batch(ctx, queries, g0, g1, &db));
    TRY(fragment_stage(ctx, sk->params, db, &fs));          // once per sub-batch, whatever the number of index chunks
    TRY(upload_luts(sk, fs.maxS));
    for (IndexChunk *ch : sk->chunks) {
      int32_t nCand = 0;
      TRY(map_stage(ctx, sk, ch, fs, &nCand));
      TRY(reduce_stage(ctx, sk, ch, fs, nCand, g1 - g0));
    }
    TRY(collect_rows(ctx, sk, fs, g1 - g0, firstQueryId + g0, &rows));
    g0 = g1;
  }
  *m = rows.n; *out = rows.release();
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  return ANI_OK;
}


}  // extern "C"
