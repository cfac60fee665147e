// engine_sketch.hip — reference minimizer records (≙ Sketch::build, src/map/include/winSketch.hpp:124-176) and query fragment
// sketches (≙ the sketching half of Map::doL1Mapping, src/map/include/computeMap.hpp:252-274); kept fragment sets and their wire format.
#include "host/engine.hpp"
#include "kernels/reduce.hpp"
#include "kernels/sketch.hpp"

namespace anih {
using namespace ani;

int check_params(const ani_params_t *p)
{
  if (!p) return fail(ANI_ERR_ARG, "null parameters");
  if (p->kmerSize < 1 || p->kmerSize > 16) return fail(ANI_ERR_ARG, "kmerSize must be in 1..16 (hash_t is 32 bit, parseCmdArgs.hpp:142)");
  if (p->windowSize < 1 || p->windowSize > ani::kTile - ani::kTPB) return fail(ANI_ERR_LIMIT, "windowSize %d outside 1..%d", p->windowSize,
      ani::kTile - ani::kTPB);
  if (p->fragLen < 1 || p->fragLen <= 20) return fail(ANI_ERR_ARG, "fragLen must exceed 20 (bin width is fragLen-20, computeCoreIdentity.hpp:194)");
  return ANI_OK;
}

// -----------------------------------------------------------------------------------------------------
// reference minimizer records of a device-resident batch (position order, global seqIds)
// -----------------------------------------------------------------------------------------------------
// fragment table of a device-resident batch (computeMap.hpp:132-190): per-contig fragment prefix on the host (returned in
// fragStart, nContigs + 1 entries), fragment descriptors + fragment -> (genome, running id) expanded on the device
int frag_tables(ani_ctx *ctx, const ani_params_t &p, const DeviceBatch &db, FragSet *qr, std::vector<uint32_t> *fragStartOut, const FragArrays &fa, bool alloc,
                FragArrays *owned)
{
  const int k = p.kmerSize, w = p.windowSize, L = p.fragLen;
  std::vector<uint32_t> &fragStart = *fragStartOut;
  fragStart.assign((size_t)db.nContigs + 1, 0);
  std::vector<int32_t> cGenome((size_t)db.nContigs + 1, 0), cQBase((size_t)db.nContigs + 1, 0);
  qr->genomeFragments.assign(db.nGenomes, 0);
  uint64_t nF64 = 0;
  for (int32_t g = 0; g < db.nGenomes; g++) {
    int32_t seqCounter = 0;
    for (int32_t c = db.genomeContigStart[g]; c < db.genomeContigStart[g + 1]; c++) {
      fragStart[c] = (uint32_t)nF64; cGenome[c] = g; cQBase[c] = seqCounter;
      const int32_t len = db.contigLen[c];
      if (len < w || len < k || len < L) continue;            // :138
      const int32_t fc = len / L;                              // :152
      seqCounter += fc; nF64 += (uint64_t)fc;
      if (nF64 > 0x3fffffffull) return fail(ANI_ERR_LIMIT, "too many fragments in one query batch");
    }
    qr->genomeFragments[g] = seqCounter;
  }
  fragStart[db.nContigs] = (uint32_t)nF64;
  const size_t nF = (size_t)nF64;
  qr->nFrag = (int32_t)nF; qr->maxS = 0; qr->nHashes = 0;
  ctx->counters.queryGenomes += (uint64_t)db.nGenomes; ctx->counters.queryFragments += nF; ctx->counters.queryBases += db.totalBases;
  if (nF == 0) return ANI_OK;
  FragArrays arr = fa;
  if (alloc) {
    HIP_TRY(pool_malloc((void **)&owned->fragOff, nF * 4)); HIP_TRY(pool_malloc((void **)&owned->fragS, nF * 4));
    HIP_TRY(pool_malloc((void **)&owned->fragGenome, nF * 4)); HIP_TRY(pool_malloc((void **)&owned->fragQSeq, nF * 4));
    arr = *owned;
  }
  TRY(ctx->frags.ensure(nF * sizeof(FragDesc)));
  const size_t nc1 = (size_t)db.nContigs + 1;
  TRY(ctx->unitStart.ensure(nc1 * 4)); TRY(ctx->unitAux.ensure(nc1 * 8));
  HIP_TRY(hipMemcpyAsync(ctx->unitStart.p, fragStart.data(), nc1 * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->unitAux.p, cGenome.data(), nc1 * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->unitAux.as<int32_t>() + nc1, cQBase.data(), nc1 * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_expand_frags, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(),
      (const int32_t *)ctx->unitAux.as<int32_t>(),
                     (const int32_t *)(ctx->unitAux.as<int32_t>() + nc1), db.nContigs, (uint32_t)nF, L, ctx->frags.as<FragDesc>(), arr.fragGenome,
                         arr.fragQSeq);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  qr->fragOff = arr.fragOff; qr->fragS = arr.fragS; qr->fragGenome = arr.fragGenome; qr->fragQSeq = arr.fragQSeq; qr->genomeBase = 0;
  return ANI_OK;
}

inline bool fusable(const ani_params_t *p) { return p->fragLen + p->windowSize - 1 <= kTile && p->fragLen >= p->windowSize && p->fragLen >= p->kmerSize; }

// The fragment sketches of a batch as they come out of the sketch kernels lie in 64 partly filled stripes of the pool (pool_take).
// A set that is kept is packed: sketches back to back in fragment order, exactly nHashes entries, fragOff rewritten in place.
int compact_fragment_pool(ani_ctx *ctx, const uint32_t *pool, uint32_t *fragOff /* in: striped offsets, out: packed offsets */, const int32_t *fragS,
    size_t nF, uint64_t nHashes, uint32_t **out)
{
  *out = nullptr;
  HIP_TRY(pool_malloc((void **)out, (nHashes ? nHashes : 1) * 4));
  if (nF == 0) return ANI_OK;
  TRY(ctx->scanTmpC.ensure(nF * 4)); TRY(ctx->scanTmpD.ensure(nF * 4));
  // s = -1 marks an overflowed fragment
  hipLaunchKernelGGL(k_clamp_counts, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, (int32_t)nF, fragS, (const int32_t *)nullptr, ctx->scanTmpC.as<int32_t>());
  uint64_t total = 0;
  int rc = device_scan(ctx, ctx->scanTmpC.as<int32_t>(), ctx->scanTmpD.as<uint32_t>(), (uint32_t)nF, &total);
  if (rc == ANI_OK && total != nHashes) rc = fail(ANI_ERR_INTERNAL, "fragment sketch pool: %llu hashes counted, %llu in the sketches",
      (unsigned long long)nHashes, (unsigned long long)total);
  if (rc != ANI_OK) { pool_free(*out); *out = nullptr; return rc; }
  hipLaunchKernelGGL(k_pack_fragment_pool, dim3(grid_for(nF * 64)), dim3(256), 0, ctx->stream, pool, fragOff, (const int32_t *)ctx->scanTmpC.as<int32_t>(),
      (const uint32_t *)ctx->scanTmpD.as<uint32_t>(), (uint32_t)nF, *out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { pool_free(*out); *out = nullptr; HIP_TRY(e); }
  return ANI_OK;
}

int sketch_records(ani_ctx *ctx, const ani_params_t *p, const DeviceBatch &db, int32_t seqIdBase, uint32_t **dRecords, size_t *nOut, FusedOut *fused)
{
  const int k = p->kmerSize, w = p->windowSize, L = p->fragLen;
  *dRecords = nullptr; *nOut = 0;
  std::vector<uint32_t> fragStart;
  size_t nF = 0;
  if (fused) {
    TRY(frag_tables(ctx, *p, db, fused->fs, &fragStart, FragArrays(), true, fused->arr));
    nF = (size_t)fused->fs->nFrag;
  }
  // per-contig tile prefix; the 1.6 M tile descriptors of a 1000-genome batch are expanded on the device
  std::vector<uint32_t> tileStart((size_t)db.nContigs + 1, 0);
  const int stride = kTile - (w - 1);
  uint64_t positions = 0, nT64 = 0;
  for (int32_t c = 0; c < db.nContigs; c++) {
    tileStart[c] = (uint32_t)nT64;
    const int32_t len = db.contigLen[c];
    if (len < w || len < k) continue;                       // winSketch.hpp:153
    const int32_t nPos = len - k + 1;
    positions += (uint64_t)nPos;
    uint64_t cnt;
    if (!fused) {
      cnt = 1;                                              // tiles B = 0, stride, ... while B + (w-1) < nPos
      if (nPos > w - 1) cnt = ((uint64_t)(nPos - (w - 1)) + stride - 1) / stride;
    } else {
      // one tile per whole fragment (it emits the windows that end inside the fragment), then plain tiles for the rest of the contig
      const int32_t nFragC = (int32_t)(fragStart[c + 1] - fragStart[c]);
      const int32_t base = nFragC > 0 ? nFragC * L - (w - 1) : 0;
      uint64_t tail = nFragC > 0 ? 0 : 1;
      if (nPos > base + (w - 1)) tail = ((uint64_t)(nPos - (base + w - 1)) + stride - 1) / stride;
      cnt = (uint64_t)nFragC + tail;
    }
    nT64 += cnt;
    if (nT64 > 0x7fffffffull) return fail(ANI_ERR_LIMIT, "too many tiles in one reference batch");
  }
  tileStart[db.nContigs] = (uint32_t)nT64;
  const size_t nT = (size_t)nT64;
  if (nT == 0) return ANI_OK;
  TRY(ctx->tiles.ensure(nT * sizeof(TileDesc))); TRY(ctx->tileMeta.ensure(nT * sizeof(TileMeta)));
  TRY(ctx->tileCnt.ensure(nT * 4)); TRY(ctx->tileDrop.ensure(nT)); TRY(ctx->tileOff.ensure((nT + 1) * 4));
  TRY(ctx->unitStart.ensure(tileStart.size() * 4));
  HIP_TRY(hipMemcpyAsync(ctx->unitStart.p, tileStart.data(), tileStart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!fused)
    hipLaunchKernelGGL(k_expand_tiles, dim3(grid_for(nT)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(), db.nContigs,
        (uint32_t)nT, stride,
                       ctx->tiles.as<TileDesc>());
  else {
    TRY(ctx->tileInfo.ensure(nT * sizeof(FusedInfo))); TRY(ctx->unitAux.ensure(fragStart.size() * 4));
    HIP_TRY(hipMemcpyAsync(ctx->unitAux.p, fragStart.data(), fragStart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_expand_fused, dim3(grid_for(nT)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(),
        (const uint32_t *)ctx->unitAux.as<uint32_t>(), db.nContigs,
                       (uint32_t)nT, L, w, stride, ctx->tiles.as<TileDesc>(), ctx->tileInfo.as<FusedInfo>());
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));

  uint64_t cap = (uint64_t)((double)positions * 2.6 / (w + 1)) + 64 * nT + 4096;
  if (cap > positions + nT) cap = positions + nT;
  uint64_t qcap = fused ? (uint64_t)nF * (uint64_t)((2.6 * L) / (w + 1) + 32) + 1024 : 0;
  unsigned long long host[CNT_N];
  for (int attempt = 0;; attempt++) {
    cap = (uint64_t)stripe_cap(cap) * kPoolStripes; qcap = (uint64_t)stripe_cap(qcap) * kPoolStripes;      // whole stripes (common.hpp: pool_take)
    if (cap > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "reference batch yields more than 2^32 minimizers; split the reference list");
    if (qcap > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "query batch sketch exceeds 2^32 hashes");
    TRY(ctx->poolHash.ensure(cap * 4)); TRY(ctx->poolWpos.ensure(cap * 4));
    if (fused && (!*fused->qPool || attempt > 0)) {
      if (*fused->qPool) { pool_free(*fused->qPool); *fused->qPool = nullptr; }
      HIP_TRY(pool_malloc((void **)fused->qPool, (qcap ? qcap : 1) * 4));
    }
    TRY(zero_counters(ctx));
    {
      StageTimer tm(ctx, &ctx->counters.msSketch);
      if (!fused)
        hipLaunchKernelGGL(k_sketch_tiles, dim3((unsigned)nT), dim3(kTPB), 0, ctx->stream, db.dPacked, db.dAscii, db.dContigOff,
                           db.dContigLen, db.dContigMode, ctx->tiles.as<TileDesc>(), k, w, ctx->poolHash.as<uint32_t>(),
                           ctx->poolWpos.as<int32_t>(), stripe_cap(cap), cur_ptr(ctx, POOL_REF), ctx->tileMeta.as<TileMeta>());
      else
        hipLaunchKernelGGL(k_sketch_fused, dim3((unsigned)nT), dim3(kTPB), 0, ctx->stream, db.dPacked, db.dAscii, db.dContigOff, db.dContigLen, db.dContigMode,
                           ctx->tiles.as<TileDesc>(), ctx->tileInfo.as<FusedInfo>(), k, w, L, ctx->poolHash.as<uint32_t>(), ctx->poolWpos.as<int32_t>(),
                               stripe_cap(cap),
                           cur_ptr(ctx, POOL_REF), ctx->tileMeta.as<TileMeta>(), *fused->qPool, stripe_cap(qcap), cur_ptr(ctx,
                               POOL_Q), fused->arr->fragOff, fused->arr->fragS,
                           (int *)cnt_ptr(ctx, CNT_MAXS));
    }
    HIP_TRY(hipGetLastError());
    TRY(read_counters(ctx, host));
    const bool refFits = ctx->poolMaxStripe[POOL_REF] <= stripe_cap(cap), qFits = !fused || ctx->poolMaxStripe[POOL_Q] <= stripe_cap(qcap);
    if (refFits && qFits) break;
    if (attempt > 2) return fail(ANI_ERR_INTERNAL, "minimizer pool did not converge");
    if (!refFits) cap = grown_cap(ctx->poolMaxStripe[POOL_REF]);          // margin: the same input must not grow the pool again next time
    if (!qFits) qcap = grown_cap(ctx->poolMaxStripe[POOL_Q]);
  }
  if (fused) {
    const int maxS = (int)(uint32_t)host[CNT_MAXS];
    if (maxS >= 0x7fffffff) return fail(ANI_ERR_LIMIT, "a query fragment produced more than %d minimizers", kFragHashCap);
    const uint64_t nHashes = ctx->poolUsed[POOL_Q];
    ctx->counters.querySketchHashes += nHashes;
    // the sketches lie in 64 partly filled stripes: a kept set is packed into fragment order (exact size; this is what travels between GPUs)
    uint32_t *packed = nullptr;
    TRY(compact_fragment_pool(ctx, *fused->qPool, fused->arr->fragOff, fused->arr->fragS, nF, nHashes, &packed));
    pool_free(*fused->qPool); *fused->qPool = packed;
    fused->fs->maxS = maxS; fused->fs->nHashes = nHashes; fused->fs->poolSize = nHashes; fused->fs->qPool = *fused->qPool;
  }
  StageTimer tm(ctx, &ctx->counters.msSketch);
  hipLaunchKernelGGL(k_sketch_tile_counts, dim3(grid_for(nT)), dim3(256), 0, ctx->stream, ctx->tiles.as<TileDesc>(),
                     ctx->tileMeta.as<TileMeta>(), (int)nT, ctx->tileCnt.as<int32_t>(), ctx->tileDrop.as<uint8_t>());
  uint64_t total = 0;
  TRY(device_scan(ctx, ctx->tileCnt.as<int32_t>(), ctx->tileOff.as<uint32_t>(), (uint32_t)nT, &total));
  if (total >= 0x7ffffff0ull) return fail(ANI_ERR_LIMIT, "reference batch yields >= 2^31 minimizers; split the reference list");
  if (total == 0) return ANI_OK;
  uint32_t *rec = nullptr;
  HIP_TRY(pool_malloc((void **)&rec, total * 12));
  hipLaunchKernelGGL(k_sketch_gather, dim3((unsigned)nT), dim3(kTPB), 0, ctx->stream, ctx->tiles.as<TileDesc>(), ctx->tileMeta.as<TileMeta>(),
                     ctx->tileDrop.as<uint8_t>(), ctx->tileOff.as<uint32_t>(), ctx->poolHash.as<uint32_t>(), ctx->poolWpos.as<int32_t>(),
                     seqIdBase, rec);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { pool_free(rec); HIP_TRY(e); }
  *dRecords = rec; *nOut = (size_t)total;
  return ANI_OK;
}

// -----------------------------------------------------------------------------------------------------
// query path for one device-resident sub-batch
// -----------------------------------------------------------------------------------------------------
int fragment_stage(ani_ctx *ctx, const ani_params_t &p, const DeviceBatch &db, FragSet *qr)
{
  if (ctx->timerPending.size() > 4096) flush_timers(ctx);      // no stage timer is open here
  const int k = p.kmerSize, w = p.windowSize, L = p.fragLen;
  // worst case sizes are known from the contig lengths: make room before the tables point into the buffers
  {
    uint64_t nF = 0;
    for (int32_t c = 0; c < db.nContigs; c++) { const int32_t len = db.contigLen[c]; if (!(len < w || len < k || len < L)) nF += (uint64_t)(len / L); }
    TRY(ctx->fragOff.ensure((nF + 1) * 4)); TRY(ctx->fragS.ensure((nF + 1) * 4)); TRY(ctx->fragGenome.ensure((nF + 1) * 4));
    TRY(ctx->fragQSeq.ensure((nF + 1) * 4));
  }
  FragArrays fa; fa.fragOff = ctx->fragOff.as<uint32_t>(); fa.fragS = ctx->fragS.as<int32_t>(); fa.fragGenome = ctx->fragGenome.as<int32_t>();
  fa.fragQSeq = ctx->fragQSeq.as<int32_t>();
  std::vector<uint32_t> fragStart;
  TRY(frag_tables(ctx, p, db, qr, &fragStart, fa, false, nullptr));
  const size_t nF = (size_t)qr->nFrag;
  if (nF == 0) return ANI_OK;

  unsigned long long host[CNT_N];
  // ---- fragment sketches ----
  uint64_t qcap = (uint64_t)nF * (uint64_t)((2.6 * L) / (w + 1) + 32) + 1024;
  for (int attempt = 0;; attempt++) {
    if (qcap > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "query batch sketch exceeds 2^32 hashes");
    qcap = (uint64_t)stripe_cap(qcap) * kPoolStripes;
    TRY(ctx->qPool.ensure(qcap * 4));
    TRY(zero_counters(ctx));
    {
      StageTimer tm(ctx, &ctx->counters.msFragSketch);
      if (L - k + 1 <= kTile)
        hipLaunchKernelGGL((k_fragment_sketch<true>), dim3((unsigned)nF), dim3(kTPB), 0, ctx->stream, db.dPacked, db.dAscii, db.dContigOff, db.dContigMode,
                         ctx->frags.as<FragDesc>(), L, k, w, ctx->qPool.as<uint32_t>(), stripe_cap(qcap), cur_ptr(ctx, POOL_Q),
                         ctx->fragOff.as<uint32_t>(), ctx->fragS.as<int32_t>(), (int *)cnt_ptr(ctx, CNT_MAXS));
      else
        hipLaunchKernelGGL((k_fragment_sketch<false>), dim3((unsigned)nF), dim3(kTPB), 0, ctx->stream, db.dPacked, db.dAscii, db.dContigOff, db.dContigMode,
                         ctx->frags.as<FragDesc>(), L, k, w, ctx->qPool.as<uint32_t>(), stripe_cap(qcap), cur_ptr(ctx, POOL_Q),
                         ctx->fragOff.as<uint32_t>(), ctx->fragS.as<int32_t>(), (int *)cnt_ptr(ctx, CNT_MAXS));
    }
    HIP_TRY(hipGetLastError());
    TRY(read_counters(ctx, host));
    if (ctx->poolMaxStripe[POOL_Q] <= stripe_cap(qcap)) break;
    if (attempt > 2) return fail(ANI_ERR_INTERNAL, "query sketch pool did not converge");
    qcap = grown_cap(ctx->poolMaxStripe[POOL_Q]);
  }
  const int maxS = (int)(uint32_t)host[CNT_MAXS];
  if (maxS >= 0x7fffffff) return fail(ANI_ERR_LIMIT, "a query fragment produced more than %d minimizers", kFragHashCap);
  ctx->counters.querySketchHashes += ctx->poolUsed[POOL_Q];
  qr->maxS = maxS; qr->nHashes = ctx->poolUsed[POOL_Q]; qr->poolSize = qcap;          // the pool has holes between its stripes: fragOff reaches up to qcap
  qr->qPool = ctx->qPool.as<uint32_t>();
  return ANI_OK;
}

void fragset_release(ani_fragset *f)
{
  void *tables[] = {f->arr.fragOff, f->arr.fragS, f->arr.fragGenome, f->arr.fragQSeq};
  if (!f->borrowed || f->ownTables) for (void *q : tables) if (q) pool_free(q);
  if (!f->borrowed && f->qPool) pool_free(f->qPool);
  delete f;
}
void fragset_finish(ani_fragset *f)
{
  f->genomeFragStart.assign(f->fs.genomeFragments.size() + 1, 0);
  for (size_t g = 0; g < f->fs.genomeFragments.size(); g++) f->genomeFragStart[g + 1] = f->genomeFragStart[g] + f->fs.genomeFragments[g];
}
// the fragment sketches fragment_stage left in the context's buffers -> arrays owned by the kept set (pool packed on the way)
int keep_fragment_arrays(ani_ctx *ctx, ani_fragset *f)
{
  const size_t nF = (size_t)f->fs.nFrag;
  if (!nF) return ANI_OK;
  hipError_t e = pool_malloc((void **)&f->arr.fragOff, nF * 4); if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragS, nF * 4);
  if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragGenome, nF * 4); if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragQSeq, nF * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(f->arr.fragOff, f->fs.fragOff, nF * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(f->arr.fragS, f->fs.fragS, nF * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(f->arr.fragGenome, f->fs.fragGenome, nF * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(f->arr.fragQSeq, f->fs.fragQSeq, nF * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "keeping the fragment sketches failed: %s", hipGetErrorString(e));
  TRY(compact_fragment_pool(ctx, f->fs.qPool, f->arr.fragOff, f->arr.fragS, nF, f->fs.nHashes, &f->qPool));
  f->fs.fragOff = f->arr.fragOff; f->fs.fragS = f->arr.fragS; f->fs.fragGenome = f->arr.fragGenome; f->fs.fragQSeq = f->arr.fragQSeq;
  f->fs.qPool = f->qPool; f->fs.poolSize = f->fs.nHashes;
  return ANI_OK;
}

// sketch one uploaded batch as references, optionally keeping its fragment sketches
int records_of_batch(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, int32_t seqIdBase, void **devRecords, size_t *n, ani_fragset *keep)
{
  // slices of ~2^30 bases keep the temporary pools small; a single slice is handed over as it is.  With `keep` the batch is one slice
  // (the caller — the command line, the bench — already works in such slices).
  std::vector<RecordPart> parts;
  auto cleanup = [&]() { for (auto &q : parts) if (q.rec) pool_free(q.rec); };
  size_t total = 0;
  int32_t g0 = 0;
  while (g0 < refs->nGenomes) {
    int32_t g1 = g0; uint64_t bases = 0;
    while (g1 < refs->nGenomes && (g1 == g0 || keep || bases < (1ull << 30))) {
      for (int32_t c = refs->genomeContigStart[g1]; c < refs->genomeContigStart[g1 + 1]; c++) bases += (uint64_t)refs->contigLen[c];
      g1++;
    }
    DeviceBatch db;
    int rc = upload_batch(ctx, refs, g0, g1, &db);
    RecordPart pt; pt.g0 = g0; pt.g1 = g1;
    if (rc == ANI_OK) {
      if (keep && fusable(p)) {
        FusedOut fo{&keep->fs, &keep->arr, &keep->qPool};
        rc = sketch_records(ctx, p, db, seqIdBase + refs->genomeContigStart[g0], &pt.rec, &pt.n, &fo);
      } else {
        rc = sketch_records(ctx, p, db, seqIdBase + refs->genomeContigStart[g0], &pt.rec, &pt.n);
        if (rc == ANI_OK && keep) {           // fragments that do not fit one tile: the two passes stay separate
          rc = fragment_stage(ctx, *p, db, &keep->fs);
          if (rc == ANI_OK) rc = keep_fragment_arrays(ctx, keep);
        }
      }
    }
    if (rc != ANI_OK) { cleanup(); return rc; }
    parts.push_back(pt); total += pt.n;
    g0 = g1;
  }
  uint32_t *all = nullptr;
  if (parts.size() == 1) { all = parts[0].rec; parts[0].rec = nullptr; }
  else if (total) {
    hipError_t e = pool_malloc((void **)&all, total * 12);
    size_t o = 0;
    for (size_t i = 0; i < parts.size() && e == hipSuccess; i++) {
      if (parts[i].n) e = hipMemcpyAsync(all + 3 * o, parts[i].rec, parts[i].n * 12, hipMemcpyDeviceToDevice, ctx->stream);
      o += parts[i].n;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { if (all) pool_free(all); cleanup();
      return fail(e == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "gathering %zu records failed: %s", total, hipGetErrorString(e)); }
  }
  cleanup();
  if (ctx->timerPending.size() > 1024) flush_timers(ctx);
  *devRecords = all; *n = total;
  return ANI_OK;
}

}  // namespace anih

extern "C" {

int ani_sketch_records(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, int32_t seqIdBase, void **devRecords, size_t *n)
{
  if (!ctx || !devRecords || !n) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(refs));
  HIP_TRY(hipSetDevice(ctx->device));
  return records_of_batch(ctx, p, refs, seqIdBase, devRecords, n, nullptr);
}

int ani_sketch_records_self(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *genomes, int32_t seqIdBase, void **devRecords, size_t *n,
    ani_fragset **frags)
{
  if (!ctx || !devRecords || !n || !frags) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(genomes));
  HIP_TRY(hipSetDevice(ctx->device));
  ani_fragset *f = new ani_fragset();
  f->ctx = ctx; f->device = ctx->device; f->params = *p;
  const int rc = records_of_batch(ctx, p, genomes, seqIdBase, devRecords, n, f);
  if (rc != ANI_OK) { fragset_release(f); return rc; }
  fragset_finish(f);
  *frags = f;
  return ANI_OK;
}

int ani_fragset_build(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *queries, ani_fragset **out)
{
  if (!ctx || !out) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(queries));
  HIP_TRY(hipSetDevice(ctx->device));
  ani_fragset *f = new ani_fragset();
  f->ctx = ctx; f->device = ctx->device; f->params = *p;
  DeviceBatch db;
  int rc = upload_batch(ctx, queries, 0, queries->nGenomes, &db);
  if (rc == ANI_OK) rc = fragment_stage(ctx, *p, db, &f->fs);
  if (rc == ANI_OK) rc = keep_fragment_arrays(ctx, f);
  if (rc != ANI_OK) { fragset_release(f); return rc; }
  fragset_finish(f);
  *out = f;
  return ANI_OK;
}

void ani_fragset_free(ani_fragset *f)
{
  if (!f) return;
  (void)hipSetDevice(f->device);
  fragset_release(f);
}

// -----------------------------------------------------------------------------------------------------
// Wire format of a kept fragment set: ONE device buffer = 256-byte header, genome -> fragment counts, the four per-fragment arrays
// and the hash pool, every section 256-byte aligned.  A multi-GPU run moves query fragment sketches between the GPUs in this form
// (RCCL send/recv or all-gather of plain bytes; a third of the size of the minimizer records) and maps them where they arrive.
// -----------------------------------------------------------------------------------------------------
namespace {
struct FragWireHeader {
  char magic[8]; uint32_t version; int32_t kmerSize, windowSize, fragLen; float percentageIdentity;
  int32_t nFrag, nGenomes, maxS; uint64_t nHashes, poolSize;
  uint64_t offGenomeFragments, offFragOff, offFragS, offFragGenome, offFragQSeq, offPool, totalBytes;
};
static_assert(sizeof(FragWireHeader) <= 256, "wire header fits its slot");
inline uint64_t wire_align(uint64_t x) { return (x + 255) / 256 * 256; }
FragWireHeader wire_header(const ani_fragset *f)
{
  FragWireHeader h; memset(&h, 0, sizeof h);
  memcpy(h.magic, "ANIFRAGS", 8); h.version = 1;
  h.kmerSize = f->params.kmerSize; h.windowSize = f->params.windowSize; h.fragLen = f->params.fragLen; h.percentageIdentity = f->params.percentageIdentity;
  h.nFrag = f->fs.nFrag; h.nGenomes = (int32_t)f->fs.genomeFragments.size(); h.maxS = f->fs.maxS; h.nHashes = f->fs.nHashes; h.poolSize = f->fs.poolSize;
  const uint64_t nF = (uint64_t)h.nFrag;
  h.offGenomeFragments = 256;
  h.offFragOff = wire_align(h.offGenomeFragments + (uint64_t)h.nGenomes * 4);
  h.offFragS = wire_align(h.offFragOff + nF * 4);
  h.offFragGenome = wire_align(h.offFragS + nF * 4);
  h.offFragQSeq = wire_align(h.offFragGenome + nF * 4);
  h.offPool = wire_align(h.offFragQSeq + nF * 4);
  h.totalBytes = wire_align(h.offPool + h.poolSize * 4);
  return h;
}
// One check for every header that arrives in a buffer (ani_fragset_unpack, every slot of ani_fragset_unpack_merged — those come from
// peers): magic, version, counts in range, and EVERY offset equal to what the layout rule gives for these counts, so that no field of
// the header is used as a device offset unchecked.  Returns nullptr or what is wrong.
const char *validate_wire_header(const FragWireHeader &h, uint64_t bytesAvailable)
{
  if (memcmp(h.magic, "ANIFRAGS", 8) != 0 || h.version != 1) return "not a packed fragment set";
  if (h.nFrag < 0 || h.nGenomes < 0 || h.maxS < 0 || h.poolSize > 0xfffffff0ull || h.nHashes > h.poolSize) return "inconsistent counts";
  ani_fragset tmp; tmp.fs.nFrag = h.nFrag; tmp.fs.poolSize = h.poolSize; tmp.fs.genomeFragments.assign((size_t)h.nGenomes, 0);
  const FragWireHeader want = wire_header(&tmp);       // the layout follows from the counts: the offsets in the buffer must be these
  if (want.offGenomeFragments != h.offGenomeFragments || want.offFragOff != h.offFragOff || want.offFragS != h.offFragS
      || want.offFragGenome != h.offFragGenome ||
      want.offFragQSeq != h.offFragQSeq || want.offPool != h.offPool || want.totalBytes != h.totalBytes || h.totalBytes > bytesAvailable)
    return "truncated or malformed";
  return nullptr;
}
}  // namespace

int ani_fragset_pack_bytes(const ani_fragset *f, size_t *bytes)
{
  if (!f || !bytes) return fail(ANI_ERR_ARG, "null argument");
  *bytes = (size_t)wire_header(f).totalBytes;
  return ANI_OK;
}

int ani_fragset_pack(ani_ctx *ctx, const ani_fragset *f, void *devBuf, size_t cap, size_t *bytes)
{
  if (!ctx || !f || !devBuf) return fail(ANI_ERR_ARG, "null argument");
  if (f->device != ctx->device) return fail(ANI_ERR_ARG, "fragment set lives on device %d, context on %d", f->device, ctx->device);
  HIP_TRY(hipSetDevice(ctx->device));
  const FragWireHeader h = wire_header(f);
  if (h.totalBytes > cap) return fail(ANI_ERR_ARG, "buffer of %zu bytes is too small for a fragment set of %llu bytes", cap, (unsigned long long)h.totalBytes);
  uint8_t *b = (uint8_t *)devBuf;
  const size_t nF = (size_t)h.nFrag;
  HIP_TRY(hipMemcpyAsync(b, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
  if (h.nGenomes) HIP_TRY(hipMemcpyAsync(b + h.offGenomeFragments, f->fs.genomeFragments.data(), (size_t)h.nGenomes * 4, hipMemcpyHostToDevice, ctx->stream));
  if (nF) {
    HIP_TRY(hipMemcpyAsync(b + h.offFragOff, f->fs.fragOff, nF * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(b + h.offFragS, f->fs.fragS, nF * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(b + h.offFragGenome, f->fs.fragGenome, nF * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(b + h.offFragQSeq, f->fs.fragQSeq, nF * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (h.poolSize) HIP_TRY(hipMemcpyAsync(b + h.offPool, f->fs.qPool, (size_t)h.poolSize * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));          // the header and the table are host memory of this call
  if (bytes) *bytes = (size_t)h.totalBytes;
  return ANI_OK;
}

// A view of a packed set: the arrays stay in devBuf (which the caller keeps alive until the set is freed)
int ani_fragset_unpack(ani_ctx *ctx, const void *devBuf, size_t bytes, ani_fragset **out)
{
  if (!ctx || !devBuf || !out) return fail(ANI_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if (bytes < 256) return fail(ANI_ERR_ARG, "not a packed fragment set");
  FragWireHeader h;
  HIP_TRY(hipMemcpy(&h, devBuf, sizeof h, hipMemcpyDeviceToHost));
  if (const char *why = validate_wire_header(h, bytes)) return fail(ANI_ERR_ARG, "packed fragment set: %s", why);
  ani_fragset tmp; tmp.params.kmerSize = h.kmerSize; tmp.params.windowSize = h.windowSize; tmp.params.fragLen = h.fragLen;
  tmp.params.percentageIdentity = h.percentageIdentity;
  TRY(check_params(&tmp.params));
  tmp.fs.nFrag = h.nFrag; tmp.fs.maxS = h.maxS; tmp.fs.nHashes = h.nHashes; tmp.fs.poolSize = h.poolSize; tmp.fs.genomeFragments.assign((size_t)h.nGenomes, 0);
  ani_fragset *f = new ani_fragset();
  f->ctx = ctx; f->device = ctx->device; f->params = tmp.params; f->borrowed = true;
  f->fs = tmp.fs;
  if (h.nGenomes) {
    const hipError_t e = hipMemcpy(f->fs.genomeFragments.data(), (const uint8_t *)devBuf + h.offGenomeFragments, (size_t)h.nGenomes * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete f; HIP_TRY(e); }
  }
  int64_t sum = 0;
  for (int32_t v : f->fs.genomeFragments) { if (v < 0) { sum = -1; break; } sum += v; }
  if (sum != (int64_t)h.nFrag) { delete f; return fail(ANI_ERR_ARG, "packed fragment set: genome table does not add up to the fragment count"); }
  uint8_t *b = (uint8_t *)const_cast<void *>(devBuf);
  f->arr.fragOff = (uint32_t *)(b + h.offFragOff); f->arr.fragS = (int32_t *)(b + h.offFragS);
  f->arr.fragGenome = (int32_t *)(b + h.offFragGenome); f->arr.fragQSeq = (int32_t *)(b + h.offFragQSeq);
  f->qPool = (uint32_t *)(b + h.offPool);
  f->fs.fragOff = f->arr.fragOff; f->fs.fragS = f->arr.fragS; f->fs.fragGenome = f->arr.fragGenome; f->fs.fragQSeq = f->arr.fragQSeq; f->fs.qPool = f->qPool;
  f->fs.genomeBase = 0;
  fragset_finish(f);
  *out = f;
  return ANI_OK;
}

// ONE set over several packed sets that lie in one device buffer at a fixed pitch — what an all-gather of the ranks' packed sets
// delivers.  The hash pools stay where they are (the merged pool IS the buffer: sketch offsets are rebased, so the buffer must span
// less than 2^32 hashes); the per-fragment tables are copied into tables of the set's own.  slotQueryBase[i] = query id of slot i's
// first genome, or < 0 to leave the slot out (a rank's own slot).  Mapping the merged set costs one pass of the L1 / L2 kernels where
// the sets one by one cost one pass each — and a set that meets a foreign reference shard is all launch latency.
int ani_fragset_unpack_merged(ani_ctx *ctx, const void *devBuf, size_t slotBytes, int32_t nSlots, const int32_t *slotQueryBase, ani_fragset **out)
{
  if (!ctx || !devBuf || !out || nSlots < 0 || (nSlots && !slotQueryBase)) return fail(ANI_ERR_ARG, "null argument");
  if (slotBytes % 256 != 0 || slotBytes < 256) return fail(ANI_ERR_ARG, "slot pitch must be a multiple of 256 bytes");
  if ((uint64_t)slotBytes * (uint64_t)nSlots / 4 > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "gathered fragment sets span more than 2^32 hashes");
  HIP_TRY(hipSetDevice(ctx->device));
  const uint8_t *base = (const uint8_t *)devBuf;
  std::vector<FragWireHeader> hs((size_t)nSlots);
  for (int32_t i = 0; i < nSlots; i++) if (slotQueryBase[i] >= 0) HIP_TRY(hipMemcpyAsync(&hs[i], base + (size_t)i * slotBytes, sizeof(FragWireHeader),
      hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ani_fragset *f = new ani_fragset();
  f->ctx = ctx; f->device = ctx->device; f->borrowed = true; f->ownTables = true;
  auto bail = [&](int rc) { fragset_release(f); return rc; };
  uint64_t nF = 0; int32_t nG = 0; bool first = true; int64_t nextQueryId = 0;
  for (int32_t i = 0; i < nSlots; i++) {
    if (slotQueryBase[i] < 0) continue;
    const FragWireHeader &h = hs[i];
    if (const char *why = validate_wire_header(h, slotBytes)) return bail(fail(ANI_ERR_ARG, "slot %d: %s (a peer that packed nothing, or a stale buffer?)", i,
        why));
    // the streamed re-order of map_fragsets finds a query's set by a lower bound over the genomes' query ids: the used slots must
    // come in ascending, non-overlapping query-id order (ranks own contiguous genome ranges in rank order)
    if ((int64_t)slotQueryBase[i] < nextQueryId) return bail(fail(ANI_ERR_ARG,
        "slot %d: query ids must ascend over the used slots (slot starts at %d, the slots before it end at %lld)", i, slotQueryBase[i],
            (long long)nextQueryId));
    nextQueryId = (int64_t)slotQueryBase[i] + h.nGenomes;
    ani_fragset tmp; tmp.params.kmerSize = h.kmerSize; tmp.params.windowSize = h.windowSize; tmp.params.fragLen = h.fragLen;
    tmp.params.percentageIdentity = h.percentageIdentity;
    if (first) { f->params = tmp.params; const int rc = check_params(&f->params); if (rc != ANI_OK) return bail(rc); first = false; }
    else if (f->params.kmerSize != h.kmerSize || f->params.windowSize != h.windowSize || f->params.fragLen != h.fragLen
        || f->params.percentageIdentity != h.percentageIdentity)
      return bail(fail(ANI_ERR_ARG, "slot %d was sketched with other parameters", i));
    nF += (uint64_t)h.nFrag; nG += h.nGenomes;
    if (nF > 0x3fffffffull) return bail(fail(ANI_ERR_LIMIT, "too many fragments in the merged set"));
  }
  // nothing to merge: an empty set
  if (first) { f->params.kmerSize = 16; f->params.windowSize = 1; f->params.fragLen = 3000; f->params.percentageIdentity = 80.0f; }
  f->fs.nFrag = (int32_t)nF; f->fs.poolSize = (uint64_t)slotBytes * (uint64_t)nSlots / 4; f->fs.genomeBase = 0;
  f->qPool = (uint32_t *)const_cast<void *>(devBuf); f->fs.qPool = f->qPool;
  if (nF) {
    hipError_t e = pool_malloc((void **)&f->arr.fragOff, nF * 4); if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragS, nF * 4);
    if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragGenome, nF * 4); if (e == hipSuccess) e = pool_malloc((void **)&f->arr.fragQSeq, nF * 4);
    if (e != hipSuccess) return bail(fail(e == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "tables of the merged fragment set: %s",
        hipGetErrorString(e)));
  }
  uint64_t fAt = 0; int32_t gAt = 0;
  std::vector<int32_t> gf;
  for (int32_t i = 0; i < nSlots; i++) {
    if (slotQueryBase[i] < 0) continue;
    const FragWireHeader &h = hs[i];
    const uint8_t *b = base + (size_t)i * slotBytes;
    gf.assign((size_t)h.nGenomes, 0);
    if (h.nGenomes) { const hipError_t e = hipMemcpy(gf.data(), b + h.offGenomeFragments, (size_t)h.nGenomes * 4, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return bail(fail(ANI_ERR_DEVICE, "reading slot %d: %s", i, hipGetErrorString(e))); }
    int64_t sum = 0;
    for (int32_t v : gf) { if (v < 0) { sum = -1; break; } sum += v; }
    if (sum != (int64_t)h.nFrag) return bail(fail(ANI_ERR_ARG, "slot %d: genome table does not add up to the fragment count", i));
    for (int32_t g = 0; g < h.nGenomes; g++) { f->fs.genomeFragments.push_back(gf[g]); f->genomeQueryId.push_back(slotQueryBase[i] + g); }
    if (h.nFrag) {
      const uint32_t addOff = (uint32_t)(((size_t)i * slotBytes + h.offPool) / 4);
      hipLaunchKernelGGL(k_fragset_rebase, dim3(std::min<unsigned>(grid_for((size_t)h.nFrag), 65535u)), dim3(256), 0, ctx->stream, (uint32_t)h.nFrag,
          (const uint32_t *)(b + h.offFragOff), (const int32_t *)(b + h.offFragS),
                         (const int32_t *)(b + h.offFragGenome), (const int32_t *)(b + h.offFragQSeq), addOff, gAt,
                         f->arr.fragOff + fAt, f->arr.fragS + fAt, f->arr.fragGenome + fAt, f->arr.fragQSeq + fAt);
    }
    f->fs.maxS = std::max(f->fs.maxS, h.maxS); f->fs.nHashes += h.nHashes;
    fAt += (uint64_t)h.nFrag; gAt += h.nGenomes;
  }
  { const hipError_t e = hipGetLastError(); const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || e2 != hipSuccess) return bail(fail(ANI_ERR_DEVICE, "merging fragment sets failed")); }
  f->fs.fragOff = f->arr.fragOff; f->fs.fragS = f->arr.fragS; f->fs.fragGenome = f->arr.fragGenome; f->fs.fragQSeq = f->arr.fragQSeq;
  fragset_finish(f);
  *out = f;
  return ANI_OK;
}

int ani_fragset_info(const ani_fragset *f, int32_t *nGenomes, int64_t *nFragments, uint64_t *nHashes)
{
  if (!f) return fail(ANI_ERR_ARG, "null argument");
  if (nGenomes) *nGenomes = (int32_t)f->fs.genomeFragments.size();
  if (nFragments) *nFragments = f->fs.nFrag;
  if (nHashes) *nHashes = f->fs.nHashes;
  return ANI_OK;
}

int ani_query_sketch(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *query, uint32_t **hashes, uint64_t **offsets, size_t *nFragments)
{
  if (!ctx || !hashes || !offsets || !nFragments) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(query));
  HIP_TRY(hipSetDevice(ctx->device));
  DeviceBatch db; FragSet fs;
  TRY(upload_batch(ctx, query, 0, query->nGenomes, &db));
  TRY(fragment_stage(ctx, *p, db, &fs));
  const size_t nF = (size_t)fs.nFrag;
  std::vector<uint32_t> off(nF); std::vector<int32_t> s(nF);
  if (nF) {
    HIP_TRY(hipMemcpy(off.data(), ctx->fragOff.p, nF * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s.data(), ctx->fragS.p, nF * 4, hipMemcpyDeviceToHost));
  }
  std::vector<uint64_t> offs(nF + 1, 0);
  for (size_t f = 0; f < nF; f++) offs[f + 1] = offs[f] + (uint64_t)(s[f] > 0 ? s[f] : 0);
  std::vector<uint32_t> pool((size_t)fs.poolSize), h(offs[nF]);       // the pool as the kernel left it: 64 partly filled stripes
  if (fs.poolSize) HIP_TRY(hipMemcpy(pool.data(), ctx->qPool.p, (size_t)fs.poolSize * 4, hipMemcpyDeviceToHost));
  for (size_t f = 0; f < nF; f++)
    if (s[f] > 0) memcpy(h.data() + offs[f], pool.data() + off[f], (size_t)s[f] * 4);
  size_t dummy;
  TRY(to_host_malloc(h, hashes, &dummy));
  const int rc = to_host_malloc(offs, offsets, &dummy);
  if (rc != ANI_OK) { free(*hashes); *hashes = nullptr; return rc; }
  *nFragments = nF;
  return ANI_OK;
}


}  // extern "C"
