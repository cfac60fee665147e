// engine_core.hip — life cycle of a context, the caching device allocator, counters and stage timers, the device-wide prefix sum,
// the host scalars of skch::Stat (src/map/include/map_stats.hpp) and the synthetic benchmark genomes.
#include "host/engine.hpp"
#include "kernels/scan.hpp"
#include "kernels/synth.hpp"

namespace anih {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}

DevicePool g_pools[64];

int zero_counters(ani_ctx *c)
{
  HIP_TRY(hipMemsetAsync(c->dCounters.p, 0, kCounterWords * 8, c->stream));
  return ANI_OK;
}
int zero_cursors(ani_ctx *c, int pool)
{
  HIP_TRY(hipMemsetAsync(cur_ptr(c, pool), 0, (size_t)ani::kPoolStripes * ani::kPoolStripeWords * 8, c->stream));
  return ANI_OK;
}
int read_counters(ani_ctx *c, unsigned long long *host)
{
  std::vector<unsigned long long> &all = c->hostCounters;
  all.resize(kCounterWords);
  HIP_TRY(hipMemcpyAsync(all.data(), c->dCounters.p, kCounterWords * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < CNT_N; i++) host[i] = all[i];
  for (int s = 1; s < ani::kStatStripes; s++)
    for (int i = 0; i < CNT_N; i++) {
      if (i == CNT_MAXS) host[i] = std::max<unsigned long long>(host[i] & 0xffffffffull, all[(size_t)s * CNT_N + i] & 0xffffffffull);
      else host[i] += all[(size_t)s * CNT_N + i];
    }
  for (int p = 0; p < POOL_N; p++) {
    c->poolUsed[p] = 0; c->poolMaxStripe[p] = 0;
    const unsigned long long *cur = all.data() + (size_t)ani::kStatStripes * CNT_N + (size_t)p * ani::kPoolStripes * ani::kPoolStripeWords;
    for (int r = 0; r < ani::kPoolStripes; r++) { const unsigned long long v = cur[(size_t)r * ani::kPoolStripeWords]; c->poolUsed[p] += v;
      c->poolMaxStripe[p] = std::max(c->poolMaxStripe[p], v); }
  }
  return ANI_OK;
}
int pinned_buffer(ani_ctx *c, int slot, size_t bytes, void **out)
{
  if (c->pinnedCap[slot] < bytes) {
    if (c->pinned[slot]) (void)hipHostFree(c->pinned[slot]);
    c->pinned[slot] = nullptr; c->pinnedCap[slot] = 0;
    const size_t cap = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc(&c->pinned[slot], cap, 0));
    c->pinnedCap[slot] = cap;
  }
  *out = c->pinned[slot];
  return ANI_OK;
}

void add_counters(ani_counters_t *d, const ani_counters_t *s)
{
#define ADD(f) d->f += s->f
  ADD(refBases); ADD(refMinimizers); ADD(refUniqueHashes); ADD(queryGenomes); ADD(queryFragments); ADD(queryBases); ADD(querySketchHashes); ADD(seedHits);
  ADD(l1Candidates);
  ADD(l2WindowEntries); ADD(l2Steps); ADD(l2QueryHashes); ADD(l2WindowEntriesB); ADD(l2QueryHashesB); ADD(l2Launches); ADD(l2FastCandidates);
  ADD(l2SlowCandidates);
  ADD(l2SlowLimit); ADD(l2SlowDup); ADD(l2SlowOverflow); ADD(mappings); ADD(cgiRows); ADD(indexChunks); ADD(l1Probes); ADD(l2ChunkHalvings);
  ADD(indexChunkBuilds);
  ADD(l1BigFragments); ADD(l1MidFragments); ADD(l1TinyFragments);
  ADD(msSketch); ADD(msIndex); ADD(msFragSketch); ADD(msL1); ADD(msL2); ADD(msReduce); ADD(msL2Kernel); ADD(msL2Ranges); ADD(msL2Codes); ADD(msL2Slow);
  ADD(msL2SimB);
  ADD(msL1Probe); ADD(msL1Main); ADD(msL1Big); ADD(msL1Tiny);
#undef ADD
}

void flush_timers(ani_ctx *c)
{
  if (c->timerPending.empty()) { c->timerUsed = 0; return; }
  for (const auto &p : c->timerPending) {
    (void)hipEventSynchronize(c->timerEvents[p.b]);
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->timerEvents[p.a], c->timerEvents[p.b]) == hipSuccess) *p.acc += ms;
  }
  c->timerPending.clear(); c->timerUsed = 0;
}
size_t timer_event(ani_ctx *c)
{
  if (c->timerUsed == c->timerEvents.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); c->timerEvents.push_back(e); }
  return c->timerUsed++;
}
int device_scan(ani_ctx *c, const int32_t *in, uint32_t *out, uint32_t n, uint64_t *total, uint64_t limit)
{
  using namespace ani;
  *total = 0;
  if (n == 0) return ANI_OK;
  const uint32_t nb1 = (n + kScanPerBlock - 1) / kScanPerBlock;
  TRY(c->scanTmpA.ensure((size_t)nb1 * 4)); TRY(c->scanTmpB.ensure((size_t)nb1 * 4));
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb1), dim3(kTPB), 0, c->stream, in, out, n, c->scanTmpA.as<int32_t>());
  uint32_t *tot = nullptr;
  TRY(pinned_buffer(c, 2, (size_t)nb1 * 8, (void **)&tot));
  uint32_t *off = tot + nb1;
  HIP_TRY(hipMemcpyAsync(tot, c->scanTmpA.p, (size_t)nb1 * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  uint64_t run = 0;
  for (uint32_t i = 0; i < nb1; i++) { off[i] = (uint32_t)run; run += (uint64_t)tot[i]; }     // block totals are < 2^31 each (see k_scan_blocks)
  if (run > limit) return fail(ANI_ERR_LIMIT, "prefix sum of %llu elements exceeds the limit of %llu", (unsigned long long)run, (unsigned long long)limit);
  *total = run;
  if (nb1 > 1) {
    HIP_TRY(hipMemcpyAsync(c->scanTmpB.p, off, (size_t)nb1 * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, c->stream, out, n, (const uint32_t *)c->scanTmpB.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));       // `off` (pinned, reused by the next scan) has been consumed
  }
  return ANI_OK;
}

}  // namespace anih

extern "C" {

const char *ani_last_error(void) { return g_err.c_str(); }
void ani_free(void *p) { free(p); }
void ani_device_free(ani_ctx *ctx, void *p)
{
  if (!p) return;
  if (ctx) (void)hipSetDevice(ctx->device);      // the pools are per device: the block goes back to the pool of the context that owns it
  pool_free(p);
}

int ani_init(int device, ani_ctx **out)
{
  if (!out) return fail(ANI_ERR_ARG, "null output");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(ANI_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(ANI_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  ani_ctx *c = new ani_ctx();
  c->device = device;
  memset(&c->counters, 0, sizeof c->counters);
  HIP_TRY(hipStreamCreate(&c->stream));
  {
    // ANI_TEST_SIDE_PRIORITY=low|high: queue priority of the side stream (index side work, the L2 simulation) — an A/B knob
    const char *ev = getenv("ANI_TEST_SIDE_PRIORITY");
    int least = 0, greatest = 0;
    if (ev && (!strcmp(ev, "low") || !strcmp(ev, "high")) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
      HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamDefault, !strcmp(ev, "low") ? least : greatest));
    else HIP_TRY(hipStreamCreate(&c->stream2));
  }
  if (const char *ev = getenv("ANI_SUBBATCH_FRAGS")) { const long long v = atoll(ev); if (v > 0) c->subBatchFragments = (uint64_t)v; }
  if (const char *ev = getenv("ANI_TEST_L2_CHUNK")) { const long long v = atoll(ev); if (v >= 1) c->l2ChunkCandidates = (size_t)v; }
  if (const char *ev = getenv("ANI_TEST_L2_CODE_LIMIT")) { const long long v = atoll(ev); if (v >= 1) c->l2CodeLimit = (uint64_t)v; }
  if (const char *ev = getenv("ANI_MAX_INDEX_MINIMIZERS")) { const long long v = atoll(ev); if (v >= 1) c->maxIndexMinimizers = (uint64_t)v; }
  if (const char *ev = getenv("ANI_TEST_L1_FILTER_MIN")) c->l1FilterMin = atoi(ev);
  if (const char *ev = getenv("ANI_TEST_L1_LDS_MAX")) c->l1LdsMax = std::max(0, std::min(atoi(ev), 4080 /* ani::kL1HitCapMax */));
  if (const char *ev = getenv("ANI_TEST_DUP_PAIR_CAP")) { const long long v = atoll(ev); if (v >= 1) c->dupPairCap = (uint64_t)v; }
  if (const char *ev = getenv("ANI_TEST_L1_TINY")) c->l1Tiny = strcmp(ev, "0") != 0;
  if (const char *ev = getenv("ANI_TEST_L2_OVERLAP")) c->l2Overlap = strcmp(ev, "0") != 0;
  if (const char *ev = getenv("ANI_TEST_L1_HIT_LIMIT")) { const long long v = atoll(ev);
    if (v >= 1) c->l1HitLimit = std::min<uint64_t>((uint64_t)v, 0x7ffffff0ull); }
  if (const char *ev = getenv("ANI_TEST_CAND_POOL_MIN")) { const long long v = atoll(ev); if (v >= 1) c->candPoolMin = (uint64_t)v; }
  if (const char *ev = getenv("ANI_TEST_L1_BIG_GROUP_HITS")) { const long long v = atoll(ev); if (v >= 1) c->l1BigGroupHits = (uint64_t)v; }
  if (const char *ev = getenv("ANI_TEST_L1_BIG_GROUP_FRAGS")) { const long long v = atoll(ev); if (v >= 1) c->l1BigGroupFrags = (uint64_t)v; }
  if (const char *ev = getenv("ANI_MAX_RESIDENT_CHUNKS")) { const long long v = atoll(ev);
    if (v >= 0) c->maxResidentChunks = (int32_t)std::min<long long>(v, 1 << 20); }
  if (const char *ev = getenv("ANI_STREAM_CHUNK_MINIMIZERS")) { const long long v = atoll(ev); if (v >= 1) c->streamChunkMinimizers = (uint64_t)v; }
  for (int i = 0; i < 2; i++) { HIP_TRY(hipEventCreateWithFlags(&c->evSimA[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->evSetDone[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->evIndex[i], hipEventDisableTiming)); }
  int rc = c->dCounters.ensure(kCounterWords * 8);
  if (rc != ANI_OK) { delete c; return rc; }
  *out = c;
  return ANI_OK;
}

void ani_shutdown(ani_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  DevBuf *bufs[] = {&c->dCounters, &c->seqPacked, &c->seqAscii, &c->contigOff, &c->contigLen, &c->contigMode, &c->sortTmp, &c->unitStart, &c->unitAux,
      &c->tiles, &c->tileInfo, &c->tileMeta, &c->tileCnt,
                    &c->tileDrop, &c->tileOff, &c->poolHash, &c->poolWpos, &c->scanTmpA, &c->scanTmpB, &c->scanTmpC, &c->scanTmpD, &c->frags, &c->fragOff,
                        &c->fragS,
                    &c->fragGenome, &c->fragQSeq, &c->qPool, &c->l1FragDesc, &c->probeFirst, &c->probeCnt, &c->l1MidList, &c->l1SmallList, &c->l1BigList,
                        &c->l1BigHitsA, &c->l1BigHitsB, &c->l1BigV, &c->l1BigTbl, &c->l1BigHash, &c->candFrag, &c->candSeq, &c->candStart, &c->candEnd,
                        &c->fragCandOff, &c->fragCandCnt,
                    &c->fragCandCntClamped, &c->fragHits, &c->fragOrdOff, &c->fragOrder, &c->fragOrderTmp, &c->ocFrag, &c->ocSeq, &c->ocStart, &c->ocEnd,
                        &c->l2Scratch, &c->l2Ranges[0], &c->l2CodeCount[0], &c->l2CodeOff[0], &c->l2Codes[0], &c->l2SlowFlag[0], &c->l2ClassList[0],
                        &c->l2Order[0], &c->l2LenHist[0],
                    &c->l2Ranges[1], &c->l2CodeCount[1], &c->l2CodeOff[1], &c->l2Codes[1], &c->l2SlowFlag[1], &c->l2ClassList[1], &c->l2Order[1],
                        &c->l2LenHist[1], &c->l2SlowList, &c->l2Best,
                    &c->l2First, &c->l2Last, &c->refStart, &c->idBits, &c->keepFlags, &c->keepOff, &c->mapOut, &c->bins, &c->queryFragments, &c->rows};
  for (DevBuf *b : bufs) b->release();
  for (hipEvent_t e : c->timerEvents) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; i++) { if (c->evSimA[i]) (void)hipEventDestroy(c->evSimA[i]); if (c->evSetDone[i]) (void)hipEventDestroy(c->evSetDone[i]);
    if (c->evIndex[i]) (void)hipEventDestroy(c->evIndex[i]); }
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  for (int i = 0; i < 5; i++) if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  cur_pool().trim();
}

int ani_device_copy(ani_ctx *c, void *dst, const void *src, size_t bytes)
{
  if (!c || (bytes && (!dst || !src))) return fail(ANI_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  if (bytes) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
  return ANI_OK;
}

int ani_device_alloc(ani_ctx *c, size_t bytes, void **out)
{
  if (!c || !out) return fail(ANI_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(pool_malloc(out, bytes ? bytes : 1));
  return ANI_OK;
}

int ani_device_memory(ani_ctx *c, size_t *freeBytes, size_t *totalBytes)
{
  if (!c) return fail(ANI_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (freeBytes) *freeBytes = f;
  if (totalBytes) *totalBytes = t;
  return ANI_OK;
}

// device-to-device copy between two contexts (the same or different GPUs; over xGMI when peer access is possible), synchronous
int ani_device_copy_peer(ani_ctx *dstCtx, void *dst, ani_ctx *srcCtx, const void *src, size_t bytes)
{
  if (!dstCtx || !srcCtx || (bytes && (!dst || !src))) return fail(ANI_ERR_ARG, "null argument");
  if (!bytes) return ANI_OK;
  HIP_TRY(hipSetDevice(dstCtx->device));
  if (dstCtx->device == srcCtx->device) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dstCtx->stream));
  else {
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, dstCtx->device, srcCtx->device);
    if (can) { hipError_t e = hipDeviceEnablePeerAccess(srcCtx->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
      (void)hipGetLastError(); }
    HIP_TRY(hipMemcpyPeerAsync(dst, dstCtx->device, src, srcCtx->device, bytes, dstCtx->stream));
  }
  HIP_TRY(hipStreamSynchronize(dstCtx->stream));
  return ANI_OK;
}

int ani_get_counters(ani_ctx *c, ani_counters_t *out)
{
  if (!c || !out) return fail(ANI_ERR_ARG, "null argument");
  flush_timers(c);
  *out = c->counters;
  return ANI_OK;
}
int ani_reset_counters(ani_ctx *c)
{
  if (!c) return fail(ANI_ERR_ARG, "null argument");
  flush_timers(c);
  memset(&c->counters, 0, sizeof c->counters);
  return ANI_OK;
}

int ani_recommended_window(int k, int fragLen)
{
  // fixed arguments of the CLI: p_value 1e-3, alphabet 4, identity 80, reference size 5e6 (parseCmdArgs.hpp:118-127, :225-228)
  return ani::stat::recommended_window_size(1e-03, k, 4, 80.0f, fragLen, 5000000);
}
int ani_params_default(ani_params_t *p, int kmerSize, int fragLen)
{
  if (!p) return fail(ANI_ERR_ARG, "null parameters");
  p->kmerSize = kmerSize > 0 ? kmerSize : 16;
  p->fragLen = fragLen > 0 ? fragLen : 3000;
  p->percentageIdentity = 80.0f;
  if (p->kmerSize > 16) return fail(ANI_ERR_ARG, "kmerSize must be <= 16");
  p->windowSize = ani_recommended_window(p->kmerSize, p->fragLen);
  return ANI_OK;
}
int ani_min_hits_relaxed(int s, int k, float identity) { return ani::stat::estimate_minimum_hits_relaxed(s, k, identity); }
int ani_identity(int shared, int s, int k, float *nucIdentity, float *upperBound)
{
  if (s <= 0 || shared < 0 || shared > s || !nucIdentity) return fail(ANI_ERR_ARG, "invalid (shared, sketchSize)");
  ani::stat::identity(shared, s, k, nucIdentity, upperBound);
  return ANI_OK;
}

int ani_synth_packed(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, void *devOut)
{
  return ani_synth_packed_clusters(ctx, seed, variant, firstGenomeId, nGenomes, genomeLen, 20, devOut);
}

int ani_synth_packed_clusters(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, int32_t clusterSize,
    void *devOut)
{
  if (!ctx || !devOut || nGenomes < 0 || genomeLen <= 0 || firstGenomeId < 0 || clusterSize < 1) return fail(ANI_ERR_ARG, "invalid argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t words = (size_t)nGenomes * (((size_t)genomeLen + 15) / 16);
  if (words == 0) return ANI_OK;
  hipLaunchKernelGGL(ani::k_synth_packed, dim3(grid_for(words, 256, 65535)), dim3(256), 0, ctx->stream, seed, variant, firstGenomeId, nGenomes, genomeLen,
      clusterSize, (uint32_t *)devOut);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ANI_OK;
}


}  // extern "C"
