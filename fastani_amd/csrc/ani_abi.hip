// ani_abi.hip — C-ABI of the MI355X-native ANI engine (include/ani_abi.h): host orchestration of the HIP kernels.
//
// The three reference seams this file stands in for (paths relative to /root/reference):
//   skch::Sketch::Sketch   src/map/include/winSketch.hpp:109      -> ani_sketch_build / ani_sketch_from_records
//   skch::Map::Map         src/map/include/computeMap.hpp:93      -> ani_map_query
//   cgi::computeCGI        src/cgi/include/computeCoreIdentity.hpp:166 -> ani_compute_cgi
// plus the fused query loop of src/cgi/core_genome_identity.cpp:81-106 -> ani_map_cgi_batch.
// There is no CPU code path behind these entry points: without a usable HIP device every call fails.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <thread>
#include <chrono>
#include <memory>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ani_abi.h"
#include "host/stats.hpp"
#include "kernels/common.hpp"
#include "kernels/index.hpp"
#include "kernels/l1.hpp"
#include "kernels/l2.hpp"
#include "kernels/reduce.hpp"
#include "kernels/scan.hpp"
#include "kernels/sketch.hpp"
#include "kernels/synth.hpp"

// kernels/radix.hpp through sort_device.hip (hand-written LSD radix sort; two-phase calls: tmp == nullptr returns the workspace size)
extern "C" int ani_sort_keys_u64_bits(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream);
extern "C" int ani_sort_pairs_u64_u32(const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                      size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream);
extern "C" int ani_sort_index(const void *const *pieceRec, const size_t *pieceN, int nPieces, uint32_t seqBase, size_t n,
                              uint32_t *mHash, int32_t *mSeq, int32_t *mWpos, uint32_t *tmpK, uint64_t *tmpV, uint32_t *sHash, uint64_t *sSW,
                              void *tmp, size_t *tmpBytes, hipStream_t stream, hipEvent_t soaReady, hipStream_t sideStream);
extern "C" int ani_sort_check(void *tmp, hipStream_t stream);

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}

// -----------------------------------------------------------------------------------------------------
// Caching device allocator: hipMalloc/hipFree of multi-GB arrays cost milliseconds each and a many-to-many run builds and
// drops the same-sized index arrays over and over; freed blocks are kept and handed out again (best fit within 25 %).
// On allocation failure the cache is dropped and the request retried.
// -----------------------------------------------------------------------------------------------------
struct DevicePool {
  std::mutex mu;
  std::unordered_map<void *, size_t> live;
  std::multimap<size_t, void *> cache;
  size_t cachedBytes = 0;
  hipError_t alloc(void **out, size_t bytes)
  {
    if (bytes == 0) bytes = 1;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.lower_bound(bytes);
    // reuse only a closely fitting block: a looser fit lets a long-lived buffer capture the block a per-sketch array of a
    // different size will ask for again, and that array then needs fresh memory in the middle of a later step
    // ANI_POOL_POISON=<byte> (tests): every block handed out is filled with that byte first, so that a kernel which reads memory
    // it (or an earlier kernel) has not written sees the same garbage every time instead of whatever the previous owner left
    static const int poison = getenv("ANI_POOL_POISON") ? (int)strtol(getenv("ANI_POOL_POISON"), nullptr, 0) : -1;
    if (it != cache.end() && it->first <= bytes + bytes / 16 + (1u << 20)) {
      *out = it->second; live[*out] = it->first; cachedBytes -= it->first; cache.erase(it);
      if (poison >= 0) { (void)hipDeviceSynchronize(); (void)hipMemset(*out, poison, bytes); (void)hipDeviceSynchronize(); }
      return hipSuccess;
    }
    static const bool trace = getenv("ANI_POOL_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, bytes);
    size_t freed = 0; bool loose = false;
    if (e != hipSuccess) {
      // The device is full while this pool sits on cached blocks.  Dropping the whole cache is what NOT to do: hipFree costs ~40 ms
      // per GB on this stack (a 125 GB cache: 4.8 s, measured in the warm step of the 10 000 x 10 000 run, profiles/r03l).  First any
      // cached block that is large enough serves, whatever its slack; else cached blocks go one at a time, largest first, until the
      // request fits.
      (void)hipGetLastError();
      if (it != cache.end()) {
        *out = it->second; live[*out] = it->first; cachedBytes -= it->first; cache.erase(it);
        e = hipSuccess; loose = true;
      } else {
        while (e != hipSuccess && free_largest_locked(&freed)) { e = hipMalloc(out, bytes); if (e != hipSuccess) (void)hipGetLastError(); }
        if (e == hipSuccess) live[*out] = bytes;
      }
    } else live[*out] = bytes;
    if (trace) fprintf(stderr, "[ani pool] hipMalloc %.1f MB: %.2f ms%s (cache %.1f MB in %zu blocks%s)\n", bytes / 1048576.0,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), e == hipSuccess ? "" : " FAILED", cachedBytes / 1048576.0, cache.size(),
                       loose ? "; device full: served by a larger cached block" : freed ? "; device full: cached blocks freed" : "");
    if (e == hipSuccess && poison >= 0) { (void)hipDeviceSynchronize(); (void)hipMemset(*out, poison, bytes); (void)hipDeviceSynchronize(); }
    return e;
  }
  // frees the largest cached block; false if the cache is empty
  bool free_largest_locked(size_t *freedBytes)
  {
    if (cache.empty()) return false;
    auto last = std::prev(cache.end());
    (void)hipFree(last->second);
    cachedBytes -= last->first; if (freedBytes) *freedBytes += last->first;
    cache.erase(last);
    return true;
  }
  bool free_largest() { std::lock_guard<std::mutex> g(mu); return free_largest_locked(nullptr); }
  void release(void *p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(p);
    if (it == live.end()) { (void)hipFree(p); return; }
    cache.emplace(it->second, p); cachedBytes += it->second; live.erase(it);
  }
  void trim_locked() { for (auto &kv : cache) (void)hipFree(kv.second); cache.clear(); cachedBytes = 0; }
  void trim() { std::lock_guard<std::mutex> g(mu); trim_locked(); }
};
// Two pools per device ordinal (one process normally drives one GPU): class 0 for what lives and dies with a sketch (its arrays
// and the transient buffers of its build), class 1 for the grow-only buffers of a context.  Kept apart so that a context buffer
// growing between two sketch builds cannot capture a block the next build will ask for again — fresh device memory in the
// middle of a steady-state step costs up to ~30 us/MB on this stack (a 4.5 GB hipMalloc: 135 ms).
DevicePool g_pools[64][2];
inline DevicePool &cur_pool(int cls) { int d = 0; (void)hipGetDevice(&d); return g_pools[(d < 0 || d >= 64) ? 0 : d][cls]; }
inline hipError_t pool_malloc(void **p, size_t bytes, int cls = 0)
{
  hipError_t e = cur_pool(cls).alloc(p, bytes);           // gives up only with its own cache empty
  while (e != hipSuccess && cur_pool(cls ^ 1).free_largest()) { (void)hipGetLastError(); e = cur_pool(cls).alloc(p, bytes); }   // then the other class's cache, block by block
  return e;
}
inline void pool_free(void *p)
{
  if (!p) return;
  for (int cls = 0; cls < 2; cls++) {
    DevicePool &pl = cur_pool(cls);
    { std::lock_guard<std::mutex> g(pl.mu); if (!pl.live.count(p)) continue; }
    pl.release(p);
    return;
  }
  (void)hipFree(p);
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE,       \
                                      "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_ != ANI_OK) return rc_; } while (0)

// grow-only device buffer
struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  int ensure(size_t bytes)
  {
    if (bytes <= cap) return ANI_OK;
    if (p) { pool_free(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = pool_malloc(&p, want, 1);
    if (e != hipSuccess) { p = nullptr; return fail(ANI_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    cap = want;
    return ANI_OK;
  }
  void release() { if (p) pool_free(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return (T *)p; }
};

// host-side parallel loop for the ingest path (ANI_HOST_THREADS overrides the thread count; small jobs stay serial)
template <class F>
void parallel_for(size_t n, uint64_t work, F f)
{
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > 64) nt = 64;
  if (const char *ev = getenv("ANI_HOST_THREADS")) { int v = atoi(ev); if (v >= 1) nt = (unsigned)v; }
  if (nt > n) nt = (unsigned)n;
  if (nt <= 1 || work < (1u << 22)) { for (size_t i = 0; i < n; i++) f(i); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
  for (auto &x : th) x.join();
}

// 1-D grid for n work items.  Kernels without a grid-stride loop are launched with the default (uncapped: gridDim.x may reach
// 2^31 - 1 on HIP, and every count on this path is < 2^31); kernels that loop pass the cap they want.
inline unsigned grid_for(size_t n, unsigned block = 256, unsigned maxBlocks = 0x7fffffffu)
{
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > maxBlocks) g = maxBlocks;
  return (unsigned)g;
}

}  // namespace

// =====================================================================================================
struct ani_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t subBatchFragments = 1u << 20, subBatchBinBytes = (uint64_t)8 << 30;   // sub-batch bounds of ani_map_cgi_batch (env ANI_SUBBATCH_FRAGS)
  size_t l2ChunkCandidates = (size_t)1 << 21;                                      // L2 chunk size (env ANI_L2_CHUNK, tests)
  uint64_t l2CodeLimit = 0xfffffff0ull;                                             // 16-bit code entries per L2 chunk (32-bit offsets; env ANI_L2_CODE_LIMIT, tests)
  uint64_t maxIndexMinimizers = 1700000000ull;                                      // minimizers per index chunk (env ANI_MAX_INDEX_MINIMIZERS); indices are 32 bit
  int l1FilterMin = ani::kL1FilterMinHits, l1LdsMax = ani::kL1HitCapMax;           // env ANI_L1_FILTER_MIN / ANI_L1_LDS_MAX, read by ani_init (tests: per engine, not per process)
  uint64_t dupPairCap = 0;                                                           // first guess of the same-hash link list (env ANI_DUP_PAIR_CAP, tests: forces the rerun)
  bool l1Tiny = true;                                                               // env ANI_L1_TINY=0: fragments with <= 64 seed hits take the workgroup path like the others
  bool l2Overlap = false;                                                           // env ANI_L2_OVERLAP=1 (see the L2 loop)
  uint64_t l1HitLimit = 0x7ffffff0ull;                                              // seed hits per fragment and index chunk (32-bit hit offsets; env ANI_L1_HIT_LIMIT, tests)
  uint64_t candPoolMin = 4096;                                                      // floor of the L1 candidate pool, per stripe (env ANI_CAND_POOL_MIN, tests: forces the retry path)
  uint64_t l1BigGroupHits = 1ull << 27, l1BigGroupFrags = 1ull << 20;              // seed hits / fragments per group of the batched global-memory L1 path (env ANI_L1_BIG_GROUP_HITS / _FRAGS, tests)
  int32_t maxResidentChunks = 0;                                                    // index chunks of one reference set kept on the device (env ANI_MAX_RESIDENT_CHUNKS; 0 = decide from the free memory)
  uint64_t streamChunkMinimizers = 1000000000ull;                                   // chunk size once a set is streamed (env ANI_STREAM_CHUNK_MINIMIZERS): the build's transient arrays must fit beside the records
  std::vector<std::unique_ptr<ani::stat::Luts>> lutCache;
  void *pinned[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; size_t pinnedCap[5] = {0, 0, 0, 0, 0};   // page-locked staging: 0/1 result reads, 2 prefix-sum block totals, 3/4 ingest (packed / raw bytes)
  hipStream_t stream2 = nullptr;          // side stream: latency-bound launches that can run under the main simulation kernel
  hipEvent_t evSimA[2] = {nullptr, nullptr}, evSetDone[2] = {nullptr, nullptr};
  // stage timers: event pairs are recorded as the launches go out and read back lazily (flush_timers), never by blocking the host
  std::vector<hipEvent_t> timerEvents; size_t timerUsed = 0;
  struct PendingTimer { size_t a, b; double *acc; };
  std::vector<PendingTimer> timerPending;
  ani_counters_t counters;
  std::vector<unsigned long long> hostCounters;                 // read_counters: the raw block
  unsigned long long poolUsed[3] = {0, 0, 0}, poolMaxStripe[3] = {0, 0, 0};
  double candPerFrag = 12.0;      // estimate that sizes the L1 candidate pool: the LARGEST need of the last 32 batches that fill the pool stripes evenly
  double candSeen[32] = {0}; int candSeenAt = 0;   // (a ring of fragment sets alternates between sets with ~16 candidates per fragment — the rank's own genomes — and sets
                                                    //  with < 1: an estimate that follows the batches down runs the L1 kernels twice for every dense one)
  uint64_t smallBatchCandCap = 0; // ... and what the last batch too small for that estimate needed (a few hundred fragments against a species-dense index, call after call)
  // scalar device counters (array of 16 x u64)
  DevBuf dCounters;
  // workspaces reused across calls
  DevBuf seqPacked, seqAscii, contigOff, contigLen, contigMode;
  DevBuf sortTmp, unitStart, unitAux, tiles, tileInfo, tileMeta, tileCnt, tileDrop, tileOff, poolHash, poolWpos;
  DevBuf scanTmpA, scanTmpB, scanTmpC, scanTmpD;
  DevBuf frags, fragOff, fragS, fragGenome, fragQSeq, qPool;
  DevBuf probeFirst, probeCnt, l1MidList, l1SmallList, l1BigList, l1BigHitsA, l1BigHitsB, l1BigV, l1BigTbl, l1BigHash, candFrag, candSeq, candStart, candEnd, fragCandOff, fragCandCnt, fragCandCntClamped, fragHits, fragOrdOff, fragOrder, fragOrderTmp;
  DevBuf ocFrag, ocSeq, ocStart, ocEnd;
  DevBuf l2Scratch, l2Best, l2First, l2Last, refStart, idBits, keepFlags, keepOff, mapOut;
  DevBuf l2Ranges[2], l2CodeCount[2], l2CodeOff[2], l2Codes[2], l2SlowFlag[2], l2ClassList[2], l2Order[2], l2LenHist[2];   // two chunk sets (see the L2 loop)
  DevBuf l2SlowList;
  DevBuf bins, queryFragments, rows;
};

// One index over a contiguous run of reference genomes: < 2^31 minimizers, 32-bit indices, chunk-local seqIds.  A reference set
// larger than that is a list of chunks (the reference's own answer to big databases is the same split, per OpenMP thread:
// computeCoreIdentity.hpp:457-487, scripts/splitDatabase.sh) — exact, because a (query, reference) result does not depend on
// what else is in the index.
struct RecordPiece { const uint32_t *rec; size_t n; };   // n 12-byte records on the device
struct IndexChunk {
  uint32_t n = 0;
  int32_t c0 = 0, nContigs = 0, g0 = 0, nGenomes = 0;   // global ids of the first contig / genome, and counts
  uint64_t nUnique = 0;
  // The index arrays proper (everything but the small reducer tables at the end) can be dropped and rebuilt from the records
  // (`pieces`, slices of the record parts the sketch keeps) when the reference set is larger than the device memory: see ensure_chunk.
  bool resident = false, everBuilt = false;
  uint64_t lastUse = 0;
  std::vector<RecordPiece> pieces;
  // device arrays
  uint32_t *mHash = nullptr; int32_t *mSeq = nullptr, *mWpos = nullptr;
  uint32_t *sHash = nullptr, *mWin = nullptr;
  uint32_t *dupBits = nullptr; uint64_t *dupList = nullptr; uint32_t nDup = 0;   // same-hash links of near duplicates (index.hpp: DupLinks)
  uint64_t *sSW = nullptr;
  ani::TableSlot *table = nullptr; uint32_t tableSlots = 0;     // order-preserving probe table over the distinct hashes (index.hpp)
  int32_t *contigFirstMin = nullptr, *contigGenome = nullptr;
  uint32_t *contigBinBase = nullptr, *genomeBinStart = nullptr, *posBase = nullptr, *posSample = nullptr;
  uint32_t totalBins = 0, totalPosBins = 0;
  int32_t maxContigLen = 0;
};

struct ani_sketch {
  ani_ctx *ctx = nullptr;
  int device = 0;
  ani_params_t params;
  uint64_t n = 0;                          // minimizers over all chunks
  int32_t nContigs = 0, nGenomes = 0;
  uint64_t totalLen = 0;
  uint64_t nUnique = 0; bool uniqueExact = false;   // distinct hashes over all chunks (computed on demand when there are several)
  std::vector<int32_t> contigLen, genomeContigStart;
  std::vector<std::string> genomeNames;    // optional (ani_sketch_save / _load carry them)
  std::vector<IndexChunk *> chunks;
  uint32_t maxChunkBins = 0;
  // Streaming mode (reference sets whose index does not fit the device: BASELINE configs[4]; the reference's answer is the same
  // split-and-loop, computeCoreIdentity.hpp:457-487 / scripts/splitDatabase.sh): the sketch keeps the 12-byte records (`kept`,
  // a quarter of the index's size) and at most `maxResident` chunks' index arrays; the others are rebuilt on demand (ensure_chunk).
  bool streaming = false; int32_t maxResident = 0; uint64_t useClock = 0;
  std::vector<void *> kept;                // record buffers owned by the sketch (streaming mode)
  std::vector<uint64_t> genomeRecStart;    // first record of every genome in the position-ordered stream (nGenomes + 1)
  // LUTs
  ani::stat::Luts *luts = nullptr;        // host LUTs, shared by every sketch of the context with the same (k, identity cutoff)
  int32_t *dMinHits = nullptr, *dMinShared = nullptr; uint32_t *dIdLUT = nullptr; int dLutMaxS = 0;
};

namespace {

enum { CNT_POOL = 0, CNT_QPOOL = 1, CNT_CAND = 2, CNT_HITS = 3, CNT_ENTRIES = 4, CNT_STEPS = 5, CNT_ROWS = 6, CNT_UNIQ = 7,
       CNT_MAXS = 8 /* int */, CNT_NEG = 9 /* uint */, CNT_SUMQ = 10, CNT_REASON = 11 /* ..14 */, CNT_CLASSB = 15, CNT_LISTM = 16, CNT_LISTL = 17, CNT_LISTBIG = 18, CNT_ENTRIES_B = 19, CNT_SUMQ_B = 20, CNT_STEPS_B = 21, CNT_TINY = 22, CNT_SMALL = 23, CNT_N = 24 };

unsigned long long *cnt_ptr(ani_ctx *c, int i) { return c->dCounters.as<unsigned long long>() + i; }

static_assert(CNT_N == ani::kStatStripeWords, "counter block size");
// Device counter block: kStatStripes copies of the CNT_N counters (statistics are spread over the stripes by the kernels,
// stat_slot(), and summed here; list cursors live in stripe 0), then the striped pool cursors (common.hpp: pool_take): three pools
// (reference minimizers, fragment-sketch hashes, L1 candidates) x kPoolStripes cursors, one per 128-byte line.
enum { POOL_REF = 0, POOL_Q = 1, POOL_CAND = 2, POOL_N = 3 };
constexpr size_t kCursorWords = (size_t)POOL_N * ani::kPoolStripes * ani::kPoolStripeWords;
constexpr size_t kCounterWords = (size_t)ani::kStatStripes * CNT_N + kCursorWords;
unsigned long long *cur_ptr(ani_ctx *c, int pool) { return c->dCounters.as<unsigned long long>() + (size_t)ani::kStatStripes * CNT_N + (size_t)pool * ani::kPoolStripes * ani::kPoolStripeWords; }
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
// host[]: the summed statistics (CNT_MAXS: the maximum over the stripes); c->poolUsed / c->poolMaxStripe: per pool, the entries
// requested in total and by the fullest stripe (the launch overflowed iff that exceeds the stripe capacity it was given)
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
    for (int r = 0; r < ani::kPoolStripes; r++) { const unsigned long long v = cur[(size_t)r * ani::kPoolStripeWords]; c->poolUsed[p] += v; c->poolMaxStripe[p] = std::max(c->poolMaxStripe[p], v); }
  }
  return ANI_OK;
}
// entries per stripe for a pool meant to hold `cap` entries in all; and the capacity to retry with after an overflow
inline uint32_t stripe_cap(uint64_t cap) { return (uint32_t)((cap + ani::kPoolStripes - 1) / ani::kPoolStripes); }
inline uint64_t grown_cap(unsigned long long maxStripe) { return (uint64_t)(maxStripe + maxStripe / 16 + 64) * ani::kPoolStripes; }

// HIP-event bracket on the launch stream; `slot` selects an event pair so that timers can nest
// page-locked host staging buffer `slot`, at least `bytes` long (device->host copies into pageable memory crawl)
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

// HIP-event stage timer.  Both events are recorded on the stream the timed launches go to; the elapsed time is added to *acc when
// the timers are flushed (at a point where the host waits for the device anyway), so timing never serialises host and device.
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
struct StageTimer {
  ani_ctx *c; double *acc; size_t a; hipStream_t st;
  StageTimer(ani_ctx *c_, double *acc_, int /*slot*/ = 0, hipStream_t stream = nullptr) : c(c_), acc(acc_), st(stream ? stream : c_->stream)
  {
    a = timer_event(c);
    (void)hipEventRecord(c->timerEvents[a], st);
  }
  ~StageTimer()
  {
    const size_t b = timer_event(c);
    (void)hipEventRecord(c->timerEvents[b], st);
    c->timerPending.push_back({a, b, acc});
  }
};

// device-wide exclusive scan of int32 counts (n < 2^31, every count >= 0) -> uint32 offsets; the total comes back through *total.
// One device level of 2048-element blocks; the block totals (n / 2048 of them: 1024 for an L2 chunk, ~800 for a sub-batch's
// fragment table) are scanned on the host in 64 bits, so a total beyond 2^32 is detected exactly and reported as ANI_ERR_LIMIT
// (the L2 chunk loop halves its chunk on that) instead of wrapping.
int device_scan(ani_ctx *c, const int32_t *in, uint32_t *out, uint32_t n, uint64_t *total, uint64_t limit = 0xfffffff0ull)
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

// -----------------------------------------------------------------------------------------------------
// A batch of genomes resident on the device.
// -----------------------------------------------------------------------------------------------------
struct DeviceBatch {
  int32_t nGenomes = 0, nContigs = 0;
  std::vector<int32_t> genomeContigStart, contigLen;
  std::vector<int64_t> contigOff;      // words (packed) or bytes (ascii), per contig, into its own buffer
  std::vector<uint8_t> contigPacked;
  const uint32_t *dPacked = nullptr; const uint8_t *dAscii = nullptr;
  const int64_t *dContigOff = nullptr; const int32_t *dContigLen = nullptr; const uint8_t *dContigMode = nullptr;
  uint64_t totalBases = 0;
};

int check_batch(const ani_seq_batch_t *b)
{
  if (!b || b->nGenomes < 0 || b->nContigs < 0 || (b->nContigs && (!b->genomeContigStart || !b->contigLen)))
    return fail(ANI_ERR_ARG, "invalid sequence batch");
  if (b->layout == ANI_SEQ_DEVICE_BATCH) return b->data ? ANI_OK : fail(ANI_ERR_ARG, "sequence batch without its device batch handle");
  if (b->nContigs && !b->data) return fail(ANI_ERR_ARG, "sequence batch without data");
  if (b->layout != ANI_SEQ_HOST_ASCII && b->layout != ANI_SEQ_DEVICE_PACKED2 && b->layout != ANI_SEQ_HOST_ASCII_PTRS) return fail(ANI_ERR_ARG, "unknown sequence layout %d", b->layout);
  if (b->layout != ANI_SEQ_HOST_ASCII_PTRS && b->nContigs && !b->contigOffset) return fail(ANI_ERR_ARG, "sequence batch without contig offsets");
  for (int32_t c = 0; c < b->nContigs; c++) if (b->contigLen[c] < 0) return fail(ANI_ERR_LIMIT, "contig %d has a negative length (>= 2^31 bases?)", c);
  return ANI_OK;
}

// 8 ASCII bases -> 16 bits of 2-bit codes (A0 C1 G2 T3, either case) + "all eight are A/C/G/T", without a table: bits 1..2 of
// the byte give A0 C1 G3 T2, x ^ (x >> 1) swaps the last two; the byte is then rebuilt from its code and compared.
inline uint32_t pack8(uint64_t x, bool *pure)
{
  const uint64_t k01 = 0x0101010101010101ull;
  uint64_t c = (x >> 1) & (3 * k01);
  c ^= (c >> 1) & k01;
  const uint64_t lo = c & k01, hi = (c >> 1) & k01, both = lo & hi;
  // code -> upper-case letter: 'A' + {0, 2, 6, 19}
  const uint64_t rec = 0x41 * k01 + (lo << 1) + (hi << 1) + (hi << 2) + both + (both << 1) + (both << 3);
  *pure = ((x & ~(0x20 * k01)) == rec);
  c = (c | (c >> 6)) & 0x000F000F000F000Full;
  c = (c | (c >> 12)) & 0x000000FF000000FFull;
  c = (c | (c >> 24)) & 0xFFFFull;
  return (uint32_t)c;
}

// A batch of genomes in device memory that outlives the call that uploaded it (ani_batch_upload): the all-vs-all command line
// sketches it as references and maps it as queries without reading or packing the files a second time.
}  // namespace
struct ani_dev_batch {
  ani_ctx *ctx = nullptr; int device = 0;
  DeviceBatch db;
  void *bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // packed, ascii, contigOff, contigLen, contigMode
};
struct ani_fragset;
namespace {

// genomes [g0, g1) of `b` -> device (packs pure-ACGT contigs to 2 bits per base on the way).  `keep` = allocate the device
// arrays for a persistent ani_dev_batch instead of using the context's staging buffers.
int upload_batch(ani_ctx *ctx, const ani_seq_batch_t *b, int32_t g0, int32_t g1, DeviceBatch *out, ani_dev_batch *keep = nullptr)
{
  if (b->layout == ANI_SEQ_DEVICE_BATCH) {       // a view of genomes [g0, g1) of an uploaded batch
    const ani_dev_batch *src = (const ani_dev_batch *)b->data;
    if (src->device != ctx->device) return fail(ANI_ERR_ARG, "device batch lives on device %d, context on %d", src->device, ctx->device);
    if (g1 > src->db.nGenomes) return fail(ANI_ERR_ARG, "genome range beyond the device batch");
    const DeviceBatch &sb = src->db;
    const int32_t c0 = sb.genomeContigStart[g0], c1 = sb.genomeContigStart[g1];
    out->nGenomes = g1 - g0; out->nContigs = c1 - c0;
    out->genomeContigStart.resize(out->nGenomes + 1);
    for (int32_t g = g0; g <= g1; g++) out->genomeContigStart[g - g0] = sb.genomeContigStart[g] - c0;
    out->contigLen.assign(sb.contigLen.begin() + c0, sb.contigLen.begin() + c1);
    out->totalBases = 0;
    for (int32_t c = 0; c < out->nContigs; c++) out->totalBases += (uint64_t)out->contigLen[c];
    out->dPacked = sb.dPacked; out->dAscii = sb.dAscii;
    out->dContigOff = sb.dContigOff + c0; out->dContigLen = sb.dContigLen + c0; out->dContigMode = sb.dContigMode + c0;
    return ANI_OK;
  }
  const int32_t c0 = b->genomeContigStart[g0], c1 = b->genomeContigStart[g1];
  out->nGenomes = g1 - g0; out->nContigs = c1 - c0;
  out->genomeContigStart.resize(out->nGenomes + 1);
  for (int32_t g = g0; g <= g1; g++) out->genomeContigStart[g - g0] = b->genomeContigStart[g] - c0;
  out->contigLen.assign(b->contigLen + c0, b->contigLen + c1);
  out->contigOff.resize(out->nContigs); out->contigPacked.resize(out->nContigs);
  out->totalBases = 0;
  for (int32_t c = 0; c < out->nContigs; c++) out->totalBases += (uint64_t)out->contigLen[c];
  const size_t nc = (size_t)out->nContigs;
  void *dPacked = nullptr, *dAscii = nullptr;

  if (b->layout == ANI_SEQ_DEVICE_PACKED2) {
    for (int32_t c = 0; c < out->nContigs; c++) { out->contigOff[c] = b->contigOffset[c0 + c]; out->contigPacked[c] = 1; }
    dPacked = const_cast<void *>(b->data);
  } else {
    // Ingest: classify every contig (pure ACGT -> 2 bits per base, anything else -> raw bytes) and pack, on host threads, straight
    // into page-locked staging (contigs are split into segments of 4 Mbases so that one long chromosome still spreads over the
    // threads).  Segments are packed optimistically; a contig with any other byte is copied raw in a second, rare, pass.
    const uint8_t *flat = b->layout == ANI_SEQ_HOST_ASCII ? (const uint8_t *)b->data : nullptr;
    const uint8_t *const *ptrs = b->layout == ANI_SEQ_HOST_ASCII_PTRS ? (const uint8_t *const *)b->data : nullptr;
    auto contig_ptr = [&](int32_t c) -> const uint8_t * { return flat ? flat + b->contigOffset[c0 + c] : ptrs[c0 + c]; };
    const int32_t nC = out->nContigs;
    struct Seg { int32_t c; int32_t lo, hi; };
    std::vector<Seg> segs;
    const int32_t kSeg = 1 << 22;                       // multiple of 16: segments pack whole words
    size_t nWordsAll = 0;
    std::vector<size_t> wordOff((size_t)nC + 1, 0);
    for (int32_t c = 0; c < nC; c++) {
      wordOff[c] = nWordsAll; nWordsAll += ((size_t)out->contigLen[c] + 15) / 16 + 2;                          // +2 words of slack for the 3-word fetch
      for (int32_t lo = 0; lo < out->contigLen[c]; lo += kSeg) segs.push_back(Seg{c, lo, std::min(out->contigLen[c], lo + kSeg)});
    }
    wordOff[nC] = nWordsAll;
    uint32_t *hPacked = nullptr;
    TRY(pinned_buffer(ctx, 3, nWordsAll * 4 + 64, (void **)&hPacked));
    std::vector<uint8_t> segImpure(segs.size(), 0);
    parallel_for(segs.size(), out->totalBases, [&](size_t i) {
      const Seg &sg = segs[i];
      const uint8_t *sp = contig_ptr(sg.c);
      uint32_t *dst = hPacked + wordOff[sg.c];
      bool allPure = true;
      int32_t x = sg.lo;
      for (; x + 16 <= sg.hi; x += 16) {
        uint64_t a, bb; memcpy(&a, sp + x, 8); memcpy(&bb, sp + x + 8, 8);
        bool p1, p2;
        const uint32_t wd = pack8(a, &p1) | (pack8(bb, &p2) << 16);
        allPure &= p1 & p2;
        dst[x >> 4] = wd;
      }
      if (x < sg.hi) {                                  // last, partial word of the contig
        uint8_t tail[16]; memset(tail, 'A', 16); memcpy(tail, sp + x, (size_t)(sg.hi - x));
        uint64_t a, bb; memcpy(&a, tail, 8); memcpy(&bb, tail + 8, 8);
        bool p1, p2;
        uint32_t wd = pack8(a, &p1) | (pack8(bb, &p2) << 16);
        allPure &= p1 & p2;
        const int nb = sg.hi - x;
        if (nb < 16) wd &= (1u << (2 * nb)) - 1u;
        dst[x >> 4] = wd;
      }
      if (sg.hi == out->contigLen[sg.c]) { const size_t e = ((size_t)sg.hi + 15) / 16; dst[e] = 0; dst[e + 1] = 0; }
      segImpure[i] = !allPure;
    });
    for (int32_t c = 0; c < nC; c++) if (out->contigLen[c] == 0) { hPacked[wordOff[c]] = 0; hPacked[wordOff[c] + 1] = 0; }
    std::vector<uint8_t> impure(nC, 0);
    for (size_t i = 0; i < segs.size(); i++) impure[segs[i].c] |= segImpure[i];
    size_t nBytes = 0;
    for (int32_t c = 0; c < nC; c++) {
      out->contigPacked[c] = !impure[c];
      if (!impure[c]) out->contigOff[c] = (int64_t)wordOff[c];
      else { out->contigOff[c] = (int64_t)nBytes; nBytes += ((size_t)out->contigLen[c] + 3) & ~(size_t)3; }
    }
    uint8_t *hAscii = nullptr;
    if (nBytes) {
      TRY(pinned_buffer(ctx, 4, nBytes + 64, (void **)&hAscii));
      for (int32_t c = 0; c < nC; c++) if (impure[c]) memcpy(hAscii + out->contigOff[c], contig_ptr(c), (size_t)out->contigLen[c]);
    }
    if (keep) {
      HIP_TRY(pool_malloc(&keep->bufs[0], nWordsAll * 4 + 64)); HIP_TRY(pool_malloc(&keep->bufs[1], nBytes + 64));
      dPacked = keep->bufs[0]; dAscii = keep->bufs[1];
    } else {
      TRY(ctx->seqPacked.ensure(nWordsAll * 4 + 64)); TRY(ctx->seqAscii.ensure(nBytes + 64));
      dPacked = ctx->seqPacked.p; dAscii = ctx->seqAscii.p;
    }
    if (nWordsAll) HIP_TRY(hipMemcpyAsync(dPacked, hPacked, nWordsAll * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nBytes) HIP_TRY(hipMemcpyAsync(dAscii, hAscii, nBytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));    // the staging buffers are reused by the next upload
  }
  out->dPacked = (const uint32_t *)dPacked; out->dAscii = (const uint8_t *)dAscii;
  void *dOff = nullptr, *dLen = nullptr, *dMode = nullptr;
  if (keep) {
    HIP_TRY(pool_malloc(&keep->bufs[2], nc * 8 + 8)); HIP_TRY(pool_malloc(&keep->bufs[3], nc * 4 + 4)); HIP_TRY(pool_malloc(&keep->bufs[4], nc + 4));
    dOff = keep->bufs[2]; dLen = keep->bufs[3]; dMode = keep->bufs[4];
  } else {
    TRY(ctx->contigOff.ensure(nc * 8 + 8)); TRY(ctx->contigLen.ensure(nc * 4 + 4)); TRY(ctx->contigMode.ensure(nc + 4));
    dOff = ctx->contigOff.p; dLen = ctx->contigLen.p; dMode = ctx->contigMode.p;
  }
  if (nc) {
    HIP_TRY(hipMemcpyAsync(dOff, out->contigOff.data(), nc * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dLen, out->contigLen.data(), nc * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dMode, out->contigPacked.data(), nc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  out->dContigOff = (const int64_t *)dOff; out->dContigLen = (const int32_t *)dLen; out->dContigMode = (const uint8_t *)dMode;
  return ANI_OK;
}

int check_params(const ani_params_t *p)
{
  if (!p) return fail(ANI_ERR_ARG, "null parameters");
  if (p->kmerSize < 1 || p->kmerSize > 16) return fail(ANI_ERR_ARG, "kmerSize must be in 1..16 (hash_t is 32 bit, parseCmdArgs.hpp:142)");
  if (p->windowSize < 1 || p->windowSize > ani::kTile - ani::kTPB) return fail(ANI_ERR_LIMIT, "windowSize %d outside 1..%d", p->windowSize, ani::kTile - ani::kTPB);
  if (p->fragLen < 1 || p->fragLen <= 20) return fail(ANI_ERR_ARG, "fragLen must exceed 20 (bin width is fragLen-20, computeCoreIdentity.hpp:194)");
  return ANI_OK;
}

}  // namespace

namespace {
using namespace ani;

// -----------------------------------------------------------------------------------------------------
// reference minimizer records of a device-resident batch (position order, global seqIds)
// -----------------------------------------------------------------------------------------------------
// fragment table + fragment sketches of one device-resident query sub-batch; the arrays live in the context's buffers
// (frags, fragOff, fragS, fragGenome, fragQSeq, qPool) and stay valid across the index chunks the sub-batch is mapped against
struct FragSet {
  int32_t nFrag = 0, maxS = 0;
  uint64_t nHashes = 0;              // sketch hashes of these fragments
  uint64_t poolSize = 0;             // length of the pool fragOff points into (>= nHashes: a slice of a kept set points into the whole pool)
  std::vector<int32_t> genomeFragments;
  // device arrays (the context's buffers after fragment_stage, or a slice of a kept ani_fragset)
  const uint32_t *qPool = nullptr, *fragOff = nullptr; const int32_t *fragS = nullptr, *fragGenome = nullptr, *fragQSeq = nullptr;
  int32_t genomeBase = 0;            // fragGenome values are relative to the set the slice was cut from: query index = fragGenome - genomeBase
};

// device arrays of a fragment-sketch set: the context's buffers (fragment_stage) or arrays of their own (ani_fragset)
struct FragArrays { uint32_t *fragOff = nullptr; int32_t *fragS = nullptr, *fragGenome = nullptr, *fragQSeq = nullptr; };

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
  hipLaunchKernelGGL(k_expand_frags, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(), (const int32_t *)ctx->unitAux.as<int32_t>(),
                     (const int32_t *)(ctx->unitAux.as<int32_t>() + nc1), db.nContigs, (uint32_t)nF, L, ctx->frags.as<FragDesc>(), arr.fragGenome, arr.fragQSeq);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  qr->fragOff = arr.fragOff; qr->fragS = arr.fragS; qr->fragGenome = arr.fragGenome; qr->fragQSeq = arr.fragQSeq; qr->genomeBase = 0;
  return ANI_OK;
}

// `fused` (optional): the batch's genomes are queries as well — their fragment sketches come out of the same pass over the
// hashes (k_sketch_fused) into arrays owned by the caller's ani_fragset.  Only when a fragment plus its w-1 lead-in fits one tile.
struct FusedOut { FragSet *fs; FragArrays *arr; uint32_t **qPool; };
inline bool fusable(const ani_params_t *p) { return p->fragLen + p->windowSize - 1 <= kTile && p->fragLen >= p->windowSize && p->fragLen >= p->kmerSize; }

// The fragment sketches of a batch as they come out of the sketch kernels lie in 64 partly filled stripes of the pool (pool_take).
// A set that is kept is packed: sketches back to back in fragment order, exactly nHashes entries, fragOff rewritten in place.
int compact_fragment_pool(ani_ctx *ctx, const uint32_t *pool, uint32_t *fragOff /* in: striped offsets, out: packed offsets */, const int32_t *fragS, size_t nF, uint64_t nHashes, uint32_t **out)
{
  *out = nullptr;
  HIP_TRY(pool_malloc((void **)out, (nHashes ? nHashes : 1) * 4));
  if (nF == 0) return ANI_OK;
  TRY(ctx->scanTmpC.ensure(nF * 4)); TRY(ctx->scanTmpD.ensure(nF * 4));
  hipLaunchKernelGGL(k_clamp_counts, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, (int32_t)nF, fragS, (const int32_t *)nullptr, ctx->scanTmpC.as<int32_t>());   // s = -1 marks an overflowed fragment
  uint64_t total = 0;
  int rc = device_scan(ctx, ctx->scanTmpC.as<int32_t>(), ctx->scanTmpD.as<uint32_t>(), (uint32_t)nF, &total);
  if (rc == ANI_OK && total != nHashes) rc = fail(ANI_ERR_INTERNAL, "fragment sketch pool: %llu hashes counted, %llu in the sketches", (unsigned long long)nHashes, (unsigned long long)total);
  if (rc != ANI_OK) { pool_free(*out); *out = nullptr; return rc; }
  hipLaunchKernelGGL(k_pack_fragment_pool, dim3(grid_for(nF * 64)), dim3(256), 0, ctx->stream, pool, fragOff, (const int32_t *)ctx->scanTmpC.as<int32_t>(), (const uint32_t *)ctx->scanTmpD.as<uint32_t>(), (uint32_t)nF, *out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { pool_free(*out); *out = nullptr; HIP_TRY(e); }
  return ANI_OK;
}

int sketch_records(ani_ctx *ctx, const ani_params_t *p, const DeviceBatch &db, int32_t seqIdBase, uint32_t **dRecords, size_t *nOut, FusedOut *fused = nullptr)
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
    hipLaunchKernelGGL(k_expand_tiles, dim3(grid_for(nT)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(), db.nContigs, (uint32_t)nT, stride,
                       ctx->tiles.as<TileDesc>());
  else {
    TRY(ctx->tileInfo.ensure(nT * sizeof(FusedInfo))); TRY(ctx->unitAux.ensure(fragStart.size() * 4));
    HIP_TRY(hipMemcpyAsync(ctx->unitAux.p, fragStart.data(), fragStart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_expand_fused, dim3(grid_for(nT)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->unitStart.as<uint32_t>(), (const uint32_t *)ctx->unitAux.as<uint32_t>(), db.nContigs,
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
                           ctx->tiles.as<TileDesc>(), ctx->tileInfo.as<FusedInfo>(), k, w, L, ctx->poolHash.as<uint32_t>(), ctx->poolWpos.as<int32_t>(), stripe_cap(cap),
                           cur_ptr(ctx, POOL_REF), ctx->tileMeta.as<TileMeta>(), *fused->qPool, stripe_cap(qcap), cur_ptr(ctx, POOL_Q), fused->arr->fragOff, fused->arr->fragS,
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

// the index arrays of a chunk (not its reducer tables): dropped when the chunk is evicted in streaming mode
void free_chunk_index(IndexChunk *ch)
{
  void **ptrs[] = {(void **)&ch->mWin, (void **)&ch->sSW, (void **)&ch->dupBits, (void **)&ch->dupList, (void **)&ch->mHash, (void **)&ch->mSeq, (void **)&ch->mWpos,
                   (void **)&ch->sHash, (void **)&ch->table, (void **)&ch->contigFirstMin, (void **)&ch->posSample};
  for (void **q : ptrs) if (*q) { pool_free(*q); *q = nullptr; }
  ch->resident = false; ch->nDup = 0;
}
void free_chunk(IndexChunk *ch)
{
  if (!ch) return;
  free_chunk_index(ch);
  void *ptrs[] = {ch->contigGenome, ch->contigBinBase, ch->genomeBinStart, ch->posBase};
  for (void *q : ptrs) if (q) pool_free(q);
  delete ch;
}
void free_sketch_device(ani_sketch *sk)
{
  for (IndexChunk *ch : sk->chunks) free_chunk(ch);
  sk->chunks.clear();
  for (void *q : sk->kept) if (q) pool_free(q);
  sk->kept.clear();
  void *ptrs[] = {sk->dMinHits, sk->dMinShared, sk->dIdLUT};
  for (void *q : ptrs) if (q) pool_free(q);
  sk->dMinHits = sk->dMinShared = nullptr; sk->dIdLUT = nullptr;
}

int upload_luts(ani_sketch *sk, int maxS)
{
  if (maxS <= sk->dLutMaxS) return ANI_OK;
  int target = std::max(maxS, 512);
  if (target > sk->params.fragLen) target = std::max(maxS, std::min(target, sk->params.fragLen));
  // the LUTs depend on (k, cutoff) only and cost milliseconds of host arithmetic: one growing copy per context
  ani_ctx *ctx = sk->ctx;
  sk->luts = nullptr;
  for (auto &l : ctx->lutCache) if (l->k == sk->params.kmerSize && l->identityCutoff == sk->params.percentageIdentity) sk->luts = l.get();
  if (!sk->luts) { ctx->lutCache.emplace_back(new ani::stat::Luts()); sk->luts = ctx->lutCache.back().get(); }
  sk->luts->extend(sk->params.kmerSize, sk->params.percentageIdentity, target);
  if (sk->dMinHits) { pool_free(sk->dMinHits); pool_free(sk->dMinShared); pool_free(sk->dIdLUT); sk->dMinHits = sk->dMinShared = nullptr; sk->dIdLUT = nullptr; }
  const size_t n1 = (size_t)target + 1, n2 = ani::stat::Luts::off(target + 1);
  HIP_TRY(pool_malloc((void **)&sk->dMinHits, n1 * 4)); HIP_TRY(pool_malloc((void **)&sk->dMinShared, n1 * 4)); HIP_TRY(pool_malloc((void **)&sk->dIdLUT, n2 * 4 + 4));
  HIP_TRY(hipMemcpy(sk->dMinHits, sk->luts->minHits.data(), n1 * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(sk->dMinShared, sk->luts->minShared.data(), n1 * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(sk->dIdLUT, sk->luts->idBits.data(), n2 * 4, hipMemcpyHostToDevice));
  sk->dLutMaxS = target;
  return ANI_OK;
}

// -----------------------------------------------------------------------------------------------------
// One index chunk (≙ Sketch::index, winSketch.hpp:181-193) over the genomes [g0, g0 + nGenomes) = contigs [c0, c0 + nContigs) of
// a reference set.  new_chunk fills the small tables that depend on the contig lengths only (reducer bins, sampled-position
// bins); build_chunk_index builds the index arrays from the chunk's records (`pieces`: position order, set-global seqIds) and can
// be repeated after free_chunk_index.
// -----------------------------------------------------------------------------------------------------
int new_chunk(ani_ctx *ctx, const ani_params_t *p, size_t n, const int32_t *contigLenAll, const int32_t *gcsAll, int32_t g0, int32_t nGenomes, IndexChunk **out)
{
  if (n >= 0x7ffffff0ull) return fail(ANI_ERR_LIMIT, "index chunk of %zu minimizers exceeds 2^31", n);
  IndexChunk *sk = new IndexChunk();
  const int32_t c0 = gcsAll[g0], nContigs = gcsAll[g0 + nGenomes] - c0;
  const int32_t *contigLen = contigLenAll + c0;
  sk->n = (uint32_t)n; sk->c0 = c0; sk->nContigs = nContigs; sk->g0 = g0; sk->nGenomes = nGenomes;
  auto bail = [&](int rc) { free_chunk(sk); return rc; };
#define SK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
  // contig -> genome, bins (computeCoreIdentity.hpp:31-42, :194); all chunk-local
  std::vector<int32_t> cg((size_t)nContigs + 1, nGenomes);
  std::vector<uint32_t> binBase((size_t)nContigs + 1), gBin((size_t)nGenomes + 1), posBase((size_t)nContigs + 1);
  const int32_t binW = p->fragLen - 20;
  uint64_t run = 0, runPos = 0;
  for (int32_t g = 0; g < nGenomes; g++) {
    gBin[g] = (uint32_t)run;
    for (int32_t c = gcsAll[g0 + g] - c0; c < gcsAll[g0 + g + 1] - c0; c++) {
      cg[c] = g; binBase[c] = (uint32_t)run; run += (uint64_t)(contigLen[c] / binW) + 1;
      sk->maxContigLen = std::max(sk->maxContigLen, contigLen[c]);
      posBase[c] = (uint32_t)runPos; runPos += ((uint64_t)contigLen[c] >> ani::kPosSampleShift) + 1;      // bins of the sampled position index
      if (run > 0xfffffff0ull || runPos > 0xfffffff0ull) return bail(fail(ANI_ERR_LIMIT, "index chunk has more than 2^32 position bins"));
    }
  }
  gBin[nGenomes] = (uint32_t)run; binBase[nContigs] = (uint32_t)run; posBase[nContigs] = (uint32_t)runPos;
  sk->totalBins = (uint32_t)run; sk->totalPosBins = (uint32_t)runPos;
  SK_HIP(pool_malloc((void **)&sk->contigGenome, ((size_t)nContigs + 1) * 4)); SK_HIP(pool_malloc((void **)&sk->contigBinBase, ((size_t)nContigs + 1) * 4));
  SK_HIP(pool_malloc((void **)&sk->genomeBinStart, ((size_t)nGenomes + 1) * 4)); SK_HIP(pool_malloc((void **)&sk->posBase, ((size_t)nContigs + 1) * 4));
  SK_HIP(hipMemcpyAsync(sk->contigGenome, cg.data(), cg.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  SK_HIP(hipMemcpyAsync(sk->contigBinBase, binBase.data(), binBase.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  SK_HIP(hipMemcpyAsync(sk->genomeBinStart, gBin.data(), gBin.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  SK_HIP(hipMemcpyAsync(sk->posBase, posBase.data(), posBase.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  SK_HIP(hipStreamSynchronize(ctx->stream));      // the host tables above die at scope exit
#undef SK_HIP
  *out = sk;
  return ANI_OK;
}

int build_chunk_index(ani_ctx *ctx, const ani_params_t *p, IndexChunk *sk)
{
  const size_t n = sk->n;
  const int32_t c0 = sk->c0, nContigs = sk->nContigs;
  auto bail = [&](int rc) { free_chunk_index(sk); return rc; };
#define SK_TRY(expr) do { int rc_ = (expr); if (rc_ != ANI_OK) return bail(rc_); } while (0)
#define SK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
  const size_t n4 = (n ? n : 1) * 4;
  const size_t bitWords = (n + 31) / 32 + 1;
  SK_HIP(pool_malloc((void **)&sk->mHash, n4)); SK_HIP(pool_malloc((void **)&sk->mSeq, n4)); SK_HIP(pool_malloc((void **)&sk->mWpos, n4));
  SK_HIP(pool_malloc((void **)&sk->sHash, n4)); SK_HIP(pool_malloc((void **)&sk->sSW, 2 * n4));
  SK_HIP(pool_malloc((void **)&sk->mWin, n4)); SK_HIP(pool_malloc((void **)&sk->dupBits, bitWords * 4));
  SK_HIP(pool_malloc((void **)&sk->contigFirstMin, ((size_t)nContigs + 1) * 4));
  SK_HIP(pool_malloc((void **)&sk->posSample, ((size_t)sk->totalPosBins + 1) * 4));
  {
    StageTimer tm(ctx, &ctx->counters.msIndex);
    const int32_t cmw = p->fragLen - (p->windowSize - 1) - (p->kmerSize - 1);
    const bool winLinks = n && cmw >= 1 && cmw + 2 <= (int32_t)kWinMask;        // otherwise the L2 fast path is off (map_stage) and the window links are never read
    uint32_t *tmpK = nullptr; uint64_t *tmpV = nullptr;
    SK_HIP(pool_malloc((void **)&tmpK, n4));
    { const hipError_t ev = pool_malloc((void **)&tmpV, 2 * n4); if (ev != hipSuccess) { pool_free(tmpK); SK_HIP(ev); } }
    SK_HIP(hipMemsetAsync(sk->dupBits, 0, bitWords * 4, ctx->stream));
    // Main stream: the index sort (radix.hpp) — its histogram read of the records writes the position-ordered SoA arrays, its passes
    // are bound by memory bandwidth.  Side stream, as soon as the SoA arrays exist (evSimA[0]): everything that needs positions only —
    // contig slices, the window links of the L2 event stream (binary searches over LDS-staged positions: latency- and LDS-bound),
    // the sampled position index — runs underneath the sort's passes instead of after them.
    struct SideJoin { ani_ctx *c; ~SideJoin() { (void)hipStreamSynchronize(c->stream2); } } sideJoin{ctx};   // also on the error paths below
    bool sorting = false;
    if (n) {
      std::vector<const void *> recs; std::vector<size_t> cnts;
      for (const RecordPiece &pc : sk->pieces) if (pc.n) { recs.push_back(pc.rec); cnts.push_back(pc.n); }
      size_t tb = 0;
      int rc = ani_sort_index(recs.data(), cnts.data(), (int)recs.size(), (uint32_t)c0, n, sk->mHash, sk->mSeq, sk->mWpos, tmpK, tmpV, sk->sHash, sk->sSW, nullptr, &tb, ctx->stream, nullptr, nullptr);
      if (rc == 0) { rc = ctx->sortTmp.ensure(tb + 256); if (rc == ANI_OK) rc = ani_sort_index(recs.data(), cnts.data(), (int)recs.size(), (uint32_t)c0, n, sk->mHash, sk->mSeq, sk->mWpos, tmpK, tmpV, sk->sHash, sk->sSW, ctx->sortTmp.p, &tb, ctx->stream, ctx->evSimA[0], ctx->stream2); }
      if (rc != 0) { pool_free(tmpK); pool_free(tmpV); return bail(rc < 0 && rc >= ANI_ERR_INTERNAL ? rc : fail(rc > 9000 ? ANI_ERR_INTERNAL : ANI_ERR_DEVICE, "radix sort of the index failed (%d)", rc)); }
      sorting = true;                               // the passes are in flight; stream2 waits for the SoA arrays
    }
    hipLaunchKernelGGL(k_index_contig_first, dim3(grid_for((size_t)nContigs + 1, 256, 65535)), dim3(256), 0, ctx->stream2, sk->mSeq, (uint32_t)n, nContigs, sk->contigFirstMin);
    if (winLinks)
      hipLaunchKernelGGL(k_index_window_links, dim3((unsigned)((n + kWinBlock - 1) / kWinBlock)), dim3(256), 0, ctx->stream2, (const int32_t *)sk->mSeq, (const int32_t *)sk->mWpos,
                         (const int32_t *)sk->contigFirstMin, (uint32_t)n, cmw - 1, (int32_t)((2 * (int64_t)cmw) / (p->windowSize + 1)), sk->mWin);
    if (nContigs) hipLaunchKernelGGL(k_index_pos_sample, dim3(grid_for((size_t)sk->totalPosBins + 1, 256, 65535)), dim3(256), 0, ctx->stream2, sk->mWpos, sk->contigFirstMin, sk->posBase, nContigs,
                                    sk->totalPosBins, (uint32_t)n, sk->posSample);
    if (sorting) {
      const int rc = ani_sort_check(ctx->sortTmp.p, ctx->stream);
      if (rc != 0) { pool_free(tmpK); pool_free(tmpV); return bail(fail(rc > 9000 ? ANI_ERR_INTERNAL : ANI_ERR_DEVICE, "radix sort of the index failed (%d)", rc)); }
    }
    pool_free(tmpK); pool_free(tmpV);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(ctx->evSimA[1], ctx->stream2));
    SK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evSimA[1], 0));            // the links kernel needs the contig slices and ORs into the window links
    // same-hash links of near duplicates (index.hpp: DupLinks) and the number of distinct hashes
    unsigned long long host[CNT_N];
    host[CNT_UNIQ] = 0;
    if (n) {
      uint32_t pairCap = (uint32_t)std::min<uint64_t>(n, ctx->dupPairCap ? ctx->dupPairCap : n / 64 + 4096);     // first guess; a repetitive reference reruns with the exact count
      for (int attempt = 0;; attempt++) {
        uint64_t *pairs = nullptr;
        SK_HIP(pool_malloc((void **)&pairs, (size_t)pairCap * 16));
        { const int rz = zero_counters(ctx); if (rz != ANI_OK) { pool_free(pairs); return bail(rz); } }
        hipLaunchKernelGGL(k_index_links, dim3(grid_for(n, 256, 8192)), dim3(256), 0, ctx->stream, sk->sHash, (const uint64_t *)sk->sSW, (uint32_t)n, sk->mWpos, sk->contigFirstMin,
                           cmw, pairs, pairCap, (unsigned int *)cnt_ptr(ctx, CNT_NEG), sk->dupBits, winLinks ? sk->mWin : (uint32_t *)nullptr, cnt_ptr(ctx, CNT_UNIQ));
        { const int rr = read_counters(ctx, host); if (rr != ANI_OK) { pool_free(pairs); return bail(rr); } }
        const uint64_t nPairs = (uint32_t)host[CNT_NEG];
        if (nPairs > pairCap) {                                  // a repetitive reference: again with room for every pair (bits and flags are idempotent)
          pool_free(pairs);
          if (attempt > 0) return bail(fail(ANI_ERR_INTERNAL, "same-hash links did not converge"));
          pairCap = (uint32_t)nPairs;
          continue;
        }
        sk->nDup = (uint32_t)(2 * nPairs);
        if (nPairs) {
          // the half-records sorted by (entry, kind): bisection finds an entry's links (keys are unique: every key bit is significant up to the entry's)
          int keyBits = 33; while (keyBits < 64 && (1ull << (keyBits - 32)) <= (uint64_t)n) keyBits++;
          SK_HIP(pool_malloc((void **)&sk->dupList, (size_t)sk->nDup * 8));
          size_t tb = 0;
          int rc = ani_sort_keys_u64_bits(pairs, sk->dupList, sk->nDup, keyBits, nullptr, &tb, ctx->stream);
          if (rc == 0) { rc = ctx->sortTmp.ensure(tb + 256); if (rc == ANI_OK) rc = ani_sort_keys_u64_bits(pairs, sk->dupList, sk->nDup, keyBits, ctx->sortTmp.p, &tb, ctx->stream); }
          if (rc != 0) { pool_free(pairs); return bail(rc < 0 && rc >= ANI_ERR_INTERNAL ? rc : fail(ANI_ERR_DEVICE, "radix sort of the same-hash links failed (%d)", rc)); }
        }
        pool_free(pairs);
        break;
      }
    }
    // probe table: the distinct hashes in an order-preserving open-addressing table, load 0.5 (index.hpp)
    {
      sk->nUnique = host[CNT_UNIQ];                           // k_index_links counted the distinct hashes
      const uint32_t nSlots = (uint32_t)std::min<uint64_t>(0x7ffffff0ull, std::max<uint64_t>(1024, (uint64_t)sk->nUnique * 2));
      const uint32_t nb = (uint32_t)((n + kTableBlock - 1) / kTableBlock);
      std::vector<int32_t> cnt(nb ? nb : 1), best(nb ? nb : 1);
      int64_t lastP = -1;
      if (n) {
        SK_TRY(ctx->scanTmpA.ensure((size_t)nb * 4)); SK_TRY(ctx->scanTmpB.ensure((size_t)nb * 4));
        hipLaunchKernelGGL(k_table_block_totals, dim3(nb), dim3(kTPB), 0, ctx->stream, (const uint32_t *)sk->sHash, (uint32_t)n, p->windowSize, nSlots,
                           ctx->scanTmpA.as<int32_t>(), ctx->scanTmpB.as<int32_t>());
        SK_HIP(hipMemcpyAsync(cnt.data(), ctx->scanTmpA.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        SK_HIP(hipMemcpyAsync(best.data(), ctx->scanTmpB.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        SK_HIP(hipStreamSynchronize(ctx->stream));
        int64_t before = 0, run = INT32_MIN;                              // distinct hashes before the block; max(slot - global index) over them
        for (uint32_t b = 0; b < nb; b++) {
          const int32_t c = cnt[b], bb = best[b];
          cnt[b] = (int32_t)before; best[b] = (int32_t)std::max<int64_t>(run, INT32_MIN);
          if (c > 0) { run = std::max<int64_t>(run, (int64_t)bb - before); lastP = before + c - 1 + run; }
          before += c;
        }
      }
      const uint64_t alloc = std::max<uint64_t>(nSlots, (uint64_t)(lastP + 1)) + 2;        // the clusters at the end may run past nSlots
      if (alloc > 0x7ffffff0ull) return bail(fail(ANI_ERR_LIMIT, "probe table of %llu slots", (unsigned long long)alloc));
      SK_HIP(pool_malloc((void **)&sk->table, alloc * sizeof(TableSlot)));
      // (measured, round 4: writing the empty slots from k_table_scatter instead — every entry with the gap in front of it — made that
      //  kernel 1.3 ms slower per 4 x 10^8 minimizers and saved 0.5 ms of fill: the device fills 9.6 GB in 0.65 ms)
      SK_HIP(hipMemsetAsync(sk->table, 0xff, alloc * sizeof(TableSlot), ctx->stream));
      if (n) {
        SK_HIP(hipMemcpyAsync(ctx->scanTmpA.p, cnt.data(), (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
        SK_HIP(hipMemcpyAsync(ctx->scanTmpB.p, best.data(), (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_table_scatter, dim3(nb), dim3(kTPB), 0, ctx->stream, (const uint32_t *)sk->sHash, (uint32_t)n, p->windowSize, nSlots,
                           (const int32_t *)ctx->scanTmpA.as<int32_t>(), (const int32_t *)ctx->scanTmpB.as<int32_t>(), sk->table);
      }
      const TableSlot sentinel{0xffffffffu, (uint32_t)n | 0x80000000u, 0u};
      SK_HIP(hipMemcpyAsync(sk->table + (alloc - 2), &sentinel, sizeof sentinel, hipMemcpyHostToDevice, ctx->stream));
      SK_HIP(hipStreamSynchronize(ctx->stream));                             // cnt / best / sentinel are host memory
      sk->tableSlots = nSlots;
    }
    SK_HIP(hipGetLastError());
  }
  SK_HIP(hipStreamSynchronize(ctx->stream));
  if (!sk->everBuilt) ctx->counters.refMinimizers += n;
  ctx->counters.indexChunkBuilds++;
  sk->resident = true; sk->everBuilt = true;
#undef SK_TRY
#undef SK_HIP
  return ANI_OK;
}

// A reference set = its contig/genome tables + the LUTs; index chunks are attached by add_chunks.
ani_sketch *new_sketch(ani_ctx *ctx, const ani_params_t *p, const int32_t *contigLen, int32_t nContigs, const int32_t *genomeContigStart, int32_t nGenomes)
{
  ani_sketch *sk = new ani_sketch();
  sk->ctx = ctx; sk->device = ctx->device; sk->params = *p; sk->nContigs = nContigs; sk->nGenomes = nGenomes;
  sk->contigLen.assign(contigLen, contigLen + nContigs);
  sk->genomeContigStart.assign(genomeContigStart, genomeContigStart + nGenomes + 1);
  for (int32_t c = 0; c < nContigs; c++) sk->totalLen += (uint64_t)contigLen[c];
  return sk;
}

// Streaming mode: make chunk i's index arrays resident, evicting the least recently used ones beyond the set's limit (`pin`, if
// >= 0, is never evicted and raises the limit to two: exact_unique compares pairs of chunks).
int ensure_chunk(ani_sketch *sk, size_t i, int pin = -1)
{
  IndexChunk *ch = sk->chunks[i];
  ch->lastUse = ++sk->useClock;
  if (ch->resident) return ANI_OK;
  auto evict_one = [&]() -> bool {
    IndexChunk *victim = nullptr;
    for (size_t x = 0; x < sk->chunks.size(); x++) {
      IndexChunk *c = sk->chunks[x];
      if (!c->resident || x == i || (int)x == pin) continue;
      if (!victim || c->lastUse < victim->lastUse) victim = c;
    }
    if (!victim) return false;
    (void)hipStreamSynchronize(sk->ctx->stream); (void)hipStreamSynchronize(sk->ctx->stream2);   // nothing in flight reads the arrays that go back to the pool
    free_chunk_index(victim);
    return true;
  };
  const int limit = std::max<int>(sk->maxResident > 0 ? sk->maxResident : (int)sk->chunks.size(), pin >= 0 ? 2 : 1);
  for (;;) {
    int res = 0;
    for (IndexChunk *c : sk->chunks) res += c->resident;
    if (res < limit || !evict_one()) break;
  }
  int rc = build_chunk_index(sk->ctx, &sk->params, ch);
  while (rc == ANI_ERR_NOMEM && evict_one()) rc = build_chunk_index(sk->ctx, &sk->params, ch);     // estimate too optimistic: make room, try again
  return rc;
}

// A piece of a position-ordered record stream with global seqIds: n records on the device that belong to genomes [g0, g1).
struct RecordPart { uint32_t *rec = nullptr; size_t n = 0; int32_t g0 = 0, g1 = 0; bool owned = true; };

// Cut the record parts into index chunks at genome borders (each chunk <= ctx->maxIndexMinimizers records, balanced) and build
// them.  Owned parts are released as soon as the chunks that need them exist — unless the set is streamed: then the sketch takes
// the records over (unowned parts are copied) and builds index arrays on demand.
int add_chunks(ani_ctx *ctx, ani_sketch *sk, std::vector<RecordPart> &parts)
{
  const int32_t *gcs = sk->genomeContigStart.data();
  // records per genome: first record of every contig inside its part (binary search on the device), then differences
  std::vector<uint64_t> genomeRecs((size_t)sk->nGenomes, 0), genomeOffInPart((size_t)sk->nGenomes, 0);
  std::vector<int32_t> genomePart((size_t)sk->nGenomes, -1);
  uint64_t total = 0;
  for (size_t pi = 0; pi < parts.size(); pi++) {
    const RecordPart &pt = parts[pi];
    if (pt.g1 <= pt.g0) continue;
    const int32_t c0 = gcs[pt.g0], nc = gcs[pt.g1] - c0;
    std::vector<uint64_t> first((size_t)nc + 1, 0);
    if (pt.n) {
      TRY(ctx->unitAux.ensure(((size_t)nc + 1) * 8));
      hipLaunchKernelGGL(k_records_contig_first, dim3(grid_for((size_t)nc + 1, 256, 65535)), dim3(256), 0, ctx->stream, (const uint32_t *)pt.rec, (uint64_t)pt.n, c0, nc,
                         ctx->unitAux.as<uint64_t>());
      HIP_TRY(hipMemcpyAsync(first.data(), ctx->unitAux.p, ((size_t)nc + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (first[nc] != pt.n || first[0] != 0) return fail(ANI_ERR_ARG, "minimizer records carry seqIds outside their genome range [%d, %d)", c0, c0 + nc);
    }
    for (int32_t g = pt.g0; g < pt.g1; g++) {
      genomePart[g] = (int32_t)pi; genomeOffInPart[g] = first[gcs[g] - c0]; genomeRecs[g] = first[gcs[g + 1] - c0] - first[gcs[g] - c0];
    }
    total += pt.n;
  }
  for (int32_t g = 0; g < sk->nGenomes; g++) if (genomePart[g] < 0) return fail(ANI_ERR_INTERNAL, "genome %d is in no record part", g);
  sk->genomeRecStart.assign((size_t)sk->nGenomes + 1, 0);
  for (int32_t g = 0; g < sk->nGenomes; g++) sk->genomeRecStart[g + 1] = sk->genomeRecStart[g] + genomeRecs[g];
  // Resident or streamed?  The index takes ~45 bytes per minimizer and its build another ~25 of transient
  // arrays; a set that does not fit beside a working-set reserve keeps its 12-byte records instead and at most `maxResident`
  // chunks' arrays (ANI_MAX_RESIDENT_CHUNKS forces a limit: the tests stream tiny sets that way).
  uint64_t maxN = std::min<uint64_t>(ctx->maxIndexMinimizers, 0x7fffffe0ull);
  int32_t resident = ctx->maxResidentChunks;
  if (resident == 0) {
    size_t freeB = 0, totB = 0;
    if (hipMemGetInfo(&freeB, &totB) == hipSuccess) {
      uint64_t cached = 0;
      for (int cls = 0; cls < 2; cls++) { DevicePool &pl = cur_pool(cls); std::lock_guard<std::mutex> g(pl.mu); cached += pl.cachedBytes; }
      const uint64_t avail = (uint64_t)freeB + cached, reserve = std::min<uint64_t>((uint64_t)32 << 30, (uint64_t)totB / 8);
      const uint64_t largest = std::min<uint64_t>(total, maxN);
      // (Streamed chunks of a fixed 10^9 minimizers.  Measured: chunks as large as the free memory allows — 4 instead of 5 for
      // 10 000 x 5 Mbp — save a probe pass, but more fragments then have > 2048 seed hits per chunk and move to LDS class M:
      // the 10 000 x 10 000 step stayed at 6.1 s.)
      if (45 * total + 25 * largest + reserve > avail) { resident = 1; maxN = std::min<uint64_t>(maxN, ctx->streamChunkMinimizers); }
    }
  }
  // balanced chunk sizes: ceil(total / max) chunks of about total / that
  const uint64_t nCh = std::max<uint64_t>(1, (total + maxN - 1) / maxN);
  const uint64_t target = std::min<uint64_t>(maxN, (total + nCh - 1) / nCh + (total / nCh) / 50 + 1);
  // the plan: genome ranges and the record slices they are built from
  int32_t g0 = 0;
  while (g0 < sk->nGenomes || (sk->nGenomes == 0 && sk->chunks.empty())) {
    int32_t g1 = g0; uint64_t n = 0;
    while (g1 < sk->nGenomes && (g1 == g0 || n + genomeRecs[g1] <= target)) { n += genomeRecs[g1]; g1++; }
    if (n >= 0x7ffffff0ull) return fail(ANI_ERR_LIMIT, "reference genome %d alone yields %llu minimizers (>= 2^31)", g0, (unsigned long long)n);
    IndexChunk *ch = nullptr;
    TRY(new_chunk(ctx, &sk->params, (size_t)n, sk->contigLen.data(), gcs, g0, g1 - g0, &ch));
    for (int32_t g = g0; g < g1;) {                 // runs of genomes inside one part
      const int32_t pi = genomePart[g]; int32_t h = g; uint64_t m = 0;
      while (h < g1 && genomePart[h] == pi) { m += genomeRecs[h]; h++; }
      if (m) ch->pieces.push_back(RecordPiece{parts[pi].rec + 3 * genomeOffInPart[g], (size_t)m});
      g = h;
    }
    sk->chunks.push_back(ch); ctx->counters.indexChunks++;
    sk->n += n; sk->maxChunkBins = std::max(sk->maxChunkBins, ch->totalBins);
    g0 = g1;
    if (sk->nGenomes == 0) break;
  }
  sk->streaming = resident > 0 && (size_t)resident < sk->chunks.size();
  sk->maxResident = sk->streaming ? resident : 0;
  if (sk->streaming) {
    // the sketch takes the records over: owned parts as they are, the others copied (the chunks' pieces point into them)
    for (RecordPart &pt : parts) {
      if (!pt.rec || !pt.n) continue;
      if (pt.owned) { sk->kept.push_back(pt.rec); pt.rec = nullptr; continue; }     // taken over: the caller's cleanup finds nothing to free
      uint32_t *cp = nullptr;
      HIP_TRY(pool_malloc((void **)&cp, pt.n * 12));
      sk->kept.push_back(cp);
      HIP_TRY(hipMemcpyAsync(cp, pt.rec, pt.n * 12, hipMemcpyDeviceToDevice, ctx->stream));
      for (IndexChunk *ch : sk->chunks) for (RecordPiece &pc : ch->pieces)
        if (pc.rec >= pt.rec && pc.rec < pt.rec + 3 * pt.n) pc.rec = cp + (pc.rec - pt.rec);
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  } else {
    std::vector<int32_t> partLastUse(parts.size(), -1);
    for (int32_t g = 0; g < sk->nGenomes; g++) partLastUse[genomePart[g]] = g;
    for (size_t c = 0; c < sk->chunks.size(); c++) {
      IndexChunk *ch = sk->chunks[c];
      TRY(build_chunk_index(ctx, &sk->params, ch));
      ch->pieces.clear();                              // the records may go away now
      for (size_t pi = 0; pi < parts.size(); pi++)
        if (parts[pi].owned && parts[pi].rec && partLastUse[pi] < ch->g0 + ch->nGenomes) { pool_free(parts[pi].rec); parts[pi].rec = nullptr; }
    }
  }
  sk->nUnique = 0;
  for (IndexChunk *ch : sk->chunks) sk->nUnique += ch->nUnique;
  sk->uniqueExact = sk->chunks.size() <= 1 && !sk->streaming;
  ctx->counters.refBases += sk->totalLen; ctx->counters.refUniqueHashes += sk->nUnique;
  return upload_luts(sk, 512);
}

// distinct hashes over all chunks = sum of the chunks' own counts - hashes that already occur in an earlier chunk
int exact_unique(ani_sketch *sk)
{
  if (sk->uniqueExact) return ANI_OK;
  ani_ctx *ctx = sk->ctx;
  uint64_t dup = 0;
  for (size_t c = 1; c < sk->chunks.size(); c++) {
    IndexChunk *C = sk->chunks[c];
    if (!C->n) continue;
    TRY(ensure_chunk(sk, c));
    uint8_t *seen = nullptr;
    HIP_TRY(pool_malloc((void **)&seen, C->n));
    hipError_t e = hipMemsetAsync(seen, 0, C->n, ctx->stream);
    for (size_t x = 0; x < c && e == hipSuccess; x++) {
      IndexChunk *E = sk->chunks[x];
      if (!E->n) continue;
      { const int rcE = ensure_chunk(sk, x, (int)c); if (rcE != ANI_OK) { pool_free(seen); return rcE; } }
      hipLaunchKernelGGL(k_index_mark_shared, dim3(grid_for(C->n, 256, 65535)), dim3(256), 0, ctx->stream, (const uint32_t *)C->sHash, C->n, (const TableSlot *)E->table,
                         E->tableSlots, sk->params.windowSize, seen);
    }
    int rc = e == hipSuccess ? zero_counters(ctx) : fail(ANI_ERR_DEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    unsigned long long host[CNT_N];
    if (rc == ANI_OK) {
      hipLaunchKernelGGL(k_count_flags, dim3(grid_for(C->n, 256, 4096)), dim3(256), 0, ctx->stream, (const uint8_t *)seen, C->n, cnt_ptr(ctx, CNT_UNIQ));
      rc = read_counters(ctx, host);
    }
    pool_free(seen);
    TRY(rc);
    dup += host[CNT_UNIQ];
  }
  sk->nUnique = 0;                 // (a streamed set knows a chunk's own count only once the chunk has been built: every chunk has been, above)
  for (IndexChunk *ch : sk->chunks) { if (!ch->everBuilt && ch->n) return fail(ANI_ERR_INTERNAL, "index chunk never built"); sk->nUnique += ch->nUnique; }
  sk->nUnique -= dup; sk->uniqueExact = true;
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
    TRY(ctx->fragOff.ensure((nF + 1) * 4)); TRY(ctx->fragS.ensure((nF + 1) * 4)); TRY(ctx->fragGenome.ensure((nF + 1) * 4)); TRY(ctx->fragQSeq.ensure((nF + 1) * 4));
  }
  FragArrays fa; fa.fragOff = ctx->fragOff.as<uint32_t>(); fa.fragS = ctx->fragS.as<int32_t>(); fa.fragGenome = ctx->fragGenome.as<int32_t>(); fa.fragQSeq = ctx->fragQSeq.as<int32_t>();
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
  { const char *ev = getenv("ANI_FRAG_ORDER");
    if (fs.genomeFragments.size() >= 2 && fs.fragQSeq && !(ev && !strcmp(ev, "plain"))) {
      TRY(ctx->fragOrder.ensure(nF * 4)); TRY(ctx->fragOrderTmp.ensure(nF * 20));
      uint64_t *keyIn = ctx->fragOrderTmp.as<uint64_t>(), *keyOut = keyIn + nF; uint32_t *idxIn = (uint32_t *)(keyOut + nF);
      hipLaunchKernelGGL(k_frag_order_keys, dim3(grid_for(nF)), dim3(256), 0, ctx->stream, fs.fragQSeq, (int32_t)nF, keyIn, idxIn);
      size_t tb = 0;
      int keyBits = 1;                              // keys are running fragment ids inside a genome: 11 bits for 5 Mbp genomes, two 8-bit passes
      { int32_t mx = 0; for (int32_t v : fs.genomeFragments) mx = std::max(mx, v); while (keyBits < 32 && (1ll << keyBits) <= (long long)mx) keyBits++; }
      int rc = ani_sort_pairs_u64_u32(keyIn, keyOut, idxIn, ctx->fragOrder.as<uint32_t>(), nF, keyBits, nullptr, &tb, ctx->stream);
      if (rc == 0) { TRY(ctx->sortTmp.ensure(tb + 256)); rc = ani_sort_pairs_u64_u32(keyIn, keyOut, idxIn, ctx->fragOrder.as<uint32_t>(), nF, keyBits, ctx->sortTmp.p, &tb, ctx->stream); }
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
  uint64_t ccap = std::min<uint64_t>(std::max<uint64_t>((uint64_t)((double)nF * ctx->candPerFrag) + ctx->candPoolMin, ctx->candPoolMin * kPoolStripes), kCandLimit);   // at least 4096 per stripe (4 MB): a few heavy fragments fit without a retry
  const bool smallBatch = nF < 16 * (size_t)kPoolStripes;
  if (smallBatch) ccap = std::min<uint64_t>(std::max<uint64_t>(ccap, ctx->smallBatchCandCap), kCandLimit);
  TRY(ctx->l1MidList.ensure(nF * 4)); TRY(ctx->l1BigList.ensure(nF * 4));
  unsigned nMid = 0, nBig = 0; unsigned long long nTiny = 0, nSmall = 0;
  std::vector<int32_t> bigFrags, bigInfo;          // fragments beyond the LDS classes and their (sketch size, seed hits)
  unsigned long long hitsTotal = 0;
  uint32_t probeOverflow = 0;                       // fragments k_l1_probe marked with >= 2^31 seed hits: latched after attempt 0 (the probe runs once)
  for (int attempt = 0;; attempt++) {
    ccap = (uint64_t)stripe_cap(ccap) * kPoolStripes;
    TRY(ctx->candFrag.ensure(ccap * 4)); TRY(ctx->candSeq.ensure(ccap * 4)); TRY(ctx->candStart.ensure(ccap * 4)); TRY(ctx->candEnd.ensure(ccap * 4));
    if (attempt == 0) TRY(zero_counters(ctx));
    else TRY(zero_cursors(ctx, POOL_CAND));            // only the candidate pool is redone: k_l1_probe's results (CNT_HITS, CNT_NEG = its overflow marker, the class lists) stay
    L1Args a;
    a.qPool = fs.qPool; a.fragOff = fs.fragOff; a.fragS = fs.fragS; a.nFrag = (int32_t)nF;
    a.table = sk->table; a.tableSlots = sk->tableSlots; a.sSW = sk->sSW; a.bucketW = w; a.nIndex = sk->n;
    a.minHitsLUT = set->dMinHits; a.lutMaxS = set->dLutMaxS; a.L = L;
    a.candFrag = ctx->candFrag.as<int32_t>(); a.candSeq = ctx->candSeq.as<int32_t>(); a.candStart = ctx->candStart.as<int32_t>(); a.candEnd = ctx->candEnd.as<int32_t>();
    a.candCap = stripe_cap(ccap); a.candCount = cur_ptr(ctx, POOL_CAND);
    a.fragCandOff = ctx->fragCandOff.as<uint32_t>(); a.fragCandCnt = ctx->fragCandCnt.as<int32_t>(); a.fragHits = ctx->fragHits.as<int32_t>();
    a.sumHits = cnt_ptr(ctx, CNT_HITS); a.tinyCount = cnt_ptr(ctx, CNT_TINY); a.smallCount = cnt_ptr(ctx, CNT_SMALL);
    a.filterShift = 1; while (a.filterShift < 30 && (1 << a.filterShift) < 2 * L) a.filterShift++;
    a.fragOrder = fragOrder;
    a.filterMinHits = ctx->l1FilterMin; a.ldsHitCap = ctx->l1LdsMax; a.tinyPath = ctx->l1Tiny ? 1 : 0;
    a.probeFirst = ctx->probeFirst.as<uint32_t>(); a.probeCnt = ctx->probeCnt.as<uint32_t>();
    a.midList = ctx->l1MidList.as<int32_t>(); a.midCount = (unsigned int *)cnt_ptr(ctx, CNT_LISTM);
    a.bigList = ctx->l1BigList.as<int32_t>(); a.bigCount = (unsigned int *)cnt_ptr(ctx, CNT_LISTBIG);
    a.overflowCount = (unsigned int *)cnt_ptr(ctx, CNT_NEG); a.hitLimit = ctx->l1HitLimit;
    {
      StageTimer tm(ctx, &ctx->counters.msL1);
      if (attempt == 0) {
        { StageTimer tk(ctx, &ctx->counters.msL1Probe, 1); hipLaunchKernelGGL(k_l1_probe, dim3(pad8((nF + kL1ProbeFrags - 1) / kL1ProbeFrags)), dim3(kTPB), 0, ctx->stream, a); }
        // the probe routed the fragments to their classes: how many of each decides what is launched
        unsigned long long cls[CNT_N];
        TRY(read_counters(ctx, cls));
        nMid = (unsigned)cls[CNT_LISTM]; nBig = (unsigned)cls[CNT_LISTBIG]; nTiny = cls[CNT_TINY]; nSmall = cls[CNT_SMALL];
        ctx->counters.l1TinyFragments += nTiny;
      }
      if (nTiny && ctx->l1Tiny) { StageTimer tk(ctx, &ctx->counters.msL1Tiny, 1); hipLaunchKernelGGL(k_l1_tiny, dim3(pad8((nF + kL1TinyFrags - 1) / kL1TinyFrags)), dim3(kTPB), 0, ctx->stream, a); }
      {
        StageTimer tk(ctx, &ctx->counters.msL1Main, 1);
        if (2 * nSmall >= nF || getenv("ANI_L1_DENSE"))                  // the usual case: (nearly) every fragment is of class S, one workgroup per fragment
          hipLaunchKernelGGL((k_l1<0, kL1HitCapSmall>), dim3(pad8(nF)), dim3(kTPB), 0, ctx->stream, a, (const int32_t *)nullptr);
        else {                                                            // a fragment set against a foreign shard / chunk: class S is the exception, listed
          TRY(ctx->l1SmallList.ensure((nSmall + 1) * 4));
          HIP_TRY(hipMemsetAsync(cnt_ptr(ctx, CNT_LISTL), 0, 8, ctx->stream));
          hipLaunchKernelGGL(k_l1_list, dim3(grid_for(nF, kTPB)), dim3(kTPB), 0, ctx->stream, a, ctx->l1SmallList.as<int32_t>(), (unsigned int *)cnt_ptr(ctx, CNT_LISTL));
          if (nSmall) hipLaunchKernelGGL((k_l1<0, kL1HitCapSmall>), dim3((unsigned)nSmall), dim3(kTPB), 0, ctx->stream, a, (const int32_t *)ctx->l1SmallList.as<int32_t>());
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
          if (tiles > 0x7ffffff0ull || hashes > 0xfffffff0ull) return fail(ANI_ERR_LIMIT, "group of oversized fragments with %llu seed hits", (unsigned long long)hits);
          // group tables: [frag n][sOff n+1][tileFirst n+1] as 32-bit words, then hitOff (64-bit) — one upload
          const size_t w32 = n + 2 * (n + 1), tblBytes = ((w32 * 4 + 7) / 8) * 8 + (n + 1) * 8;
          TRY(ctx->l1BigTbl.ensure(tblBytes)); TRY(ctx->l1BigHash.ensure((hashes ? hashes : 1) * 4));
          TRY(ctx->l1BigHitsA.ensure((hits ? hits : 1) * 8)); TRY(ctx->l1BigHitsB.ensure((hits ? hits : 1) * 8)); TRY(ctx->l1BigV.ensure((hits ? hits : 1) * 4 + (size_t)nBig * 8));
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
            int rc = ani_sort_keys_u64_bits(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), (size_t)hits, shiftRank + rankBits, nullptr, &tbytes, ctx->stream);
            if (rc == 0) { TRY(ctx->sortTmp.ensure(tbytes + 256)); rc = ani_sort_keys_u64_bits(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), (size_t)hits, shiftRank + rankBits, ctx->sortTmp.p, &tbytes, ctx->stream); }
            if (rc != 0) return fail(ANI_ERR_DEVICE, "radix sort of seed hits failed (%d)", rc);
            hipLaunchKernelGGL(k_l1_big_unpack, dim3(grid_for((size_t)hits, 256, 65535)), dim3(256), 0, ctx->stream, ctx->l1BigHitsB.as<uint64_t>(), (uint64_t)hits, shiftSeq, shiftRank);
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
  // after 34 GB of pool; seen in the parity suite under ANI_POOL_POISON).  The estimate follows the batches down as well as up.
  if (!smallBatch) {                             // (a one-to-many query of 1666 fragments counts: without its update every call ran the L1 kernels twice)
    const double seen = 1.25 * (double)ctx->poolMaxStripe[POOL_CAND] * kPoolStripes / (double)nF;
    ctx->candSeen[ctx->candSeenAt++ & 31] = seen;
    double mx = 12.0;
    for (double v : ctx->candSeen) mx = std::max(mx, v);
    ctx->candPerFrag = mx;
  } else ctx->smallBatchCandCap = std::min<uint64_t>(grown_cap(ctx->poolMaxStripe[POOL_CAND]), (uint64_t)1 << 26);   // the next small batch starts from what this one needed (bounded: 1 GB of pool)
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
                         ctx->fragOrdOff.as<uint32_t>(), fragOrder, (int32_t)nF, ctx->candSeq.as<int32_t>(), ctx->candStart.as<int32_t>(), ctx->candEnd.as<int32_t>(),
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
    a.candFrag = ctx->ocFrag.as<int32_t>(); a.candSeq = ctx->ocSeq.as<int32_t>(); a.candStart = ctx->ocStart.as<int32_t>(); a.candEnd = ctx->ocEnd.as<int32_t>();
    a.nCand = (int32_t)nCand; a.qPool = fs.qPool; a.fragOff = fs.fragOff; a.fragS = fs.fragS;
    a.mHash = sk->mHash; a.mWpos = sk->mWpos; a.dup = DupLinks{sk->dupList, sk->nDup, sk->dupBits}; a.mWin = sk->mWin; a.posBase = sk->posBase; a.posSample = sk->posSample;
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
      fa.codes = nullptr; fa.slowFlag = ctx->l2SlowFlag[p].as<int32_t>(); fa.fragCandOff = ctx->fragOrdOff.as<uint32_t>(); fa.fragOrder = fragOrder; fa.nFrag = (int32_t)nF;
      // fragments that own candidates c0 and c1-1 (ordOff is non-decreasing; fragments without candidates repeat a value)
      const int32_t fA = (int32_t)(std::upper_bound(ordOff, ordOff + nF, (uint32_t)c0) - ordOff) - 1;
      const int32_t fB = (int32_t)(std::upper_bound(ordOff, ordOff + nF, (uint32_t)(c1 - 1)) - ordOff) - 1;
      fa.fragBase = fA;
      { const char *ev = getenv("ANI_L2_PATH"); fa.allowFast = (ev && !strcmp(ev, "general")) ? 0 : (ev && !strcmp(ev, "classB")) ? 2 : 1; }
      if (L - (w - 1) - (k - 1) + 2 > (int)kWinMask || L - (w - 1) - (k - 1) < 1) fa.allowFast = 0;      // the 14-bit window links need cmw + 2 < 2^14 (and a window at all)
      {
        StageTimer tk(ctx, &ctx->counters.msL2Ranges, 1);
        hipLaunchKernelGGL(k_l2_ranges, dim3(grid_for(n)), dim3(kTPB), 0, ctx->stream, fa);
      }
      uint64_t nCodes = 0;
      {
        const int rc = device_scan(ctx, fa.codeCount, ctx->l2CodeOff[p].as<uint32_t>(), (uint32_t)n, &nCodes, ctx->l2CodeLimit);
        if (rc == ANI_ERR_LIMIT && n > 1) { chunk = (n + 1) / 2; ctx->counters.l2ChunkHalvings++; continue; }     // very long candidate ranges: smaller chunk, same candidates again
        TRY(rc);
      }
      TRY(ctx->l2Codes[p].ensure((nCodes + 64) * 2));
      fa.codes = ctx->l2Codes[p].as<uint32_t>();
      if (nCodes) {
        {
          // order the chunk's candidates by code-stream length (longest first) for the simulation
          StageTimer tk(ctx, &ctx->counters.msL2Ranges, 1);
          HIP_TRY(hipMemsetAsync(ctx->l2LenHist[p].p, 0, kL2LenBuckets * 4, ctx->stream));
          HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ctx->l2LenHist[p].as<unsigned int>() + kL2LenBuckets), (int)n, 1, ctx->stream));   // list length for the simulation launch
          hipLaunchKernelGGL(k_l2_len_hist, dim3(grid_for(n, kTPB, 2048)), dim3(kTPB), 0, ctx->stream, (const int32_t *)fa.codeCount, (int32_t)n, ctx->l2LenHist[p].as<unsigned int>());
          hipLaunchKernelGGL(k_l2_len_scan, dim3(1), dim3(kTPB), 0, ctx->stream, ctx->l2LenHist[p].as<unsigned int>());
          hipLaunchKernelGGL(k_l2_len_scatter, dim3(grid_for(n, kTPB)), dim3(kTPB), 0, ctx->stream, (const int32_t *)fa.codeCount, (int32_t)c0, (int32_t)n,
                             ctx->l2LenHist[p].as<unsigned int>(), ctx->l2Order[p].as<int32_t>());
        }
        {
          StageTimer tk(ctx, &ctx->counters.msL2Codes, 1);
          fa.nFragChunk = fB - fA + 1;
          hipLaunchKernelGGL(k_l2_codes, dim3((unsigned)((fa.nFragChunk + 7) / 8 * 8)), dim3(kTPB), 0, ctx->stream, fa);
        }
        // ANI_L2_OVERLAP=1 puts the simulation (VALU-bound) on the side stream so that the next chunk's ranges / codes kernels (memory-
        // and latency-bound) can start beside it.  Measured (round 3, 1000 x 1000, profiles/r03f_bench_overlap.json.log): the L2 stage
        // 94.8 -> 90.8 ms, the step 221.5 -> 216.8 ms — all of it from the tails: a codes workgroup (25 KiB of LDS) does not fit the
        // 16 KiB slot a retiring simulation workgroup frees.  Off by default: with it the per-kernel times (bench line, rocprofv3) are
        // those of kernels that share the machine (k_l2_codes reads 77 ms instead of 39.6), and the dominant-kernel accounting of the
        // measurement contract stops meaning what it says.
        const bool overlap = ctx->l2Overlap;
        hipStream_t simStream = ctx->stream;
        if (overlap) {
          HIP_TRY(hipEventRecord(ctx->evSimA[p], ctx->stream));            // codes of this chunk are written
          HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->evSimA[p], 0));
          simStream = ctx->stream2;
        }
        {
          StageTimer tk(ctx, &ctx->counters.msL2Kernel, 1, simStream);
          hipLaunchKernelGGL((k_l2_sim<L2GeomA>), dim3(grid_for(n, kL2SimTPB)), dim3(kL2SimTPB), 0, simStream, fa, (const int32_t *)ctx->l2Order[p].as<int32_t>(),
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
        hipLaunchKernelGGL((k_l2_sim<L2GeomB>), dim3(grid_for(n, kL2SimTPB)), dim3(kL2SimTPB), 0, ctx->stream2, fb, (const int32_t *)ctx->l2ClassList[p].as<int32_t>(),
                           (const unsigned int *)cnt_ptr(ctx, CNT_CLASSB));
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
        hipLaunchKernelGGL(k_l2, dim3((unsigned)(lanes / kTPB)), dim3(kTPB), 0, ctx->stream, sa, (const int32_t *)ctx->l2SlowList.as<int32_t>(), (int32_t)nSlowTotal, (int32_t)base);
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

// 1-way / 2-way / mean for the candidates produced by query_stages; rows appended to `rows`
// result rows in a malloc'ed, geometrically grown buffer that is handed to the caller as it is (ani_free releases it)
struct RowBuf {
  ani_cgi_t *p = nullptr; size_t n = 0, cap = 0;
  ani_cgi_t *grow(size_t extra)
  {
    if (n + extra > cap || !p) {
      size_t want = std::max<size_t>(n + extra, cap + cap / 2 + 1024);
      void *q = realloc(p, want * sizeof(ani_cgi_t));
      if (!q) return nullptr;
      p = (ani_cgi_t *)q; cap = want;
    }
    return p + n;
  }
  ~RowBuf() { free(p); }
  ani_cgi_t *release() { ani_cgi_t *r = p ? p : (ani_cgi_t *)malloc(sizeof(ani_cgi_t)); p = nullptr; return r; }
};

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
int collect_rows(ani_ctx *ctx, ani_sketch *set, const FragSet &fs, int32_t nQuery, int32_t firstQueryId, RowBuf *rows, const IndexChunk *block = nullptr)
{
  const int32_t nCols = block ? block->nGenomes : set->nGenomes, col0 = block ? block->g0 : 0;
  const size_t nPairs = (size_t)nQuery * (size_t)nCols;
  if (nPairs == 0) return ANI_OK;
  uint32_t *dense = nullptr;
  TRY(pinned_buffer(ctx, 1, nPairs * 8 + 8, (void **)&dense));
  HIP_TRY(hipMemcpyAsync(dense, ctx->rows.p, nPairs * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  size_t m = 0;
  for (size_t p = 0; p < nPairs; p++) m += dense[p] != 0;
  ani_cgi_t *out = rows->grow(m);
  if (!out) return fail(ANI_ERR_NOMEM, "host allocation of %zu result rows failed", m);
  rows->n += m;
  for (int32_t qi = 0; qi < nQuery; qi++) {                         // query ascending, reference ascending
    const uint32_t *cnt = dense + (size_t)qi * (size_t)nCols, *idb = cnt + nPairs;
    for (int32_t g = 0; g < nCols; g++) {
      if (!cnt[g]) continue;
      ani_cgi_t r; r.refGenomeId = col0 + g; r.qryGenomeId = firstQueryId + qi; r.countSeq = (int32_t)cnt[g];
      r.totalQueryFragments = fs.genomeFragments[qi];
      memcpy(&r.identity, &idb[g], 4);
      *out++ = r;
    }
  }
  ctx->counters.cgiRows += m;
  return ANI_OK;
}

template <class T> int to_host_malloc(const std::vector<T> &v, T **out, size_t *n)
{
  *n = v.size();
  *out = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  if (!v.empty()) memcpy(*out, v.data(), v.size() * sizeof(T));
  return ANI_OK;
}

}  // namespace

// Fragment sketches of a batch of query genomes, kept on the device (ani_fragset_build / ani_sketch_records_self): mapping them
// against several reference sets or index chunks — or mapping genomes that were sketched as references — hashes nothing twice.
struct ani_fragset {
  ani_ctx *ctx = nullptr; int device = 0;
  ani_params_t params;
  FragSet fs;                      // host tables + device pointers into the arrays below
  FragArrays arr; uint32_t *qPool = nullptr;
  bool borrowed = false;           // the arrays live in a caller's buffer (ani_fragset_unpack): not freed with the set
  std::vector<int64_t> genomeFragStart;     // prefix of fs.genomeFragments
};

namespace {
void fragset_release(ani_fragset *f)
{
  void *ptrs[] = {f->arr.fragOff, f->arr.fragS, f->arr.fragGenome, f->arr.fragQSeq, f->qPool};
  if (!f->borrowed) for (void *q : ptrs) if (q) pool_free(q);
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
    if (e != hipSuccess) { if (all) pool_free(all); cleanup(); return fail(e == hipErrorOutOfMemory ? ANI_ERR_NOMEM : ANI_ERR_DEVICE, "gathering %zu records failed: %s", total, hipGetErrorString(e)); }
  }
  cleanup();
  if (ctx->timerPending.size() > 1024) flush_timers(ctx);
  *devRecords = all; *n = total;
  return ANI_OK;
}

// Map + reduce for the fragments of `fs` (a whole set or a slice of one) against every index chunk (all resident); rows appended
int map_fragset(ani_ctx *ctx, ani_sketch *sk, const FragSet &fs, int32_t nQuery, int32_t firstQueryId, RowBuf *rows)
{
  TRY(upload_luts(sk, fs.maxS));
  for (IndexChunk *ch : sk->chunks) {
    int32_t nCand = 0;
    TRY(map_stage(ctx, sk, ch, fs, &nCand));
    TRY(reduce_stage(ctx, sk, ch, fs, nCand, nQuery));
  }
  return collect_rows(ctx, sk, fs, nQuery, firstQueryId, rows);
}

// Sub-batches of kept fragment sets, mapped against a whole reference set.  A resident set is walked sub-batch by sub-batch (every
// chunk per sub-batch, one dense result table per sub-batch).  A streamed set is walked CHUNK by chunk — build the chunk's index,
// map every sub-batch of every set against it, drop it — so that each chunk is built once per call however many query genomes
// there are (the reference's own loop has the same shape: per reference split, all queries; core_genome_identity.cpp:55-106);
// the rows of a sub-batch then come chunk by chunk and are put back into (query, reference) order at the end.
struct SubBatch { const ani_fragset *set; int32_t g0, g1, firstQueryId; FragSet v; };
int map_fragsets(ani_ctx *ctx, ani_sketch *sk, const std::vector<const ani_fragset *> &sets, const std::vector<int32_t> &firstQueryIds, RowBuf *rows)
{
  std::vector<SubBatch> sub;
  int maxS = 0;
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
      while (g1 < nG && (g1 == g0 || ((uint64_t)(f->genomeFragStart[g1] - f->genomeFragStart[g0]) < ctx->subBatchFragments && (uint64_t)(g1 - g0) < maxQ))) g1++;
      const int64_t fA = f->genomeFragStart[g0], fB = f->genomeFragStart[g1];
      SubBatch sb; sb.set = f; sb.g0 = g0; sb.g1 = g1; sb.firstQueryId = firstQueryIds[si] + g0;
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
    for (const SubBatch &sb : sub) TRY(map_fragset(ctx, sk, sb.v, sb.g1 - sb.g0, sb.firstQueryId, rows));
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
      TRY(collect_rows(ctx, sk, sb.v, sb.g1 - sb.g0, sb.firstQueryId, &part[i], ch));
    }
  }
  // a sub-batch's rows are (chunk, query, reference)-ordered and a chunk's references all precede the next chunk's: a stable
  // distribution by query — one counting pass, one scatter; the 7.8e7 rows of a 10 000 x 10 000 run are not comparison-sorted —
  // restores (query, reference) order
  for (size_t i = 0; i < sub.size(); i++) {
    RowBuf &pb = part[i];
    if (!pb.n) continue;
    ani_cgi_t *out = rows->grow(pb.n);
    if (!out) return fail(ANI_ERR_NOMEM, "host allocation of %zu result rows failed", pb.n);
    const int32_t q0 = sub[i].firstQueryId, nq = sub[i].g1 - sub[i].g0;
    std::vector<size_t> start((size_t)nq + 1, 0);
    for (size_t r = 0; r < pb.n; r++) start[(size_t)(pb.p[r].qryGenomeId - q0) + 1]++;
    for (int32_t q = 0; q < nq; q++) start[(size_t)q + 1] += start[(size_t)q];
    for (size_t r = 0; r < pb.n; r++) out[start[(size_t)(pb.p[r].qryGenomeId - q0)]++] = pb.p[r];
    rows->n += pb.n;
    free(pb.p); pb.p = nullptr; pb.n = pb.cap = 0;                  // host memory of a big run: give each part back as soon as it is merged
  }
  return ANI_OK;
}
}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
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
  HIP_TRY(hipStreamCreate(&c->stream2));
  if (const char *ev = getenv("ANI_SUBBATCH_FRAGS")) { const long long v = atoll(ev); if (v > 0) c->subBatchFragments = (uint64_t)v; }
  if (const char *ev = getenv("ANI_L2_CHUNK")) { const long long v = atoll(ev); if (v >= 1) c->l2ChunkCandidates = (size_t)v; }
  if (const char *ev = getenv("ANI_L2_CODE_LIMIT")) { const long long v = atoll(ev); if (v >= 1) c->l2CodeLimit = (uint64_t)v; }
  if (const char *ev = getenv("ANI_MAX_INDEX_MINIMIZERS")) { const long long v = atoll(ev); if (v >= 1) c->maxIndexMinimizers = (uint64_t)v; }
  if (const char *ev = getenv("ANI_L1_FILTER_MIN")) c->l1FilterMin = atoi(ev);
  if (const char *ev = getenv("ANI_L1_LDS_MAX")) c->l1LdsMax = std::max(0, std::min(atoi(ev), (int)ani::kL1HitCapMax));
  if (const char *ev = getenv("ANI_DUP_PAIR_CAP")) { const long long v = atoll(ev); if (v >= 1) c->dupPairCap = (uint64_t)v; }
  if (const char *ev = getenv("ANI_L1_TINY")) c->l1Tiny = strcmp(ev, "0") != 0;
  if (const char *ev = getenv("ANI_L2_OVERLAP")) c->l2Overlap = !strcmp(ev, "1");
  if (const char *ev = getenv("ANI_L1_HIT_LIMIT")) { const long long v = atoll(ev); if (v >= 1) c->l1HitLimit = std::min<uint64_t>((uint64_t)v, 0x7ffffff0ull); }
  if (const char *ev = getenv("ANI_CAND_POOL_MIN")) { const long long v = atoll(ev); if (v >= 1) c->candPoolMin = (uint64_t)v; }
  if (const char *ev = getenv("ANI_L1_BIG_GROUP_HITS")) { const long long v = atoll(ev); if (v >= 1) c->l1BigGroupHits = (uint64_t)v; }
  if (const char *ev = getenv("ANI_L1_BIG_GROUP_FRAGS")) { const long long v = atoll(ev); if (v >= 1) c->l1BigGroupFrags = (uint64_t)v; }
  if (const char *ev = getenv("ANI_MAX_RESIDENT_CHUNKS")) { const long long v = atoll(ev); if (v >= 0) c->maxResidentChunks = (int32_t)std::min<long long>(v, 1 << 20); }
  if (const char *ev = getenv("ANI_STREAM_CHUNK_MINIMIZERS")) { const long long v = atoll(ev); if (v >= 1) c->streamChunkMinimizers = (uint64_t)v; }
  for (int i = 0; i < 2; i++) { HIP_TRY(hipEventCreateWithFlags(&c->evSimA[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->evSetDone[i], hipEventDisableTiming)); }
  int rc = c->dCounters.ensure(kCounterWords * 8);
  if (rc != ANI_OK) { delete c; return rc; }
  *out = c;
  return ANI_OK;
}

void ani_shutdown(ani_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  DevBuf *bufs[] = {&c->dCounters, &c->seqPacked, &c->seqAscii, &c->contigOff, &c->contigLen, &c->contigMode, &c->sortTmp, &c->unitStart, &c->unitAux, &c->tiles, &c->tileInfo, &c->tileMeta, &c->tileCnt,
                    &c->tileDrop, &c->tileOff, &c->poolHash, &c->poolWpos, &c->scanTmpA, &c->scanTmpB, &c->scanTmpC, &c->scanTmpD, &c->frags, &c->fragOff, &c->fragS,
                    &c->fragGenome, &c->fragQSeq, &c->qPool, &c->probeFirst, &c->probeCnt, &c->l1MidList, &c->l1SmallList, &c->l1BigList, &c->l1BigHitsA, &c->l1BigHitsB, &c->l1BigV, &c->l1BigTbl, &c->l1BigHash, &c->candFrag, &c->candSeq, &c->candStart, &c->candEnd, &c->fragCandOff, &c->fragCandCnt,
                    &c->fragCandCntClamped, &c->fragHits, &c->fragOrdOff, &c->fragOrder, &c->fragOrderTmp, &c->ocFrag, &c->ocSeq, &c->ocStart, &c->ocEnd, &c->l2Scratch, &c->l2Ranges[0], &c->l2CodeCount[0], &c->l2CodeOff[0], &c->l2Codes[0], &c->l2SlowFlag[0], &c->l2ClassList[0], &c->l2Order[0], &c->l2LenHist[0],
                    &c->l2Ranges[1], &c->l2CodeCount[1], &c->l2CodeOff[1], &c->l2Codes[1], &c->l2SlowFlag[1], &c->l2ClassList[1], &c->l2Order[1], &c->l2LenHist[1], &c->l2SlowList, &c->l2Best,
                    &c->l2First, &c->l2Last, &c->refStart, &c->idBits, &c->keepFlags, &c->keepOff, &c->mapOut, &c->bins, &c->queryFragments, &c->rows};
  for (DevBuf *b : bufs) b->release();
  for (hipEvent_t e : c->timerEvents) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; i++) { if (c->evSimA[i]) (void)hipEventDestroy(c->evSimA[i]); if (c->evSetDone[i]) (void)hipEventDestroy(c->evSetDone[i]); }
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  for (int i = 0; i < 5; i++) if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  cur_pool(0).trim(); cur_pool(1).trim();
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
    if (can) { hipError_t e = hipDeviceEnablePeerAccess(srcCtx->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e); (void)hipGetLastError(); }
    HIP_TRY(hipMemcpyPeerAsync(dst, dstCtx->device, src, srcCtx->device, bytes, dstCtx->stream));
  }
  HIP_TRY(hipStreamSynchronize(dstCtx->stream));
  return ANI_OK;
}

int ani_batch_upload(ani_ctx *ctx, const ani_seq_batch_t *genomes, ani_dev_batch **out)
{
  if (!ctx || !out) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_batch(genomes));
  if (genomes->layout == ANI_SEQ_DEVICE_BATCH || genomes->layout == ANI_SEQ_DEVICE_PACKED2) return fail(ANI_ERR_ARG, "ani_batch_upload takes host sequences");
  HIP_TRY(hipSetDevice(ctx->device));
  ani_dev_batch *b = new ani_dev_batch();
  b->ctx = ctx; b->device = ctx->device;
  const int rc = upload_batch(ctx, genomes, 0, genomes->nGenomes, &b->db, b);
  if (rc != ANI_OK) { ani_batch_free(b); return rc; }
  *out = b;
  return ANI_OK;
}

void ani_batch_free(ani_dev_batch *b)
{
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void *q : b->bufs) if (q) pool_free(q);
  delete b;
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

int ani_sketch_records(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, int32_t seqIdBase, void **devRecords, size_t *n)
{
  if (!ctx || !devRecords || !n) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(refs));
  HIP_TRY(hipSetDevice(ctx->device));
  return records_of_batch(ctx, p, refs, seqIdBase, devRecords, n, nullptr);
}

int ani_sketch_records_self(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *genomes, int32_t seqIdBase, void **devRecords, size_t *n, ani_fragset **frags)
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
  if (memcmp(h.magic, "ANIFRAGS", 8) != 0 || h.version != 1) return fail(ANI_ERR_ARG, "not a packed fragment set");
  ani_fragset tmp; tmp.params.kmerSize = h.kmerSize; tmp.params.windowSize = h.windowSize; tmp.params.fragLen = h.fragLen; tmp.params.percentageIdentity = h.percentageIdentity;
  TRY(check_params(&tmp.params));
  if (h.nFrag < 0 || h.nGenomes < 0 || h.maxS < 0 || h.poolSize > 0xfffffff0ull || h.nHashes > h.poolSize) return fail(ANI_ERR_ARG, "packed fragment set with inconsistent counts");
  tmp.fs.nFrag = h.nFrag; tmp.fs.maxS = h.maxS; tmp.fs.nHashes = h.nHashes; tmp.fs.poolSize = h.poolSize; tmp.fs.genomeFragments.assign((size_t)h.nGenomes, 0);
  const FragWireHeader want = wire_header(&tmp);       // the layout follows from the counts: the offsets in the buffer must be these
  if (want.offFragOff != h.offFragOff || want.offFragS != h.offFragS || want.offFragGenome != h.offFragGenome || want.offFragQSeq != h.offFragQSeq ||
      want.offPool != h.offPool || want.totalBytes != h.totalBytes || h.totalBytes > bytes)
    return fail(ANI_ERR_ARG, "packed fragment set is truncated or malformed");
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
  f->fs.fragOff = f->arr.fragOff; f->fs.fragS = f->arr.fragS; f->fs.fragGenome = f->arr.fragGenome; f->fs.fragQSeq = f->arr.fragQSeq; f->fs.qPool = f->qPool; f->fs.genomeBase = 0;
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

int ani_map_cgi_fragset(ani_ctx *ctx, const ani_sketch *skc, const ani_fragset *f, int32_t firstQueryId, ani_cgi_t **out, size_t *m)
{
  return ani_map_cgi_fragsets(ctx, skc, 1, &f, &firstQueryId, out, m);
}

int ani_map_cgi_fragsets(ani_ctx *ctx, const ani_sketch *skc, int32_t nSets, const ani_fragset *const *frags, const int32_t *firstQueryIds, ani_cgi_t **out, size_t *m)
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

int ani_sketch_from_records(ani_ctx *ctx, const ani_params_t *p, const void *devRecords, size_t n, const int32_t *contigLen, int32_t nContigs,
                            const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out)
{
  if (!ctx || !out || (n && !devRecords) || nContigs < 0 || nGenomes < 0 || (nContigs && !contigLen) || !genomeContigStart)
    return fail(ANI_ERR_ARG, "invalid argument");
  if (genomeContigStart[0] != 0 || genomeContigStart[nGenomes] != nContigs) return fail(ANI_ERR_ARG, "genomeContigStart does not cover the contig table");
  TRY(check_params(p));
  HIP_TRY(hipSetDevice(ctx->device));
  ani_sketch *sk = new_sketch(ctx, p, contigLen, nContigs, genomeContigStart, nGenomes);
  std::vector<RecordPart> parts(1);
  parts[0].rec = (uint32_t *)devRecords; parts[0].n = n; parts[0].g0 = 0; parts[0].g1 = nGenomes; parts[0].owned = false;
  const int rc = add_chunks(ctx, sk, parts);
  if (rc != ANI_OK) { free_sketch_device(sk); delete sk; return rc; }
  *out = sk;
  return ANI_OK;
}

namespace {
int sketch_from_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, const void *const *devRecords, const uint64_t *n,
                      const int32_t *partGenomeStart, const int32_t *contigLen, int32_t nContigs,
                      const int32_t *genomeContigStart, int32_t nGenomes, bool adopt, ani_sketch **out, bool *consumed);
}
int ani_sketch_from_record_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, const void *const *devRecords, const uint64_t *n,
                                 const int32_t *partGenomeStart, const int32_t *contigLen, int32_t nContigs,
                                 const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out)
{
  bool consumed = false;
  return sketch_from_parts(ctx, p, nParts, devRecords, n, partGenomeStart, contigLen, nContigs, genomeContigStart, nGenomes, false, out, &consumed);
}
int ani_sketch_adopt_record_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, void *const *devRecords, const uint64_t *n,
                                  const int32_t *partGenomeStart, const int32_t *contigLen, int32_t nContigs,
                                  const int32_t *genomeContigStart, int32_t nGenomes, ani_sketch **out)
{
  bool consumed = false;
  const int rc = sketch_from_parts(ctx, p, nParts, devRecords, n, partGenomeStart, contigLen, nContigs, genomeContigStart, nGenomes, true, out, &consumed);
  if (!consumed && ctx && devRecords && nParts > 0) {        // rejected before anything was taken over: the buffers are the library's all the same
    (void)hipSetDevice(ctx->device);
    for (int32_t i = 0; i < nParts; i++) if (devRecords[i]) pool_free(devRecords[i]);
  }
  return rc;
}
namespace {
int sketch_from_parts(ani_ctx *ctx, const ani_params_t *p, int32_t nParts, const void *const *devRecords, const uint64_t *n,
                      const int32_t *partGenomeStart, const int32_t *contigLen, int32_t nContigs,
                      const int32_t *genomeContigStart, int32_t nGenomes, bool adopt, ani_sketch **out, bool *consumed)
{
  if (!ctx || !out || nParts < 0 || (nParts && (!devRecords || !n || !partGenomeStart)) || nContigs < 0 || nGenomes < 0 || (nContigs && !contigLen) || !genomeContigStart)
    return fail(ANI_ERR_ARG, "invalid argument");
  if (genomeContigStart[0] != 0 || genomeContigStart[nGenomes] != nContigs) return fail(ANI_ERR_ARG, "genomeContigStart does not cover the contig table");
  if (nParts && (partGenomeStart[0] != 0 || partGenomeStart[nParts] != nGenomes)) return fail(ANI_ERR_ARG, "partGenomeStart does not cover the genomes");
  if (!nParts && nGenomes) return fail(ANI_ERR_ARG, "no record parts for %d genomes", nGenomes);
  TRY(check_params(p));
  HIP_TRY(hipSetDevice(ctx->device));
  std::vector<RecordPart> parts((size_t)nParts);
  for (int32_t i = 0; i < nParts; i++) {
    if (partGenomeStart[i + 1] < partGenomeStart[i] || (n[i] && !devRecords[i])) return fail(ANI_ERR_ARG, "record part %d is malformed", i);
    parts[i].rec = (uint32_t *)devRecords[i]; parts[i].n = (size_t)n[i]; parts[i].g0 = partGenomeStart[i]; parts[i].g1 = partGenomeStart[i + 1]; parts[i].owned = false;
  }
  // adopted buffers: a streamed set keeps them as they are (no copy of the records: a set near the device's capacity has no room
  // for one), a resident set releases each as soon as the chunks that need it are built; whatever add_chunks did not take goes back here
  for (auto &q : parts) q.owned = adopt;
  *consumed = true;
  ani_sketch *sk = new_sketch(ctx, p, contigLen, nContigs, genomeContigStart, nGenomes);
  const int rc = add_chunks(ctx, sk, parts);
  if (adopt) for (auto &q : parts) if (q.rec) { pool_free(q.rec); q.rec = nullptr; }
  if (rc != ANI_OK) { free_sketch_device(sk); delete sk; return rc; }
  *out = sk;
  return ANI_OK;
}
}  // namespace

int ani_sketch_build(ani_ctx *ctx, const ani_params_t *p, const ani_seq_batch_t *refs, ani_sketch **out)
{
  if (!ctx || !out) return fail(ANI_ERR_ARG, "null argument");
  TRY(check_params(p)); TRY(check_batch(refs));
  HIP_TRY(hipSetDevice(ctx->device));
  // references are sketched in slices of genomes so that the temporary pools stay small; the slices' records (position order,
  // global seqIds) are then cut into index chunks at genome borders
  std::vector<RecordPart> parts;
  auto cleanup = [&]() { for (auto &q : parts) if (q.rec && q.owned) pool_free(q.rec); };
  int32_t g0 = 0;
  while (g0 < refs->nGenomes) {
    int32_t g1 = g0; uint64_t bases = 0;
    while (g1 < refs->nGenomes && (g1 == g0 || bases < (1ull << 30))) {
      for (int32_t c = refs->genomeContigStart[g1]; c < refs->genomeContigStart[g1 + 1]; c++) bases += (uint64_t)refs->contigLen[c];
      g1++;
    }
    DeviceBatch db;
    int rc = upload_batch(ctx, refs, g0, g1, &db);
    RecordPart pt; pt.g0 = g0; pt.g1 = g1;
    if (rc == ANI_OK) rc = sketch_records(ctx, p, db, refs->genomeContigStart[g0], &pt.rec, &pt.n);
    if (rc != ANI_OK) { cleanup(); return rc; }
    parts.push_back(pt);
    g0 = g1;
  }
  ani_sketch *sk = new_sketch(ctx, p, refs->contigLen, refs->nContigs, refs->genomeContigStart, refs->nGenomes);
  const int rc = add_chunks(ctx, sk, parts);
  cleanup();
  if (rc != ANI_OK) { free_sketch_device(sk); delete sk; return rc; }
  *out = sk;
  return ANI_OK;
}

void ani_sketch_destroy(ani_sketch *sk)
{
  if (!sk) return;
  (void)hipSetDevice(sk->device);       // the context may be gone already: device memory goes back to the per-device pools
  free_sketch_device(sk);
  delete sk;
}

int ani_sketch_export(const ani_sketch *sk, ani_minimizer_t **out, size_t *n)
{
  if (!sk || !out || !n) return fail(ANI_ERR_ARG, "null argument");
  ani_ctx *ctx = sk->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  *n = (size_t)sk->n;
  *out = (ani_minimizer_t *)malloc((sk->n ? (size_t)sk->n : 1) * sizeof(ani_minimizer_t));
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  size_t o = 0;
  for (const IndexChunk *ch : sk->chunks) {
    if (ch->n == 0) continue;
    if (!ch->resident) {                     // streamed set: the chunk's records are at hand in the export layout already
      for (const RecordPiece &pc : ch->pieces) {
        const hipError_t ec = hipMemcpy(*out + o, pc.rec, pc.n * 12, hipMemcpyDeviceToHost);
        if (ec != hipSuccess) { free(*out); *out = nullptr; HIP_TRY(ec); }
        o += pc.n;
      }
      continue;
    }
    uint32_t *tmp = nullptr;
    hipError_t e = pool_malloc((void **)&tmp, (size_t)ch->n * 12);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ani::k_index_join, dim3(grid_for(ch->n, 256, 65535u * 8u)), dim3(256), 0, ctx->stream, ch->mHash, ch->mSeq, ch->mWpos, ch->n, (uint32_t)ch->c0, tmp);
      e = hipMemcpyAsync(*out + o, tmp, (size_t)ch->n * 12, hipMemcpyDeviceToHost, ctx->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (tmp) pool_free(tmp);
    if (e != hipSuccess || e2 != hipSuccess) { free(*out); *out = nullptr; HIP_TRY(e); HIP_TRY(e2); }
    o += ch->n;
  }
  return ANI_OK;
}

int ani_sketch_stats(const ani_sketch *skc, uint64_t *nMinimizers, uint64_t *nUnique, uint64_t *totalLength, int32_t *nContigs, int32_t *nGenomes)
{
  if (!skc) return fail(ANI_ERR_ARG, "null sketch");
  ani_sketch *sk = const_cast<ani_sketch *>(skc);
  if (nUnique) {
    HIP_TRY(hipSetDevice(sk->device));
    TRY(exact_unique(sk));          // several index chunks: hashes shared between chunks are counted once (lazily, here)
    *nUnique = sk->nUnique;
  }
  if (nMinimizers) *nMinimizers = sk->n;
  if (totalLength) *totalLength = sk->totalLen;
  if (nContigs) *nContigs = sk->nContigs;
  if (nGenomes) *nGenomes = sk->nGenomes;
  return ANI_OK;
}

int ani_sketch_chunks(const ani_sketch *sk, int32_t *nChunks, int32_t *firstGenome, int32_t cap)
{
  if (!sk || !nChunks) return fail(ANI_ERR_ARG, "null argument");
  *nChunks = (int32_t)sk->chunks.size();
  if (firstGenome) for (int32_t i = 0; i < cap && i < *nChunks; i++) firstGenome[i] = sk->chunks[i]->g0;
  return ANI_OK;
}

int ani_sketch_residency(const ani_sketch *sk, int32_t *streaming, int32_t *maxResident, int32_t *residentNow)
{
  if (!sk) return fail(ANI_ERR_ARG, "null sketch");
  if (streaming) *streaming = sk->streaming ? 1 : 0;
  if (maxResident) *maxResident = sk->streaming ? sk->maxResident : (int32_t)sk->chunks.size();
  if (residentNow) { int32_t r = 0; for (const IndexChunk *ch : sk->chunks) r += ch->resident; *residentNow = r; }
  return ANI_OK;
}

// -----------------------------------------------------------------------------------------------------
// Persistent sketch file (the reference rebuilds its sketch in every run and every thread).
//   header (4096 B) : magic "ANISKTCH", version, the parameters the sketch depends on, counts, section offsets
//   contigLen        int32[nContigs]
//   genomeContigStart int32[nGenomes + 1]
//   genomeRecStart   uint64[nGenomes + 1]   first record of every genome: a reader can take any genome range (one rank's shard)
//   names            nGenomes NUL-terminated strings (may be empty)
//   records          12-byte (hash, seqId, wpos) minimizer records, position order, global seqIds = skch::MinimizerInfo, the
//                    layout of ani_sketch_records / ani_sketch_from_records
// Sections start on 4096-byte boundaries, so the file can be mmap'ed and the record section handed to hipMemcpy as it is.  The
// hash-ordered index is not stored: rebuilding it on the device (radix sort) is faster than reading it back.
// -----------------------------------------------------------------------------------------------------
namespace {
struct SketchFileHeader {
  char magic[8]; uint32_t version, headerBytes;
  int32_t kmerSize, windowSize, fragLen; float percentageIdentity;
  int32_t nContigs, nGenomes; uint64_t nRecords;
  uint64_t offContigLen, offGcs, offGenomeRec, offNames, namesBytes, offRecords;
};
constexpr uint64_t kFileAlign = 4096;
inline uint64_t align_up(uint64_t x) { return (x + kFileAlign - 1) / kFileAlign * kFileAlign; }
bool write_all(int fd, const void *p, size_t n) { const char *c = (const char *)p; while (n) { const ssize_t w = ::write(fd, c, n); if (w <= 0) return false; c += w; n -= (size_t)w; } return true; }
bool pad_to(int fd, uint64_t *pos, uint64_t target) { static const char z[4096] = {0}; while (*pos < target) { const size_t n = (size_t)std::min<uint64_t>(4096, target - *pos); if (!write_all(fd, z, n)) return false; *pos += n; } return true; }
}  // namespace

int ani_sketch_save(const ani_sketch *sk, const char *path, const char *const *genomeNames)
{
  if (!sk || !path) return fail(ANI_ERR_ARG, "null argument");
  ani_ctx *ctx = sk->ctx;
  HIP_TRY(hipSetDevice(sk->device));
  const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return fail(ANI_ERR_ARG, "cannot create %s", path);
  std::string names;
  for (int32_t g = 0; g < sk->nGenomes; g++) { names += genomeNames && genomeNames[g] ? genomeNames[g] : (g < (int32_t)sk->genomeNames.size() ? sk->genomeNames[g].c_str() : ""); names.push_back('\0'); }
  const std::vector<uint64_t> &genomeRec = sk->genomeRecStart;      // first record of every genome (add_chunks)
  if (genomeRec.size() != (size_t)sk->nGenomes + 1) { ::close(fd); return fail(ANI_ERR_INTERNAL, "sketch without its genome record table"); }
  SketchFileHeader h; memset(&h, 0, sizeof h);
  memcpy(h.magic, "ANISKTCH", 8); h.version = 1; h.headerBytes = (uint32_t)kFileAlign;
  h.kmerSize = sk->params.kmerSize; h.windowSize = sk->params.windowSize; h.fragLen = sk->params.fragLen; h.percentageIdentity = sk->params.percentageIdentity;
  h.nContigs = sk->nContigs; h.nGenomes = sk->nGenomes; h.nRecords = sk->n;
  h.offContigLen = kFileAlign;
  h.offGcs = align_up(h.offContigLen + (uint64_t)sk->nContigs * 4);
  h.offGenomeRec = align_up(h.offGcs + ((uint64_t)sk->nGenomes + 1) * 4);
  h.offNames = align_up(h.offGenomeRec + ((uint64_t)sk->nGenomes + 1) * 8);
  h.namesBytes = names.size();
  h.offRecords = align_up(h.offNames + names.size());
  uint64_t pos = 0;
  bool ok = write_all(fd, &h, sizeof h); pos += sizeof h;
  ok = ok && pad_to(fd, &pos, h.offContigLen) && write_all(fd, sk->contigLen.data(), (size_t)sk->nContigs * 4); pos += (uint64_t)sk->nContigs * 4;
  ok = ok && pad_to(fd, &pos, h.offGcs) && write_all(fd, sk->genomeContigStart.data(), ((size_t)sk->nGenomes + 1) * 4); pos += ((uint64_t)sk->nGenomes + 1) * 4;
  ok = ok && pad_to(fd, &pos, h.offGenomeRec) && write_all(fd, genomeRec.data(), genomeRec.size() * 8); pos += genomeRec.size() * 8;
  ok = ok && pad_to(fd, &pos, h.offNames) && write_all(fd, names.data(), names.size()); pos += names.size();
  ok = ok && pad_to(fd, &pos, h.offRecords);
  // records: joined on the device into 12-byte records, through page-locked staging, 64 M records at a time
  const size_t kPiece = (size_t)64 << 20;
  for (const IndexChunk *ch : sk->chunks) {
    if (!ch->resident) {                     // streamed set: written from the records the sketch keeps
      for (const RecordPiece &pc : ch->pieces)
        for (size_t o = 0; ok && o < pc.n; o += kPiece) {
          const size_t m = std::min<size_t>(kPiece, pc.n - o);
          void *host = nullptr;
          if (pinned_buffer(ctx, 0, m * 12, &host) != ANI_OK) { ok = false; break; }
          ok = hipMemcpy(host, pc.rec + 3 * o, m * 12, hipMemcpyDeviceToHost) == hipSuccess && write_all(fd, host, m * 12);
        }
      continue;
    }
    for (size_t o = 0; ok && o < ch->n; o += kPiece) {
      const size_t m = std::min<size_t>(kPiece, ch->n - o);
      uint32_t *tmp = nullptr; void *host = nullptr;
      if (pool_malloc((void **)&tmp, m * 12) != hipSuccess || pinned_buffer(ctx, 0, m * 12, &host) != ANI_OK) { if (tmp) pool_free(tmp); ok = false; break; }
      hipLaunchKernelGGL(ani::k_index_join, dim3(grid_for(m, 256, 65535u * 8u)), dim3(256), 0, ctx->stream, ch->mHash + o, ch->mSeq + o, ch->mWpos + o, (uint32_t)m, (uint32_t)ch->c0, tmp);
      hipError_t e = hipMemcpyAsync(host, tmp, m * 12, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      pool_free(tmp);
      ok = e == hipSuccess && write_all(fd, host, m * 12);
    }
  }
  if (::close(fd) != 0) ok = false;
  if (!ok) { ::unlink(path); return fail(ANI_ERR_DEVICE, "writing %s failed", path); }
  return ANI_OK;
}

// genomes [g0, g1) of a sketch file (g1 < 0: to the end) -> a sketch on this context; seqIds / genome ids are renumbered from 0
int ani_sketch_load(ani_ctx *ctx, const char *path, int32_t g0, int32_t g1, ani_sketch **out)
{
  if (!ctx || !path || !out) return fail(ANI_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) return fail(ANI_ERR_ARG, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < sizeof(SketchFileHeader)) { ::close(fd); return fail(ANI_ERR_ARG, "%s is not a sketch file", path); }
  const size_t fileBytes = (size_t)st.st_size;
  const uint8_t *base = (const uint8_t *)mmap(nullptr, fileBytes, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (base == MAP_FAILED) return fail(ANI_ERR_NOMEM, "mmap of %s failed", path);
  struct Unmap { const uint8_t *p; size_t n; ~Unmap() { munmap((void *)p, n); } } unmap{base, fileBytes};
  SketchFileHeader h; memcpy(&h, base, sizeof h);
  if (memcmp(h.magic, "ANISKTCH", 8) != 0 || h.version != 1) return fail(ANI_ERR_ARG, "%s is not a version-1 sketch file", path);
  // section extents without overflow: a section [off, off + count * size) lies inside the file iff off <= fileBytes and count <= (fileBytes - off) / size
  auto inside = [&](uint64_t off, uint64_t count, uint64_t size) { return off <= fileBytes && count <= (fileBytes - off) / size; };
  if (h.nContigs < 0 || h.nGenomes < 0 || !inside(h.offRecords, h.nRecords, 12) || !inside(h.offNames, h.namesBytes, 1) ||
      !inside(h.offGenomeRec, (uint64_t)h.nGenomes + 1, 8) || !inside(h.offGcs, (uint64_t)h.nGenomes + 1, 4) || !inside(h.offContigLen, (uint64_t)h.nContigs, 4) ||
      (h.offRecords | h.offGenomeRec | h.offGcs | h.offContigLen) % 4 != 0 || h.offGenomeRec % 8 != 0)
    return fail(ANI_ERR_ARG, "%s is truncated", path);
  {
    // the tables are used as copy offsets and sizes below: validate them instead of trusting the file
    const int32_t *gcsT = (const int32_t *)(base + h.offGcs), *clenT = (const int32_t *)(base + h.offContigLen);
    const uint64_t *grecT = (const uint64_t *)(base + h.offGenomeRec);
    bool ok = gcsT[0] == 0 && gcsT[h.nGenomes] == h.nContigs && grecT[0] == 0 && grecT[h.nGenomes] == h.nRecords;
    for (int32_t g = 0; ok && g < h.nGenomes; g++) ok = gcsT[g] <= gcsT[g + 1] && grecT[g] <= grecT[g + 1];
    for (int32_t c = 0; ok && c < h.nContigs; c++) ok = clenT[c] >= 0;
    if (!ok) return fail(ANI_ERR_ARG, "%s has inconsistent genome / contig tables", path);
  }
  if (g1 < 0) g1 = h.nGenomes;
  if (g0 < 0 || g0 > g1 || g1 > h.nGenomes) return fail(ANI_ERR_ARG, "genome range [%d, %d) outside the file's %d genomes", g0, g1, h.nGenomes);
  ani_params_t p; p.kmerSize = h.kmerSize; p.windowSize = h.windowSize; p.fragLen = h.fragLen; p.percentageIdentity = h.percentageIdentity;
  TRY(check_params(&p));
  const int32_t *gcsF = (const int32_t *)(base + h.offGcs), *clenF = (const int32_t *)(base + h.offContigLen);
  const uint64_t *grec = (const uint64_t *)(base + h.offGenomeRec);
  const int32_t c0 = gcsF[g0], c1 = gcsF[g1], nG = g1 - g0;
  std::vector<int32_t> gcs((size_t)nG + 1);
  for (int32_t g = 0; g <= nG; g++) gcs[g] = gcsF[g0 + g] - c0;
  ani_sketch *sk = new_sketch(ctx, &p, clenF + c0, c1 - c0, gcs.data(), nG);
  { const char *nm = (const char *)(base + h.offNames), *end = nm + h.namesBytes;
    for (int32_t g = 0; g < h.nGenomes && nm < end; g++) { const size_t l = strnlen(nm, (size_t)(end - nm)); if (g >= g0 && g < g1) sk->genomeNames.emplace_back(nm, l); nm += l + 1; } }
  // records to the device in pieces of whole genomes (<= 64 M records), seqIds rebased to the range's first contig
  std::vector<RecordPart> parts;
  auto bail = [&](int rc) { for (auto &q : parts) if (q.rec) pool_free(q.rec); free_sketch_device(sk); delete sk; return rc; };
  const uint64_t kPiece = (uint64_t)64 << 20;
  int32_t ga = g0;
  while (ga < g1) {
    int32_t gb = ga + 1;
    while (gb < g1 && grec[gb + 1] - grec[ga] <= kPiece) gb++;
    RecordPart pt; pt.g0 = ga - g0; pt.g1 = gb - g0; pt.n = (size_t)(grec[gb] - grec[ga]);
    if (pt.n) {
      if (pool_malloc((void **)&pt.rec, pt.n * 12) != hipSuccess) return bail(fail(ANI_ERR_NOMEM, "device allocation of %zu records failed", pt.n));
      parts.push_back(pt);
      hipError_t e = hipMemcpyAsync(pt.rec, base + h.offRecords + grec[ga] * 12, pt.n * 12, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess && c0) hipLaunchKernelGGL(ani::k_records_rebase, dim3(grid_for(pt.n, 256, 65535u * 8u)), dim3(256), 0, ctx->stream, pt.rec, (uint64_t)pt.n, c0);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) return bail(fail(ANI_ERR_DEVICE, "copying records to the device failed: %s", hipGetErrorString(e)));
    } else parts.push_back(pt);
    ga = gb;
  }
  if (parts.empty()) { RecordPart pt; pt.g0 = 0; pt.g1 = nG; parts.push_back(pt); }
  const int rc = add_chunks(ctx, sk, parts);
  for (auto &q : parts) if (q.rec) { pool_free(q.rec); q.rec = nullptr; }
  if (rc != ANI_OK) return bail(rc);
  *out = sk;
  return ANI_OK;
}

int ani_sketch_file_info(const char *path, ani_params_t *p, int32_t *nContigs, int32_t *nGenomes, uint64_t *nMinimizers)
{
  if (!path) return fail(ANI_ERR_ARG, "null argument");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(ANI_ERR_ARG, "cannot open %s", path);
  SketchFileHeader h;
  const size_t got = fread(&h, 1, sizeof h, f);
  fclose(f);
  if (got != sizeof h || memcmp(h.magic, "ANISKTCH", 8) != 0 || h.version != 1) return fail(ANI_ERR_ARG, "%s is not a version-1 sketch file", path);
  if (p) { p->kmerSize = h.kmerSize; p->windowSize = h.windowSize; p->fragLen = h.fragLen; p->percentageIdentity = h.percentageIdentity; }
  if (nContigs) *nContigs = h.nContigs;
  if (nGenomes) *nGenomes = h.nGenomes;
  if (nMinimizers) *nMinimizers = h.nRecords;
  return ANI_OK;
}

// genome name / contig lengths of a (loaded) sketch: what the command line needs to print results without the FASTA files
const char *ani_sketch_genome_name(const ani_sketch *sk, int32_t g) { return (sk && g >= 0 && g < (int32_t)sk->genomeNames.size()) ? sk->genomeNames[g].c_str() : ""; }
int ani_sketch_tables(const ani_sketch *sk, const int32_t **contigLen, const int32_t **genomeContigStart)
{
  if (!sk) return fail(ANI_ERR_ARG, "null sketch");
  if (contigLen) *contigLen = sk->contigLen.data();
  if (genomeContigStart) *genomeContigStart = sk->genomeContigStart.data();
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
    if (i && (mappings[i - 1].querySeqId > a.querySeqId || (mappings[i - 1].querySeqId == a.querySeqId && mappings[i - 1].refSeqId > a.refSeqId))) ordered = false;
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
    int rc = ani_sort_pairs_u64_u32(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), ctx->keepFlags.as<uint32_t>(), ctx->keepOff.as<uint32_t>(), n, 64, nullptr, &tb, ctx->stream);
    if (rc == 0) { TRY(ctx->sortTmp.ensure(tb + 256)); rc = ani_sort_pairs_u64_u32(ctx->l1BigHitsA.as<uint64_t>(), ctx->l1BigHitsB.as<uint64_t>(), ctx->keepFlags.as<uint32_t>(), ctx->keepOff.as<uint32_t>(), n, 64, ctx->sortTmp.p, &tb, ctx->stream); }
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

int ani_synth_packed(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, void *devOut)
{
  return ani_synth_packed_clusters(ctx, seed, variant, firstGenomeId, nGenomes, genomeLen, 20, devOut);
}

int ani_synth_packed_clusters(ani_ctx *ctx, uint64_t seed, uint64_t variant, int32_t firstGenomeId, int32_t nGenomes, int32_t genomeLen, int32_t clusterSize, void *devOut)
{
  if (!ctx || !devOut || nGenomes < 0 || genomeLen <= 0 || firstGenomeId < 0 || clusterSize < 1) return fail(ANI_ERR_ARG, "invalid argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t words = (size_t)nGenomes * (((size_t)genomeLen + 15) / 16);
  if (words == 0) return ANI_OK;
  hipLaunchKernelGGL(ani::k_synth_packed, dim3(grid_for(words, 256, 65535)), dim3(256), 0, ctx->stream, seed, variant, firstGenomeId, nGenomes, genomeLen, clusterSize, (uint32_t *)devOut);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ANI_OK;
}

}  // extern "C"

