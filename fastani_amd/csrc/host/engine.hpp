// engine.hpp — what the host-side units of libfastani_amd.so share: the device allocator, the context, the reference set with its
// index chunks, fragment sets, and the functions one unit calls in another.  The library is built from
//   engine_core.hip    life cycle, allocator, counters, timers, prefix sum, host scalars (skch::Stat), synthetic genomes
//   engine_ingest.hip  sequence batches: classify, 2-bit pack, upload                                  
//   engine_sketch.hip  minimizer records and fragment sketches (≙ Sketch::build, Map::doL1Mapping's sketch), kept fragment sets + wire format
//   engine_index.hip   index chunks (≙ Sketch::index), chunk streaming, sketch file                    (winSketch.hpp:181-193; section 8 f3)
// engine_map.hip     L1 + L2 + identity + reducer (≙ Map::mapQuery ... cgi::computeCGI)              (computeMap.hpp:112-545, computeCoreIdentity.hpp:166-298)
//   engine_l2.hip      the launches of k_l2_codes / k_l2_sim (a unit of its own for its compiler flags: build_lib.sh)
//   sort_device.hip    the radix sort
// Kernels live in kernels/*.hpp (internal linkage: a unit compiles the ones it launches).  Nothing here crosses the C-ABI.
#pragma once
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
#include <condition_variable>
#include <functional>
#include <memory>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/ani_abi.h"
#include "stats.hpp"
#include "../kernels/common.hpp"

// kernels/radix.hpp through sort_device.hip (hand-written LSD radix sort; two-phase calls: tmp == nullptr returns the workspace size)
extern "C" int ani_sort_keys_u64_bits(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream);
extern "C" int ani_sort_keys_u64_range(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int beginBit, int endBit, void *tmp, size_t *tmpBytes,
    hipStream_t stream, int async);
extern "C" int ani_sort_pairs_u64_u32(const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                      size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream);
extern "C" int ani_sort_index(const void *const *pieceRec, const size_t *pieceN, int nPieces, uint32_t seqBase, size_t n,
                              uint32_t *mHash, int32_t *mSeq, int32_t *mWpos, uint32_t *tmpK, uint64_t *tmpV, uint32_t *sHash, uint64_t *sSW,
                              void *tmp, size_t *tmpBytes, hipStream_t stream, hipEvent_t soaReady, hipStream_t sideStream);
extern "C" int ani_sort_check(void *tmp, hipStream_t stream);

namespace ani { struct TableSlot; }

namespace anih {

extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);

// -----------------------------------------------------------------------------------------------------
// Device allocator (round 6: one arena per device instead of two caches of whole blocks).
//
// Why: fresh device memory costs up to 20 - 40 us per MB on some hosts (a 4.5 GB hipMalloc: 135 ms) and hipFree ~40 ms per GB; a
// many-to-many run builds and drops same-sized index arrays over and over, and a cold command-line run needs ~29 GB once.  The
// round-2..5 pools kept freed blocks WHOLE and reused one only for a request within 6 % of its size, so the 11.8 GB the index build
// hands back (sort buffers, link candidates, minimizer records) could not serve the 3.2 GB of mapping buffers asked for a moment
// later: that was fresh memory again (profiles/r06a_e2e_pool_trace.txt).
//
// Now: memory is taken from the driver in SEGMENTS and handed out as EXTENTS of them — best fit, split, coalesced with free
// neighbours on release — so anything that is free can serve anything that fits.
//   * large requests (>= 32 MiB): a miss takes a fresh segment of exactly the request's size (size classes as before), so the hole a
//     dropped index array leaves is the hole the next build's array of that size fits;
//   * small requests share 256 MiB segments of their own and never chip at a large hole;
//   * the device is full: wholly free segments go back to the driver one at a time, largest first, until the request fits;
//   * reserve(): a segment made on ANOTHER thread ahead of need (ani_pool_prewarm_index: the command line reserves what sketching and
//     indexing its input will take while the readers parse the first files); a request that only a promised segment can serve waits
//     for it — up to a deadline — instead of taking fresh memory beside it.
// The lock is not held while the driver works.  ANI_POOL_TRACE=1 prints every segment taken from the driver.
// -----------------------------------------------------------------------------------------------------
struct DevicePool {
  static constexpr size_t kGran = 512, kSmallLimit = (size_t)32 << 20, kSmallSegment = (size_t)256 << 20;
  struct Ext { size_t size; bool free; char *seg; };
  struct Seg { size_t size; bool small; };
  std::mutex mu;
  std::map<char *, Ext> ext;                               // every extent of every segment, by address
  std::multimap<size_t, char *> freeBySize[2];             // free extents: [0] large segments, [1] small segments
  std::map<char *, Seg> segs;
  size_t cachedBytes = 0, liveBytes = 0, segBytes = 0;     // cachedBytes = free bytes inside the segments
  uint64_t freshCalls = 0, freshBytes = 0; double freshMs = 0;   // what the driver was asked for so far (ani_pool_stats)
  std::multiset<size_t> promised; std::condition_variable cvPromised;

  // the size a request of `bytes` is served with: large blocks in classes of 1/32 .. 1/64 of their size (<= 3 % slack) — the index
  // chunks of a streamed reference set are cut at genome borders and differ by a few 10^-4 of their size; with classes the hole one
  // leaves fits the next (profiles/r04c5b: without them the 39 chunk builds of the 90 000-genome run spent as long in fresh memory
  // and hipFree as in kernels)
  static size_t class_size(size_t bytes)
  {
    if (bytes == 0) bytes = 1;
    static const bool classes = !(getenv("ANI_TEST_POOL_CLASSES") && !strcmp(getenv("ANI_TEST_POOL_CLASSES"), "0"));
    if (classes && bytes >= ((size_t)64 << 20)) {
      int lg = 63; while (!((bytes >> lg) & 1)) lg--;
      const size_t gran = (size_t)1 << (lg - 5);
      bytes = (bytes + gran - 1) / gran * gran;
    }
    return (bytes + kGran - 1) / kGran * kGran;
  }
  void free_insert_locked(char *a, const Ext &e) { freeBySize[segs[e.seg].small ? 1 : 0].emplace(e.size, a); }
  void free_erase_locked(char *a, const Ext &e)
  {
    auto &m = freeBySize[segs[e.seg].small ? 1 : 0];
    for (auto r = m.equal_range(e.size); r.first != r.second; ++r.first) if (r.first->second == a) { m.erase(r.first); return; }
  }
  // hands out the first `bytes` of the free extent at `a`; the rest stays free
  void *take_locked(char *a, size_t bytes)
  {
    auto it = ext.find(a);
    Ext e = it->second;
    free_erase_locked(a, e);
    if (e.size > bytes) {
      const Ext rest{e.size - bytes, true, e.seg};
      ext.emplace(a + bytes, rest); free_insert_locked(a + bytes, rest);
      it->second.size = bytes;
    }
    it->second.free = false;
    cachedBytes -= it->second.size; liveBytes += it->second.size;
    return a;
  }
  void add_segment_locked(char *p, size_t bytes, bool small)
  {
    segs[p] = Seg{bytes, small}; segBytes += bytes;
    const Ext e{bytes, true, p};
    ext.emplace(p, e); free_insert_locked(p, e); cachedBytes += bytes;
  }
  // gives the largest wholly free segment back to the driver; false if there is none
  bool free_largest_locked(size_t *freedBytes)
  {
    char *best = nullptr; size_t bestSize = 0;
    for (auto &kv : segs) {
      auto it = ext.find(kv.first);
      if (it != ext.end() && it->second.free && it->second.size == kv.second.size && kv.second.size > bestSize) { best = kv.first; bestSize = kv.second.size; }
    }
    if (!best) return false;
    free_erase_locked(best, ext[best]); ext.erase(best);
    segs.erase(best); segBytes -= bestSize; cachedBytes -= bestSize;
    (void)hipFree(best);
    if (freedBytes) *freedBytes += bestSize;
    return true;
  }
  bool free_largest() { std::lock_guard<std::mutex> g(mu); return free_largest_locked(nullptr); }
  bool promised_fit_locked(size_t bytes) const { return !promised.empty() && *promised.rbegin() >= bytes; }

  hipError_t alloc(void **out, size_t bytes)
  {
    bytes = class_size(bytes);
    const bool small = bytes < kSmallLimit;
    // ANI_TEST_POOL_POISON=<byte> (tests): every block handed out is filled with that byte first, so that a kernel which reads memory
    // it (or an earlier kernel) has not written sees the same garbage every time instead of whatever the previous owner left
    static const int poison = getenv("ANI_TEST_POOL_POISON") ? (int)strtol(getenv("ANI_TEST_POOL_POISON"), nullptr, 0) : -1;
    static const bool trace = getenv("ANI_POOL_TRACE") != nullptr;
    auto handout = [&](void *p) {
      *out = p;
      if (poison >= 0) { (void)hipDeviceSynchronize(); (void)hipMemset(p, poison, bytes); (void)hipDeviceSynchronize(); }
      return hipSuccess;
    };
    std::unique_lock<std::mutex> g(mu);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(10);
    for (;;) {
      auto &m = freeBySize[small ? 1 : 0];
      auto it = m.lower_bound(bytes);
      if (it != m.end()) { void *p = take_locked(it->second, bytes); g.unlock(); return handout(p); }
      // a segment that will fit is being made on another thread: its tail is shorter than a second fresh allocation beside it
      if (!small && promised_fit_locked(bytes) && std::chrono::steady_clock::now() < deadline) { cvPromised.wait_for(g, std::chrono::milliseconds(5));
        continue; }
      break;
    }
    const size_t segSize = small ? kSmallSegment : bytes;
    g.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, segSize);
    size_t freed = 0;
    g.lock();
    if (e != hipSuccess) {
      // the device is full: wholly free segments go back one at a time, largest first (hipFree costs ~40 ms per GB on this stack: a
      // 125 GB cache dropped at once took 4.8 s in the warm step of the 10 000 x 10 000 run, profiles/r03l) — but first another thread
      // may have released something that fits in the meantime
      (void)hipGetLastError();
      auto &m = freeBySize[small ? 1 : 0];
      auto it = m.lower_bound(bytes);
      if (it != m.end()) { void *q = take_locked(it->second, bytes); g.unlock(); return handout(q); }
      while (e != hipSuccess && free_largest_locked(&freed)) { e = hipMalloc(&p, segSize); if (e != hipSuccess) (void)hipGetLastError(); }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (trace) fprintf(stderr, "[ani pool] hipMalloc %.1f MB: %.2f ms%s (%.1f MB free in %zu segments of %.1f MB%s)\n", segSize / 1048576.0, ms,
        e == hipSuccess ? "" : " FAILED",
                       cachedBytes / 1048576.0, segs.size(), segBytes / 1048576.0, freed ? "; device full: free segments returned" : "");
    if (e != hipSuccess) return e;
    freshCalls++; freshBytes += segSize; freshMs += ms;
    add_segment_locked((char *)p, segSize, small);
    void *q = take_locked((char *)p, bytes);
    g.unlock();
    return handout(q);
  }
  // false: not a block of this pool
  bool release(void *p)
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = ext.find((char *)p);
    if (it == ext.end() || it->second.free) return false;
    it->second.free = true;
    cachedBytes += it->second.size; liveBytes -= it->second.size;
    auto nx = std::next(it);
    if (nx != ext.end() && nx->second.free && nx->second.seg == it->second.seg) { free_erase_locked(nx->first, nx->second);
      it->second.size += nx->second.size; ext.erase(nx); }
    if (it != ext.begin()) {
      auto pv = std::prev(it);
      if (pv->second.free && pv->second.seg == it->second.seg) { free_erase_locked(pv->first, pv->second); pv->second.size += it->second.size; ext.erase(it);
        it = pv; }
    }
    free_insert_locked(it->first, it->second);
    return true;
  }
  // Reserving (ani_pool_prewarm_index): a fresh large segment of an announced size, touched (hipMemset on `s`) and entered as free.
  void promise(size_t bytes) { std::lock_guard<std::mutex> g(mu); promised.insert(class_size(bytes)); }
  hipError_t reserve(size_t bytes, hipStream_t s)           // one promised segment
  {
    bytes = class_size(bytes);
    static const bool trace = getenv("ANI_POOL_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) { e = hipMemsetAsync(p, 0, bytes, s); if (e == hipSuccess) e = hipStreamSynchronize(s); if (e != hipSuccess) { (void)hipFree(p);
      p = nullptr; } }
    else (void)hipGetLastError();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = promised.find(bytes); if (it != promised.end()) promised.erase(it);
      if (p) { add_segment_locked((char *)p, bytes, false); freshCalls++; freshBytes += bytes; freshMs += ms; }
    }
    if (trace) fprintf(stderr, "[ani pool] reserved %.1f MB ahead of need: %.2f ms%s\n", bytes / 1048576.0, ms, p ? "" : " FAILED");
    cvPromised.notify_all();
    return e;
  }
  void drop_promises(const std::vector<size_t> &sizes)      // (error path: nothing more will come)
  {
    { std::lock_guard<std::mutex> g(mu); for (size_t b : sizes) { auto it = promised.find(class_size(b)); if (it != promised.end()) promised.erase(it); } }
    cvPromised.notify_all();
  }
  void trim() { std::lock_guard<std::mutex> g(mu); while (free_largest_locked(nullptr)) {} }
};

// One pool per device ordinal (one process normally drives one GPU).  (`cls` of the earlier two-pool design — 0: what lives and dies
// with a sketch, 1: the grow-only buffers of a context — is kept in the signatures as documentation of the caller's intent.)
extern DevicePool g_pools[64];
inline DevicePool &cur_pool(int /*cls*/ = 0) { int d = 0; (void)hipGetDevice(&d); return g_pools[(d < 0 || d >= 64) ? 0 : d]; }
inline hipError_t pool_malloc(void **p, size_t bytes, int /*cls*/ = 0) { return cur_pool().alloc(p, bytes); }
inline void pool_free(void *p)
{
  if (!p) return;
  if (!cur_pool().release(p)) (void)hipFree(p);
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

// host-side parallel loop for the ingest path (ANI_HOST_THREADS overrides the thread count; small jobs stay serial).
// The workers are started once and live as long as the process: a slice of the command line's input is one call, and sixty-four
// thread creations per call — each an mmap and an munmap of a stack, in an address space whose reader threads map and unmap sequence
// buffers all the time — cost more than the packing they were created for.
struct HostPool {
  std::mutex mu, runMu; std::condition_variable cvWork, cvDone;
  std::function<void(size_t)> job; size_t n = 0; std::atomic<size_t> next{0}; unsigned wanted = 0, started = 0, active = 0; uint64_t epoch = 0;
  std::vector<std::thread> th;
  void worker(unsigned id)
  {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cvWork.wait(lk, [&]() { return epoch != seen && id < wanted; });
        seen = epoch;
      }
      for (size_t i; (i = next.fetch_add(1)) < n;) job(i);
      { std::lock_guard<std::mutex> lk(mu); active--; }
      cvDone.notify_all();
    }
  }
  void run(size_t count, unsigned nt, const std::function<void(size_t)> &f)
  {
    std::lock_guard<std::mutex> one(runMu);             // one loop at a time (a second caller waits its turn)
    {
      std::lock_guard<std::mutex> lk(mu);
      while (started + 1 < nt) { const unsigned id = started++; th.emplace_back([this, id]() { worker(id); }); th.back().detach(); }
      job = f; n = count; next = 0; wanted = nt - 1; active = nt - 1; epoch++;
    }
    cvWork.notify_all();
    for (size_t i; (i = next.fetch_add(1)) < count;) f(i);                    // the caller works too
    std::unique_lock<std::mutex> lk(mu);
    cvDone.wait(lk, [&]() { return active == 0; });
    wanted = 0;
  }
};
inline HostPool &host_pool() { static HostPool *p = new HostPool(); return *p; }      // never destroyed: its threads are detached
template <class F>
void parallel_for(size_t n, uint64_t work, F f)
{
  // (beside 32 reader threads: 16-32 packers measured best, 64 starve the readers; profiles/r04ak_e2e_probe.txt)
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > 32) nt = 32;
  if (const char *ev = getenv("ANI_HOST_THREADS")) { int v = atoi(ev); if (v >= 1) nt = (unsigned)v; }
  if (nt > n) nt = (unsigned)n;
  uint64_t minWork = 1u << 22;
  // test knob: small inputs through the pool
  if (const char *ev = getenv("ANI_TEST_HOST_PAR_MIN_WORK")) { const long long v = atoll(ev); if (v >= 0) minWork = (uint64_t)v; }
  if (nt <= 1 || work < minWork) { for (size_t i = 0; i < n; i++) f(i); return; }
  host_pool().run(n, nt, std::function<void(size_t)>(f));
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

}  // namespace anih
using namespace anih;

// =====================================================================================================
struct ani_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t subBatchFragments = 1u << 20, subBatchBinBytes = (uint64_t)8 << 30;   // sub-batch bounds of ani_map_cgi_batch (env ANI_SUBBATCH_FRAGS)
  size_t l2ChunkCandidates = (size_t)1 << 21;                                      // L2 chunk size (env ANI_TEST_L2_CHUNK, tests)
  // 16-bit code entries per L2 chunk (32-bit offsets; env ANI_TEST_L2_CODE_LIMIT, tests)
  uint64_t l2CodeLimit = 0xfffffff0ull;
  // minimizers per index chunk (env ANI_MAX_INDEX_MINIMIZERS); indices are 32 bit
  uint64_t maxIndexMinimizers = 1700000000ull;
  // (engine_map.hip checks them against kernels/l1.hpp)           // env ANI_TEST_L1_FILTER_MIN / ANI_TEST_L1_LDS_MAX, read by ani_init (tests: per engine, not
  // per process)
  int l1FilterMin = 300 /* ani::kL1FilterMinHits */, l1LdsMax = 4080 /* ani::kL1HitCapMax */;
  // first guess of the same-hash link list (env ANI_TEST_DUP_PAIR_CAP, tests: forces the rerun)
  uint64_t dupPairCap = 0;
  // env ANI_TEST_L1_TINY=0: fragments with <= 64 seed hits take the workgroup path like the others
  bool l1Tiny = true;
  // the L2 simulation on the side stream, beside the next chunk's ranges / codes kernels (env ANI_TEST_L2_OVERLAP=0 switches it off; see the L2 loop)
  bool l2Overlap = true;
  // seed hits per fragment and index chunk (32-bit hit offsets; env ANI_TEST_L1_HIT_LIMIT, tests)
  uint64_t l1HitLimit = 0x7ffffff0ull;
  // floor of the L1 candidate pool, per stripe (env ANI_TEST_CAND_POOL_MIN, tests: forces the retry path)
  uint64_t candPoolMin = 4096;
  // seed hits / fragments per group of the batched global-memory L1 path (env ANI_TEST_L1_BIG_GROUP_HITS / _FRAGS, tests)
  uint64_t l1BigGroupHits = 1ull << 27, l1BigGroupFrags = 1ull << 20;
  // index chunks of one reference set kept on the device (env ANI_MAX_RESIDENT_CHUNKS; 0 = decide from the free memory)
  int32_t maxResidentChunks = 0;
  // chunk size once a set is streamed (env ANI_STREAM_CHUNK_MINIMIZERS): the build's transient arrays must fit beside the records
  uint64_t streamChunkMinimizers = 1000000000ull;
  std::vector<std::unique_ptr<ani::stat::Luts>> lutCache;
  // page-locked staging: 0/1 result reads, 2 prefix-sum block totals, 3/4 ingest (packed / raw bytes)
  void *pinned[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; size_t pinnedCap[5] = {0, 0, 0, 0, 0};
  hipStream_t stream2 = nullptr;          // side stream: latency-bound launches that can run under the main simulation kernel
  // evIndex: build_chunk_index's own (sort -> side work -> join)
  hipEvent_t evSimA[2] = {nullptr, nullptr}, evSetDone[2] = {nullptr, nullptr}, evIndex[2] = {nullptr, nullptr};
  // stage timers: event pairs are recorded as the launches go out and read back lazily (flush_timers), never by blocking the host
  std::vector<hipEvent_t> timerEvents; size_t timerUsed = 0;
  struct PendingTimer { size_t a, b; double *acc; };
  std::vector<PendingTimer> timerPending;
  ani_counters_t counters;
  std::vector<unsigned long long> hostCounters;                 // read_counters: the raw block
  unsigned long long poolUsed[3] = {0, 0, 0}, poolMaxStripe[3] = {0, 0, 0};
  double candPerFrag = 12.0;      // estimate that sizes the L1 candidate pool: the LARGEST need of the last 32 batches that fill the pool stripes evenly
  // (a ring of fragment sets alternates between sets with ~16 candidates per fragment — the rank's own genomes — and sets
  double candSeen[32] = {0}; int candSeenAt = 0;
                                                    //  with < 1: an estimate that follows the batches down runs the L1 kernels twice for every dense one)
  // ... and what the last batch too small for that estimate needed (a few hundred fragments against a species-dense index, call after call)
  uint64_t smallBatchCandCap = 0;
  // scalar device counters (array of 16 x u64)
  DevBuf dCounters;
  // workspaces reused across calls
  DevBuf seqPacked, seqAscii, contigOff, contigLen, contigMode;
  DevBuf sortTmp, unitStart, unitAux, tiles, tileInfo, tileMeta, tileCnt, tileDrop, tileOff, poolHash, poolWpos;
  DevBuf scanTmpA, scanTmpB, scanTmpC, scanTmpD;
  DevBuf frags, fragOff, fragS, fragGenome, fragQSeq, qPool;
  DevBuf l1FragDesc;                      // per fragment, in processing order: what the L1 gather kernels need of it (kernels/l1.hpp: L1FragDesc)
  DevBuf probeFirst, probeCnt, l1MidList, l1SmallList, l1BigList, l1BigHitsA, l1BigHitsB, l1BigV, l1BigTbl, l1BigHash, candFrag, candSeq, candStart, candEnd,
      fragCandOff, fragCandCnt, fragCandCntClamped, fragHits, fragOrdOff, fragOrder, fragOrderTmp;
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
  // added to refGenomeId in the CGI rows of the batch entry points (ani_sketch_set_ref_id_base: a shard or block of a larger set)
  int32_t refIdBase = 0;
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

namespace anih {

enum { CNT_POOL = 0, CNT_QPOOL = 1, CNT_CAND = 2, CNT_HITS = 3, CNT_ENTRIES = 4, CNT_STEPS = 5, CNT_ROWS = 6, CNT_UNIQ = 7,
       CNT_MAXS = 8 /* int */, CNT_NEG = 9 /* uint */, CNT_SUMQ = 10, CNT_REASON = 11 /* ..14 */, CNT_CLASSB = 15, CNT_LISTM = 16, CNT_LISTL = 17,
           CNT_LISTBIG = 18, CNT_ENTRIES_B = 19, CNT_SUMQ_B = 20, CNT_STEPS_B = 21, CNT_TINY = 22, CNT_SMALL = 23, CNT_N = 24 };


inline unsigned long long *cnt_ptr(ani_ctx *c, int i) { return c->dCounters.as<unsigned long long>() + i; }

static_assert(CNT_N == ani::kStatStripeWords, "counter block size");
// Device counter block: kStatStripes copies of the CNT_N counters (statistics are spread over the stripes by the kernels,
// stat_slot(), and summed here; list cursors live in stripe 0), then the striped pool cursors (common.hpp: pool_take): three pools
// (reference minimizers, fragment-sketch hashes, L1 candidates) x kPoolStripes cursors, one per 128-byte line.
enum { POOL_REF = 0, POOL_Q = 1, POOL_CAND = 2, POOL_N = 3 };
constexpr size_t kCursorWords = (size_t)POOL_N * ani::kPoolStripes * ani::kPoolStripeWords;
constexpr size_t kCounterWords = (size_t)ani::kStatStripes * CNT_N + kCursorWords;
inline unsigned long long *cur_ptr(ani_ctx *c,
    int pool) { return c->dCounters.as<unsigned long long>() + (size_t)ani::kStatStripes * CNT_N + (size_t)pool * ani::kPoolStripes * ani::kPoolStripeWords; }
int zero_counters(ani_ctx *c);
int zero_cursors(ani_ctx *c, int pool);
// host[]: the summed statistics (CNT_MAXS: the maximum over the stripes); c->poolUsed / c->poolMaxStripe: per pool, the entries
// requested in total and by the fullest stripe (the launch overflowed iff that exceeds the stripe capacity it was given)
int read_counters(ani_ctx *c, unsigned long long *host);
// entries per stripe for a pool meant to hold `cap` entries in all; and the capacity to retry with after an overflow
inline uint32_t stripe_cap(uint64_t cap) { return (uint32_t)((cap + ani::kPoolStripes - 1) / ani::kPoolStripes); }
inline uint64_t grown_cap(unsigned long long maxStripe) { return (uint64_t)(maxStripe + maxStripe / 16 + 64) * ani::kPoolStripes; }

// page-locked host staging buffer `slot`, at least `bytes` long (device->host copies into pageable memory crawl)
int pinned_buffer(ani_ctx *c, int slot, size_t bytes, void **out);
// HIP-event stage timer.  Both events are recorded on the stream the timed launches go to; the elapsed time is added to *acc when
// the timers are flushed (at a point where the host waits for the device anyway), so timing never serialises host and device.
void flush_timers(ani_ctx *c);
void add_counters(ani_counters_t *dst, const ani_counters_t *src);     // dst += src, field by field
size_t timer_event(ani_ctx *c);
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
int device_scan(ani_ctx *c, const int32_t *in, uint32_t *out, uint32_t n, uint64_t *total, uint64_t limit = 0xfffffff0ull);

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
int check_batch(const ani_seq_batch_t *b);
int check_params(const ani_params_t *p);
}  // namespace anih

// A batch of genomes in device memory that outlives the call that uploaded it (ani_batch_upload): the all-vs-all command line
// sketches it as references and maps it as queries without reading or packing the files a second time.
struct ani_dev_batch {
  ani_ctx *ctx = nullptr; int device = 0;
  DeviceBatch db;
  void *bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // packed, ascii, contigOff, contigLen, contigMode
};
struct ani_fragset;

namespace anih {
// genomes [g0, g1) of `b` -> device (packs pure-ACGT contigs to 2 bits per base on the way).  `keep` = allocate the device
// arrays for a persistent ani_dev_batch instead of using the context's staging buffers.
int upload_batch(ani_ctx *ctx, const ani_seq_batch_t *b, int32_t g0, int32_t g1, DeviceBatch *out, ani_dev_batch *keep = nullptr);

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
// `fused` (optional): the batch's genomes are queries as well — their fragment sketches come out of the same pass over the
// hashes (k_sketch_fused) into arrays owned by the caller's ani_fragset.  Only when a fragment plus its w-1 lead-in fits one tile.
struct FusedOut { FragSet *fs; FragArrays *arr; uint32_t **qPool; };
// A piece of a position-ordered record stream with global seqIds: n records on the device that belong to genomes [g0, g1).
struct RecordPart { uint32_t *rec = nullptr; size_t n = 0; int32_t g0 = 0, g1 = 0; bool owned = true; };
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

}  // namespace anih

// Fragment sketches of a batch of query genomes, kept on the device (ani_fragset_build / ani_sketch_records_self): mapping them
// against several reference sets or index chunks — or mapping genomes that were sketched as references — hashes nothing twice.
struct ani_fragset {
  ani_ctx *ctx = nullptr; int device = 0;
  ani_params_t params;
  FragSet fs;                      // host tables + device pointers into the arrays below
  FragArrays arr; uint32_t *qPool = nullptr;
  bool borrowed = false;           // the arrays live in a caller's buffer (ani_fragset_unpack): not freed with the set
  std::vector<int64_t> genomeFragStart;     // prefix of fs.genomeFragments
  std::vector<int32_t> genomeQueryId;       // merged sets (ani_fragset_unpack_merged): the query id of every genome, relative to the call's firstQueryId
  bool ownTables = false;                   // ... whose per-fragment tables are the set's own while the pool is borrowed
};


namespace anih {
// ---- engine_sketch.hip ----
int sketch_records(ani_ctx *ctx, const ani_params_t *p, const DeviceBatch &db, int32_t seqIdBase, uint32_t **dRecords, size_t *nOut, FusedOut *fused = nullptr);
int fragment_stage(ani_ctx *ctx, const ani_params_t &p, const DeviceBatch &db, FragSet *qr);
// ---- engine_index.hip ----
void free_chunk_index(IndexChunk *ch);
void free_chunk(IndexChunk *ch);
void free_sketch_device(ani_sketch *sk);
int upload_luts(ani_sketch *sk, int maxS);
ani_sketch *new_sketch(ani_ctx *ctx, const ani_params_t *p, const int32_t *contigLen, int32_t nContigs, const int32_t *genomeContigStart, int32_t nGenomes);
int ensure_chunk(ani_sketch *sk, size_t i, int pin = -1);
int add_chunks(ani_ctx *ctx, ani_sketch *sk, std::vector<RecordPart> &parts);
int exact_unique(ani_sketch *sk);

template <class T> int to_host_malloc(const std::vector<T> &v, T **out, size_t *n)
{
  *n = v.size();
  *out = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
  if (!*out) return fail(ANI_ERR_NOMEM, "host allocation failed");
  if (!v.empty()) memcpy(*out, v.data(), v.size() * sizeof(T));
  return ANI_OK;
}
}  // namespace anih
