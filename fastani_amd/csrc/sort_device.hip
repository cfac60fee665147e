// sort_device.hip — host side of the hand-written radix sort (kernels/radix.hpp): the hash-ordered half of the reference index
// (≙ filling minimizerPosLookupIndex, src/map/include/winSketch.hpp:181-193), the seed hits of the batched L1 path
// (src/map/include/computeMap.hpp:320) and the small orderings of the host orchestration.  No library sort is linked.
//
// Every entry point has the two-phase shape the caching allocator wants: tmp == nullptr returns the workspace size in *tmpBytes.
// Workspace = [digit histograms / offsets: 8 x 256 x u64][tile counters: 64 x u32][error flag][look-back status: tiles x 256 x u64]
// [ping-pong buffer for sorts of more than one pass].  The calls return when the sort is done (they check the error flag).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>

#include "kernels/radix.hpp"

namespace {
using namespace ani;

constexpr size_t kHistBytes = (size_t)kRadixMaxPasses * kRadixDigits * 8;
constexpr int kMaxLaunches = 1024;                          // tile counters: one per (pass, input buffer)
constexpr size_t kCtrBytes = (size_t)kMaxLaunches * 4 + 64;  // + the error flag (at the end) and padding

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
inline uint64_t tiles_of(uint64_t n) { return (n + kRadixTile - 1) / kRadixTile; }

struct Workspace {
  unsigned long long *hist; unsigned int *counters, *err; unsigned long long *status; unsigned char *pong;
  size_t statusBytes;
};
// `tiles`: look-back tiles of one pass (partial tiles at buffer ends included); `pongBytes`: ping-pong storage
size_t workspace_bytes(uint64_t tiles,
    size_t pongBytes) { return align256(kHistBytes) + align256(kCtrBytes) + align256(tiles * kRadixDigits * 8) + align256(pongBytes) + 256; }
Workspace carve(void *tmp, uint64_t tiles)
{
  Workspace w;
  unsigned char *p = (unsigned char *)(((uintptr_t)tmp + 255) / 256 * 256);
  w.hist = (unsigned long long *)p; p += align256(kHistBytes);
  w.counters = (unsigned int *)p; w.err = w.counters + kMaxLaunches; p += align256(kCtrBytes);
  w.status = (unsigned long long *)p; w.statusBytes = tiles * kRadixDigits * 8; p += align256(w.statusBytes);
  w.pong = p;
  return w;
}
int finish(const Workspace &w, hipStream_t stream)
{
  unsigned int err = 0;
  hipError_t e = hipMemcpyAsync(&err, w.err, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  // test knob: ANI_TEST_SORT_FAIL_EVERY=<k> makes every k-th completed sort of the process report a given-up look-back (k = 2: every
  // sort fails once and succeeds when it is repeated)
  if (const char *ev = getenv("ANI_TEST_SORT_FAIL_EVERY")) {
    static std::atomic<unsigned> done{0};
    const int k = atoi(ev);
    if (k > 0 && (done.fetch_add(1) + 1) % (unsigned)k == 0) return 9001;
  }
  return err ? 9001 : 0;                                    // a look-back gave up (kRadixSpinLimit): the output is incomplete
}
// A look-back that gives up (9001) is a scheduling surprise — the device shared with another process, a pre-empted queue, a debugger —
// not a property of the data: the sort is run ONCE more (histograms, counters, status words and the error flag are re-zeroed by every
// run; the inputs are only read) before the caller sees the error.
constexpr int kSortAttempts = 2;

// generic: keys (and values) in arrays, bits [beginBit, endBit) significant (the bits below beginBit travel with the key, unordered).
// `async`: the call returns with the passes in flight (no retry); the caller completes it with ani_sort_check on the same stream
// and must leave `tmp` alone until then.
template <class KeyT, class ValT>
int sort_arrays(const KeyT *keysIn, KeyT *keysOut, const ValT *valsIn, ValT *valsOut, size_t n, int beginBit, int endBit, void *tmp, size_t *tmpBytes,
    hipStream_t stream, bool async = false)
{
  constexpr bool kHasVal = !std::is_same<ValT, RadixNoVal>::value;
  if (endBit > (int)sizeof(KeyT) * 8) endBit = (int)sizeof(KeyT) * 8;
  if (beginBit < 0) beginBit = 0;
  if (endBit < beginBit + 1) endBit = beginBit + 1;
  const int P = (endBit - beginBit + kRadixBits - 1) / kRadixBits;
  const uint64_t tiles = tiles_of(n);
  const size_t keyBytes = align256(n * sizeof(KeyT)), valBytes = kHasVal ? align256(n * sizeof(ValT)) : 0;
  const size_t need = workspace_bytes(tiles, P > 1 ? keyBytes + valBytes : 0);
  if (!tmp) { *tmpBytes = need; return 0; }
  if (n == 0) return 0;
  if (n > 0xfffffff0ull) return 9002;
  if (*tmpBytes < need) return 9003;
  Workspace w = carve(tmp, tiles);
  KeyT *pongK = (KeyT *)w.pong; ValT *pongV = (ValT *)(w.pong + keyBytes);
  const bool rerunnable = (const void *)keysIn != (const void *)keysOut && (!kHasVal || (const void *)valsIn != (const void *)valsOut);
  int rc = 0;
  for (int attempt = 0; attempt < kSortAttempts; attempt++) {
    hipError_t e = hipMemsetAsync(w.hist, 0, align256(kHistBytes) + align256(kCtrBytes), stream);
    if (e != hipSuccess) return (int)e;
    const unsigned hg = (unsigned)std::min<uint64_t>((n + kTPB * 8 - 1) / (kTPB * 8), 4096);
    ArraySrc<KeyT, ValT> in{keysIn, valsIn};
    hipLaunchKernelGGL((k_radix_histogram<KeyT, ArraySrc<KeyT, ValT>>), dim3(hg ? hg : 1), dim3(kTPB), 0, stream, in, (uint64_t)n, beginBit, endBit, P,
        w.hist, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(kTPB), 0, stream, w.hist, P);
    for (int p = 0; p < P; p++) {
      const bool toOut = ((P - 1 - p) & 1) == 0;
      ArraySrc<KeyT, ValT> src = p == 0 ? in : (toOut ? ArraySrc<KeyT, ValT>{pongK, pongV} : ArraySrc<KeyT, ValT>{keysOut, valsOut});
      KeyT *dk = toOut ? keysOut : pongK; ValT *dv = toOut ? valsOut : pongV;
      e = hipMemsetAsync(w.status, 0, w.statusBytes, stream);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((k_radix_pass<KeyT, ValT, ArraySrc<KeyT, ValT>>), dim3((unsigned)tiles), dim3(kRadixTPB), 0, stream, src, dk, dv, (uint64_t)n,
          beginBit + p * kRadixBits, endBit,
                         (const unsigned long long *)(w.hist + p * kRadixDigits), w.status, w.counters + p, 0u, w.err);
    }
    if (async) return 0;
    rc = finish(w, stream);
    if (rc != 9001 || !rerunnable) break;
  }
  return rc;
}
}  // namespace

// (64-bit key, 32-bit payload), the low endBit bits of the key significant: the processing order of the fragments (key = running
// fragment id inside its genome) and foreign mapping lists handed to ani_compute_cgi (key = querySeqId << 32 | refSeqId)
extern "C" int ani_sort_pairs_u64_u32(const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                      size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  return sort_arrays<uint64_t, uint32_t>(keysIn, keysOut, valsIn, valsOut, n, 0, endBit, tmp, tmpBytes, stream);
}

// 64-bit keys, the low endBit bits significant (the batched L1 path packs (fragment, seqId, wpos) into as few bits as the index
// chunk needs)
extern "C" int ani_sort_keys_u64_bits(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int endBit, void *tmp, size_t *tmpBytes, hipStream_t stream)
{
  return sort_arrays<uint64_t, ani::RadixNoVal>(keysIn, keysOut, (const ani::RadixNoVal *)nullptr, (ani::RadixNoVal *)nullptr, n, 0, endBit, tmp, tmpBytes,
      stream);
}

// 64-bit keys ordered by their bits [beginBit, endBit) only; `async` != 0: the passes stay in flight, ani_sort_check(tmp, stream)
// completes the call.  (The same-hash half-records of the index: entry << 32 | kind << 31 | other are unique in (entry, kind), so the
// 31 low bits need no pass — four passes instead of eight, on the side stream.)
extern "C" int ani_sort_keys_u64_range(const uint64_t *keysIn, uint64_t *keysOut, size_t n, int beginBit, int endBit, void *tmp, size_t *tmpBytes,
    hipStream_t stream, int async)
{
  return sort_arrays<uint64_t, ani::RadixNoVal>(keysIn, keysOut, (const ani::RadixNoVal *)nullptr, (ani::RadixNoVal *)nullptr, n, beginBit, endBit, tmp,
      tmpBytes, stream, async != 0);
}

// The index sort (Sketch::index, winSketch.hpp:181-193).  Input: the chunk's minimizer records as `nPieces` device buffers of
// 12-byte records in position order (seqIds global: seqBase = first contig of the chunk).  Output: the position-ordered SoA arrays
// mHash / mSeq / mWpos (written by the histogram read) and the hash-ordered sHash / sSW (stable: every hash's occurrences stay in
// (seqId, wpos) order).  tmpK / tmpV: n-element ping-pong arrays.  soaReady / sideStream (optional): the event is recorded once the SoA
// arrays are written and sideStream is made to wait for it; the call then returns with the passes still running — the caller queues its
// side work and calls ani_sort_check.  Four 8-bit passes: SoA -> tmp -> (sHash, sSW) -> tmp -> (sHash, sSW).
extern "C" int ani_sort_index(const void *const *pieceRec, const size_t *pieceN, int nPieces, uint32_t seqBase, size_t n,
                              uint32_t *mHash, int32_t *mSeq, int32_t *mWpos, uint32_t *tmpK, uint64_t *tmpV, uint32_t *sHash, uint64_t *sSW,
                              void *tmp, size_t *tmpBytes, hipStream_t stream, hipEvent_t soaReady, hipStream_t sideStream)
{
  constexpr int P = 4;
  const uint64_t tiles = tiles_of(n);
  const size_t need = workspace_bytes(tiles, 0);
  if (!tmp) { *tmpBytes = need; return 0; }
  if (n == 0) return 0;
  if (n > 0x7ffffff0ull) return 9002;
  if (*tmpBytes < need) return 9003;
  if (nPieces + P > kMaxLaunches) return 9004;
  Workspace w = carve(tmp, tiles);
  const bool async = soaReady && sideStream;
  int rc = 0;
  // (a caller with a side stream repeats the call without one when ani_sort_check reports 9001)
  for (int attempt = 0; attempt < (async ? 1 : kSortAttempts); attempt++) {
  hipError_t e = hipMemsetAsync(w.hist, 0, align256(kHistBytes) + align256(kCtrBytes), stream);
  if (e != hipSuccess) return (int)e;
  size_t o = 0;
  for (int i = 0; i < nPieces; i++) {
    if (!pieceN[i]) continue;
    const unsigned hg = (unsigned)std::min<uint64_t>((pieceN[i] + kTPB * 8 - 1) / (kTPB * 8), 4096);
    RecordSrc src{(const uint32_t *)pieceRec[i], seqBase};
    hipLaunchKernelGGL((k_radix_histogram<uint32_t, RecordSrc>), dim3(hg ? hg : 1), dim3(kTPB), 0, stream, src, (uint64_t)pieceN[i], 0, 32, P, w.hist,
        mHash + o, mSeq + o, mWpos + o);
    o += pieceN[i];
  }
  // the SoA arrays are complete: work that only needs positions may start on the caller's side stream, underneath the passes
  if (async) {
    e = hipEventRecord(soaReady, stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(sideStream, soaReady, 0);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(kTPB), 0, stream, w.hist, P);
  // pass 0 reads the SoA arrays the histogram pass has just written (three coalesced streams; the 12-byte records would be three
  // strided ones), so the record buffers are read exactly once
  e = hipMemsetAsync(w.status, 0, w.statusBytes, stream);
  if (e != hipSuccess) return (int)e;
  int launch = 0;
  {
    SoaSrc src{mHash, mSeq, mWpos};
    hipLaunchKernelGGL((k_radix_pass<uint32_t, uint64_t, SoaSrc>), dim3((unsigned)tiles_of(n)), dim3(kRadixTPB), 0, stream, src, tmpK, tmpV, (uint64_t)n, 0, 32,
                       (const unsigned long long *)w.hist, w.status, w.counters + launch, 0u, w.err);
    launch++;
  }
  for (int p = 1; p < P; p++) {
    const bool toOut = (p & 1) == 1;                         // tmp -> S -> tmp -> S
    ArraySrc<uint32_t, uint64_t> src = toOut ? ArraySrc<uint32_t, uint64_t>{tmpK, tmpV} : ArraySrc<uint32_t, uint64_t>{sHash, sSW};
    e = hipMemsetAsync(w.status, 0, w.statusBytes, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_radix_pass<uint32_t, uint64_t, ArraySrc<uint32_t, uint64_t>>), dim3((unsigned)tiles_of(n)), dim3(kRadixTPB), 0, stream, src,
        toOut ? sHash : tmpK, toOut ? sSW : tmpV,
                       (uint64_t)n, p * kRadixBits, 32, (const unsigned long long *)(w.hist + p * kRadixDigits), w.status, w.counters + launch, 0u, w.err);
    launch++;
  }
  if (async) return 0;                                         // with a side stream the caller queues its side work first, then ani_sort_check
  rc = finish(w, stream);
  if (rc != 9001) break;
  }
  return rc;
}

// completes an ani_sort_index call that was given a side stream: waits for the sort and checks its error flag
extern "C" int ani_sort_check(void *tmp, hipStream_t stream)
{
  return finish(carve(tmp, 0), stream);
}
