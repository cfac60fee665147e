// sketch.hpp — winnowed-minimizer kernels (reference sketch tiles + query fragment sketches).
//
// Restates skch::CommonFunc::addMinimizers (src/map/include/commonFunc.hpp:91-167) in stateless form
// (SURVEY.md App. A.1/A.2):
//   h(i)   = min(hashFwd(i), hashBwd(i)), position skipped iff hashFwd == hashBwd            (:126-134)
//   P(i)   = rightmost non-skipped j in (i-w, i] minimising h(j)                             (:137-148)
//   emit (h(P(i)), seqId, i-w+1) at every non-skipped i >= w-1 whose P(i) differs from P at the previous
//   such position                                                                            (:152-161)
//
// Work decomposition: one workgroup (256 threads) per tile of kTile = 3072 consecutive k-mer start positions of
// one contig; each thread owns kPer = 12 consecutive positions.  The thread packs its 27 bases into a 64-bit
// register (2-bit codes), keeps the forward and reverse-complement k-mer as rolling 2x64-bit ASCII words and
// hashes both strands per position.  Canonical hashes go to LDS as 64-bit keys (hash<<32 | reversed position,
// so that `min` implements "smallest hash, rightmost on ties"); the w-wide window minimum is computed by
// log2(w) doubling rounds over LDS.  Tiles overlap by w-1 positions.  Argmin positions are monotone in i, so
// "P at the previous non-skipped position" is a max-scan.
#pragma once
#include "common.hpp"

namespace ani {

constexpr int kPer = 12;
constexpr int kTile = kTPB * kPer;   // 3072 k-mer positions per tile
constexpr uint64_t kSkipKey = ~0ull;

struct TileDesc {          // one per tile, built on the host from contig lengths
  int32_t contig;          // index into the batch's contig table
  int32_t firstPos;        // B: first k-mer start position hashed by this tile
};
struct TileMeta {          // produced by k_sketch_tiles
  uint32_t off;            // offset of this tile's records in the temporary pool
  int32_t cnt;             // records emitted (the first one is provisional, see firstP)
  int32_t firstP;          // absolute argmin position behind the first emitted record, -1 if none
  int32_t lastP;           // absolute argmin position at the last non-skipped window of the tile, -1 if none
};

// ------------------------------------------------------------------------------------------------
// Per-thread strand hashing of kPer consecutive positions.
//   PACKED: seq = const uint32_t* (16 bases per word), off in words;  else seq = const uint8_t*, off in bytes.
// key[j] = (canonical hash << 32) | (kTile-1-local) or kSkipKey (skipped / beyond the contig end)
// ------------------------------------------------------------------------------------------------
// 2-bit packed input, k = 16 (the default and by far the common case).  The thread's 28 bases become two byte arrays of seven dwords
// each — F: the ASCII letters in order, R: the letters of the reverse complement (R byte b = comp(S[i0 + 27 - b])) — with one
// v_perm_b32 per dword (four 2-bit codes select from "ACGT" / "TGCA"); the 16-byte k-mer of a position is then four dwords cut out of
// F at byte offset j and out of R at byte offset 12 - j with v_alignbyte_b32.  (The general form below decodes letter by letter and
// rolls four 64-bit words byte-wise: ~55 vector instructions per position where this takes ~13.)
__device__ __forceinline__ void hash_positions_packed16(const uint32_t *wds, int32_t len, int32_t i0, int local0, uint64_t (&key)[kPer])
{
  const int32_t nWords = (len + 15) >> 4;
  const int32_t w0 = i0 >> 4;
  const int sh = (i0 & 15) * 2;
  const uint64_t a = w0 < nWords ? wds[w0] : 0u, b = w0 + 1 < nWords ? wds[w0 + 1] : 0u, c = w0 + 2 < nWords ? wds[w0 + 2] : 0u;
  const uint64_t lo = a | (b << 32);
  const uint64_t bits = sh ? ((lo >> sh) | (c << (64 - sh))) : lo;       // codes of bases i0 .. i0 + 31 (27 are used)
  uint32_t F[7], R[7];
#pragma unroll
  for (int d = 0; d < 7; d++) {
    const uint32_t c8 = (uint32_t)(bits >> (8 * d)) & 0xffu;
    const uint32_t sp = spread_codes4(c8);
    F[d] = perm_b32(0u, 0x54474341u, sp);                                 // "ACGT"[code]
    R[6 - d] = perm_b32(0u, 0x41434754u, perm_b32(0u, sp, 0x00010203u));  // "TGCA"[code], the four letters in reverse order
  }
  const int32_t nPos = len - 16 + 1;
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    uint32_t f[4], r[4];
    const int fq = j >> 2, fr = j & 3, rq = (12 - j) >> 2, rr = (12 - j) & 3;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      f[t] = fr ? alignbyte(F[fq + t + 1], F[fq + t], (uint32_t)fr) : F[fq + t];
      r[t] = rr ? alignbyte(R[rq + t + 1], R[rq + t], (uint32_t)rr) : R[rq + t];
    }
    const uint32_t hf = murmur32_k16(((uint64_t)f[1] << 32) | f[0], ((uint64_t)f[3] << 32) | f[2]);
    const uint32_t hb = murmur32_k16(((uint64_t)r[1] << 32) | r[0], ((uint64_t)r[3] << 32) | r[2]);
    const bool ok = (i0 + j < nPos) && (hf != hb);                      // commonFunc.hpp:131
    const uint32_t h = hf < hb ? hf : hb;                               // :134
    key[j] = ok ? (((uint64_t)h << 32) | (uint32_t)(kTile - 1 - (local0 + j))) : kSkipKey;
  }
}

template <bool PACKED>
__device__ __forceinline__ void hash_positions(const void *seq, int64_t off, int32_t len, int32_t i0, int local0,
                                               int k, uint64_t (&key)[kPer])
{
  if (PACKED && k == 16) { hash_positions_packed16((const uint32_t *)seq + off, len, i0, local0, key); return; }
  // ---- fetch the 27 bases [i0, i0+27) as ASCII on demand ----
  uint64_t bits = 0;                  // PACKED: 2-bit codes of bases i0.. in bits 2j..2j+1
  const uint8_t *bytes = nullptr;
  if (PACKED) {
    const uint32_t *wds = (const uint32_t *)seq + off;
    const int32_t nWords = (len + 15) >> 4;
    const int32_t w0 = i0 >> 4;
    const int sh = (i0 & 15) * 2;
    uint64_t a = w0 < nWords ? wds[w0] : 0u;
    uint64_t b = w0 + 1 < nWords ? wds[w0 + 1] : 0u;
    uint64_t c = w0 + 2 < nWords ? wds[w0 + 2] : 0u;
    uint64_t lo = a | (b << 32);
    bits = sh ? ((lo >> sh) | (c << (64 - sh))) : lo;
  } else {
    bytes = (const uint8_t *)seq + off;
  }
  auto fwd_byte = [&](int j) -> uint64_t {
    if (PACKED) return code_to_ascii((uint32_t)(bits >> (2 * j)) & 3u);
    int32_t p = i0 + j;
    return p < len ? ascii_upper(bytes[p]) : 0u;
  };
  auto rc_byte = [&](int j) -> uint64_t {
    if (PACKED) return code_to_ascii_comp((uint32_t)(bits >> (2 * j)) & 3u);
    int32_t p = i0 + j;
    return p < len ? ascii_comp(ascii_upper(bytes[p])) : 0u;
  };

  // forward 16-byte window F = S[i..i+16) ; reverse window R: byte j = comp(S[i+15-j])
  uint64_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    flo |= fwd_byte(j) << (8 * j);
    fhi |= fwd_byte(8 + j) << (8 * j);
    rhi |= rc_byte(j) << (8 * (7 - j));
    rlo |= rc_byte(8 + j) << (8 * (7 - j));
  }
  const bool k16 = (k == 16);
  const uint64_t maskLo = k >= 8 ? ~0ull : ((1ull << (8 * k)) - 1);
  const uint64_t maskHi = k16 ? ~0ull : (k > 8 ? ((1ull << (8 * (k - 8))) - 1) : 0ull);
  const int rsh = 8 * (16 - k);       // the k-byte reverse complement is R shifted right by 16-k bytes
  const int32_t nPos = len - k + 1;

#pragma unroll
  for (int j = 0; j < kPer; j++) {
    uint32_t hf, hb;
    if (k16) {
      hf = murmur32_k16(flo, fhi);
      hb = murmur32_k16(rlo, rhi);
    } else {
      uint64_t r1, r2;
      if (rsh < 64) { r1 = (rlo >> rsh) | (rhi << (64 - rsh)); r2 = rhi >> rsh; }
      else { r1 = rsh == 64 ? rhi : (rhi >> (rsh - 64)); r2 = 0; }
      hf = murmur32_tail(flo & maskLo, fhi & maskHi, k);
      hb = murmur32_tail(r1, r2, k);
    }
    const bool ok = (i0 + j < nPos) && (hf != hb);                      // commonFunc.hpp:131
    const uint32_t h = hf < hb ? hf : hb;                               // :134
    key[j] = ok ? (((uint64_t)h << 32) | (uint32_t)(kTile - 1 - (local0 + j))) : kSkipKey;
    // roll to position i+1
    const uint64_t nb = fwd_byte(16 + j), nc = rc_byte(16 + j);
    flo = (flo >> 8) | (fhi << 56);
    fhi = (fhi >> 8) | (nb << 56);
    rhi = (rhi << 8) | (rlo >> 56);
    rlo = (rlo << 8) | nc;
  }
}

// ------------------------------------------------------------------------------------------------
// Tile core: hashes + window minimum + emission flags.  All 256 threads of the workgroup call it.
//   keys : LDS, kTile uint64
//   ws   : LDS scratch, >= kTPB + 8 ints
// Outputs per thread: em[j] (emit at own position j), W[j] (window-min key), and block-wide first/last P.
// ------------------------------------------------------------------------------------------------
// key[] = this thread's kPer position keys (hash_positions).  lo = first local position that exists for this pass (keys below
// it must be kSkipKey: a query fragment is winnowed in isolation), windows are evaluated where they end at local positions in
// [lo + w - 1, hiEmit) and at k-mer starts below nPos.
__device__ __forceinline__ void tile_winnow_keys(const uint64_t (&key)[kPer], int32_t B, int32_t nPos, int w, int lo, int hiEmit,
                                                 uint64_t *keys, int *ws,
                                                 uint64_t (&W)[kPer], bool (&em)[kPer], int &tileFirstP, int &tileLastP)
{
  const int t = threadIdx.x;
  const int local0 = t * kPer;
  bool own_ok[kPer];
  if (t == 0) ws[kTPB + 8] = -1;                   // slot of the tile's first valid P (read after the barriers below)
  if (w >= kPer && w <= 4 * kPer) {
    // ---- window minimum over (i-w, i], w >= kPer: the window of own position j is the own prefix [0, j], a whole number of
    // earlier threads' blocks and a suffix of one more block.  Prefix minima stay in registers, every thread publishes its suffix
    // minima (suffix 0 = its whole block): one barrier and ~16 LDS reads, where doubling the span takes log2(w) rounds of two
    // barriers and 2 * kPer LDS accesses each (the sketch kernels spent nearly half their wave cycles parked at barriers).
    uint64_t suf = kSkipKey, pre = kSkipKey;
#pragma unroll
    for (int j = kPer - 1; j >= 0; j--) { suf = key[j] < suf ? key[j] : suf; keys[local0 + j] = suf; }
#pragma unroll
    for (int j = 0; j < kPer; j++) { own_ok[j] = key[j] != kSkipKey; pre = key[j] < pre ? key[j] : pre; W[j] = pre; }
    block_barrier();
    const int aHi = (w - 1) / kPer;                // whole blocks before the own one that position 0 needs (1..3); position j: aHi or aHi - 1
    uint64_t allHi = kSkipKey, allLo = kSkipKey;   // minimum over the aHi / aHi - 1 preceding blocks
#pragma unroll
    for (int x = 1; x <= 3; x++) {
      if (x <= aHi) {
        const uint64_t v = t - x >= 0 ? keys[(t - x) * kPer] : kSkipKey;
        allLo = allHi; allHi = v < allHi ? v : allHi;
      }
    }
    int a = aHi, b = (w - 1) - aHi * kPer;         // position j needs w - 1 - j earlier positions = a blocks + the last b of one more
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const uint64_t whole = a == aHi ? allHi : allLo;
      const int tb = t - a - 1;
      const uint64_t part = (b > 0 && tb >= 0) ? keys[tb * kPer + kPer - b] : kSkipKey;
      uint64_t m = whole < part ? whole : part;
      W[j] = m < W[j] ? m : W[j];
      if (b == 0) { a--; b = kPer - 1; } else b--;
    }
  } else {
#pragma unroll
  for (int j = 0; j < kPer; j++) { keys[local0 + j] = key[j]; W[j] = key[j]; own_ok[j] = key[j] != kSkipKey; }

  // ---- window minimum over (i-w, i] by doubling: after the loop W = min over the last p positions ----
  int p = 1;
  while (2 * p <= w) {
    block_barrier();
    uint64_t r[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) { int q = local0 + j - p; r[j] = q >= 0 ? keys[q] : kSkipKey; }
    block_barrier();
#pragma unroll
    for (int j = 0; j < kPer; j++) { W[j] = r[j] < W[j] ? r[j] : W[j]; keys[local0 + j] = W[j]; }
    p <<= 1;
  }
  block_barrier();
  if (w > p) {
    const int d = w - p;
#pragma unroll
    for (int j = 0; j < kPer; j++) { int q = local0 + j - d; uint64_t r = q >= 0 ? keys[q] : kSkipKey; W[j] = r < W[j] ? r : W[j]; }
  }
  }

  // ---- emission: argmin positions are monotone, so "previous P" is a running maximum ----
  int P[kPer]; bool val[kPer];
  int myMax = -1, myFirst = -1;
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int l = local0 + j;
    val[j] = own_ok[j] && l >= lo + w - 1 && l < hiEmit && (B + l) < nPos;
    P[j] = val[j] ? (kTile - 1 - (int)(uint32_t)W[j]) : -1;
    if (val[j]) { if (myFirst < 0) myFirst = P[j]; myMax = P[j]; }
  }
  // max P over all earlier threads: wave scan by shuffles, one barrier for the wave totals
  const int lane = t & (kWave - 1), wv = t >> 6;
  int v = myMax;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v = o > v ? o : v; }
  int prev = __shfl_up(v, 1); if (lane == 0) prev = -1;
  if (lane == kWave - 1) ws[wv] = v;
  block_barrier();
  int lastAll = -1;
#pragma unroll
  for (int i = 0; i < kTPB / kWave; i++) { const int x = ws[i]; if (i < wv) prev = x > prev ? x : prev; lastAll = x > lastAll ? x : lastAll; }
  // first valid P in the tile: the thread whose predecessor max is -1 and that has a valid position (exactly one, if any)
  if (myFirst >= 0 && prev < 0) ws[kTPB + 8] = myFirst;
  block_barrier();
  const int firstAll = ws[kTPB + 8];
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    em[j] = val[j] && P[j] != prev;
    if (val[j]) prev = P[j];
  }
  tileFirstP = firstAll >= 0 ? B + firstAll : -1;
  tileLastP = lastAll >= 0 ? B + lastAll : -1;
  block_barrier();
}

template <bool PACKED>
__device__ __forceinline__ void tile_winnow(const void *seq, int64_t off, int32_t len, int32_t B, int k, int w,
                                            uint64_t *keys, int *ws,
                                            uint64_t (&W)[kPer], bool (&em)[kPer], int &tileFirstP, int &tileLastP)
{
  uint64_t key[kPer];
  hash_positions<PACKED>(seq, off, len, B + (int)threadIdx.x * kPer, (int)threadIdx.x * kPer, k, key);
  tile_winnow_keys(key, B, len - k + 1, w, 0, kTile, keys, ws, W, em, tileFirstP, tileLastP);
}

// ------------------------------------------------------------------------------------------------
// Reference sketch, pass 1: one workgroup per tile; records go to a temporary pool in tile-arrival order.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kTPB) void k_sketch_tiles(const uint32_t *__restrict__ packed, const uint8_t *__restrict__ ascii,
                                                       const int64_t *__restrict__ contigOff, const int32_t *__restrict__ contigLen,
                                                       const uint8_t *__restrict__ contigMode /* 1 = 2-bit packed, 0 = raw bytes */,
                                                       const TileDesc *__restrict__ tiles, int k, int w,
                                                       uint32_t *__restrict__ poolHash, int32_t *__restrict__ poolWpos,
                                                       uint32_t poolCap, unsigned long long *__restrict__ poolCount,
                                                       TileMeta *__restrict__ meta)
{
  __shared__ uint64_t keys[kTile];
  __shared__ int ws[kTPB + 16];
  __shared__ unsigned long long sBase;
  const TileDesc td = tiles[blockIdx.x];
  const int64_t off = contigOff[td.contig];
  const int32_t len = contigLen[td.contig];
  uint64_t W[kPer]; bool em[kPer]; int firstP, lastP;
  if (contigMode[td.contig]) tile_winnow<true>(packed, off, len, td.firstPos, k, w, keys, ws, W, em, firstP, lastP);
  else tile_winnow<false>(ascii, off, len, td.firstPos, k, w, keys, ws, W, em, firstP, lastP);
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kPer; j++) cnt += em[j];
  int total; int rank = block_excl_scan(cnt, ws, &total);
  if (threadIdx.x == 0) sBase = pool_take(poolCount, poolCap, (unsigned long long)total);
  block_barrier();
  const unsigned long long base = sBase & ~kPoolOverflowBit;
  if (!(sBase & kPoolOverflowBit)) {
#pragma unroll
    for (int j = 0; j < kPer; j++)
      if (em[j]) {
        poolHash[base + rank] = (uint32_t)(W[j] >> 32);
        poolWpos[base + rank] = td.firstPos + (int)threadIdx.x * kPer + j - w + 1;     // currentWindowId, commonFunc.hpp:124
        rank++;
      }
  }
  if (threadIdx.x == 0) {
    TileMeta m; m.off = (uint32_t)base; m.cnt = total; m.firstP = firstP; m.lastP = lastP;
    meta[blockIdx.x] = m;
  }
}

// Tile / fragment tables are expanded on the device from per-contig prefix arrays (a 1000-genome batch has 1.6 M of each):
//   unit u belongs to the contig c with start[c] <= u < start[c+1];  its offset inside the contig is (u - start[c]) * step
__device__ __forceinline__ int32_t owner_of(const uint32_t *__restrict__ start, int32_t nContigs, uint32_t u)
{
  int32_t lo = 0, hi = nContigs;                      // last c with start[c] <= u  (start[] is non-decreasing, start[nContigs] = total)
  while (hi - lo > 1) { int32_t mid = (lo + hi) >> 1; if (start[mid] <= u) lo = mid; else hi = mid; }
  return lo;
}

static __global__ void k_expand_tiles(const uint32_t *__restrict__ tileStart, int32_t nContigs, uint32_t nTiles, int32_t stride, TileDesc *__restrict__ tiles)
{
  const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
  if (T >= nTiles) return;
  const int32_t c = owner_of(tileStart, nContigs, T);
  tiles[T] = TileDesc{c, (int32_t)((T - tileStart[c]) * (uint32_t)stride)};
}

// pass 2: decide, per tile, whether its provisional first record repeats the argmin that the nearest earlier
// non-empty tile of the same contig ended with (then it is not a new minimizer, commonFunc.hpp:155).
static __global__ void k_sketch_tile_counts(const TileDesc *__restrict__ tiles, const TileMeta *__restrict__ meta, int nTiles,
                                     int32_t *__restrict__ outCnt /* [nTiles] */, uint8_t *__restrict__ dropFirst)
{
  int T = blockIdx.x * blockDim.x + threadIdx.x;
  if (T >= nTiles) return;
  const TileMeta m = meta[T];
  int drop = 0;
  if (m.cnt > 0) {
    const int c = tiles[T].contig;
    for (int U = T - 1; U >= 0 && tiles[U].contig == c; U--) {
      const int lp = meta[U].lastP;
      if (lp >= 0) { drop = (lp == m.firstP); break; }
    }
  }
  dropFirst[T] = (uint8_t)drop;
  outCnt[T] = m.cnt - drop;
}

// pass 3: gather the tile-ordered records into position order as 12-byte skch::MinimizerInfo records.
static __global__ __launch_bounds__(kTPB) void k_sketch_gather(const TileDesc *__restrict__ tiles, const TileMeta *__restrict__ meta,
                                                        const uint8_t *__restrict__ dropFirst, const uint32_t *__restrict__ outOff,
                                                        const uint32_t *__restrict__ poolHash, const int32_t *__restrict__ poolWpos,
                                                        int32_t seqIdBase, uint32_t *__restrict__ records /* 3 words each */)
{
  const int T = blockIdx.x;
  const TileMeta m = meta[T];
  const int drop = dropFirst[T];
  const int n = m.cnt - drop;
  const size_t dst = outOff[T];
  const int32_t seqId = seqIdBase + tiles[T].contig;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const uint32_t src = m.off + drop + r;
    uint32_t *rec = records + 3 * (dst + r);
    rec[0] = poolHash[src]; rec[1] = (uint32_t)seqId; rec[2] = (uint32_t)poolWpos[src];
  }
}

// ------------------------------------------------------------------------------------------------
// Query fragments: one workgroup per fragment (computeMap.hpp:157-175 aliases bytes [i*L, i*L+L) of the
// contig; :260 winnows them in isolation; :268-274 sort + unique by hash => sketch size s).
// Sorted unique hashes go to a pool; per fragment (offset, s).
// ------------------------------------------------------------------------------------------------
struct FragDesc {
  int32_t contig;     // contig index in the query batch
  int32_t start;      // first base of the fragment inside the contig (i * fragLen)
};

static __global__ void k_expand_frags(const uint32_t *__restrict__ fragStart, const int32_t *__restrict__ contigGenomeLocal,
                               const int32_t *__restrict__ contigQSeqBase, int32_t nContigs, uint32_t nFrags, int32_t fragLen,
                               FragDesc *__restrict__ frags, int32_t *__restrict__ fragGenome, int32_t *__restrict__ fragQSeq)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nFrags) return;
  const int32_t c = owner_of(fragStart, nContigs, f);
  const int32_t i = (int32_t)(f - fragStart[c]);
  frags[f] = FragDesc{c, i * fragLen};
  fragGenome[f] = contigGenomeLocal[c];
  fragQSeq[f] = contigQSeqBase[c] + i;            // running fragment id inside the query genome (computeMap.hpp:175,:188)
}

constexpr int kFragHashCap = 4096;   // minimizers one fragment may produce before sort/unique (LDS staging)
static_assert(kFragHashCap <= kBlockSortMax, "block_sort (common.hpp) sorts at most kBlockSortMax keys");

template <bool PACKED>
__device__ __forceinline__ void fragment_sketch_body(const void *__restrict__ seq, const int64_t *__restrict__ contigOff,
                                                     const FragDesc *__restrict__ frags, int fragLen, int k, int w,
                                                     uint32_t *__restrict__ pool, uint32_t poolCap,
                                                     unsigned long long *__restrict__ poolCount,
                                                     uint32_t *__restrict__ fragOff, int32_t *__restrict__ fragS,
                                                     int *__restrict__ maxS,
                                                     uint64_t *keys, uint32_t *hbuf, int *ws, unsigned long long *sBasePtr)
{
  const FragDesc fd = frags[blockIdx.x];
  // The fragment is winnowed as a "virtual contig" that starts at base fd.start of the real contig and ends at
  // fd.start + fragLen: tile_winnow only emits for windows that lie completely inside its first tile position
  // onwards (local >= w-1) and for k-mers that end before the virtual end, i.e. in isolation from the
  // neighbouring fragments.
  const int64_t off = contigOff[fd.contig];
  const int32_t vlen = fd.start + fragLen;
  const int32_t nPos = fragLen - k + 1;
  const int stride = kTile - (w - 1);
  int produced = 0;
  int prevLastP = -1;
  bool overflow = false;
  for (int32_t B = 0; B == 0 || B + (w - 1) < nPos; B += stride) {
    uint64_t W[kPer]; bool em[kPer]; int firstP, lastP;
    tile_winnow<PACKED>(seq, off, vlen, fd.start + B, k, w, keys, ws, W, em, firstP, lastP);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) cnt += em[j];
    int total; int rank = block_excl_scan(cnt, ws, &total);
    const int drop = (total > 0 && prevLastP >= 0 && firstP == prevLastP) ? 1 : 0;   // commonFunc.hpp:155 across tiles
#pragma unroll
    for (int j = 0; j < kPer; j++)
      if (em[j]) {
        const int r = produced + rank - drop;
        if (rank >= drop) { if (r < kFragHashCap) hbuf[r] = (uint32_t)(W[j] >> 32); }
        rank++;
      }
    produced += total - drop;
    if (produced > kFragHashCap) { overflow = true; produced = kFragHashCap; }
    if (lastP >= 0) prevLastP = lastP;
    block_barrier();
  }
  // ---- sort + unique (computeMap.hpp:268-274) ----
  const int n = produced;
  block_sort<uint32_t>(hbuf, n);
  const int per = (n + kTPB - 1) / kTPB;
  const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  int cntU = 0;
  for (int i = lo; i < hi; i++) cntU += (i == 0 || hbuf[i] != hbuf[i - 1]);
  int s; int r = block_excl_scan(cntU, ws, &s);
  if (threadIdx.x == 0) {
    *sBasePtr = pool_take(poolCount, poolCap, (unsigned long long)s);
    fragOff[blockIdx.x] = (uint32_t)*sBasePtr;
    fragS[blockIdx.x] = overflow ? -1 : s;
    // 0x7fffffff: a fragment exceeded kFragHashCap minimizers (host reports a limit error)
    atomicMax((int *)stat_slot((unsigned long long *)maxS), overflow ? 0x7fffffff : s);
  }
  block_barrier();
  const unsigned long long base = *sBasePtr & ~kPoolOverflowBit;
  if (!(*sBasePtr & kPoolOverflowBit))
    for (int i = lo; i < hi; i++)
      if (i == 0 || hbuf[i] != hbuf[i - 1]) pool[base + r++] = hbuf[i];
}

// SINGLE: the fragment fits one tile (fragLen - k + 1 <= kTile, the default 3 kb fragments): the minimizer staging buffer then
// reuses the window-key array, which is dead once the tile is winnowed — 26 KiB of LDS per workgroup instead of 42.
template <bool SINGLE>
static __global__ __launch_bounds__(kTPB, SINGLE ? 4 : 3) void k_fragment_sketch(const uint32_t *__restrict__ packed, const uint8_t *__restrict__ ascii,
                                                          const int64_t *__restrict__ contigOff, const uint8_t *__restrict__ contigMode,
                                                          const FragDesc *__restrict__ frags, int fragLen, int k, int w,
                                                          uint32_t *__restrict__ pool, uint32_t poolCap, unsigned long long *__restrict__ poolCount,
                                                          uint32_t *__restrict__ fragOff, int32_t *__restrict__ fragS, int *__restrict__ maxS)
{
  __shared__ uint64_t keys[kTile];
  __shared__ uint32_t hbufOwn[SINGLE ? 1 : kFragHashCap];
  __shared__ int ws[kTPB + 16];
  __shared__ unsigned long long sBase;
  static_assert(kFragHashCap * 4 <= kTile * 8, "staging buffer fits the key array");
  uint32_t *hbuf = SINGLE ? (uint32_t *)keys : hbufOwn;
  if (contigMode[frags[blockIdx.x].contig])
    fragment_sketch_body<true>(packed, contigOff, frags, fragLen, k, w, pool, poolCap, poolCount, fragOff, fragS, maxS, keys, hbuf, ws, &sBase);
  else
    fragment_sketch_body<false>(ascii, contigOff, frags, fragLen, k, w, pool, poolCap, poolCount, fragOff, fragS, maxS, keys, hbuf, ws, &sBase);
}

// ------------------------------------------------------------------------------------------------
// All-vs-all: a genome that is reference AND query is hashed once.  One workgroup per fragment-aligned tile: the tile of
// fragment f of a contig starts w-1 positions before the fragment, so that the same kTile strand hashes give (1) the
// contig-continuous reference minimizers whose windows end inside the fragment's positions and (2), with the positions outside
// the fragment masked, the fragment's own sketch (winnowed in isolation, computeMap.hpp:260).  Hashing is ~2/3 of either
// kernel's instructions; the second window-minimum pass and the sort are what remains of the fragment kernel.  Tiles behind a
// contig's last whole fragment (and contigs shorter than a fragment) are reference-only.  Needs fragLen + w - 1 <= kTile.
// ------------------------------------------------------------------------------------------------
struct FusedInfo {
  int32_t frag;         // fragment index in the batch, -1 = reference-only tile
  int32_t emitEnd;      // reference minimizers are emitted for windows ending at local positions < emitEnd
  int32_t fragLocal0;   // local position of the fragment's first base
};

static __global__ void k_expand_fused(const uint32_t *__restrict__ tileStart, const uint32_t *__restrict__ fragStart, int32_t nContigs, uint32_t nTiles,
                               int32_t fragLen, int32_t w, int32_t stride, TileDesc *__restrict__ tiles, FusedInfo *__restrict__ info)
{
  const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
  if (T >= nTiles) return;
  const int32_t c = owner_of(tileStart, nContigs, T);
  const int32_t t = (int32_t)(T - tileStart[c]);
  const int32_t nFragC = (int32_t)(fragStart[c + 1] - fragStart[c]);
  if (t < nFragC) {
    const int32_t B = t == 0 ? 0 : t * fragLen - (w - 1);
    tiles[T] = TileDesc{c, B};
    info[T] = FusedInfo{(int32_t)fragStart[c] + t, t * fragLen - B + fragLen, t * fragLen - B};
  } else {
    const int32_t base = nFragC > 0 ? nFragC * fragLen - (w - 1) : 0;
    tiles[T] = TileDesc{c, base + (t - nFragC) * stride};
    info[T] = FusedInfo{-1, kTile, 0};
  }
}

// sorted unique hashes of the n minimizer hashes in hbuf -> pool (computeMap.hpp:268-274); shared by the fragment kernels
__device__ __forceinline__ void fragment_finish(uint32_t *hbuf, int n, bool overflow, int frag, uint32_t *__restrict__ pool, uint32_t poolCap,
                                                unsigned long long *__restrict__ poolCount, uint32_t *__restrict__ fragOff, int32_t *__restrict__ fragS,
                                                int *__restrict__ maxS, int *ws, unsigned long long *sBasePtr)
{
  block_sort<uint32_t>(hbuf, n);
  const int per = (n + kTPB - 1) / kTPB;
  const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  int cntU = 0;
  for (int i = lo; i < hi; i++) cntU += (i == 0 || hbuf[i] != hbuf[i - 1]);
  int s; int r = block_excl_scan(cntU, ws, &s);
  if (threadIdx.x == 0) {
    *sBasePtr = pool_take(poolCount, poolCap, (unsigned long long)s);
    fragOff[frag] = (uint32_t)*sBasePtr;
    fragS[frag] = overflow ? -1 : s;
    // 0x7fffffff: a fragment exceeded kFragHashCap minimizers (host reports a limit error)
    atomicMax((int *)stat_slot((unsigned long long *)maxS), overflow ? 0x7fffffff : s);
  }
  block_barrier();
  const unsigned long long base = *sBasePtr & ~kPoolOverflowBit;
  if (!(*sBasePtr & kPoolOverflowBit))
    for (int i = lo; i < hi; i++)
      if (i == 0 || hbuf[i] != hbuf[i - 1]) pool[base + r++] = hbuf[i];
}

// one wave per fragment: its sketch from the striped pool to its place in the packed pool; fragOff is rewritten
static __global__ void k_pack_fragment_pool(const uint32_t *__restrict__ pool, uint32_t *__restrict__ fragOff, const int32_t *__restrict__ s,
    const uint32_t *__restrict__ newOff,
                                     uint32_t nFrag, uint32_t *__restrict__ out)
{
  // (64-bit: 64 threads per fragment pass 2^32 at 2^26 fragments)
  const uint32_t f = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (f >= nFrag) return;
  const uint32_t src = fragOff[f], dst = newOff[f];
  const int n = s[f];
  for (int i = (int)lane; i < n; i += 64) out[dst + i] = pool[src + i];
  ANI_WAVE_SYNC();                                  // every lane has read fragOff[f]
  if (lane == 0) fragOff[f] = dst;
}

// the per-fragment tables of one packed fragment set (ani_fragset_pack) into their place in the tables of a merged set: sketch
// offsets rebased to the merged pool, genome numbers to the merged genome list
static __global__ void k_fragset_rebase(uint32_t n, const uint32_t *__restrict__ srcOff, const int32_t *__restrict__ srcS,
    const int32_t *__restrict__ srcGenome,
                                        const int32_t *__restrict__ srcQSeq, uint32_t addOff, int32_t addGenome,
                                        uint32_t *__restrict__ dstOff, int32_t *__restrict__ dstS, int32_t *__restrict__ dstGenome,
                                            int32_t *__restrict__ dstQSeq)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    dstOff[i] = srcOff[i] + addOff; dstS[i] = srcS[i]; dstGenome[i] = srcGenome[i] + addGenome; dstQSeq[i] = srcQSeq[i];
  }
}

template <bool PACKED>
__device__ __forceinline__ void fused_tile_body(const void *__restrict__ seq, int64_t off, int32_t len, const TileDesc td, const FusedInfo fi, int k, int w,
    int fragLen,
                                                uint32_t *__restrict__ poolHash, int32_t *__restrict__ poolWpos, uint32_t poolCap,
                                                    unsigned long long *__restrict__ poolCount,
                                                TileMeta *__restrict__ meta,
                                                uint32_t *__restrict__ qPool, uint32_t qCap, unsigned long long *__restrict__ qCount,
                                                uint32_t *__restrict__ fragOff, int32_t *__restrict__ fragS, int *__restrict__ maxS,
                                                uint64_t *keys, int *ws, unsigned long long *sBasePtr)
{
  const int local0 = (int)threadIdx.x * kPer;
  uint64_t key[kPer];
  hash_positions<PACKED>(seq, off, len, td.firstPos + local0, local0, k, key);
  uint64_t W[kPer]; bool em[kPer]; int firstP, lastP;
  // ---- reference minimizers (as k_sketch_tiles, windows ending below emitEnd) ----
  tile_winnow_keys(key, td.firstPos, len - k + 1, w, 0, fi.emitEnd, keys, ws, W, em, firstP, lastP);
  {
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) cnt += em[j];
    int total; int rank = block_excl_scan(cnt, ws, &total);
    if (threadIdx.x == 0) *sBasePtr = pool_take(poolCount, poolCap, (unsigned long long)total);
    block_barrier();
    const unsigned long long base = *sBasePtr & ~kPoolOverflowBit;
    if (!(*sBasePtr & kPoolOverflowBit)) {
#pragma unroll
      for (int j = 0; j < kPer; j++)
        if (em[j]) {
          poolHash[base + rank] = (uint32_t)(W[j] >> 32);
          poolWpos[base + rank] = td.firstPos + local0 + j - w + 1;     // currentWindowId, commonFunc.hpp:124
          rank++;
        }
    }
    if (threadIdx.x == 0) { TileMeta m; m.off = (uint32_t)base; m.cnt = total; m.firstP = firstP; m.lastP = lastP; meta[blockIdx.x] = m; }
  }
  if (fi.frag < 0) return;                          // workgroup-uniform
  // ---- the fragment's own sketch (winnowed in isolation, computeMap.hpp:260) WITHOUT a second winnowing pass.  Only k-mers inside
  // the fragment exist for it: local positions f0 .. fEnd.  A window that ends at l in [f0 + w - 1, fEnd] covers fragment positions
  // only, so its minimum W and its argmin P are the ones of the contig-continuous pass above; and "P differs from the previous
  // window's" is the same decision too, except at the fragment's first window, which always emits (commonFunc.hpp:152-161: the
  // deque starts empty).  So: the fragment's minimizers = the reference minimizers emitted in that range + its first window's.
  const int f0 = fi.fragLocal0, fEnd = f0 + fragLen - k;     // last k-mer start inside the fragment (local)
  bool inR[kPer]; int myFirst = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < kPer; j++) {
    const int l = local0 + j;
    inR[j] = key[j] != kSkipKey && l >= f0 + w - 1 && l <= fEnd;
    if (inR[j] && l < myFirst) myFirst = l;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_down(myFirst, d); myFirst = o < myFirst ? o : myFirst; }
  block_barrier();                                  // the reference pass is done with ws / sBasePtr
  if ((threadIdx.x & (kWave - 1)) == 0) ws[threadIdx.x >> 6] = myFirst;
  block_barrier();
  int firstL = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < kTPB / kWave; i++) { const int x = ws[i]; firstL = x < firstL ? x : firstL; }
  uint32_t *hbuf = (uint32_t *)keys;                // dead once the tile is winnowed (kFragHashCap * 4 <= kTile * 8)
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kPer; j++) { inR[j] = inR[j] && (em[j] || local0 + j == firstL); cnt += inR[j]; }
  int total; int rank = block_excl_scan(cnt, ws, &total);
#pragma unroll
  for (int j = 0; j < kPer; j++)
    if (inR[j]) { if (rank < kFragHashCap) hbuf[rank] = (uint32_t)(W[j] >> 32); rank++; }
  const bool overflow = total > kFragHashCap;
  block_barrier();
  fragment_finish(hbuf, overflow ? kFragHashCap : total, overflow, fi.frag, qPool, qCap, qCount, fragOff, fragS, maxS, ws, sBasePtr);
}

static __global__ __launch_bounds__(kTPB, 3) void k_sketch_fused(const uint32_t *__restrict__ packed, const uint8_t *__restrict__ ascii,
                                                          const int64_t *__restrict__ contigOff, const int32_t *__restrict__ contigLen,
                                                              const uint8_t *__restrict__ contigMode,
                                                          const TileDesc *__restrict__ tiles, const FusedInfo *__restrict__ info, int k, int w, int fragLen,
                                                          uint32_t *__restrict__ poolHash, int32_t *__restrict__ poolWpos, uint32_t poolCap,
                                                              unsigned long long *__restrict__ poolCount,
                                                          TileMeta *__restrict__ meta,
                                                          uint32_t *__restrict__ qPool, uint32_t qCap, unsigned long long *__restrict__ qCount,
                                                          uint32_t *__restrict__ fragOff, int32_t *__restrict__ fragS, int *__restrict__ maxS)
{
  __shared__ uint64_t keys[kTile];
  __shared__ int ws[kTPB + 16];
  __shared__ unsigned long long sBase;
  const TileDesc td = tiles[blockIdx.x];
  const FusedInfo fi = info[blockIdx.x];
  const int64_t off = contigOff[td.contig];
  const int32_t len = contigLen[td.contig];
  if (contigMode[td.contig])
    fused_tile_body<true>(packed, off, len, td, fi, k, w, fragLen, poolHash, poolWpos, poolCap, poolCount, meta, qPool, qCap, qCount, fragOff, fragS, maxS,
        keys, ws, &sBase);
  else
    fused_tile_body<false>(ascii, off, len, td, fi, k, w, fragLen, poolHash, poolWpos, poolCap, poolCount, meta, qPool, qCap, qCount, fragOff, fragS, maxS,
        keys, ws, &sBase);
}

}  // namespace ani
