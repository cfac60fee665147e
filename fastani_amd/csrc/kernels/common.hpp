// common.hpp — device primitives shared by the gfx950 kernels of the ANI hot path.
//
//  * k-mer hashing: MurmurHash3_x64_128(kmer, k, seed 42) low 32 bits, on two 64-bit words that hold the
//    k-mer's ASCII bytes (reference: src/common/murmur3.h:226-303 via src/map/include/commonFunc.hpp:71-81).
//  * workgroup scan / bitonic sort over LDS (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Use a freshly loaded register right here: the compiler then waits for the load at this point (inside a rarely taken branch)
// instead of at a later use on the common path.
#if defined(__HIP_DEVICE_COMPILE__)
#define ANI_CONSUME(x) asm volatile("" : "+v"(x))
#else
#define ANI_CONSUME(x) (void)(x)
#endif

namespace ani {

constexpr int kWave = 64;
constexpr int kTPB = 256;             // threads per workgroup for all cooperative kernels (4 waves)
constexpr uint32_t kMurmurSeed = 42;  // commonFunc.hpp:32

// Statistics counters (sums only, never cursors) are striped over kStatStripes copies of the counter block so that the
// per-wave atomics of a large grid do not serialise on one address; the host adds the stripes up.
constexpr int kStatStripes = 32;
constexpr int kStatStripeWords = 24;   // 64-bit words per stripe (= number of counters)
__device__ __forceinline__ unsigned long long *stat_slot(unsigned long long *base) { return base + (blockIdx.x & (kStatStripes - 1)) * kStatStripeWords; }

// Bump allocation out of a pool, one request per workgroup.  A returning atomicAdd on ONE address costs 12 ns on MI355X however many
// CUs ask (tools/ubench/atomic.hip, profiles/r03_ubench_atomic.txt): with 1.67 M workgroups per launch and two such cursors per
// workgroup, k_sketch_fused took its 41 ms whatever its instruction count or occupancy.  So a pool is cut into kPoolStripes equal
// stripes, each with its own cursor on its own 128-byte line (64 cursors: 0.4 ns per request); workgroup b allocates from stripe
// b mod 64, which fills them evenly.  `cursors` = kPoolStripes cursors, kPoolStripeWords apart; `stripeCap` = entries per stripe.
// Returns the absolute offset of n entries, bit 63 set if they do not fit their stripe (the cursor keeps counting, so the host
// learns the size a retry needs).  Consumers reach the entries through stored offsets (TileMeta::off, fragOff, fragCandOff), so
// holes between the stripes are harmless.
constexpr int kPoolStripes = 64, kPoolStripeWords = 16;
constexpr unsigned long long kPoolOverflowBit = 1ull << 63;
__device__ __forceinline__ unsigned long long pool_take(unsigned long long *cursors, uint32_t stripeCap, unsigned long long n)
{
  const uint32_t r = blockIdx.x & (kPoolStripes - 1);
  const unsigned long long off = n ? atomicAdd(&cursors[r * kPoolStripeWords], n) : 0ull;
  return ((unsigned long long)r * stripeCap + off) | (off + n > (unsigned long long)stripeCap ? kPoolOverflowBit : 0ull);
}

// ---------------------------------------------------------------------------------------------
// MurmurHash3_x64_128 -> low 32 bits of h1
// ---------------------------------------------------------------------------------------------
// 64-bit rotate by a compile-time constant: two v_alignbit_b32 on the device (the generic form compiles to two 64-bit shifts
// and two ORs)
__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r)
{
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if (r >= 32) { const uint32_t t = lo; lo = hi; hi = t; r -= 32; }
  if (r == 0) return ((uint64_t)hi << 32) | lo;
  const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - r), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - r);
  return ((uint64_t)nh << 32) | nl;
#else
  return (x << r) | (x >> (64 - r));
#endif
}

// z ^ (z >> 33) touches the low word only: lo ^= hi >> 1
__host__ __device__ __forceinline__ uint64_t xorshr33(uint64_t z)
{
  const uint32_t hi = (uint32_t)(z >> 32);
  return ((uint64_t)hi << 32) | ((uint32_t)z ^ (hi >> 1));
}
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t z)
{
  z = xorshr33(z); z *= 0xff51afd7ed558ccdULL;
  z = xorshr33(z); z *= 0xc4ceb9fe1a85ec53ULL;
  return xorshr33(z);
}

// k == 16: exactly one body block, empty tail (murmur3.h:245-254, :290-298)
__host__ __device__ __forceinline__ uint32_t murmur32_k16(uint64_t k1, uint64_t k2)
{
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = kMurmurSeed, h2 = kMurmurSeed;
  k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
  k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
  h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  h1 ^= 16; h2 ^= 16;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  return (uint32_t)(h1 + h2);
}

// k < 16: no body block; k1 = bytes 0..min(k,8)-1, k2 = bytes 8..k-1, bytes beyond k are zero (murmur3.h:265-285)
__host__ __device__ __forceinline__ uint32_t murmur32_tail(uint64_t k1, uint64_t k2, int k)
{
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = kMurmurSeed, h2 = kMurmurSeed;
  if (k > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  return (uint32_t)(h1 + h2);
}

// v_perm_b32 / v_alignbyte_b32 with host stand-ins (the CPU build of the tests runs the same code)
//   perm_b32(hi, lo, sel): result byte i = byte sel_i (0..7) of the 8 bytes {hi, lo} (lo = bytes 0..3)
//   alignbyte(hi, lo, n):  the 32 bits of {hi, lo} that start n bytes up (n = 0..3)
__host__ __device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t src = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) r |= (uint32_t)((src >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
  return r;
#endif
}
__host__ __device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
  return (uint32_t)(((((uint64_t)hi << 32) | lo) >> (8 * (n & 3u))) & 0xffffffffu);
#endif
}
// 8 bits = four 2-bit codes -> one code per byte
__host__ __device__ __forceinline__ uint32_t spread_codes4(uint32_t c8)
{
  const uint32_t t = (c8 | (c8 << 12)) & 0x000F000Fu;
  return (t | (t << 6)) & 0x03030303u;
}

// ASCII byte of a 2-bit code (A=0 C=1 G=2 T=3) and of its complement
__host__ __device__ __forceinline__ uint32_t code_to_ascii(uint32_t c) { return (0x54474341u >> (8 * c)) & 0xffu; }
__host__ __device__ __forceinline__ uint32_t code_to_ascii_comp(uint32_t c) { return (0x41434754u >> (8 * c)) & 0xffu; }
// complement of a raw byte: only A,C,G,T change (commonFunc.hpp:37-54)
__host__ __device__ __forceinline__ uint32_t ascii_comp(uint32_t b)
{
  return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : b;
}
// commonFunc.hpp:61-64
__host__ __device__ __forceinline__ uint32_t ascii_upper(uint32_t b) { return (b > 96 && b < 123) ? b - 32 : b; }

// ---------------------------------------------------------------------------------------------
// Workgroup primitives (256 threads = 4 waves of 64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v)
{
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// The same scan over the DPP data path (gfx9 wave64: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast:15 into rows 1 and
// 3, row_bcast:31 into rows 2 and 3): six v_add_u32_dpp instead of six ds_bpermute round trips with their selects.  Lanes without a
// source (bound_ctrl off, `old` = 0) add nothing.  Checked against wave_incl_scan on the hardware by tools/ubench/lanexor.hip.
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
  return v;
#else
  return wave_incl_scan(v);
#endif
}

// Workgroup barrier.  HIP's __syncthreads() is fence(release) + s_barrier + fence(acquire), and the compiler of ROCm 7.2 was seen
// to drop the s_waitcnt lgkmcnt(0) that the release fence needs in front of a barrier at a loop header (the back edge of the bitonic
// sort's pass loop: four ds_write_b64 in flight, then s_barrier): a wave then passes the barrier with its LDS writes still queued
// and another wave reads the old values — about one mis-sorted fragment in 10^6, different ones in every run.  The wait is
// therefore spelled out: every kernel of this library synchronises through block_barrier().
__device__ __forceinline__ void block_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
  // asm volatile with a memory clobber: the compiler may not move an LDS store below the wait (the builtin form carries no
  // ordering against memory operations in the IR)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}
// ... and the same with the wave's global stores drained too, for code that runs over LDS or global memory alike (the global-memory
// L1 path of fragments beyond the LDS classes): one thread's stores are read by another thread after the barrier.
__device__ __forceinline__ void block_barrier_mem()
{
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}

// A value that is the same in every lane of the wave, moved to a scalar register: comparisons against it take it as the scalar
// operand, branches on it are scalar branches and a loop that carries it carries no vector copies.  (The compiler cannot know that
// something read from LDS or derived from threadIdx.x >> 6 is wave-uniform; the CPU stand-in needs nothing.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
#else
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) { return x; }
#endif
__device__ __forceinline__ int32_t wave_uniform(int32_t x) { return (int32_t)wave_uniform((uint32_t)x); }
// the value lane `l` (wave-uniform, usually a constant) holds, as a scalar
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int32_t lane_value(int32_t x, int l) { return __builtin_amdgcn_readlane(x, l); }
#else
__device__ __forceinline__ int32_t lane_value(int32_t x, int l) { return __shfl(x, l); }
#endif
__device__ __forceinline__ uint64_t wave_uniform(uint64_t x) { return (uint64_t)wave_uniform((uint32_t)x) | ((uint64_t)wave_uniform((uint32_t)(x >> 32))
    << 32); }

// exclusive scan of one int per thread across the workgroup; *total = sum over all threads.
// `ws` = LDS scratch of at least 8 ints.  Contains barriers: every thread of the block must call it.
__device__ __forceinline__ int block_excl_scan(int v, int *ws, int *total)
{
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  block_barrier();                       // protect ws against a previous use
  if (lane == kWave - 1) ws[wv] = incl;
  block_barrier();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kTPB / kWave; i++) { int x = ws[i]; if (i < wv) base += x; tot += x; }
  *total = wave_uniform(tot);            // the same in every thread: a scalar for the loops and branches it bounds
  return base + incl - v;
}

// inclusive max-scan (used for "nearest predecessor with property" searches)
__device__ __forceinline__ int block_incl_maxscan(int v, int *ws)
{
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int o = __shfl_up(v, d);
    if (lane >= d) v = o > v ? o : v;
  }
  block_barrier();
  if (lane == kWave - 1) ws[wv] = v;
  block_barrier();
#pragma unroll
  for (int i = 0; i < kTPB / kWave; i++) { int x = ws[i]; if (i < wv) v = x > v ? x : v; }
  return v;
}

// In-place exclusive scan of an int array of n elements that lives in LDS or global memory, by the whole
// workgroup, any n.  Returns the total.  `ws` = LDS scratch (>= 8 ints).
__device__ inline int block_array_excl_scan(int *a, int n, int *ws)
{
  int carry = 0;
  // each pass covers kTPB*8 consecutive elements; thread t owns 8 consecutive ones
  for (int base = 0; base < n; base += kTPB * 8) {
    int first = base + (int)threadIdx.x * 8;
    int loc[8]; int sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { int idx = first + j; int x = idx < n ? a[idx] : 0; loc[j] = sum; sum += x; }
    int tot; int off = block_excl_scan(sum, ws, &tot);
#pragma unroll
    for (int j = 0; j < 8; j++) { int idx = first + j; if (idx < n) a[idx] = carry + off + loc[j]; }
    carry += tot;
  }
  block_barrier_mem();
  return carry;
}

// Wave-level rendezvous for data exchanged through LDS inside ONE wave: the LDS performs a wave's operations in order, so no
// counter wait is needed, only that the compiler keeps the order (and, in the CPU stand-in where lanes are fibers, a real rendezvous).
#if defined(__HIP_DEVICE_COMPILE__)
#define ANI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define ANI_WAVE_SYNC() (void)__ballot(1)
#endif

// Bitonic sort of n2 (power of two) keys in LDS, ascending, by the whole workgroup.
// Two consecutive stages (strides 2h and h) are fused: a thread loads the four elements i0 + {0, h, 2h, 3h}, runs the four
// compare-exchanges of both stages in registers and stores them back — half the LDS passes of the plain network.
// Each wave owns a segment of n2 / 4 keys (one wave all of them if n2 <= 256): a pass whose stride stays inside a segment touches no other
// wave's keys, and because one wave's LDS operations are performed in order such passes follow each other with a wave-level
// rendezvous only.  Workgroup barriers remain around the few passes that cross segments (3 of the 55 stages at n2 = 1024).
template <class K>
__device__ __forceinline__ void bitonic_ce(K &x, K &y, bool up) { if ((x > y) == up) { const K t = x; x = y; y = t; } }

template <class K>
__device__ __forceinline__ void bitonic_quad(K *a, int q, int h, int stride, int size)
{
  const int i0 = ((q & ~(h - 1)) << 2) | (q & (h - 1));      // index with the `h` and `stride` bits cleared
  const bool up = ((i0 & size) == 0);
  K x0 = a[i0], x1 = a[i0 + h], x2 = a[i0 + stride], x3 = a[i0 + stride + h];
  bitonic_ce(x0, x2, up); bitonic_ce(x1, x3, up);            // stage `stride`
  bitonic_ce(x0, x1, up); bitonic_ce(x2, x3, up);            // stage `h`
  a[i0] = x0; a[i0 + h] = x1; a[i0 + stride] = x2; a[i0 + stride + h] = x3;
}

template <class K>
__device__ inline void block_bitonic_sort(K *a, int n2)
{
  constexpr int kWaves = kTPB / kWave;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int seg = n2 > kWaves * kWave ? n2 / kWaves : n2;           // keys per wave segment (up to 256 keys: one wave sorts alone, a quad per lane)
  const bool mine = wv * seg < n2;                                  // this wave has a segment (wave-uniform)
  block_barrier();                                                  // the keys are complete
  for (int size = 2; size <= n2; size <<= 1) {
    int stride = size >> 1;
    while (stride >= 2) {
      const int h = stride >> 1;
      if (stride >= seg) {                                          // the quads span segments: all waves, barriers around the pass
        block_barrier();
        for (int q = threadIdx.x; q < (n2 >> 2); q += kTPB) bitonic_quad(a, q, h, stride, size);
        block_barrier();
      } else if (mine) {
        ANI_WAVE_SYNC();
        for (int q = wv * (seg >> 2) + lane; q < (wv + 1) * (seg >> 2); q += kWave) bitonic_quad(a, q, h, stride, size);
      }
      stride >>= 2;
    }
    if (stride == 1) {
      if (seg < 2) {                                                // n2 == 1 never gets here; kept for completeness
        block_barrier();
      }
      if (mine) {
        ANI_WAVE_SYNC();
        for (int t = wv * (seg >> 1) + lane; t < (wv + 1) * (seg >> 1); t += kWave) {
          const int lo = 2 * t;
          const bool up = ((lo & size) == 0);
          K x = a[lo], y = a[lo + 1];
          if ((x > y) == up) { a[lo] = y; a[lo + 1] = x; }
        }
      }
    }
  }
  block_barrier();
}

// ---------------------------------------------------------------------------------------------
// Bitonic sort in REGISTERS.
//
// block_bitonic_sort above moves every key through LDS once per two stages: at 1024 64-bit keys that is 30 passes, and the L1
// kernel spent its time in the LDS pipe (SQ counters, profiles/r03c_pmc_sq_summary.txt: 1.07e9 LDS instructions and 3.9e9
// bank-conflict cycles per step against 4.9e9 vector instructions).  Here a thread keeps KPT consecutive keys of the sequence in
// registers (element e = t * KPT + r).  A stage whose stride stays inside a thread is compare-exchanges between registers; a stride
// of KPT * m, m < 64, pairs a lane with lane ^ m of the same wave: the partner's key comes over the DPP / permlane data paths
// (lane_xor below; no LDS bandwidth, no bank conflicts, no barrier); only strides that pair two WAVES go through LDS (3 of the 55
// stages at 1024 keys in four waves).
// ---------------------------------------------------------------------------------------------
// value of lane ^ M for M = 1, 2, 4, 8, 16, 32.  (Checked lane by lane on an MI355X: tools/ubench/lanexor.hip.)
template <int M> __device__ __forceinline__ uint32_t lane_xor(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);         // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {
    int r = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);                               // row_shl:4 into lanes 0-3, 8-11 of a row
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xA, false);                      // row_shr:4 into lanes 4-7, 12-15
  }
  else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, true);   // row_ror:8
  // v_permlane16_swap (gfx950): odd rows of one <-> even rows of the other
  else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
  // v_permlane32_swap: upper half of one <-> lower half of the other
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
#else
  return __shfl_xor(x, M);
#endif
}
template <int M> __device__ __forceinline__ uint64_t lane_xor(uint64_t x)
{
#if !defined(__HIP_DEVICE_COMPILE__)
  return __shfl_xor(x, M);
#endif
  return ((uint64_t)lane_xor<M>((uint32_t)(x >> 32)) << 32) | lane_xor<M>((uint32_t)x);
}

// this element keeps the smaller (keepMin) or the larger of its own key and its partner's
template <class K> __device__ __forceinline__ K sort_pick(K own, K partner, bool keepMin) { return ((partner < own) == keepMin) ? partner : own; }

// one stage with thread stride m (1..32): every key against the same register of lane ^ m
template <class K, int KPT, int M> __device__ __forceinline__ void sort_lane_stage(K (&k)[KPT], bool keepMin)
{
#pragma unroll
  for (int r = 0; r < KPT; r++) k[r] = sort_pick(k[r], lane_xor<M>(k[r]), keepMin);
}
// strides KPT/2 .. 1 of a merge whose direction `up` is the same for all keys of the thread
template <class K, int KPT> __device__ __forceinline__ void sort_thread_merge(K (&k)[KPT], bool up)
{
#pragma unroll
  for (int st = KPT >> 1; st >= 1; st >>= 1)
#pragma unroll
    for (int r = 0; r < KPT; r++)
      if ((r & st) == 0) bitonic_ce(k[r], k[r + st], up);
}

// Sorts the n2 = 64 * WAVES * KPT keys a[0 .. n2) ascending; waves >= WAVES return at once (WAVES = 1: no workgroup barrier inside, the
// caller's barriers frame it).  WAVES is 1 or kTPB / kWave.
template <class K, int KPT, int WAVES>
__device__ __forceinline__ void block_sort_regs(K *a)
{
  constexpr int NT = kWave * WAVES;                      // threads that hold keys
  const int t = threadIdx.x;
  if (WAVES == 1 && t >= kWave) return;
  K k[KPT];
  // the input is unordered, so any assignment of keys to (thread, register) will do: the conflict-free one
#pragma unroll
  for (int r = 0; r < KPT; r++) k[r] = a[r * NT + t];
  if (WAVES > 1) block_barrier();                        // every key is in a register: `a` is free for the cross-wave stages
  // sizes 2 .. KPT: inside the thread; the direction of an element is bit `size` of e = t * KPT + r
#pragma unroll
  for (int size = 2; size <= KPT; size <<= 1)
#pragma unroll
    for (int st = size >> 1; st >= 1; st >>= 1)
#pragma unroll
      for (int r = 0; r < KPT; r++)
        if ((r & st) == 0) bitonic_ce(k[r], k[r + st], size < KPT ? (r & size) == 0 : (t & 1) == 0);
  // sizes 2 KPT .. n2: thread strides m = size / (2 KPT) .. 1, then the thread's own merge
  for (int ts = 2; ts <= NT; ts <<= 1) {                 // ts = size / KPT
    const bool up = (t & ts) == 0;
    for (int m = ts >> 1; m >= 1; m >>= 1) {
      const bool keepMin = ((t & m) == 0) == up;
      if (WAVES > 1 && m >= kWave) {                     // partner in another wave: through LDS
#pragma unroll
        for (int r = 0; r < KPT; r++) a[r * NT + t] = k[r];
        block_barrier();
#pragma unroll
        for (int r = 0; r < KPT; r++) k[r] = sort_pick(k[r], a[r * NT + (t ^ m)], keepMin);
        block_barrier();
      } else {
        switch (m) {                                     // wave-uniform
          case 32: sort_lane_stage<K, KPT, 32>(k, keepMin); break;
          case 16: sort_lane_stage<K, KPT, 16>(k, keepMin); break;
          case 8: sort_lane_stage<K, KPT, 8>(k, keepMin); break;
          case 4: sort_lane_stage<K, KPT, 4>(k, keepMin); break;
          case 2: sort_lane_stage<K, KPT, 2>(k, keepMin); break;
          default: sort_lane_stage<K, KPT, 1>(k, keepMin); break;
        }
      }
    }
    sort_thread_merge<K, KPT>(k, up);
  }
#pragma unroll
  for (int r = 0; r < KPT; r++) a[t * KPT + r] = k[r];
}

// 64 x KPT keys held by ONE wave (any wave of a workgroup), lane l owning the elements l * KPT .. l * KPT + KPT - 1 of the sequence,
// sorted ascending: the register network of block_sort_regs for a single wave, addressed by lane.  No LDS, no barrier.
template <class K, int KPT> __device__ __forceinline__ void wave_sort_regs(K (&k)[KPT])
{
  const int lane = threadIdx.x & (kWave - 1);
  // sizes 2 .. KPT: inside the lane; the direction of an element is bit `size` of e = lane * KPT + r
#pragma unroll
  for (int size = 2; size <= KPT; size <<= 1)
#pragma unroll
    for (int st = size >> 1; st >= 1; st >>= 1)
#pragma unroll
      for (int r = 0; r < KPT; r++)
        if ((r & st) == 0) bitonic_ce(k[r], k[r + st], size < KPT ? (r & size) == 0 : (lane & 1) == 0);
  for (int ts = 2; ts <= kWave; ts <<= 1) {              // ts = size / KPT
    const bool up = (lane & ts) == 0;                    // (ts = 64: every lane)
    for (int m = ts >> 1; m >= 1; m >>= 1) {
      const bool keepMin = ((lane & m) == 0) == up;
      switch (m) {                                       // wave-uniform
        case 32: sort_lane_stage<K, KPT, 32>(k, keepMin); break;
        case 16: sort_lane_stage<K, KPT, 16>(k, keepMin); break;
        case 8: sort_lane_stage<K, KPT, 8>(k, keepMin); break;
        case 4: sort_lane_stage<K, KPT, 4>(k, keepMin); break;
        case 2: sort_lane_stage<K, KPT, 2>(k, keepMin); break;
        default: sort_lane_stage<K, KPT, 1>(k, keepMin); break;
      }
    }
    sort_thread_merge<K, KPT>(k, up);
  }
}

constexpr int kBlockSortMax = 4096;
// Sorts a[0 .. n) ascending, n <= kBlockSortMax (the caller's array has room for next_pow2(max(n, 64)) keys; the tail is padded with ~0).
// Frames itself with workgroup barriers: the keys are complete before, the sorted sequence is visible after.
__host__ __device__ __forceinline__ int next_pow2_dev(int n) { int p = 1; while (p < n) p <<= 1; return p; }
template <class K>
// (inlined: an out-of-line copy would see `a` as a generic pointer and use flat instead of ds instructions)
__device__ __forceinline__ void block_sort(K *a, int n)
{
  int n2 = next_pow2_dev(n); n2 = n2 < kWave ? kWave : n2;
  for (int i = n + (int)threadIdx.x; i < n2; i += kTPB) a[i] = (K)~(K)0;
#ifdef ANI_SORT_LDS                                          // build switch for A/B measurements: the LDS network above
  block_bitonic_sort<K>(a, n2);
  return;
#endif
  block_barrier();
  switch (n2) {                                           // workgroup-uniform
    case 64: block_sort_regs<K, 1, 1>(a); break;
    case 128: block_sort_regs<K, 2, 1>(a); break;
    case 256: block_sort_regs<K, 4, 1>(a); break;
    case 512: block_sort_regs<K, 2, kTPB / kWave>(a); break;
    case 1024: block_sort_regs<K, 4, kTPB / kWave>(a); break;
    case 2048: block_sort_regs<K, 8, kTPB / kWave>(a); break;
    case 4096: block_sort_regs<K, 16, kTPB / kWave>(a); break;
    default: __builtin_trap();                            // n > kBlockSortMax: the caller's class limits rule it out — never sort a prefix silently
  }
  block_barrier();
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A grid padded to a multiple of 8 is turned inside out:
// XCD x works through the x-th eighth of the items in order, so items that are neighbours in the list (and share data) meet in
// one L2 at about the same time.  The caller drops indices >= its item count.
__device__ __forceinline__ int xcd_item(int block, int gridBlocks) { return (block & 7) * (gridBlocks >> 3) + (block >> 3); }
__host__ __device__ __forceinline__ unsigned pad8(size_t n) { return (unsigned)((n + 7) / 8 * 8); }

__host__ __device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace ani
