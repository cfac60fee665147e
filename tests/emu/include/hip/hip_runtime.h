/* TEST INFRASTRUCTURE ONLY — a tiny single-threaded CPU stand-in for <hip/hip_runtime.h>.
 *
 * Purpose: this container has no GPU and GPU-box minutes are scarce, so the `-m "not gpu"` tests
 * compile the UNMODIFIED product sources (fastani_amd/csrc: kernels + host orchestration) with g++
 * against this header and run them on the CPU, purely to debug host logic and kernel control flow
 * before a gpurun.  The result (tests/emu/libfastani_emu.so) is loaded by tests only.  The product
 * library is built by hipcc for gfx950 against the real ROCm headers and contains none of this; the
 * package loader (fastani_amd/_lib.py) never looks at tests/emu.
 *
 * Model: blocks run one after another; the threads of a block are ucontext fibers that switch only at
 * __syncthreads() and at wave collectives (__shfl*, __ballot, ...), which rendezvous the 64 lanes of
 * a wave.  A wave collective reached by only part of a wave's live lanes while the others sit in a
 * block barrier is reported as an error (it would be a divergence bug on the GPU too).
 */
#ifndef ANI_HIP_EMU_RUNTIME_H
#define ANI_HIP_EMU_RUNTIME_H

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <functional>
#include <algorithm>
#include <chrono>
#include <vector>

#define ANI_HIP_EMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }

namespace hipemu {
extern emu_uint3 g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void sync_block();
/* deposits v for this lane, returns pointer to the 64 deposited values of the wave; *mask = live lanes.
 * Must be followed by wave_done() after the values were read. */
const uint64_t *wave_gather(uint64_t v, uint64_t *mask);
void wave_done();
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
static const int warpSize = 64;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

/* ---- runtime API subset ---- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct emu_stream *hipStream_t;
typedef struct emu_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof *p); strcpy(p->name, "hip-emu (CPU, tests only)"); strcpy(p->gcnArchName, "emu");
  p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = 1; return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t = 0) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
#define hipStreamDefault 0u
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
typedef void *hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t n, hipStream_t = 0) { for (size_t i = 0; i < n; i++) ((int *)p)[i] = v; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

/* ---- device builtins ---- */
static inline void __syncthreads() { hipemu::sync_block(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T> static inline uint64_t emu_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> static inline T emu_unbits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

template <class T> static inline T __shfl(T v, int src, int width = 64) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(emu_bits(v), &mask);
  int lane = hipemu::lane_id(); int base = lane & ~(width - 1);
  T r = emu_unbits<T>(all[base + (src & (width - 1))]);
  hipemu::wave_done(); return r;
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(emu_bits(v), &mask);
  int lane = hipemu::lane_id(); int rel = lane & (width - 1);
  T r = (rel >= (int)d) ? emu_unbits<T>(all[lane - d]) : v;
  hipemu::wave_done(); return r;
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(emu_bits(v), &mask);
  int lane = hipemu::lane_id(); int rel = lane & (width - 1);
  T r = (rel + (int)d < width) ? emu_unbits<T>(all[lane + d]) : v;
  hipemu::wave_done(); return r;
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(emu_bits(v), &mask);
  int lane = hipemu::lane_id();
  T r = emu_unbits<T>(all[lane ^ m]);
  (void)width; hipemu::wave_done(); return r;
}
// v_readlane_b32: value of (wave-uniform) lane l
static inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l); }
static inline unsigned long long __ballot(int pred) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(pred ? 1 : 0, &mask);
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (((mask >> i) & 1) && all[i]) r |= 1ull << i;
  hipemu::wave_done(); return r;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
  uint64_t mask; const uint64_t *all = hipemu::wave_gather(pred ? 1 : 0, &mask);
  int r = 1;
  for (int i = 0; i < 64; i++) if (((mask >> i) & 1) && !all[i]) r = 0;
  hipemu::wave_done(); return r;
}

static inline int __mul24(int a, int b) { return a * b; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned *p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
static inline unsigned long long atomicAdd(unsigned long long *p, int v) { unsigned long long o = *p; *p = o + (unsigned long long)v; return o; }
template <class T> static inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

#endif
