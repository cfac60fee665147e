// alloc.hip — what fresh device memory costs on this box, and whether a side thread can hide it (round 6).
//   hipcc --offload-arch=gfx950 -O2 -o alloc alloc.hip -lpthread && ./alloc
// Prints: hipMalloc / first hipMemset / second hipMemset / hipFree per size; N x 1 GiB in one thread and in two threads at once;
// the latency of small kernel launches on one thread while another thread allocates; hipHostMalloc.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }
__global__ void k_touch(unsigned *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
  CK(hipSetDevice(0));
  hipStream_t s; CK(hipStreamCreate(&s));
  { void *p; CK(hipMalloc(&p, 1 << 20)); CK(hipFree(p)); }
  for (size_t mb : {64, 256, 1024, 4096, 16384}) {
    for (int rep = 0; rep < 2; rep++) {
      void *p = nullptr; const size_t bytes = mb << 20;
      auto t = Clock::now(); CK(hipMalloc(&p, bytes)); const double tm = ms(t);
      t = Clock::now(); CK(hipMemsetAsync(p, 0, bytes, s)); CK(hipStreamSynchronize(s)); const double t1 = ms(t);
      t = Clock::now(); CK(hipMemsetAsync(p, 1, bytes, s)); CK(hipStreamSynchronize(s)); const double t2 = ms(t);
      t = Clock::now(); CK(hipFree(p)); const double tf = ms(t);
      printf("%6zu MB rep %d: hipMalloc %8.2f ms (%6.2f us/MB)  first memset %7.2f ms  second %7.2f ms  hipFree %8.2f ms (%6.2f us/MB)\n", mb, rep, tm, tm * 1e3 / mb, t1, t2, tf, tf * 1e3 / mb);
    }
  }
  // 8 x 1 GiB, one thread / two threads / four threads
  for (int nt : {1, 2, 4}) {
    std::vector<void *> ptr(8, nullptr);
    auto t = Clock::now();
    std::vector<std::thread> th;
    for (int x = 0; x < nt; x++) th.emplace_back([&, x]() { (void)hipSetDevice(0); for (int i = x; i < 8; i += nt) (void)hipMalloc(&ptr[i], (size_t)1 << 30); });
    for (auto &q : th) q.join();
    const double tm = ms(t);
    t = Clock::now();
    for (void *q : ptr) if (q) (void)hipFree(q);
    printf("8 x 1 GiB hipMalloc on %d thread(s): %8.2f ms wall (%.2f us/MB); hipFree of them %8.2f ms\n", nt, tm, tm * 1e3 / 8192, ms(t));
  }
  // launches beside an allocating thread
  {
    unsigned *d; CK(hipMalloc((void **)&d, 1 << 20));
    std::atomic<int> stop{0};
    double worst = 0, sum = 0; int n = 0;
    auto probe = [&]() {
      (void)hipSetDevice(0);
      while (!stop.load()) {
        auto t = Clock::now();
        hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d, (size_t)16384);
        (void)hipStreamSynchronize(s);
        const double x = ms(t); worst = x > worst ? x : worst; sum += x; n++;
      }
    };
    { std::thread a(probe); std::this_thread::sleep_for(std::chrono::milliseconds(300)); stop = 1; a.join(); }
    printf("launch+sync of a small kernel, machine quiet:            mean %.3f ms worst %.3f ms (%d launches)\n", sum / n, worst, n);
    stop = 0; worst = sum = 0; n = 0;
    std::vector<void *> ptr(8, nullptr);
    double tAlloc;
    { std::thread a(probe); auto t = Clock::now(); for (auto &q : ptr) (void)hipMalloc(&q, (size_t)1 << 30); tAlloc = ms(t); stop = 1; a.join(); }
    printf("the same while another thread allocates 8 x 1 GiB (%.1f ms): mean %.3f ms worst %.3f ms (%d launches)\n", tAlloc, sum / n, worst, n);
    stop = 0; worst = sum = 0; n = 0;
    { std::thread a(probe); auto t = Clock::now(); for (auto &q : ptr) (void)hipFree(q); tAlloc = ms(t); stop = 1; a.join(); }
    printf("the same while another thread frees them (%.1f ms):          mean %.3f ms worst %.3f ms (%d launches)\n", tAlloc, sum / n, worst, n);
    // a second allocating thread beside the first: small allocations (64 MB) while 8 GiB are being allocated
    stop = 0; worst = sum = 0; n = 0;
    auto small = [&]() {
      (void)hipSetDevice(0);
      while (!stop.load()) { void *q; auto t = Clock::now(); (void)hipMalloc(&q, 64 << 20); const double x = ms(t); worst = x > worst ? x : worst; sum += x; n++; (void)hipFree(q); }
    };
    { std::thread a(small); std::this_thread::sleep_for(std::chrono::milliseconds(200)); stop = 1; a.join(); }
    printf("64 MB hipMalloc, machine quiet:                              mean %.3f ms worst %.3f ms (%d)\n", sum / n, worst, n);
    stop = 0; worst = sum = 0; n = 0;
    { std::thread a(small); auto t = Clock::now(); void *big; (void)hipMalloc(&big, (size_t)8 << 30); tAlloc = ms(t); stop = 1; a.join(); (void)hipFree(big); }
    printf("64 MB hipMalloc beside one 8 GiB hipMalloc (%.1f ms):        mean %.3f ms worst %.3f ms (%d)\n", tAlloc, sum / n, worst, n);
  }
  for (size_t mb : {64, 512}) {
    void *h; auto t = Clock::now(); CK(hipHostMalloc(&h, mb << 20)); const double tm = ms(t);
    t = Clock::now(); CK(hipHostFree(h));
    printf("hipHostMalloc %zu MB: %.2f ms, hipHostFree %.2f ms\n", mb, tm, ms(t));
  }
  return 0;
}
