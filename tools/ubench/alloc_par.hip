// alloc_par.hip — on a box where fresh device memory is slow: does hipMalloc run in parallel on several threads, and does the piece size matter?
// Must be the FIRST thing that touches the device after the box came up (memory that has been allocated and released once is fast afterwards).
// Every phase takes memory that no phase before it had (everything stays allocated until the end).
//   hipcc --offload-arch=gfx950 -O2 -o alloc_par alloc_par.hip -lpthread && ./alloc_par
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }
static std::vector<void *> g_keep; static std::mutex *g_mu;
static void take(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, bytes) == hipSuccess) { std::lock_guard<std::mutex> g(*g_mu); g_keep.push_back(p); } else printf("hipMalloc(%zu) failed\n", bytes); }
static double phase(const char *what, int threads, int perThread, size_t bytes)
{
  auto t = Clock::now();
  std::vector<std::thread> th;
  for (int x = 0; x < threads; x++) th.emplace_back([=]() { (void)hipSetDevice(0); for (int i = 0; i < perThread; i++) take(bytes); });
  for (auto &q : th) q.join();
  const double d = ms(t), mb = (double)threads * perThread * (bytes >> 20);
  printf("%-62s %9.1f ms  (%6.2f us/MB over %.0f MB)\n", what, d, d * 1e3 / mb, mb);
  return d;
}
int main()
{
  std::mutex mu; g_mu = &mu;
  (void)hipSetDevice(0);
  { void *p; (void)hipMalloc(&p, 1 << 20); (void)hipFree(p); }
  const size_t G = (size_t)1 << 30;
  phase("A  4 threads x 1 x 4 GiB at once", 4, 1, 4 * G);
  phase("B  1 thread  x 4 x 4 GiB one after the other", 1, 4, 4 * G);
  phase("C  1 thread  x 16 x 1 GiB", 1, 16, G);
  phase("D  4 threads x 4 x 1 GiB", 4, 4, G);
  phase("E  1 thread  x 1 x 16 GiB", 1, 1, 16 * G);
  phase("F  1 thread  x 8 x 2 GiB", 1, 8, 2 * G);
  phase("G  8 threads x 1 x 2 GiB", 8, 1, 2 * G);
  auto t = Clock::now();
  for (void *p : g_keep) (void)hipFree(p);
  printf("hipFree of everything (%zu blocks): %.1f ms\n", g_keep.size(), ms(t));
  g_keep.clear();
  phase("H  1 thread  x 4 x 4 GiB again (memory that was released)", 1, 4, 4 * G);
  for (void *p : g_keep) (void)hipFree(p);
  return 0;
}
