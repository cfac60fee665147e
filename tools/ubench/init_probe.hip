// init_probe.hip — where does the start of the HIP runtime go?  (The command line reaches "compute contexts up" 0.11 - 0.30 s after its start.)
//   init_probe [sleep seconds before]   prints the cost of each first call, in the order ani_init makes them
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p) *p = 1; }
int main(int argc, char **argv)
{
  const double t0 = now(); double t = t0;
  auto lap = [&](const char *what) { const double n = now(); printf("  %-44s %7.1f ms   (at %6.1f)\n", what, (n - t) * 1e3, (n - t0) * 1e3); t = n; };
  int n = 0; (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount (runtime + driver start)");
  (void)hipSetDevice(0); lap("hipSetDevice");
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); lap("hipStreamCreate #1");
  (void)hipStreamCreate(&s2); lap("hipStreamCreate #2");
  hipEvent_t ev[6]; for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); lap("6 x hipEventCreate");
  void *p = nullptr; (void)hipMalloc(&p, 1 << 20); lap("hipMalloc 1 MiB (first)");
  void *q = nullptr; (void)hipMalloc(&q, (size_t)1 << 30); lap("hipMalloc 1 GiB");
  (void)hipMemsetAsync(p, 0, 1 << 20, s1); (void)hipStreamSynchronize(s1); lap("hipMemsetAsync + sync (first kernel of the runtime)");
  k_nop<<<1, 64, 0, s1>>>((int *)p); (void)hipStreamSynchronize(s1); lap("first own kernel on stream 1 (code object load)");
  k_nop<<<1, 64, 0, s2>>>((int *)p); (void)hipStreamSynchronize(s2); lap("first kernel on stream 2");
  void *h = nullptr; (void)hipHostMalloc(&h, (size_t)64 << 20); lap("hipHostMalloc 64 MiB");
  hipStream_t s3; (void)hipStreamCreate(&s3); lap("hipStreamCreate #3");
  printf("  total %.1f ms\n", (now() - t0) * 1e3);
  fflush(stdout);
  _exit(0);
}
