// exit_probe.hip — what does the END of a HIP process cost?  (The command line's wall clock has ~0.17 s behind its last mark.)
//   exit_probe <device GB> <pinned MB> <threads>   allocates that, prints a time stamp and calls _exit(0); the caller measures stamp -> process gone
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
  const double gb = argc > 1 ? atof(argv[1]) : 0; const double pinnedMb = argc > 2 ? atof(argv[2]) : 0; const int nth = argc > 3 ? atoi(argv[3]) : 0;
  (void)hipSetDevice(0);
  void *p = nullptr; (void)hipMalloc(&p, 1 << 20);
  std::vector<void *> keep;
  for (double g = 0; g < gb; g += 1.0) { void *q = nullptr; if (hipMalloc(&q, (size_t)1 << 30) == hipSuccess) { (void)hipMemset(q, 0, (size_t)1 << 30); keep.push_back(q); } }
  void *h = nullptr; if (pinnedMb > 0) { (void)hipHostMalloc(&h, (size_t)(pinnedMb * 1048576.0)); if (h) ((volatile char *)h)[0] = 1; }
  (void)hipDeviceSynchronize();
  std::vector<std::thread> th;
  for (int i = 0; i < nth; i++) th.emplace_back([]() { for (;;) pause(); });
  for (auto &t : th) t.detach();
  const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
  printf("%.6f\n", now); fflush(stdout);
  _exit(0);
}
