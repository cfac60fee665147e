#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
std::atomic<int> started{0};
int tids[256];
int main(int argc, char **argv) {
  size_t gb = atoi(argv[1]); int nth = atoi(argv[2]);
  char *p = (char *)mmap(0, gb << 30, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  for (size_t i = 0; i < (gb << 30); i += 4096) p[i] = 1;
  for (int i = 0; i < nth; i++) std::thread([i]() { tids[i] = (int)syscall(SYS_gettid); started++; for (;;) pause(); }).detach();
  while (started.load() < nth) usleep(100);
  for (int i = 0; i < nth; i++) printf("%d ", tids[i]);
  double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
  printf("\n%.6f\n", now); fflush(stdout);
  _exit(0);
}
