/* TEST INFRASTRUCTURE ONLY — fiber scheduler behind tests/emu/include/hip/hip_runtime.h. */
#include <hip/hip_runtime.h>
#include <mutex>
#include <ucontext.h>
#include <cstdlib>
#include <vector>
#include <stdexcept>

namespace hipemu {

emu_uint3 g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;

namespace {
enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
/* Context switch.  swapcontext() saves and restores the signal mask with a system call on every switch; a kernel of this library
 * switches at every barrier and wave collective of every lane (the register bitonic sort alone: hundreds of lane exchanges per
 * workgroup), and the CPU test suite spent most of its time there.  On x86-64 the switch is done by hand: callee-saved registers
 * and the stack pointer, nothing else (no signal mask, no floating-point control words: the kernels touch neither). */
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void **saveSp, void *loadSp);
asm(".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch, .-hipemu_switch\n");
#endif
struct Fiber {
#ifdef EMU_FAST_SWITCH
  void *sp = nullptr;
#else
  ucontext_t ctx;
#endif
  char *stack = nullptr;
  State st = DONE;
  emu_uint3 tid;
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> g_fibers;
#ifdef EMU_FAST_SWITCH
void *g_schedSp = nullptr;
#else
ucontext_t g_sched;
#endif
int g_cur = -1;
const std::function<void()> *g_body = nullptr;
/* per-wave exchange */
struct WaveBuf { uint64_t v[64]; uint64_t mask; };
std::vector<WaveBuf> g_waves;

inline void to_scheduler(int me) {
#ifdef EMU_FAST_SWITCH
  hipemu_switch(&g_fibers[me].sp, g_schedSp);
#else
  swapcontext(&g_fibers[me].ctx, &g_sched);
#endif
}
void fiber_entry() {
  (*g_body)();
  g_fibers[g_cur].st = DONE;
  to_scheduler(g_cur);
  abort();                       /* a finished fiber is never resumed */
}
void yield(State s) {
  int me = g_cur;
  g_fibers[me].st = s;
  to_scheduler(me);
  g_threadIdx = g_fibers[me].tid; /* restored by scheduler too; belt and braces */
}
/* makes fiber t start at fiber_entry on its own stack the next time it is switched to */
void fiber_reset(Fiber &f) {
#ifdef EMU_FAST_SWITCH
  uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
  void **sp = (void **)top;
  *--sp = nullptr;                      /* the return address fiber_entry would see: it never returns (and rsp = 8 mod 16 on entry) */
  *--sp = (void *)&fiber_entry;         /* popped by the `ret` of hipemu_switch */
  for (int i = 0; i < 6; i++) *--sp = nullptr;   /* rbp rbx r12 r13 r14 r15 */
  f.sp = (void *)sp;
#else
  getcontext(&f.ctx);
  f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = &g_sched;
  makecontext(&f.ctx, fiber_entry, 0);
#endif
}
inline void run_fiber(Fiber &f) {
#ifdef EMU_FAST_SWITCH
  hipemu_switch(&g_schedSp, f.sp);
#else
  swapcontext(&g_sched, &f.ctx);
#endif
}
}  // namespace

int lane_id() { return (int)(g_cur & 63); }

void sync_block() { yield(WAIT_BLOCK); }

const uint64_t *wave_gather(uint64_t v, uint64_t *mask) {
  int w = g_cur >> 6, l = g_cur & 63;
  g_waves[w].v[l] = v;
  yield(WAIT_WAVE);            /* rendezvous #1: everyone deposited */
  /* live mask = lanes of this wave that are not DONE */
  uint64_t m = 0;
  int n = (int)g_fibers.size();
  for (int i = 0; i < 64; i++) { int f = w * 64 + i; if (f < n && g_fibers[f].st != DONE) m |= 1ull << i; }
  *mask = m;
  return g_waves[w].v;
}
void wave_done() { yield(WAIT_WAVE); /* rendezvous #2: everyone has read */ }

static std::mutex g_launchMu;      // one kernel at a time: host threads driving several contexts (fastANI --devices) take turns
void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  std::lock_guard<std::mutex> launchLock(g_launchMu);
  size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nthreads == 0 || (size_t)grid.x * grid.y * grid.z == 0) return;
  if (g_fibers.size() < nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; i++) g_fibers[i].stack = (char *)malloc(kStack);
  }
  g_waves.resize((nthreads + 63) / 64);
  g_body = &body;
  g_blockDim = block; g_gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; bz++)
  for (unsigned by = 0; by < grid.y; by++)
  for (unsigned bx = 0; bx < grid.x; bx++) {
    g_blockIdx = {bx, by, bz};
    for (size_t t = 0; t < nthreads; t++) {
      Fiber &f = g_fibers[t];
      f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
      f.st = READY;
      fiber_reset(f);
    }
    for (;;) {
      bool any = false;
      for (size_t t = 0; t < nthreads; t++) {
        if (g_fibers[t].st != READY) continue;
        any = true;
        g_cur = (int)t; g_threadIdx = g_fibers[t].tid;
        run_fiber(g_fibers[t]);
      }
      /* everyone is now waiting or done: release waves first, then the block barrier */
      size_t done = 0, wb = 0, ww = 0;
      for (size_t t = 0; t < nthreads; t++) { State s = g_fibers[t].st; done += s == DONE; wb += s == WAIT_BLOCK; ww += s == WAIT_WAVE; }
      if (done == nthreads) break;
      bool released = false;
      if (ww) {
        for (size_t w = 0; w * 64 < nthreads; w++) {
          size_t lo = w * 64, hi = std::min(nthreads, lo + 64);
          size_t nw = 0, nb = 0;
          for (size_t t = lo; t < hi; t++) { nw += g_fibers[t].st == WAIT_WAVE; nb += g_fibers[t].st == WAIT_BLOCK; }
          if (nw && !nb) { for (size_t t = lo; t < hi; t++) if (g_fibers[t].st == WAIT_WAVE) g_fibers[t].st = READY; released = true; }
          else if (nw && nb) {
            fprintf(stderr, "hip-emu: divergent wave collective in block (%u,%u,%u) wave %zu: %zu lanes in a wave op, %zu at __syncthreads\n",
                    bx, by, bz, w, nw, nb);
            abort();
          }
        }
      }
      if (!released) {
        if (wb && wb + done == nthreads) { for (size_t t = 0; t < nthreads; t++) if (g_fibers[t].st == WAIT_BLOCK) g_fibers[t].st = READY; }
        else if (!any) { fprintf(stderr, "hip-emu: deadlock in block (%u,%u,%u)\n", bx, by, bz); abort(); }
      }
    }
  }
  g_cur = -1;
}

}  // namespace hipemu
