// TEST INFRASTRUCTURE ONLY.  Fiber-based SIMT interpreter behind tests/emu/rg_platform.h.
// One workgroup at a time; each GPU thread is a ucontext fiber scheduled round-robin.  Fibers
// switch only at __syncthreads() and wave collectives, so execution is deterministic.
#include "rg_platform.h"

#include <ucontext.h>

#include <cstdio>
#include <vector>

namespace emu {

namespace {
constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  ThreadCtx tc;
};

struct WaveState {
  std::vector<char> buf[2];
  int deposited = 0;
  long gen = 0;       // number of completed collectives
  int lanes = 0;      // live lanes in this wave
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<WaveState> g_waves;
int g_cur = -1;
int g_nthreads = 0;
int g_alive = 0;
int g_bar_arrived = 0;
long g_bar_gen = 0;
const std::function<void()>* g_body = nullptr;

void yield() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

void trampoline() {
  (*g_body)();
  Fiber& f = g_fibers[g_cur];
  f.done = true;
  g_alive--;
  WaveState& w = g_waves[f.tc.linear_tid >> 6];
  w.lanes--;
  if (w.lanes > 0 && w.deposited == w.lanes) {
    w.deposited = 0;
    w.gen++;
  }
  // a thread that exits while others wait at a barrier must not dead-lock them
  if (g_alive > 0 && g_bar_arrived == g_alive) {
    g_bar_arrived = 0;
    g_bar_gen++;
  }
  swapcontext(&f.ctx, &g_sched);
}
}  // namespace

ThreadCtx& cur() { return g_fibers[g_cur].tc; }

void sync_threads() {
  const long gen = g_bar_gen;
  if (++g_bar_arrived == g_alive) {
    g_bar_arrived = 0;
    g_bar_gen++;
    return;
  }
  while (g_bar_gen == gen) yield();
}

const void* wave_gather(const void* mine, size_t bytes) {
  Fiber& f = g_fibers[g_cur];
  WaveState& w = g_waves[f.tc.linear_tid >> 6];
  const int lane = f.tc.linear_tid & 63;
  const long gen = w.gen;
  std::vector<char>& b = w.buf[gen & 1];
  if (b.size() < 64 * bytes) b.resize(64 * bytes);
  memcpy(b.data() + (size_t)lane * bytes, mine, bytes);
  if (++w.deposited == w.lanes) {
    w.deposited = 0;
    w.gen++;
  } else {
    while (w.gen == gen) yield();
  }
  return w.buf[gen & 1].data();
}

static std::vector<char> g_dyn_lds;
char* dyn_lds() { return g_dyn_lds.data(); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t dyn_lds_bytes) {
  if (g_dyn_lds.size() < dyn_lds_bytes + 64) g_dyn_lds.resize(dyn_lds_bytes + 64);
  const int n = (int)(block.x * block.y * block.z);
  if (n <= 0 || n > kMaxThreads) {
    fprintf(stderr, "emu: bad block size %d\n", n);
    abort();
  }
  if ((int)g_fibers.size() < n) g_fibers.resize(n);
  for (int i = 0; i < n; ++i)
    if (!g_fibers[i].stack) g_fibers[i].stack = (char*)malloc(kStackBytes);
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_nthreads = n;
        g_alive = n;
        g_bar_arrived = 0;
        g_bar_gen = 0;
        g_waves.assign((n + 63) / 64, WaveState());
        for (int i = 0; i < n; ++i) {
          Fiber& f = g_fibers[i];
          f.done = false;
          f.tc.linear_tid = i;
          f.tc.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
          f.tc.bid = dim3(bx, by, bz);
          f.tc.bdim = block;
          f.tc.gdim = grid;
          g_waves[i >> 6].lanes++;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStackBytes;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, trampoline, 0);
        }
        long spins = 0;
        while (g_alive > 0) {
          for (int i = 0; i < n; ++i) {
            if (g_fibers[i].done) continue;
            g_cur = i;
            swapcontext(&g_sched, &g_fibers[i].ctx);
          }
          if (++spins > 100000000L) {
            fprintf(stderr, "emu: workgroup dead-locked (divergent barrier/collective?)\n");
            abort();
          }
        }
      }
  g_cur = -1;
  g_body = nullptr;
}

}  // namespace emu
