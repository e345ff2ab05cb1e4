// wave_emu.cpp -- fiber scheduler of the lane emulator (see wave_emu.h).  TEST INFRASTRUCTURE.
#include "wave_emu.h"

#ifdef MODLE_EMU_THREADS
#include <pthread.h>

#include <thread>
#include <vector>
#endif

namespace wave_emu {

thread_local WaveRuntime* g_rt = nullptr;
#ifdef MODLE_EMU_THREADS
// every lane a thread, every collective a barrier (wave_emu.h)
thread_local int t_lane = 0;
static pthread_barrier_t g_barrier;
void threads_barrier() { pthread_barrier_wait(&g_barrier); }
void set_lane_schedule(unsigned) {}
void run_wave(void (*body)(void*), void* arg) {
  WaveRuntime rt;
  memset(&rt, 0, sizeof(rt));
  for (int l = 0; l < kLanes; ++l)
    for (int b = 0; b < 2; ++b) rt.slots[b][l].line = -1;
  pthread_barrier_init(&g_barrier, nullptr, kLanes);
  std::vector<std::thread> lanes;
  for (int l = 0; l < kLanes; ++l)
    lanes.emplace_back([&rt, body, arg, l]() {
      g_rt = &rt;
      t_lane = l;
      body(arg);
    });
  for (auto& t : lanes) t.join();
  pthread_barrier_destroy(&g_barrier);
}
}  // namespace wave_emu
#else
static unsigned g_schedule = 0;
void set_lane_schedule(unsigned schedule) { g_schedule = schedule; }

// callee-saved register context switch (SysV x86-64)
asm(R"(
.text
.globl modle_emu_switch
.type modle_emu_switch,@function
modle_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size modle_emu_switch,.-modle_emu_switch
)");

static void lane_entry() {
  WaveRuntime* rt = g_rt;
  rt->body(rt->arg);
  const int me = rt->cur;
  rt->done[me] = true;
  int nxt = -1;
  for (int k = 1; k <= kLanes; ++k) {
    const int c = rt->order[(rt->slot_of[me] + k) % kLanes];
    if (!rt->done[c]) {
      nxt = c;
      break;
    }
  }
  void* dummy;
  if (nxt < 0) {
    modle_emu_switch(&dummy, rt->main_sp);
  } else {
    rt->cur = nxt;
    modle_emu_switch(&dummy, rt->lane_sp[nxt]);
  }
  abort();  // a finished lane is never resumed
}

void run_wave(void (*body)(void*), void* arg) {
  constexpr size_t kStack = 512 * 1024;
  WaveRuntime rt;
  memset(&rt, 0, sizeof(rt));
  rt.stacks = static_cast<char*>(aligned_alloc(64, kStack * kLanes));
  rt.body = body;
  rt.arg = arg;
  for (int l = 0; l < kLanes; ++l) {
    for (int b = 0; b < 2; ++b) rt.slots[b][l].line = -1;
    char* top = rt.stacks + kStack * (l + 1);
    uintptr_t a = reinterpret_cast<uintptr_t>(top) & ~uintptr_t(15);
    a -= 16;  // return-address slot, 16-byte aligned
    void** sp = reinterpret_cast<void**>(a);
    sp[0] = reinterpret_cast<void*>(&lane_entry);
    sp[1] = nullptr;
    sp -= 6;  // rbp rbx r12 r13 r14 r15
    for (int k = 0; k < 6; ++k) sp[k] = nullptr;
    rt.lane_sp[l] = sp;
  }
  for (int l = 0; l < kLanes; ++l) rt.order[l] = g_schedule == 1 ? kLanes - 1 - l : l;
  if (g_schedule > 1) {
    uint64_t x = g_schedule;
    for (int l = kLanes - 1; l > 0; --l) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      const int j = static_cast<int>((x >> 33) % static_cast<uint64_t>(l + 1));
      const int t = rt.order[l];
      rt.order[l] = rt.order[j];
      rt.order[j] = t;
    }
  }
  for (int l = 0; l < kLanes; ++l) rt.slot_of[rt.order[l]] = l;
  WaveRuntime* prev = g_rt;
  g_rt = &rt;
  rt.cur = rt.order[0];
  modle_emu_switch(&rt.main_sp, rt.lane_sp[rt.order[0]]);
  g_rt = prev;
  free(rt.stacks);
}

}  // namespace wave_emu
#endif
