// handover_model.cpp -- the hand-over protocol of the helper-wave mode on host threads, for
// ThreadSanitizer.
//
// TEST INFRASTRUCTURE.  On the GPU a cell may be served by up to three waves of one workgroup --
// main wave, helper, PRNG producer -- that pass work to each other through 24 words in LDS
// (modle_amd/csrc/sim_pair.h), six more for the fed stream (sim_rng.h / sim_helper.h) and a
// compare-and-swap claim for helpers that attach late.  The lane emulator runs one wave, so none of
// it had a CPU-side test; a lost hand-over on the GPU is a hung box.
//
// This program compiles THE PRODUCT'S OWN protocol code -- sim_pair.h (requests, waits, claim, open /
// close, dismissal), sim_helper.h (pair_serve, pair_feed) and the fed-stream consumer and block
// generator of sim_rng.h -- against a one-lane backend (wave_one_lane.h: a wave = one host thread,
// the protocol's words accessed with the very release / acquire / CAS operations the device code
// names) and runs main / helper / producer as std::threads.  What is replaced are the PASSES the
// helper runs (move generation, barrier states, LEF-BAR fwd, secondary filter fwd): stand-ins below
// that consume the stream at the same kind of positions and read / write the plain arrays the real
// passes hand over, with values the other side can check.  ThreadSanitizer then reports any plain
// access the protocol does not order, the value checks catch a hand-over of stale data, and a
// watchdog turns a lost hand-over into a failure instead of a hang -- by raising the abort word,
// which also shows that every spin loop of the protocol leaves when the host says so.
//
// Modes (argv[1]):
//   fixed      trios with fixed roles (launches that leave wave slots empty): several cells, all
//              three request kinds, producer sessions in every burn-in epoch
//   dynamic    launches that fill the slots: two main waves, idle waves that claim them at random
//              moments (pair_claim / pair_open / pair_close), many rounds
//   stuck      fixed roles with the test fault of sim_helper.h (the helper withholds the signals of
//              its third request): the watchdog raises the abort word, every thread must leave and
//              the main wave must report ERR_CANCELLED
// Built twice by the Makefile: `handover_model` (the code as shipped) and `handover_model_race`
// (-DMODLE_MODEL_SEEN_AFTER_CLAIM: the helper reads the request counter AFTER its claim, the race of
// DESIGN.md "A race worth writing down"): `dynamic` must pass on the first and be caught hanging by
// the watchdog on the second (exit code 3).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "wave_one_lane.h"
// clang-format off
#include "sim_cell.h"   // Cell, Rng, Workspace, WaveLds, the stream (sim_rng.h), plain data (sim_types.h)
#include "sim_pair.h"   // the main wave's side of the protocol, claim / open / close
// clang-format on
#include "host_prng.hpp"

namespace modle_dev {

// ---------------------------------------------------------------------------------------------
// Stand-ins for the passes the helper runs (and the main wave, when it has no helper).  Each one
// consumes the stream through the REAL rng_ensure / rng_advance (so the fed-stream protocol runs
// when a producer feeds the helper) and writes values the main wave can recompute.
// ---------------------------------------------------------------------------------------------
struct BoundaryCounts {
  u32 n5, n3;
};

// first output of the block that holds stream position p, read from the ring (lane 0's chunk: what
// the one lane of whoever generated the block has written)
MODLE_DEV u64 block_head(const Rng& g, u64 p) { return g.ring[ring_index(p & ~static_cast<u64>(RNG_BLOCK - 1))]; }

MODLE_DEV u32 move_value(u64 head0, u64 head1, u64 pos, u32 i) {
  return static_cast<u32>(head0 * 3 + head1 * 5 + pos * 7 + i);
}

MODLE_DEV_NOINLINE void generate_moves_by_id(Cell& c, f64, f64, u32* mv_by_id) {
  const u32 n = c.n_active;
  for (u32 base = 0; base < n; base += 64) {
    if (c.g.feed != nullptr) wave::st_release_wg(&c.g.feed[FEED_POS], static_cast<u32>(c.g.pos));
    rng_ensure(c.g, 65);
    const u64 h0 = block_head(c.g, c.g.pos), h1 = block_head(c.g, c.g.pos + 64);
    for (u32 j = 0; j < 64 && base + j < n; ++j) mv_by_id[base + j] = move_value(h0, h1, c.g.pos, base + j);
    rng_advance(c.g, 64 + (base / 64) % 2);
  }
  // (the real pass hands back what its last step evaluated beyond the last draw)
  if (n != 0) c.g.pos -= 17;
}

MODLE_DEV_NOINLINE void barriers_next_state(Cell& c) {
  const u32 nb = c.iv->n_barriers;
  u32 acc = 0;
  for (u32 base = 0; base < nb; base += 64) {
    rng_ensure(c.g, 64);
    acc += static_cast<u32>(block_head(c.g, c.g.pos)) + static_cast<u32>(c.g.pos);
    for (u32 j = 0; j < 64 && base + j < nb; ++j) c.ws.bar_active[base + j] = static_cast<u8>((acc + j) & 1u);
    rng_advance(c.g, 64);
  }
  c.n_hit[0] = acc & 0xFFu;
  c.n_hit[1] = (acc >> 8) & 0xFFu;
  for (u32 d = 0; d < 2; ++d)
    for (u32 e = 0; e < c.n_hit[d]; ++e) c.ws.hit_pos[d][e] = acc + 31 * d + e;
}

MODLE_DEV u32 coll_value(u32 pos, u32 move, u32 n3, u32 nh) { return (pos * 2654435761u) ^ move ^ (n3 << 7) ^ nh; }

template <bool FWD>
MODLE_DEV_NOINLINE void detect_lef_bar(Cell& c, BoundaryCounts bc) {
  const u32* pos = FWD ? c.ws.f_pos : c.ws.r_pos;
  const u32* mv = FWD ? c.ws.f_move : c.ws.r_move;
  u32* coll = FWD ? c.ws.f_coll : c.ws.r_coll;
  const u32* hp = c.ws.hit_pos[FWD ? 1 : 0];
  const u32 nh = c.n_hit[FWD ? 1 : 0];
  for (u32 k = 0; k < c.n_active; ++k)
    coll[k] = coll_value(pos[k], mv[k], FWD ? bc.n3 : bc.n5, nh != 0 ? hp[k % nh] : 0u);
}

template <bool FWD>
struct SecondaryFilter {
  Cell* c;
  BoundaryCounts bc;
  u32 nblk, n_cand, cap;
  void init(Cell& cell, BoundaryCounts b, u32 list_cap, bool, bool) {
    c = &cell;
    bc = b;
    cap = list_cap;
    nblk = (cell.n_active + 255) / 256;
    n_cand = 0;
  }
  void step(u32 t) {
    Workspace& ws = c->ws;
    const u32* pos = FWD ? ws.f_pos : ws.r_pos;
    u32* mv = FWD ? ws.f_move : ws.r_move;
    const u32* coll = FWD ? ws.f_coll : ws.r_coll;
    u32* list = FWD ? ws.tmp[1] : ws.tmp[0];
    for (u32 k = 256 * t; k < 256 * (t + 1) && k < c->n_active; ++k) {
      mv[k] += coll[k] & 3u;  // ("LEF-BAR move correction")
      if (((pos[k] ^ coll[k]) & 7u) == 0 && n_cand < cap) list[n_cand++] = k;
    }
  }
};

}  // namespace modle_dev

#include "sim_helper.h"  // the helper's and the producer's loops

using namespace modle_dev;

namespace {

constexpr u32 kWaves = 8;
constexpr u32 kLefs = 700;      // active LEFs of a model cell
constexpr u32 kBarriers = 300;
constexpr u32 kCap = 1024;

std::vector<u64> g_jump;
u32 g_abort = 0;  // the host's abort word (accessed with load_system_u32 by the waves)
std::atomic<u64> g_progress{0};
std::atomic<bool> g_failed{false};

#define MODEL_CHECK(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) {                                               \
      std::fprintf(stderr, "MODEL CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                         \
      std::fprintf(stderr, "\n");                                \
      g_failed = true;                                           \
    }                                                            \
  } while (0)

// everything one main wave owns on the device: its LDS (ring, lane states, snapshots, staging) and
// its slice of the workspace
struct WaveMemory {
  std::vector<u64> ring = std::vector<u64>(RNG_RING, 0), state = std::vector<u64>(RNG_STATE_WORDS, 0), snap = std::vector<u64>(8, 0);
  std::vector<u64> sort = std::vector<u64>(SORT_LDS_CAP, 0);
  std::vector<u32> stage = std::vector<u32>(STAGE_CAP, 0);
  std::vector<u32> arr[24];
  std::vector<u8> bar_active = std::vector<u8>(kBarriers + 64, 0);
  WaveMemory() {
    for (auto& a : arr) a.assign(kCap, 0);
  }
  WaveLds lds() {
    WaveLds l{};
    l.ring = ring.data();
    l.rng_state = state.data();
    l.rng_snap = snap.data();
    l.jump_table = g_jump.data();
    l.sort_lds = sort.data();
    l.stage = stage.data();
    l.abort_flag = &g_abort;
    return l;
  }
  Workspace ws() {
    Workspace w{};
    w.r_pos = arr[0].data(); w.r_id = arr[1].data(); w.r_move = arr[2].data(); w.r_coll = arr[3].data();
    w.f_pos = arr[4].data(); w.f_id = arr[5].data(); w.f_move = arr[6].data(); w.f_coll = arr[7].data();
    w.epoch = arr[8].data(); w.r_rank = arr[9].data(); w.f_rank = arr[10].data(); w.stall = arr[11].data();
    for (u32 k = 0; k < NUM_TMP; ++k) w.tmp[k] = arr[12 + k].data();
    w.bar_active = bar_active.data();
    w.hit_pos[0] = arr[22].data();
    w.hit_pos[1] = arr[22].data() + kCap / 2;
    w.hit_idx[0] = arr[23].data();
    w.hit_idx[1] = arr[23].data() + kCap / 2;
    w.capacity_lefs = kCap;
    w.capacity_barriers = kBarriers;
    return w;
  }
};

struct Workgroup {
  alignas(64) u32 pairbox[kWaves][PAIR_WORDS] = {};
  alignas(64) u32 feedbox[kWaves][PAIR_WORDS] = {};  // (device: the first lane-state words of the producer wave)
  WaveMemory mem[kWaves];
  Interval interval{};
  Params params{};
  Workgroup() {
    interval.n_barriers = kBarriers;
    interval.start = 0;
    interval.end = 1u << 30;
  }
};

// what the main wave expects of one burn-in epoch's moves + barrier states, recomputed from the
// reference stream (sequential xoshiro of the cell's seed): outputs at the head of every block
struct ReferenceStream {
  std::vector<u64> head;  // head[b] = output 512 b of the stream
  explicit ReferenceStream(const u64 seed[4], size_t blocks) : head(blocks) {
    u64 s[4] = {seed[0], seed[1], seed[2], seed[3]};
    for (size_t b = 0; b < blocks; ++b) {
      for (u32 k = 0; k < RNG_BLOCK; ++k) {
        const u64 x = modle_host::xoshiro_next(s);
        if (k == 0) head[b] = x;
      }
    }
  }
  u64 at(u64 pos) const { return head.at(pos / RNG_BLOCK); }
};

// the move generation + barrier update of one epoch on the reference stream: checks the arrays a
// helper (or this wave) has filled and returns the position the generator must come back with
u64 check_moves_and_barriers(const ReferenceStream& ref, const Workspace& ws, u64 pos, u32 n, u32 nb, const u32 n_hit[2],
                             const char* who) {
  for (int dir = 0; dir < 2; ++dir) {
    const u32* mv = ws.tmp[8 + dir];
    for (u32 base = 0; base < n; base += 64) {
      const u64 h0 = ref.at(pos), h1 = ref.at(pos + 64);
      for (u32 j = 0; j < 64 && base + j < n; ++j)
        MODEL_CHECK(mv[base + j] == move_value(h0, h1, pos, base + j), "%s: move %u of direction %d is not the stream's", who,
                    base + j, dir);
      pos += 64 + (base / 64) % 2;
    }
    if (n != 0) pos -= 17;
  }
  u32 acc = 0;
  for (u32 base = 0; base < nb; base += 64) {
    acc += static_cast<u32>(ref.at(pos)) + static_cast<u32>(pos);
    for (u32 j = 0; j < 64 && base + j < nb; ++j)
      MODEL_CHECK(ws.bar_active[base + j] == static_cast<u8>((acc + j) & 1u), "%s: barrier state %u", who, base + j);
    pos += 64;
  }
  MODEL_CHECK(n_hit[0] == (acc & 0xFFu) && n_hit[1] == ((acc >> 8) & 0xFFu), "%s: lengths of the stalling lists", who);
  return pos;
}

// One cell on a main wave: the sequence of simulate_cell (sim_epoch.h) / phase_process_collisions /
// process_secondary_both (sim_collisions.h) around the hand-overs, with the real pair_* calls.
// Returns the cell's status (0, or ERR_CANCELLED when a wait was abandoned).
u32 run_cell(Workgroup& wg, u32 w, u32 cell_no, u32 epochs, std::mt19937& rnd) {
  Cell c{};
  WaveMemory& mem = wg.mem[w];
  c.p = &wg.params;
  c.iv = &wg.interval;
  c.ws = mem.ws();
  c.lds = mem.lds();
  c.lds.mbox = wg.pairbox[w];
  c.lds.pair_dynamic = (wave::ld_acquire_wg(&wg.pairbox[w][PAIR_STATE]) != PAIR_IDLE);
  c.n_lefs = c.n_active = kLefs;
  c.pair_seq = c.lds.mbox[PAIR_REQ];  // (init_cell: the helper keeps counting across the tasks of its main wave)
  c.pair_interval = 0;
  c.g.ring = c.lds.ring;
  c.g.jump = c.lds.jump_table;
  c.g.state = c.lds.rng_state;
  c.g.snap = c.lds.rng_snap;
  u64 seed[4];
  modle_host::splitmix_seed(1000 * w + cell_no, seed);
  rng_init(c.g, seed);
  const ReferenceStream ref(seed, 16 + static_cast<size_t>(epochs) * 12);
  Workspace& ws = c.ws;
  const u32 n = c.n_active, nb = wg.interval.n_barriers;
  for (u32 e = 0; e < epochs; ++e) {
    const bool burnin = e < epochs - 2;
    // bind: the main wave draws from the stream itself (ring, lane states: its own again)
    rng_ensure(c.g, 5);
    MODEL_CHECK(block_head(c.g, c.g.pos) == ref.at(c.g.pos), "wave %u cell %u epoch %u: the ring does not hold the stream at %llu",
                w, cell_no, e, static_cast<unsigned long long>(c.g.pos));
    rng_advance(c.g, 5);
    c.pair_on = pair_helper_present(c.lds);
    const bool offload = c.pair_on && burnin;
    const u64 pos0 = c.g.pos;
    if (offload) pair_request(c, false, 0);
    // rank updates (no draws): the unit arrays the later requests hand over
    for (u32 k = 0; k < n; ++k) {
      ws.r_pos[k] = e * 1000003u + k;
      ws.f_pos[k] = e * 999983u + 2 * k + 1;
    }
    if (rnd() % 4 == 0) std::this_thread::yield();
    if (offload) {
      if (!pair_wait(c, PAIR_MOVES)) return c.error;
      // move adjustment: reads the helper's moves while the helper goes on with the barrier states
      for (u32 k = 0; k < n; ++k) {
        ws.r_move[k] = ws.tmp[8][k] & 0xFFFFu;
        ws.f_move[k] = ws.tmp[9][k] & 0xFFFFu;
      }
      if (!pair_take_back(c)) return c.error;
    } else {
      generate_moves_by_id(c, 0, 0, ws.tmp[8]);
      generate_moves_by_id(c, 0, 0, ws.tmp[9]);
      for (u32 k = 0; k < n; ++k) {
        ws.r_move[k] = ws.tmp[8][k] & 0xFFFFu;
        ws.f_move[k] = ws.tmp[9][k] & 0xFFFFu;
      }
      barriers_next_state(c);
    }
    const u64 pos1 = check_moves_and_barriers(ref, ws, pos0, n, nb, c.n_hit, offload ? "helper" : "main");
    MODEL_CHECK(c.g.pos == pos1, "wave %u cell %u epoch %u: the generator came back at %llu, expected %llu", w, cell_no, e,
                static_cast<unsigned long long>(c.g.pos), static_cast<unsigned long long>(pos1));
    // LEF-BAR detection: fwd instance on the helper, rev instance here
    const BoundaryCounts bc{e % 5, e % 7};
    if (c.pair_on) {
      pair_request_lef_bar(c, bc.n5, bc.n3);
      detect_lef_bar<false>(c, bc);
      if (!pair_wait(c, PAIR_ALL)) return c.error;
    } else {
      detect_lef_bar<false>(c, bc);
      detect_lef_bar<true>(c, bc);
    }
    for (u32 k = 0; k < n; ++k) {
      const u32 nh = c.n_hit[1];
      MODEL_CHECK(ws.f_coll[k] == coll_value(ws.f_pos[k], ws.f_move[k], bc.n3, nh != 0 ? ws.hit_pos[1][k % nh] : 0u),
                  "wave %u cell %u epoch %u: fwd collision word %u", w, cell_no, e, k);
    }
    // secondary pass: fwd filter on the helper while this wave runs the rev filter and the rev
    // resolve pass (which draws)
    std::vector<u32> f_move_before(ws.f_move, ws.f_move + n);
    SecondaryFilter<false> fr;
    fr.init(c, bc, kCap, true, true);
    u32 n_cand_fwd = 0;
    if (c.pair_on) {
      pair_request_sec_filter(c, bc.n5, bc.n3, kCap);
      for (u32 t = 0; t < fr.nblk; ++t) fr.step(t);
    } else {
      SecondaryFilter<true> ff;
      ff.init(c, bc, kCap, true, true);
      for (u32 t = 0; t < fr.nblk; ++t) {
        fr.step(t);
        ff.step(t);
      }
      n_cand_fwd = ff.n_cand;
    }
    rng_ensure(c.g, 9);  // (rev resolve pass: draws on this wave)
    rng_advance(c.g, 9);
    if (c.pair_on) {
      n_cand_fwd = pair_take_sec_filter(c);
      if (c.error != 0) return c.error;
    }
    u32 expect_cand = 0;
    for (u32 k = 0; k < n; ++k) {
      MODEL_CHECK(ws.f_move[k] == f_move_before[k] + (ws.f_coll[k] & 3u), "wave %u cell %u epoch %u: fwd move %u after the filter",
                  w, cell_no, e, k);
      if (((ws.f_pos[k] ^ ws.f_coll[k]) & 7u) == 0) {
        MODEL_CHECK(expect_cand < n_cand_fwd && ws.tmp[1][expect_cand] == k, "wave %u cell %u epoch %u: fwd candidate list", w,
                    cell_no, e);
        ++expect_cand;
      }
    }
    MODEL_CHECK(expect_cand == n_cand_fwd, "wave %u cell %u epoch %u: %u fwd candidates, expected %u", w, cell_no, e, n_cand_fwd,
                expect_cand);
    // extrusion + release: draws on this wave
    rng_ensure(c.g, 70);
    MODEL_CHECK(block_head(c.g, c.g.pos + 69) == ref.at(c.g.pos + 69), "wave %u cell %u epoch %u: ring after the hand-overs", w,
                cell_no, e);
    rng_advance(c.g, 70);
    g_progress.fetch_add(1, std::memory_order_relaxed);
    if (g_failed) return ERR_INTERNAL;
  }
  return 0;
}

// the helper's context, as the kernel builds it (modle_hip.hip): the main wave's generator, tables
// and workspace, its own staging and sort buffers
Cell helper_cell(Workgroup& wg, u32 self, u32 main_wave) {
  Cell c{};
  c.p = &wg.params;
  c.lds = wg.mem[main_wave].lds();
  c.lds.stage = wg.mem[self].stage.data();
  c.lds.sort_lds = wg.mem[self].sort.data();
  c.ws = wg.mem[main_wave].ws();
  c.g.ring = c.lds.ring;
  c.g.jump = c.lds.jump_table;
  c.g.state = c.lds.rng_state;
  c.g.snap = c.lds.rng_snap;
  return c;
}

// raises the abort word when no epoch has completed for `patience`; returns true when it had to
struct Watchdog {
  std::atomic<bool> stop{false}, fired{false};
  std::thread th;
  explicit Watchdog(std::chrono::milliseconds patience) {
    th = std::thread([this, patience] {
      u64 last = g_progress.load();
      auto since = std::chrono::steady_clock::now();
      while (!stop) {
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        const u64 now = g_progress.load();
        if (now != last) {
          last = now;
          since = std::chrono::steady_clock::now();
        } else if (std::chrono::steady_clock::now() - since > patience) {
          __atomic_store_n(&g_abort, 1u, __ATOMIC_RELEASE);
          fired = true;
          return;
        }
      }
    });
  }
  bool finish() {
    stop = true;
    th.join();
    return fired;
  }
};

int mode_fixed(u32 test_fault) {
  // pair_mains = 1: wave 0 main, wave 7 helper, wave 2 producer (modle_hip.hip)
  Workgroup wg;
  std::mt19937 rnd(7);
  u32 status = 0;
  Watchdog dog(std::chrono::milliseconds(test_fault != 0 ? 300 : 5000));
  std::thread helper([&] {
    Cell c = helper_cell(wg, 7, 0);
    pair_serve(c, &wg.interval, wg.pairbox[0], wg.feedbox[2], 0, test_fault);
  });
  std::thread producer([&] {
    WaveLds lm = wg.mem[0].lds();
    pair_feed(lm.ring, lm.jump_table, lm.rng_state, lm.rng_snap, wg.feedbox[2], &g_abort);
  });
  for (u32 cell = 0; cell < 6 && status == 0; ++cell) status = run_cell(wg, 0, cell, 40, rnd);
  pair_dismiss(wg.pairbox[0]);
  helper.join();
  producer.join();
  const bool fired = dog.finish();
  if (test_fault != 0) {
    // the helper withheld a signal: the main wave must have been released by the abort word
    if (fired && status == ERR_CANCELLED && !g_failed) {
      std::printf("stuck: the abort word released every wave; the cell reports ERR_CANCELLED\n");
      return 0;
    }
    std::fprintf(stderr, "stuck: fired=%d status=%u failed=%d\n", int(fired), status, int(g_failed.load()));
    return 1;
  }
  if (fired) return 3;
  if (status != 0 || g_failed) return 1;
  std::printf("fixed: %llu epochs, every hand-over checked\n", static_cast<unsigned long long>(g_progress.load()));
  return 0;
}

int mode_dynamic(u32 rounds) {
  std::mt19937 seed_rnd(11);
  Watchdog dog(std::chrono::milliseconds(2000));
  u64 helped = 0;
  for (u32 round = 0; round < rounds; ++round) {
    Workgroup wg;
    std::atomic<u32> status{0};
    std::atomic<u64> served_epochs{0};
    const u32 r0 = seed_rnd();
    std::vector<std::thread> th;
    // two main waves, each with a few short cells; between them they are OPEN for most of the round
    for (u32 w = 0; w < 2; ++w) {
      th.emplace_back([&, w] {
        std::mt19937 rnd(r0 + w);
        pair_open(wg.pairbox[w]);
        for (u32 cell = 0; cell < 3 && status == 0; ++cell) {
          const u64 before = g_progress.load();
          const u32 st = run_cell(wg, w, cell, 4 + rnd() % 6, rnd);
          (void)before;
          if (st != 0) status = st;
        }
        pair_close(wg.pairbox[w]);
      });
    }
    // idle waves: find the queue empty at some moment, claim a running main wave, serve it until it
    // is dismissed, look for another (the kernel's loop in modle_hip.hip)
    for (u32 w = 2; w < 5; ++w) {
      th.emplace_back([&, w] {
        std::mt19937 rnd(r0 + 100 + w);
        std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 1500));
        for (;;) {
          u32 seen = 0;
          const int main_wave = pair_claim(&wg.pairbox[0][0], kWaves, static_cast<int>(w), seen);
          if (main_wave < 0) return;
          Cell c = helper_cell(wg, w, static_cast<u32>(main_wave));
          c.lds.abort_flag = &g_abort;
          pair_serve(c, &wg.interval, wg.pairbox[main_wave], nullptr, seen, 0);
          served_epochs.fetch_add(1);
          if (wave::load_system_u32(&g_abort) != 0) return;
        }
      });
    }
    for (auto& t : th) t.join();
    helped += served_epochs.load();
    if (status != 0 || g_failed || wave::load_system_u32(&g_abort) != 0) break;
  }
  const bool fired = dog.finish();
  if (fired) {
    std::fprintf(stderr, "dynamic: a hand-over was lost (no epoch completed for 2 s): the watchdog raised the abort word\n");
    return 3;
  }
  if (g_failed) return 1;
  std::printf("dynamic: %u rounds, %llu epochs, %llu helper attachments, every hand-over checked\n", rounds,
              static_cast<unsigned long long>(g_progress.load()), static_cast<unsigned long long>(helped));
  return helped == 0 ? 1 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s fixed|dynamic|stuck [rounds]\n", argv[0]);
    return 2;
  }
  g_jump = modle_host::build_jump_table(RNG_HOP);
  const std::string mode = argv[1];
  if (mode == "fixed") return mode_fixed(0);
  if (mode == "stuck") return mode_fixed(TEST_FAULT_STUCK_HELPER);
  if (mode == "dynamic") return mode_dynamic(argc > 2 ? static_cast<u32>(std::atoi(argv[2])) : 200);
  return 2;
}
