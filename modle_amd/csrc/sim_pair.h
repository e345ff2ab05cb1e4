// sim_pair.h -- part of sim_device.h (included by it, in this order): helper-wave mode.
//
// A launch with fewer tasks than half the wave slots of the GPU (BASELINE config 1: 512 cells of
// chr1 on 2048 slots) lasts as long as its longest cell, and one wave runs a cell no faster than
// a CPU core of the reference does (scheduler_simulate.cpp:190-271: one cell per worker).  In such
// launches a second wave of the workgroup -- the helper -- takes over the part of a burn-in epoch
// that depends on nothing but the PRNG stream: generate_moves (simulation.cpp:272-330) followed by
// ExtrusionBarriers::next_state (extrusion_barriers.cpp:219-230), which come one after the other
// in the stream, while the main wave runs the two rank updates (no draws) and then the move
// adjustment.  The stream stays ONE stream consumed in the reference's order: the generator
// (ring, lane states, snapshots: LDS of the main wave; position and end of the ring: handed over)
// belongs to exactly one of the two waves at any time.
//
//   main:    bind | request | rank rev, rank fwd | wait moves | adjust moves | wait all | ...
//   helper:        | moves rev, moves fwd | signal moves | barrier states + stalling lists | signal all
//
// Second kind of request, in every epoch: LEF-BAR detection without Bernoulli trials draws nothing
// and its rev and fwd instances touch disjoint arrays (simulation_detect_collisions.cpp:122-247):
// the helper runs the fwd instance while the main wave runs the rev instance.
//
//   main:    ... boundaries | request | LEF-BAR rev | wait all | primary LEF-LEF ...
//   helper:                 | LEF-BAR fwd | signal all
//
// Third kind, in every epoch: the filter pass of the secondary LEF-LEF pass (LEF-BAR move
// corrections + the list of candidates; no draws, one direction's arrays).  The helper runs the
// fwd filter while the main wave runs the rev filter AND the rev resolve pass (which draws, and
// touches rev arrays only); the fwd resolve pass then takes the helper's list.
//
//   main:    ... primary | request | rev filter | rev resolve | wait all | fwd resolve ...
//   helper:              | fwd filter | signal all (number of candidates)
//
// This file holds the main wave's side; the helper's loop is in sim_helper.h.
//
#pragma once

namespace modle_dev {

// Hand-over words of main wave w (u32, in LDS: BlockLds::pairbox[w] in modle_hip.hip):
constexpr u32 PAIR_REQ = 0;        // main -> helper: sequence number of the request
constexpr u32 PAIR_MOVES = 1;      // helper -> main: the moves of request <seq> are in device memory
constexpr u32 PAIR_ALL = 2;        // helper -> main: barrier states, lists and generator are back
constexpr u32 PAIR_N_ACTIVE = 3;   // request: active LEFs (PAIR_EXIT: the main wave has no more tasks)
constexpr u32 PAIR_BURNIN_DONE = 4;
constexpr u32 PAIR_INTERVAL = 5;   // request: index of the task's interval
constexpr u32 PAIR_POS = 6;        // generator: stream position (2 words), there and back
constexpr u32 PAIR_GEN_END = 8;    // generator: end of the ring (2 words), there and back
constexpr u32 PAIR_N_HIT = 10;     // reply: entries of the two lists of stalling barriers (2 words);
                                   // secondary-filter reply: the number of candidates (first word)
constexpr u32 PAIR_KIND = 12;      // request: PAIR_KIND_MOVES / PAIR_KIND_LEF_BAR / PAIR_KIND_SEC_FILTER
constexpr u32 PAIR_BC = 13;        // LEF-BAR request: BoundaryCounts (2 words)
constexpr u32 PAIR_ERR = 15;       // reply: error status of the helper's side of the request (0 = none); written
                                   // before PAIR_MOVES / PAIR_ALL, folded into the cell's status by the main wave
constexpr u32 PAIR_F_POS = 16;     // LEF-BAR request: the fwd position / move arrays (the rank updates
constexpr u32 PAIR_F_MOVE = 18;    // and the move adjustment swap workspace pointers) (2 words each)
constexpr u32 PAIR_STATE = 20;     // launches that fill the slots: PAIR_IDLE / PAIR_OPEN / PAIR_TAKEN (below)
constexpr u32 PAIR_LIST_CAP = 21;  // secondary-filter request: capacity of the candidate list
constexpr u32 PAIR_Q = 22;         // secondary-filter request: the fwd candidate list (2 words)
constexpr u32 PAIR_WORDS = 24;
constexpr u32 PAIR_EXIT = 0xFFFFFFFFu;
constexpr u32 PAIR_KIND_MOVES = 0, PAIR_KIND_LEF_BAR = 1, PAIR_KIND_SEC_FILTER = 2;
// Launches that fill the wave slots have an idle tail (the queue is empty, the last cells are still
// running: 4.7 % of the slot time of BASELINE config 2).  A wave that finds the queue empty then
// becomes the helper of a main wave of its workgroup that is still running and has none:
//   PAIR_IDLE   the main wave is not running a task loop (never started, or left it)
//   PAIR_OPEN   it is, and has no helper: an idle wave may claim it (compare-and-swap to PAIR_TAKEN)
//   PAIR_TAKEN  a helper is attached; the main wave looks at the word once per epoch and hands work
//               over from then on; when it leaves its task loop it swaps PAIR_IDLE in and, if the
//               word was PAIR_TAKEN, dismisses the helper (which then looks for another main wave)
constexpr u32 PAIR_IDLE = 0, PAIR_OPEN = 1, PAIR_TAKEN = 2;

// (every spin loop below is bounded by the host through spin_nap_aborted, sim_rng.h)
MODLE_DEV void pair_put_u64(u32* m, u32 at, u64 v) {
  m[at] = static_cast<u32>(v);
  m[at + 1] = static_cast<u32>(v >> 32);
}
MODLE_DEV u64 pair_get_u64(const u32* m, u32 at) {
  return wave::uniform(static_cast<u64>(m[at]) | (static_cast<u64>(m[at + 1]) << 32));
}

// main wave: hands the generator to the helper together with what the two phases need
MODLE_DEV void pair_request(Cell& c, bool burnin_completed, u32 interval) {
  u32* m = c.lds.mbox;
  c.pair_interval = interval;
  wave::lockstep();
  if (wave::lane() == 0) {
    m[PAIR_N_ACTIVE] = c.n_active;
    m[PAIR_BURNIN_DONE] = burnin_completed ? 1u : 0u;
    m[PAIR_INTERVAL] = interval;
    m[PAIR_KIND] = PAIR_KIND_MOVES;
    pair_put_u64(m, PAIR_POS, c.g.pos);
    pair_put_u64(m, PAIR_GEN_END, c.g.gen_end);
  }
  ++c.pair_seq;
  c.ring_lent = true;
  wave::st_release_wg(&m[PAIR_REQ], c.pair_seq);
}
// main wave: the fwd instance of LEF-BAR detection goes to the helper (n5, n3: BoundaryCounts)
MODLE_DEV void pair_request_lef_bar(Cell& c, u32 n5, u32 n3) {
  u32* m = c.lds.mbox;
  wave::lockstep();
  if (wave::lane() == 0) {
    m[PAIR_N_ACTIVE] = c.n_active;
    m[PAIR_INTERVAL] = c.pair_interval;
    m[PAIR_KIND] = PAIR_KIND_LEF_BAR;
    m[PAIR_BC] = n5;
    m[PAIR_BC + 1] = n3;
    m[PAIR_N_HIT + 1] = c.n_hit[1];
    pair_put_u64(m, PAIR_F_POS, reinterpret_cast<u64>(c.ws.f_pos));
    pair_put_u64(m, PAIR_F_MOVE, reinterpret_cast<u64>(c.ws.f_move));
  }
  ++c.pair_seq;
  wave::st_release_wg(&m[PAIR_REQ], c.pair_seq);
}
MODLE_DEV bool pair_wait(Cell& c, u32 what);
// main wave: the fwd filter of the secondary pass goes to the helper
MODLE_DEV void pair_request_sec_filter(Cell& c, u32 n5, u32 n3, u32 list_cap) {
  u32* m = c.lds.mbox;
  wave::lockstep();
  if (wave::lane() == 0) {
    m[PAIR_N_ACTIVE] = c.n_active;
    m[PAIR_INTERVAL] = c.pair_interval;
    m[PAIR_KIND] = PAIR_KIND_SEC_FILTER;
    m[PAIR_BC] = n5;
    m[PAIR_BC + 1] = n3;
    m[PAIR_LIST_CAP] = list_cap;
    pair_put_u64(m, PAIR_F_POS, reinterpret_cast<u64>(c.ws.f_pos));
    pair_put_u64(m, PAIR_F_MOVE, reinterpret_cast<u64>(c.ws.f_move));
    pair_put_u64(m, PAIR_Q, reinterpret_cast<u64>(c.ws.tmp[1]));
  }
  ++c.pair_seq;
  wave::st_release_wg(&m[PAIR_REQ], c.pair_seq);
}
// ... and its answer: the number of candidates it listed (0 with c.error set when the wait failed)
MODLE_DEV u32 pair_take_sec_filter(Cell& c) {
  if (!pair_wait(c, PAIR_ALL)) return 0;
  return wave::uniform(c.lds.mbox[PAIR_N_HIT]);
}
// main wave: waits until the helper has signalled `what` (PAIR_MOVES / PAIR_ALL) for the request.
// false (and c.error set) when the host raised the abort word meanwhile or the helper reported an
// error of its own (reported with PAIR_ALL): what the helper was to produce is then not there, the
// caller leaves the epoch.
MODLE_DEV bool pair_wait(Cell& c, u32 what) {
  const u32* m = c.lds.mbox;
  u32 spins = 0;
  while (wave::uniform(wave::ld_acquire_wg(&m[what])) != c.pair_seq) {
    if (spin_nap_aborted(c.lds.abort_flag, spins)) {
      c.error = ERR_CANCELLED;
      return false;
    }
  }
  // the helper's status of the request comes with PAIR_ALL (it writes the word once per request)
  const u32 helper_error = what == PAIR_ALL ? wave::uniform(m[PAIR_ERR]) : 0u;
  if (helper_error != 0) {
    c.error = helper_error;
    return false;
  }
  return true;
}
// main wave: the generator and the lists of stalling barriers come back
MODLE_DEV bool pair_take_back(Cell& c) {
  if (!pair_wait(c, PAIR_ALL)) return false;
  const u32* m = c.lds.mbox;
  c.g.pos = pair_get_u64(m, PAIR_POS);
  c.g.gen_end = pair_get_u64(m, PAIR_GEN_END);
  c.n_hit[0] = wave::uniform(m[PAIR_N_HIT]);
  c.n_hit[1] = wave::uniform(m[PAIR_N_HIT + 1]);
  c.ring_lent = false;
  return true;
}
// main wave, once per epoch: is there a helper to hand work to?
MODLE_DEV bool pair_helper_present(const WaveLds& lds) {
  if (lds.mbox == nullptr) return false;
  if (!lds.pair_dynamic) return true;
  return wave::uniform(wave::ld_acquire_wg(&lds.mbox[PAIR_STATE])) == PAIR_TAKEN;
}
// main wave, once its task queue is empty: the helper leaves its loop
MODLE_DEV void pair_dismiss(u32* m) {
  const u32 seq = wave::uniform(m[PAIR_REQ]) + 1;
  wave::lockstep();
  // (an atomic store: a main wave that gives up on a request -- the abort word -- dismisses a helper
  // that may still be reading that request's words)
  wave::st_release_wg(&m[PAIR_N_ACTIVE], PAIR_EXIT);
  wave::st_release_wg(&m[PAIR_REQ], seq);
}
// main wave of a launch that fills the slots, when it enters / leaves its task loop: an idle wave of
// the workgroup may claim it while the word says PAIR_OPEN; on leaving the word goes back to
// PAIR_IDLE, and a helper that had claimed it is dismissed
MODLE_DEV void pair_open(u32* m) { wave::st_release_wg(&m[PAIR_STATE], PAIR_OPEN); }
MODLE_DEV void pair_close(u32* m) {
  wave::lockstep();
  u32 leader = wave::lane();
  wave::launder(leader);
  u32 old = PAIR_IDLE;
  if (leader == 0) old = wave::exchange_wg(&m[PAIR_STATE], PAIR_IDLE);
  if (wave::bcast(old, 0) == PAIR_TAKEN) pair_dismiss(m);
}
// An idle wave (`self`: its index in the workgroup) looks for a main wave of its workgroup that is
// running without a helper and claims it: returns the main wave's index or -1.  `boxes`: the
// workgroup's hand-over words, PAIR_WORDS per wave.  `seen` = the main wave's request counter as it
// was BEFORE the claim: the main wave posts requests (or the dismissal) only once it has seen the
// claim, so nothing the helper must serve carries a number at or below `seen`.  Read after the
// compare-and-swap instead, a request posted in between is taken for an old one and both waves wait
// for ever (the race that cost a 15-minute hang in round 3: tests/protocol_model keeps it as a
// regression case, -DMODLE_MODEL_SEEN_AFTER_CLAIM).
MODLE_DEV int pair_claim(u32* boxes, int n_waves, int self, u32& seen) {
  for (int w = 0; w < n_waves; ++w) {
    u32* m = boxes + static_cast<u32>(w) * PAIR_WORDS;
    if (w == self || wave::uniform(wave::ld_acquire_wg(&m[PAIR_STATE])) != PAIR_OPEN) continue;
#ifndef MODLE_MODEL_SEEN_AFTER_CLAIM
    seen = wave::uniform(wave::ld_acquire_wg(&m[PAIR_REQ]));
#endif
    wave::lockstep();
    u32 leader = wave::lane();
    wave::launder(leader);
    u32 won = 0;
    if (leader == 0) won = wave::cas_wg(&m[PAIR_STATE], PAIR_OPEN, PAIR_TAKEN) ? 1u : 0u;
    if (wave::bcast(won, 0) != 0) {
#ifdef MODLE_MODEL_SEEN_AFTER_CLAIM
      wave::model_delay();  // (the window in which the main wave posts its first request)
      seen = wave::uniform(wave::ld_acquire_wg(&m[PAIR_REQ]));
#endif
      return w;
    }
  }
  return -1;
}

}  // namespace modle_dev
