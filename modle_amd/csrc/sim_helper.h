// sim_helper.h -- part of sim_device.h (included by it, in this order): the helper wave's loop of
// the helper-wave mode (protocol and the main wave's side: sim_pair.h).
#pragma once

namespace modle_dev {

// The helper's loop.  `c` is a cell context that shares the main wave's generator, tables and
// workspace (moves, barrier states, lists, unit arrays) and has the helper's own staging and sort
// buffers; `intervals` is the launch's interval table.
// `feed`: hand-over words of the wave that produces the PRNG blocks while the helper draws the moves
// (pair_feed below), or nullptr when the workgroup has no wave to spare for it.
// `seen`: the request counter as it was when this wave became the helper -- zero at the start of the
// kernel, read BEFORE the claim for a helper that attaches later (read here, a request or the
// dismissal that the main wave posted in between would be taken for an old one: the helper would
// wait for ever).
// Every wait reads the host's abort word (c.lds.abort_flag) now and then and gives up when it is
// raised (spin_nap_aborted): the loop then ends like a dismissal.  `test_fault` (MODLE_HIP_TEST_FAULT,
// tests only): TEST_FAULT_STUCK_HELPER makes the helper withhold the signals of its third request,
// which is what a lost hand-over looks like to the main wave.
constexpr u32 TEST_FAULT_STUCK_HELPER = 1;
MODLE_DEV void pair_serve(Cell& c, const Interval* intervals, u32* m, u32* feed, u32 seen, u32 test_fault) {
  u32 fseq = 0;
  u32 served = 0;
  const u32* abort_flag = c.lds.abort_flag;
  const auto dismiss_producer = [&]() {
    if (feed != nullptr) {
      wave::lockstep();
      if (wave::lane() == 0) feed[FEED_EXIT] = 1;
      wave::st_release_wg(&feed[FEED_START], fseq + 1);
    }
  };
  for (;;) {
    u32 seq;
    u32 spins = 0;
    bool aborted = false;
    while ((seq = wave::uniform(wave::ld_acquire_wg(&m[PAIR_REQ]))) == seen) {
      if (spin_nap_aborted(abort_flag, spins)) {
        aborted = true;
        break;
      }
    }
    if (aborted) {
      dismiss_producer();
      break;
    }
    seen = seq;
    const u32 n_active = wave::uniform(wave::ld_acquire_wg(&m[PAIR_N_ACTIVE]));  // (see pair_dismiss)
    if (n_active == PAIR_EXIT) {
      dismiss_producer();
      break;
    }
    ++served;
    const bool withhold = test_fault == TEST_FAULT_STUCK_HELPER && served == 3;
    c.error = 0;
    const Interval ivg = interval_in_device_memory(intervals[wave::uniform(m[PAIR_INTERVAL])]);
    c.iv = &ivg;
    c.n_active = n_active;
    if (wave::uniform(m[PAIR_KIND]) == PAIR_KIND_LEF_BAR) {
      BoundaryCounts bc;
      bc.n5 = wave::uniform(m[PAIR_BC]);
      bc.n3 = wave::uniform(m[PAIR_BC + 1]);
      c.n_hit[1] = wave::uniform(m[PAIR_N_HIT + 1]);
      c.ws.f_pos = wave::as_global(reinterpret_cast<u32*>(pair_get_u64(m, PAIR_F_POS)));
      c.ws.f_move = wave::as_global(reinterpret_cast<move_t*>(pair_get_u64(m, PAIR_F_MOVE)));
      detect_lef_bar<true>(c, bc);  // (ends with sync_mem)
      wave::lockstep();
      if (wave::lane() == 0) m[PAIR_ERR] = c.error;
      if (!withhold) wave::st_release_wg(&m[PAIR_ALL], seq);
      continue;
    }
    if (wave::uniform(m[PAIR_KIND]) == PAIR_KIND_SEC_FILTER) {
      BoundaryCounts bc;
      bc.n5 = wave::uniform(m[PAIR_BC]);
      bc.n3 = wave::uniform(m[PAIR_BC + 1]);
      c.ws.f_pos = wave::as_global(reinterpret_cast<u32*>(pair_get_u64(m, PAIR_F_POS)));
      c.ws.f_move = wave::as_global(reinterpret_cast<move_t*>(pair_get_u64(m, PAIR_F_MOVE)));
      c.ws.tmp[1] = wave::as_global(reinterpret_cast<u32*>(pair_get_u64(m, PAIR_Q)));
      SecondaryFilter<true> ff;
      ff.init(c, bc, wave::uniform(m[PAIR_LIST_CAP]), true, true);
      for (u32 t = 0; t < ff.nblk; ++t) ff.step(t);
      wave::sync_mem();
      wave::lockstep();
      if (wave::lane() == 0) {
        m[PAIR_N_HIT] = ff.n_cand;
        m[PAIR_ERR] = c.error;
      }
      if (!withhold) wave::st_release_wg(&m[PAIR_ALL], seq);
      continue;
    }
    const bool burnin_completed = wave::uniform(m[PAIR_BURNIN_DONE]) != 0;
    c.g.pos = pair_get_u64(m, PAIR_POS);
    c.g.gen_end = pair_get_u64(m, PAIR_GEN_END);
    const Params& p = *c.p;
    if (feed != nullptr) {
      // the blocks of the stream come from the producer wave while the moves are drawn
      wave::lockstep();
      if (wave::lane() == 0) {
        feed[FEED_POS] = static_cast<u32>(c.g.pos);
        feed[FEED_GEN_END] = static_cast<u32>(c.g.gen_end);
      }
      ++fseq;
      wave::st_release_wg(&feed[FEED_START], fseq);
      c.g.feed = feed;
      c.g.feed_abort = abort_flag;
      c.g.feed_error = 0;
    }
    generate_moves_by_id(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std, as_moves(c.ws.tmp[8]));
    generate_moves_by_id(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std, as_moves(c.ws.tmp[9]));
    if (feed != nullptr) {
      // the producer stops (it may be a block ahead: the ring then ends where it says)
      wave::st_release_wg(&feed[FEED_STOP], fseq);
      u32 ack_spins = 0;
      while (wave::uniform(wave::ld_acquire_wg(&feed[FEED_ACK])) != fseq) {
        if (spin_nap_aborted(abort_flag, ack_spins)) {
          c.g.feed_error = ERR_CANCELLED;
          break;
        }
      }
      if (c.g.feed_error == 0) {
        const u32 ahead = wave::uniform(feed[FEED_GEN_END]) - static_cast<u32>(c.g.gen_end);
        c.g.gen_end = wave::known_uniform(c.g.gen_end + ahead);
      }
      c.g.feed = nullptr;
      // (a wait for the producer that was abandoned: the moves are not the stream's; the main wave
      // learns it with the signal and leaves the epoch)
      if (c.g.feed_error != 0) c.error = c.g.feed_error;
    }
    wave::sync_mem();
    // (PAIR_MOVES carries no status: the error word is written once per request, before PAIR_ALL,
    // which the main wave waits for before it uses anything but the move adjustment's arithmetic)
    if (!withhold) wave::st_release_wg(&m[PAIR_MOVES], seq);
    if (c.error != 0) {
      // nothing more to do for this request: the generator does not go back (the main wave fails the cell)
      wave::lockstep();
      if (wave::lane() == 0) m[PAIR_ERR] = c.error;
      if (!withhold) wave::st_release_wg(&m[PAIR_ALL], seq);
      continue;
    }
    barriers_next_state(c);
    wave::lockstep();
    if (wave::lane() == 0) {
      pair_put_u64(m, PAIR_POS, c.g.pos);
      pair_put_u64(m, PAIR_GEN_END, c.g.gen_end);
      m[PAIR_N_HIT] = c.n_hit[0];
      m[PAIR_N_HIT + 1] = c.n_hit[1];
      m[PAIR_ERR] = c.error;
    }
    if (!withhold) wave::st_release_wg(&m[PAIR_ALL], seq);
  }
}

#ifndef MODLE_RNG_PHILOX
// The producer's loop: blocks of the main wave's stream (ring, jump table, lane states, snapshots:
// LDS of the main wave) for the helper that draws the moves, one block ahead of it at most.
// `abort_flag`: the host's abort word; the producer leaves for good when it is raised while it waits.
MODLE_DEV void pair_feed(u64* ring, const u64* jump, u64* state, u64* snap, u32* f, const u32* abort_flag) {
  u32 seen = 0;
  for (;;) {
    u32 seq;
    u32 spins = 0;
    while ((seq = wave::uniform(wave::ld_acquire_wg(&f[FEED_START]))) == seen) {
      if (spin_nap_aborted(abort_flag, spins)) return;
    }
    seen = seq;
    if (wave::uniform(f[FEED_EXIT]) != 0) break;
    u32 gen_end = wave::uniform(f[FEED_GEN_END]);
    for (;;) {
      if (wave::uniform(wave::ld_acquire_wg(&f[FEED_STOP])) == seq) break;
      const u32 pos = wave::uniform(wave::ld_acquire_wg(&f[FEED_POS]));
      if (static_cast<i32>(gen_end - pos) <= static_cast<i32>(RNG_BLOCK - FEED_MARGIN)) {
        wave::lockstep();
        rng_gen_block_call((MODLE_LDS u64*)ring, (const MODLE_LDS u64*)jump, (MODLE_LDS u64*)state,
                           (MODLE_LDS u64*)snap, ((gen_end / RNG_BLOCK) & 1u) * RNG_BLOCK);
        gen_end += RNG_BLOCK;
        wave::sync_lds();
        wave::st_release_wg(&f[FEED_GEN_END], gen_end);
      } else if (spin_nap_aborted(abort_flag, spins)) {
        return;
      }
    }
    wave::st_release_wg(&f[FEED_ACK], seq);
  }
}
#endif

}  // namespace modle_dev
