// sim_helper.h -- part of sim_device.h (included by it, in this order): the helper wave's loop of
// the helper-wave mode (protocol and the main wave's side: sim_pair.h).
#pragma once

namespace modle_dev {

// The helper's loop.  `c` is a cell context that shares the main wave's generator, tables and
// workspace (moves, barrier states, lists, unit arrays) and has the helper's own staging and sort
// buffers; `intervals` is the launch's interval table.
MODLE_DEV void pair_serve(Cell& c, const Interval* intervals, u32* m) {
  u32 seen = wave::uniform(m[PAIR_REQ]);
  for (;;) {
    u32 seq;
    while ((seq = wave::uniform(wave::ld_acquire_wg(&m[PAIR_REQ]))) == seen) wave::nap();
    seen = seq;
    const u32 n_active = wave::uniform(m[PAIR_N_ACTIVE]);
    if (n_active == PAIR_EXIT) break;
    const Interval ivg = interval_in_device_memory(intervals[wave::uniform(m[PAIR_INTERVAL])]);
    c.iv = &ivg;
    c.n_active = n_active;
    if (wave::uniform(m[PAIR_KIND]) == PAIR_KIND_LEF_BAR) {
      BoundaryCounts bc;
      bc.n5 = wave::uniform(m[PAIR_BC]);
      bc.n3 = wave::uniform(m[PAIR_BC + 1]);
      c.n_hit[1] = wave::uniform(m[PAIR_N_HIT + 1]);
      c.ws.f_pos = wave::as_global(reinterpret_cast<u32*>(pair_get_u64(m, PAIR_F_POS)));
      c.ws.f_move = wave::as_global(reinterpret_cast<u32*>(pair_get_u64(m, PAIR_F_MOVE)));
      detect_lef_bar<true>(c, bc);  // (ends with sync_mem)
      wave::st_release_wg(&m[PAIR_ALL], seq);
      continue;
    }
    const bool burnin_completed = wave::uniform(m[PAIR_BURNIN_DONE]) != 0;
    c.g.pos = pair_get_u64(m, PAIR_POS);
    c.g.gen_end = pair_get_u64(m, PAIR_GEN_END);
    const Params& p = *c.p;
    generate_moves_by_id(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std, c.ws.tmp[8]);
    generate_moves_by_id(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std, c.ws.tmp[9]);
    wave::sync_mem();
    wave::st_release_wg(&m[PAIR_MOVES], seq);
    barriers_next_state(c);
    wave::lockstep();
    if (wave::lane() == 0) {
      pair_put_u64(m, PAIR_POS, c.g.pos);
      pair_put_u64(m, PAIR_GEN_END, c.g.gen_end);
      m[PAIR_N_HIT] = c.n_hit[0];
      m[PAIR_N_HIT + 1] = c.n_hit[1];
    }
    wave::st_release_wg(&m[PAIR_ALL], seq);
  }
}

}  // namespace modle_dev
