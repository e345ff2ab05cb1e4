// sim_barriers.h -- part of sim_device.h (included by it, in this order): ExtrusionBarriers::init_states / next_state and the per-epoch lists of stalling barriers.
#pragma once

namespace modle_dev {

// =============================================================================================
// ExtrusionBarriers::init_states / next_state (reference: extrusion_barriers.cpp:145-161,
// 219-230)
// =============================================================================================
MODLE_DEV_NOINLINE void barriers_init_states(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const bool act = i < nb;
    const f64 occ = wave::ld_sel(iv.bar_occupancy, i, act, 0.0);
    const bool draws = act && occ != 0.0;  // bernoulli(0) consumes nothing
    const u64 dm = wave::ballot(draws);
    const u32 cnt = static_cast<u32>(wave::popc64(dm));
    rng_ensure(c.g, cnt);
    const u32 k = static_cast<u32>(wave::popc64(dm & lanemask_lt(lane)));
    const bool on = draws && bernoulli_raw(rng_peek(c.g, c.g.pos + k), occ);
    if (act) c.ws.bar_active[i] = on ? 1 : 0;
    rng_advance(c.g, cnt);
  }
  wave::sync_mem();
}

// LEF-BAR detection works on the barriers that CAN stall a unit, compacted in position order: list 0
// as the rev units see them, list 1 as the fwd units do.  A barrier is on a list iff it is active
// and the blocking probability that applies to it there is not zero; an entry whose probability is
// below one costs a Bernoulli trial when a unit reaches it (lef_bar_trials_needed: round 4 -- the
// lists used to exist only when both probabilities were 0 or 1, and every other configuration,
// BASELINE configs[4] among them, searched the complete barrier set per unit).
MODLE_DEV bool lef_bar_trials_needed(const Params& p) {
  return !((p.pblock_major == 1.0 || p.pblock_major == 0.0) && (p.pblock_minor == 1.0 || p.pblock_minor == 0.0));
}
constexpr u32 HITBAR_HARD = 0x80000000u;  // the barrier blocks the direction of the list's units (the major probability applies)

// appends the barriers of one batch (index i per lane, `on`: active) to the two lists; uniform
// (major_in / minor_in: the blocking probability of that kind is not zero -- worked out once by the
// caller: a Params field read inside the loop comes back as a sixteen-register reload per batch)
MODLE_DEV void stalling_lists_append(Cell& c, u32 i, bool in, bool on, u32 bpos, u32 bdir,
                                     bool major_in, bool minor_in) {
  const u32 lane = wave::lane();
#pragma unroll
  for (u32 d = 0; d < 2; ++d) {
    const bool is_major = bdir == (d == 0 ? DIR_REV : DIR_FWD);
    const bool hit = in && on && (is_major ? major_in : minor_in);
    const u64 hm = wave::ballot(hit);
    if (hit) {
      const u32 slot = c.n_hit[d] + static_cast<u32>(wave::popc64(hm & lanemask_lt(lane)));
      c.ws.hit_pos[d][slot] = bpos;
      c.ws.hit_idx[d][slot] = i | (is_major ? HITBAR_HARD : 0u);
    }
    c.n_hit[d] += static_cast<u32>(wave::popc64(hm));
  }
}

// stand-alone construction of the lists from the current barrier states (phase-level test entry
// point; the epoch loop builds them while it updates the states)
MODLE_DEV_NOINLINE void compact_stalling_barriers(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  const bool major_in = wave::uniform(c.p->pblock_major != 0.0), minor_in = wave::uniform(c.p->pblock_minor != 0.0);
  for (u32 base = 0; base < nb; base += 64) {
    const u32 i = base + lane;
    const bool in = i < nb;
    stalling_lists_append(c, i, in, in && c.ws.bar_active[i] != 0, in ? iv.bar_pos[i] : 0,
                          in ? iv.bar_dir[i] : 0, major_in, minor_in);
  }
  wave::sync_mem();
}

MODLE_DEV_NOINLINE void barriers_next_state(Cell& c) {
  const Interval& iv = *c.iv;
  const u32 nb = wave::uniform(iv.n_barriers);
  const u32 lane = wave::lane();
  constexpr bool lists = true;
  const bool major_in = wave::uniform(c.p->pblock_major != 0.0), minor_in = wave::uniform(c.p->pblock_minor != 0.0);
  c.n_hit[0] = 0;
  c.n_hit[1] = 0;
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct BarRegs {
    u8 S[UX], D[UX];
    u32 P[UX];
    f64 I[UX], A[UX];
  };
  const auto load_bars = [&](auto op, u32 group, BarRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 iq = group + 64 * u + lane;
      r.S[u] = op(c.ws.bar_active, iq, iq < nb, u8(0), r.S[u]);
      r.I[u] = op(iv.bar_stp_inactive, iq, iq < nb, 0.0, r.I[u]);
      r.A[u] = op(iv.bar_stp_active, iq, iq < nb, 0.0, r.A[u]);
      r.D[u] = op(iv.bar_dir, iq, lists && iq < nb, u8(0), r.D[u]);
      r.P[u] = op(iv.bar_pos, iq, lists && iq < nb, 0, r.P[u]);
    }
  };
  BarRegs cur{};
  if (nb != 0) load_bars(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < nb; group += 64 * UX) {
    BarRegs g = cur;
    load_bars(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < nb) load_bars(wave::LdRaw{}, group + 64 * UX, cur);
    const u8* Sq = g.S;
    const f64* Iq = g.I;
    const f64* Aq = g.A;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 base = group + 64 * u;
      if (base >= nb) break;
      const u32 i = base + lane;
      const u32 cnt = umin(64u, nb - base);
      rng_ensure(c.g, cnt);
      u8 st = Sq[u];
      if (i < nb) {
        const f64 r = canonical_raw(rng_peek(c.g, c.g.pos + lane));
        if (!st && r > Iq[u]) {
          st = 1;
          c.ws.bar_active[i] = 1;
        } else if (st && r > Aq[u]) {
          st = 0;
          c.ws.bar_active[i] = 0;
        }
      }
      rng_advance(c.g, cnt);
      stalling_lists_append(c, i, i < nb, st != 0, g.P[u], g.D[u], major_in, minor_in);
    }
  }
  wave::sync_mem();
}

}  // namespace modle_dev
