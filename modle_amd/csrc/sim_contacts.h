// sim_contacts.h -- part of sim_device.h (included by it, in this order): sample_and_register_contacts.
#pragma once

namespace modle_dev {

// =============================================================================================
// Contact sampling (reference: src/libmodle/cpu/register_contacts.cpp)
// =============================================================================================
MODLE_DEV void matrix_increment(const Interval& iv, u64 row, u64 col) {
  // reference: contact_matrix_internal_impl.hpp:19-42, contact_matrix_dense_safe_impl.hpp:55-68
  u64 i, j;
  if (row > col) {
    i = row - col;
    j = row;
  } else {
    i = col - row;
    j = col;
  }
#ifdef MODLE_EXP_NO_OUTPUT_ATOMICS  // (measurement build, profiles/r05*/write_accounting.txt: the simulation does
  (void)i;                           // not read its outputs, so leaving the increments out changes nothing else)
  (void)j;
#else
  if (i >= iv.nrows) {
    wave::atomic_add_u64(iv.missed_updates, 1);
  } else {
    wave::atomic_inc_u32(iv.contacts + (j * iv.nrows + i));
  }
#endif
}

enum EventKind { EV_LOOP = 0, EV_TAD = 1, EV_OCC = 2 };

struct EventEval {
  u32 consumed;     // raws consumed by the event when no draw was rejected
  bool need_exact;  // a rejection happened: the consumption is not known without a replay
  bool ok;          // the event yields a registration
  u64 a, b;         // the two genomic coordinates to register
};

// lef_within_bound (reference: register_contacts.cpp:23-29); returns the unit positions
MODLE_DEV bool lef_samplable(const Cell& c, u32 i, u32& rev, u32& fwd) {
  const Workspace& ws = c.ws;
  const u32 lo = c.iv->start + 1, hi = c.iv->end - 1;
  if (ws.epoch[i] == UNBOUND) return false;
  rev = ws.r_pos[ws.r_rank[i]];
  fwd = ws.f_pos[ws.f_rank[i]];
  return rev > lo && rev < hi && fwd > lo && fwd < hi;
}

// randomize_extrusion_unit_positions / pos_within_bound (reference: register_contacts.cpp:31-63)
MODLE_DEV bool sample_lef_pair(const Cell& c, u32 rev, u32 fwd, f64 u1, f64 u2, bool noisify,
                               f64& p1, f64& p2) {
  const Params& p = *c.p;
  const f64 n1 = noisify ? genextreme_from_canonical(u1, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 a = static_cast<f64>(rev) - n1;
  const f64 n2 = noisify ? genextreme_from_canonical(u2, p.gev_mu, p.gev_sigma, p.gev_xi) : 0.0;
  const f64 b = static_cast<f64>(fwd) + n2;
  p1 = b < a ? b : a;
  p2 = b < a ? a : b;
  const f64 lo = static_cast<f64>(c.iv->start + 1), hi = static_cast<f64>(c.iv->end - 1);
  return p1 >= lo && p2 >= lo && p1 < hi && p2 < hi;
}

template <int KIND>
MODLE_DEV EventEval eval_event_fast(const Cell& c, u64 q, u64 lef_range, u64 lef_bucket,
                                    bool noisify) {
  EventEval e{1, false, false, 0, 0};
  const u64 r = rng_peek(c.g, q) / lef_bucket;
  if (r > lef_range) {
    e.need_exact = true;
    return e;
  }
  u32 rev = 0, fwd = 0;
  if (!lef_samplable(c, static_cast<u32>(r), rev, fwd)) return e;  // consumed = 1
  const u32 nz = noisify ? 2u : 0u;
  const f64 u1 = noisify ? canonical_raw(rng_peek(c.g, q + 1)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_peek(c.g, q + 2)) : 0.0;
  f64 p1, p2;
  const bool inb = sample_lef_pair(c, rev, fwd, u1, u2, noisify, p1, p2);
  e.consumed = 1 + nz;
  if (!inb) return e;
  const u64 a = static_cast<u64>(p1), b = static_cast<u64>(p2);
  if (KIND != EV_TAD) {
    e.ok = true;
    e.a = a;
    e.b = b;
    return e;
  }
  const u64 range = b - a;
  if (range == 0) {
    e.ok = true;
    e.a = a;
    e.b = a;
    return e;
  }
  const u64 bucket = uniform_int_bucket(range);
  const u64 ra = rng_peek(c.g, q + 1 + nz) / bucket;
  const u64 rb = rng_peek(c.g, q + 2 + nz) / bucket;
  if (ra > range || rb > range) {
    e.need_exact = true;
    return e;
  }
  e.consumed = 3 + nz;
  e.ok = true;
  e.a = a + ra;
  e.b = a + rb;
  return e;
}

// one sampling event replayed sequentially from g.pos; uniform
template <int KIND>
MODLE_DEV_NOINLINE EventEval eval_event_exact(Cell& c, u64 lef_range, u64 lef_bucket, bool noisify) {
  EventEval e{0, false, false, 0, 0};
  const u64 r = lef_range == 0 ? 0 : uniform_int_exact(c.g, lef_range, lef_bucket);
  u32 rev = 0, fwd = 0;
  if (!lef_samplable(c, static_cast<u32>(r), rev, fwd)) return e;
  const f64 u1 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  const f64 u2 = noisify ? canonical_raw(rng_next(c.g)) : 0.0;
  f64 p1, p2;
  if (!sample_lef_pair(c, rev, fwd, u1, u2, noisify, p1, p2)) return e;
  const u64 a = static_cast<u64>(p1), b = static_cast<u64>(p2);
  e.ok = true;
  if (KIND != EV_TAD) {
    e.a = a;
    e.b = b;
    return e;
  }
  const u64 range = b - a;
  if (range == 0) {
    e.a = a;
    e.b = a;
    return e;
  }
  const u64 bucket = uniform_int_bucket(range);
  e.a = a + uniform_int_exact(c.g, range, bucket);
  e.b = a + uniform_int_exact(c.g, range, bucket);
  return e;
}

template <int KIND>
MODLE_DEV void commit_event(const Cell& c, const EventEval& e) {
  const Interval& iv = *c.iv;
  const u64 lo = static_cast<u64>(iv.start) + 1;
  const u64 bin = c.p->bin_size;
  const u64 ba = (e.a - lo) / bin, bb = (e.b - lo) / bin;
  if (KIND == EV_OCC) {
#ifndef MODLE_EXP_NO_OUTPUT_ATOMICS
    if (iv.occupancy_1d != nullptr) {
      wave::atomic_add_u64(iv.occupancy_1d + ba, 1);
      wave::atomic_add_u64(iv.occupancy_1d + bb, 1);
    }
#else
    (void)ba;
    (void)bb;
#endif
  } else {
    matrix_increment(iv, ba, bb);
  }
}

// runs `n_events` sampling events of one kind; returns the number of registrations
template <int KIND>
MODLE_DEV_NOINLINE u64 run_events(Cell& c, u64 n_events) {
  if (n_events == 0) return 0;
  const u32 lane = wave::lane();
  const bool noisify = (c.p->sampling_strategy & CS_NOISIFY) != 0;
  const u64 lef_range = static_cast<u64>(c.n_active) - 1;
  const u64 lef_bucket = lef_range != 0 ? uniform_int_bucket(lef_range) : 1;
  const u32 stride = 1 + (noisify ? 2u : 0u) + (KIND == EV_TAD ? 2u : 0u);
  u64 registered = 0;
  u64 remaining = n_events;
  while (remaining != 0) {
    if (lef_range == 0) {
      // a single LEF: the index draw consumes nothing; keep it simple and replay sequentially
      const EventEval e = eval_event_exact<KIND>(c, lef_range, lef_bucket, noisify);
      if (e.ok && lane == 0) commit_event<KIND>(c, e);
      registered += e.ok ? 1 : 0;
      --remaining;
      continue;
    }
    // one step handles at most as many events as the PRNG ring can serve
    const u32 cntb = static_cast<u32>(umin64(umin(64u, RNG_BLOCK / stride), remaining));
    rng_ensure(c.g, cntb * stride);
    const bool act = lane < cntb;
    EventEval e{stride, false, false, 0, 0};
    if (act) e = eval_event_fast<KIND>(c, c.g.pos + static_cast<u64>(lane) * stride, lef_range,
                                       lef_bucket, noisify);
    const u64 irregular = wave::ballot(act && (e.need_exact || e.consumed != stride));
    if (irregular == 0) {
      if (act && e.ok) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(act && e.ok)));
      rng_advance(c.g, static_cast<u64>(cntb) * stride);
      remaining -= cntb;
    } else {
      const u32 f = static_cast<u32>(wave::ctz64(irregular));
      const bool commit = act && e.ok && lane < f;
      if (commit) commit_event<KIND>(c, e);
      registered += static_cast<u64>(wave::popc64(wave::ballot(commit)));
      rng_advance(c.g, static_cast<u64>(f) * stride);
      const bool needx = wave::bcast(e.need_exact, f);
      if (!needx) {
        if (lane == f && e.ok) commit_event<KIND>(c, e);
        registered += wave::bcast(e.ok, f) ? 1 : 0;
        rng_advance(c.g, wave::bcast(e.consumed, f));
      } else {
        const EventEval x = eval_event_exact<KIND>(c, lef_range, lef_bucket, noisify);
        if (x.ok && lane == 0) commit_event<KIND>(c, x);
        registered += x.ok ? 1 : 0;
      }
      remaining -= f + 1;
    }
  }
  return registered;
}

// sample_and_register_contacts (reference: register_contacts.cpp:93-120)
MODLE_DEV u64 phase_sample_contacts(Cell& c, u64 events_per_epoch, u64 num_target_contacts,
                                    u64 num_contacts, u64& events_done) {
  const Params& p = *c.p;
  u64 n_events = events_per_epoch;
  if (p.target_contact_density > 0.0)
    n_events = umin64(n_events, num_target_contacts - num_contacts);
  if (n_events == 0) return 0;
  ensure_inverse_both(c);  // events pick LEFs by id
  events_done += n_events;
  u64 n_loop;
  if (p.tad_to_loop_ratio == 0) {
    n_loop = n_events;
  } else if (!wave::f_isfinite(p.tad_to_loop_ratio)) {
    n_loop = 0;
  } else {
    n_loop = static_cast<u64>(
        binomial_exact(c.g, static_cast<i64>(n_events), 1.0 / (p.tad_to_loop_ratio + 1.0)));
  }
  u64 registered = run_events<EV_LOOP>(c, n_loop);
  registered += run_events<EV_TAD>(c, n_events - n_loop);
  if (p.track_1d) (void)run_events<EV_OCC>(c, n_events);
  return registered;
}

}  // namespace modle_dev
