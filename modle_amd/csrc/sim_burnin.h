// sim_burnin.h -- part of sim_device.h (included by it, in this order): run_burnin: loop-size statistics and the stability test.
#pragma once

namespace modle_dev {

// =============================================================================================
// Burn-in (reference: simulation.cpp:795-894)
// =============================================================================================
struct LoopStats {
  f64 avg, std;  // stats::mean / stats::standard_dev of the loop sizes (population std)
};
MODLE_DEV LoopStats loop_size_stats(Cell& c) {
  // reference: simulation.cpp:795-819 and stats/descriptive_impl.hpp:22-31, 63-101.  The mean is
  // a sum of integers below 2^53 (order independent); the squared deviations are accumulated
  // strictly left to right in LEF-id order like std::accumulate.
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  // pass A, rank order (contiguous reads, four batches in flight): every unit drops its position
  // at its LEF's slot of two id-ordered scratch arrays; the sum of all loop sizes is the sum of
  // the fwd positions minus the sum of the rev positions (released LEFs have both units at
  // UNBOUND and cancel: loop size 0, like the reference)
  u32* by_id_fwd = ws.tmp[0];
  u32* by_id_rev = ws.tmp[1];
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  u64 part = 0;
  struct UnitRegs {
    u32 fP[UX], fI[UX], rP[UX], rI[UX];
  };
  const auto load_units = [&](auto op, u32 group, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 k = group + 64 * u + lane;
      const bool act = k < n;
      r.fP[u] = op(ws.f_pos, k, act, 0, r.fP[u]);
      r.fI[u] = op(ws.f_id, k, act, 0, r.fI[u]);
      r.rP[u] = op(ws.r_pos, k, act, 0, r.rP[u]);
      r.rI[u] = op(ws.r_id, k, act, 0, r.rI[u]);
    }
  };
  UnitRegs cur;
  load_units(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_units(wave::LdRaw{}, group + 64 * UX, cur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 k = group + 64 * u + lane;
      if (k < n) {
        by_id_fwd[g.fI[u]] = g.fP[u];
        by_id_rev[g.rI[u]] = g.rP[u];
#ifdef MODLE_EXP_DOUBLE_SCATTER  // (measurement: what the two scattered stores per LEF cost -- a second pair, same pattern, dead arrays)
        ws.tmp[8][g.fI[u]] = g.fP[u];
        ws.tmp[9][g.rI[u]] = g.rP[u];
#endif
        part += static_cast<u64>(g.fP[u]) - static_cast<u64>(g.rP[u]);
      }
    }
  }
  wave::sync_mem();
#pragma unroll
  for (u32 s = 1; s < 64; s <<= 1) {
    const u64 o = wave::shfl_down(part, s);
    if (lane + s < 64) part += o;
  }
  const u64 total = wave::bcast(part, 0);
  const f64 avg = static_cast<f64>(total) / static_cast<f64>(n);
  // pass B, LEF-id order: strictly sequential accumulation like std::accumulate: every lane
  // computes its term, the terms of a batch are folded in lane order through broadcasts
  f64 ssd = 0.0;
  f64* terms = reinterpret_cast<f64*>(c.lds.stage);  // 2 x 64 terms (the buffer is idle here)
  static_assert(STAGE_CAP * sizeof(u32) >= 128 * sizeof(f64), "stage buffer too small for the fold");
  struct SizeRegs {
    u32 lf[UX], lr[UX];
  };
  const auto load_sizes = [&](u32 group, SizeRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 i = group + 64 * u + lane;
      r.lf[u] = wave::LdRaw{}(by_id_fwd, i, i < n, 0, 0u);
      r.lr[u] = wave::LdRaw{}(by_id_rev, i, i < n, 0, 0u);
    }
  };
  SizeRegs scur;
  load_sizes(0, scur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    const SizeRegs sg = scur;  // (the next group's loads are in flight during the fold)
    if (group + 64 * UX < n) load_sizes(group + 64 * UX, scur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 base = group + 64 * u;
      if (base >= n) break;
      const u32 i = base + lane;
      f64 term = 0.0;
      if (i < n) {
        const u32 ls = sg.lf[u] - sg.lr[u];
        const f64 d = static_cast<f64>(static_cast<u64>(ls)) - avg;
        term = d * d;
      }
      // lanes past the end hold +0.0, which leaves the (non-negative) running sum unchanged, so
      // all 64 terms are folded with constant indices (no loop control in the chain).  The terms
      // go through LDS: every lane reads them back in order (one address for the whole wave: a
      // broadcast) and keeps its own copy of the running sum.  Two lane broadcasts per term plus
      // the wait states between a broadcast and the addition that uses it had been two thirds of
      // the chain.
      wave::lockstep();
      terms[64 * (u & 1u) + lane] = term;
      wave::sync_lds();
#pragma unroll
      for (u32 l = 0; l < 64; ++l) ssd = ssd + terms[64 * (u & 1u) + l];
    }
  }
  return LoopStats{avg, wave::f_sqrt(ssd / static_cast<f64>(n))};
}

MODLE_DEV_NOINLINE void compute_loop_size_stats(Cell& c) {
  Workspace& ws = c.ws;
  const u32 lane = wave::lane();
  const u32 cap = c.p->hist_len;
  const LoopStats st = loop_size_stats(c);
  const f64 avg = st.avg, std = st.std;
  // push_back with pop_front at capacity (two deque<double>)
  f64* cfx = ws.hist;
  f64* avgb = ws.hist + cap;
  u32 slot;
  if (c.hist_len == cap) {
    slot = c.hist_head;
    c.hist_head = (c.hist_head + 1) % cap;
  } else {
    slot = (c.hist_head + c.hist_len) % cap;
    ++c.hist_len;
  }
  wave::lockstep();
  if (lane == 0) {
    avgb[slot] = avg;
    cfx[slot] = std / avg;
  }
  wave::sync_mem();
}

MODLE_DEV bool series_is_stable(const Cell& c, const f64* buf) {
  const u32 cap = c.p->hist_len, w = c.p->window;
  const u32 lane = wave::lane();
  const u32 ncmp = cap - w - 1;  // comparisons of consecutive window means
  u32 n_dips = 0;
  for (u32 base = 0; base < ncmp; base += 64) {
    const u32 j = base + lane;
    bool dip = false;
    if (j < ncmp) {
      f64 s1 = 0.0, s2 = 0.0;
      for (u32 t = 0; t < w; ++t) s1 = s1 + buf[(c.hist_head + j + t) % cap];
      for (u32 t = 0; t < w; ++t) s2 = s2 + buf[(c.hist_head + j + 1 + t) % cap];
      dip = (s1 / static_cast<f64>(w)) > (s2 / static_cast<f64>(w));
    }
    n_dips += static_cast<u32>(wave::popc64(wave::ballot(dip)));
  }
  const f64 r = static_cast<f64>(n_dips) / static_cast<f64>(cap - w - n_dips);
  return r >= 0.95 && r <= 1.05;
}

MODLE_DEV_NOINLINE bool evaluate_burnin(const Cell& c) {
  // reference: simulation.cpp:821-864
  const u32 cap = c.p->hist_len;
  if (c.hist_len != cap) return false;
  if (!series_is_stable(c, c.ws.hist)) return false;
  return series_is_stable(c, c.ws.hist + cap);
}

}  // namespace modle_dev
