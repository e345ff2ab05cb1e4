// sim_burnin.h -- part of sim_device.h (included by it, in this order): run_burnin: loop-size statistics and the stability test.
#pragma once

namespace modle_dev {

// =============================================================================================
// Burn-in (reference: simulation.cpp:795-894)
// =============================================================================================
struct LoopStats {
  f64 avg, std;  // stats::mean / stats::standard_dev of the loop sizes (population std)
};
MODLE_DEV LoopStats loop_size_stats_scattered(Cell& c) {
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

// Folds 64 consecutive terms (lane l holds term l; lanes past the end hold +0.0, which leaves the
// non-negative running sum unchanged) into `ssd`, strictly in lane order like std::accumulate.  The
// terms go through LDS: every lane reads them back in order (one address for the whole wave: a
// broadcast) and keeps its own copy of the running sum -- two lane broadcasts per term plus the wait
// states between a broadcast and the addition that uses it had been two thirds of the chain.
// `buf`: 64 doubles of the staging buffer (LDS operations of one wave execute in order: the reads of
// one call are performed before the writes of the next).
MODLE_DEV f64 fold_terms_in_lane_order(f64 ssd, f64 term, f64* buf) {
  wave::lockstep();
  buf[wave::lane()] = term;
  wave::sync_lds();
  // (two terms per LDS read: the chain is 64 dependent additions either way, the reads are half the
  // instructions between them)
#pragma unroll
  for (u32 l = 0; l < 64; l += 2) {
    const wave::F64x2 t = wave::lds_ld2_f64(buf + l);
    ssd = ssd + t.v[0];
    ssd = ssd + t.v[1];
  }
  return ssd;
}

// 64-lane inclusive prefix sum of doubles on the lane moves of the integer scans.  Exact -- and so
// independent of the order of the additions -- wherever fold_terms_exact uses it: multiples of one
// unit in the last place whose sum stays below 2^53 units.
MODLE_DEV f64 wave_prefix_sum_f64(f64 v) {
#define MODLE_STEP(S)                                                                            \
  {                                                                                              \
    const u64 b = __builtin_bit_cast(u64, v);                                                    \
    const u32 lo = wave::scan_move<S>(static_cast<u32>(b), 0u);                                  \
    const u32 hi = wave::scan_move<S>(static_cast<u32>(b >> 32), 0u);                            \
    v = v + __builtin_bit_cast(f64, (static_cast<u64>(hi) << 32) | lo);                          \
  }
  MODLE_SCAN_STEPS(MODLE_STEP)
#undef MODLE_STEP
  return v;
}

// The same fold -- 64 terms in lane order, every addition rounded like std::accumulate rounds it --
// without the chain of 64 dependent additions (14 cycles each: 3.4 % of the default launch).  While the
// running sum s stays inside one binade [B, 2B), u = ulp(s):
//     fl(s + t) = s + q(t),   q(t) = t rounded to the nearest multiple of u = (B + t) - B in fp64,
// unless the rounding is a tie (the parity of s / u decides) or the sum reaches 2B; and multiples of u
// whose sum stays below 2B add up exactly in any order.  So one prefix sum folds the lanes up to the
// first one that ties or crosses; that lane takes a true addition, and the lanes behind it start
// again in what may be the next binade.  A batch needs 0.24 such additions on average (ties: 64 x
// 2^-(bits of t below u), about one batch in twenty; ~12 binades per fold); a batch that needs more than
// three, and a running sum of zero, go through the chain.  The scalar rendition of exactly this is
// checked against the sequential sum in tests/fold_model (ties, crossings, zero / huge terms).
MODLE_DEV f64 fold_terms_exact(f64 ssd, f64 term, f64* buf) {
  const u32 lane = wave::lane();
  f64 s = ssd;  // (every lane holds the same value)
  u32 start = 0;
  for (u32 it = 0; it < 3; ++it) {
    const u32 ef = static_cast<u32>(wave::uniform(__builtin_bit_cast(u64, s)) >> 52) & 0x7FFu;
    if (ef <= 53u || ef == 0x7FFu) break;  // zero, tiny, not finite: the chain
    const f64 B = __builtin_bit_cast(f64, static_cast<u64>(ef) << 52);
    const f64 top = B + B;
    const f64 u_half = __builtin_bit_cast(f64, static_cast<u64>(ef - 53u) << 52);
    const bool in = lane >= start;
    const f64 t = in ? term : 0.0;
    const f64 q = (B + t) - B;
    const f64 P = wave_prefix_sum_f64(q);
    const bool bad = in && (wave::f_abs(t - q) == u_half || !(s + P < top));
    const u64 bm = wave::ballot(bad);
    if (bm == 0) return s + wave::bcast(P, 63);
    const u32 fb = static_cast<u32>(wave::ctz64(bm));
    if (fb > start) s = s + wave::bcast(P, fb - 1);
    s = s + wave::bcast(term, fb);
    start = fb + 1;
    if (start == 64) return s;
  }
  // (adding +0.0 leaves the non-negative running sum as it is)
  return fold_terms_in_lane_order(s, lane >= start ? term : 0.0, buf);
}

// The same statistics without scattering 4-byte stores over device memory (round 4).  The two
// id-ordered scatters of loop_size_stats_scattered -- one store per unit into a random line -- were
// the most expensive memory traffic of the kernel: a second pair of them cost 8.3 % of the launch
// (profiles/r04c).  Here the id order is restored in two steps:
//   A. one sweep over the units in rank order (coalesced 128-bit loads of positions and ids)
//      PARTITIONS them by windows of STATS_WINDOW consecutive LEF ids: a unit appends
//      (position, id within the window) to its window's region of a pair array -- every id occurs
//      once per direction, so window b's region is exactly entries [b W, b W + W); the next free
//      entry of every region is a counter in LDS (one returning LDS add per unit).  A wave store
//      then writes a few runs of consecutive 8-byte entries instead of 64 unrelated lines;
//   B. per window the pairs are read back (coalesced) and dropped at their id's slot of two arrays
//      in LDS (the sort buffer: W rev positions, W fwd positions), and the window is folded in id
//      order.
// The mean needs all positions first: step A sums them.  Step B needs the LDS sort buffer: the
// epoch loop computes the statistics behind the bind phase, when the list of released LEFs has
// been consumed (a LEF bound in between has both units at one position: loop size 0, exactly what
// it counted for while it was unbound).  Pair arrays: ws.sort_keys (rev) and ws.tmp[8..9] (fwd: two
// arrays that lie next to each other and hold the dead moves of the previous epoch).
constexpr u32 STATS_WINDOW = SORT_LDS_CAP;  // ids per window: two 32-bit slots each in the sort buffer
constexpr u32 STATS_MAX_WINDOWS = STAGE_CAP / 2;  // region counters (rev, fwd) in the staging buffer
constexpr u32 STATS_WINDOW_LOG2 = STATS_WINDOW == 512 ? 9 : STATS_WINDOW == 256 ? 8 : 0;
static_assert((u32(1) << STATS_WINDOW_LOG2) == STATS_WINDOW, "the window must be a power of two");
MODLE_DEV LoopStats loop_size_stats_partitioned(Cell& c, u64* pairs_r, u64* pairs_f) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u32 nblk = (n + 255) / 256;
  const u32 nwin = (n + STATS_WINDOW - 1) / STATS_WINDOW;
  u32* cursor = c.lds.stage;  // [0, nwin): rev regions, [nwin, 2 nwin): fwd regions
  wave::lockstep();
  for (u32 k = lane; k < 2 * nwin; k += 64) cursor[k] = 0;
  wave::sync_lds();
  // A. partition
  struct Blk {
    wave::U32x4 rP, rI, fP, fI;
  };
  const auto load_blk = [&](u32 t, Blk& r) {
    const u32 w = 256 * t + 4 * lane;
    const u32 wq = w < n ? w : 0u;
    r.rP = wave::ld4(ws.r_pos, wq);
    r.rI = wave::ld4(ws.r_id, wq);
    r.fP = wave::ld4(ws.f_pos, wq);
    r.fI = wave::ld4(ws.f_id, wq);
  };
  u64 part = 0;
#ifdef MODLE_SUBTIMER_STATS
  const u64 t_part = wave::clock();
#endif
  Blk cur;
  load_blk(0, cur);
  for (u32 t = 0; t < nblk; ++t) {
    const Blk g = cur;
    if (t + 1 < nblk) load_blk(t + 1, cur);
    const u32 w = 256 * t + 4 * lane;
    u32 er[4], ef[4];  // entry of the unit in its window's region
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      er[q] = 0;
      ef[q] = 0;
      if (w + q < n) {
        er[q] = wave::lds_fetch_add_u32(&cursor[g.rI.v[q] >> STATS_WINDOW_LOG2], 1u);
        ef[q] = wave::lds_fetch_add_u32(&cursor[nwin + (g.fI.v[q] >> STATS_WINDOW_LOG2)], 1u);
      }
    }
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      if (w + q < n) {
        const u32 ri = g.rI.v[q], fi = g.fI.v[q];
        pairs_r[(ri & ~(STATS_WINDOW - 1)) + er[q]] = (static_cast<u64>(ri & (STATS_WINDOW - 1)) << 32) | g.rP.v[q];
        pairs_f[(fi & ~(STATS_WINDOW - 1)) + ef[q]] = (static_cast<u64>(fi & (STATS_WINDOW - 1)) << 32) | g.fP.v[q];
        part += static_cast<u64>(g.fP.v[q]) - static_cast<u64>(g.rP.v[q]);
      }
    }
  }
  wave::sync_mem();
#pragma unroll
  for (u32 s = 1; s < 64; s <<= 1) {
    const u64 o = wave::shfl_down(part, s);
    if (lane + s < 64) part += o;
  }
  const f64 avg = static_cast<f64>(wave::bcast(part, 0)) / static_cast<f64>(n);
#ifdef MODLE_SUBTIMER_STATS
  c.ph[14] += wave::clock() - t_part;  // (the partition sweep)
#endif
  // B. per window: pairs -> LDS slots by id, then the fold in id order
  u32* slot_r = reinterpret_cast<u32*>(c.lds.sort_lds);
  u32* slot_f = slot_r + STATS_WINDOW;
  f64* terms = reinterpret_cast<f64*>(c.lds.stage);
  f64 ssd = 0.0;
  constexpr u32 UX = STATS_WINDOW / 64;  // loads of one window, all in flight
  struct PairRegs {
    u64 r[UX], f[UX];
  };
  const auto load_pairs = [&](u32 b, PairRegs& p) {
    const u32 lo = b * STATS_WINDOW;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 e = lo + 64 * u + lane;
      p.r[u] = wave::LdRaw{}(pairs_r, e, e < n, u64(0), u64(0));
      p.f[u] = wave::LdRaw{}(pairs_f, e, e < n, u64(0), u64(0));
    }
  };
  PairRegs pcur;
  load_pairs(0, pcur);
  for (u32 b = 0; b < nwin; ++b) {
    const PairRegs pg = pcur;
    if (b + 1 < nwin) load_pairs(b + 1, pcur);  // (in flight during this window's fold)
    const u32 lo = b * STATS_WINDOW;
    const u32 cnt = umin(STATS_WINDOW, n - lo);
    wave::lockstep();  // (the fold of the previous window has read its slots)
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      if (64 * u + lane < cnt) {
        slot_r[static_cast<u32>(pg.r[u] >> 32)] = static_cast<u32>(pg.r[u]);
        slot_f[static_cast<u32>(pg.f[u] >> 32)] = static_cast<u32>(pg.f[u]);
      }
    }
    wave::sync_lds();
    // the positions by LEF id, for the passes of this epoch that need a LEF's other unit
    // (fix_secondary): two coalesced stores per LEF
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 i = 64 * u + lane;
      if (i < cnt) {
        ws.by_id_pos[0][lo + i] = slot_r[i];
        ws.by_id_pos[1][lo + i] = slot_f[i];
      }
    }
#ifdef MODLE_SUBTIMER_STATS
    const u64 t_fold = wave::clock();
#endif
    for (u32 base = 0; base < cnt; base += 64) {
      const u32 i = base + lane;
      f64 term = 0.0;
      if (i < cnt) {
        const u32 ls = slot_f[i] - slot_r[i];
        const f64 d = static_cast<f64>(static_cast<u64>(ls)) - avg;
        term = d * d;
      }
      ssd = fold_terms_exact(ssd, term, terms);
    }
#ifdef MODLE_SUBTIMER_STATS
    c.ph[15] += wave::clock() - t_fold;  // (the fold alone)
#endif
  }
  wave::sync_mem();
  c.by_id_valid = true;
  return LoopStats{avg, wave::f_sqrt(ssd / static_cast<f64>(n))};
}

MODLE_DEV LoopStats loop_size_stats(Cell& c) {
  const u32 n = wave::uniform(c.n_active);
  // the fwd pairs live in two scratch arrays that must lie next to each other (the workspace is
  // carved that way: sim_types.h / modle_hip.hip device_carve; only tmp[0], tmp[1] and tmp[7] ever
  // trade places with other arrays)
  const i64 stride = (static_cast<i64>(c.ws.capacity_lefs) + 63) & ~i64(63);
  const bool adjacent = wave::uniform(static_cast<i64>(c.ws.tmp[9] - c.ws.tmp[8])) == stride;
#ifdef MODLE_EMU_TRACE_RANK  // (emulator only: which form a test exercises)
  if (wave::lane() == 0) fprintf(stderr, "loop_size_stats: n %u, %s\n", n, (n <= STATS_MAX_WINDOWS * STATS_WINDOW && adjacent) ? "partitioned" : "scattered");
#endif
  if (n <= STATS_MAX_WINDOWS * STATS_WINDOW && adjacent)
    return loop_size_stats_partitioned(c, c.ws.sort_keys, reinterpret_cast<u64*>(c.ws.tmp[8]));
  return loop_size_stats_scattered(c);
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
