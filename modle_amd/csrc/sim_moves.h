// sim_moves.h -- part of sim_device.h (included by it, in this order): generate_moves, adjust_moves_of_consecutive_extr_units, clamp_moves.
#pragma once

namespace modle_dev {

// =============================================================================================
// generate_moves (reference: simulation.cpp:272-330).  Draws are made in LEF-id order (the
// reference's stream order) and scattered to the unit's slot in rank order.
// =============================================================================================
MODLE_DEV u32 move_from_normal(f64 unit, f64 speed, f64 std) {
  const f64 v = unit * std + speed;
  return static_cast<u32>(static_cast<u64>(wave::f_round(v > 0.0 ? v : 0.0)));
}

// Queue of drawn moves (LDS, c.lds.stage): entry e lives at slot e % MOVQ_CAP, its move in the
// first half of the buffer and the low word of the stream position right after its draw in the
// second half.
constexpr u32 MOVQ_CAP = STAGE_CAP / 2;

// One step of the draw stream of generate_moves: lane l evaluates the normal-distribution attempt
// that would start at stream position pos + l (Boost's ziggurat, sim_rng.h: unit_normal_exact).
// An attempt takes one raw output (the strip's rectangle, ~98.8 %) or two (wedge test: accepted
// or rejected); which positions really start an attempt follows from the chain "an attempt that
// takes two outputs hides the position after it".  Accepted attempts are appended to the queue in
// stream order; the rare attempts whose length is data dependent beyond that (tail of the
// distribution, a uniform_01 retry) are replayed by the sequential routine.  Returns the new
// queue tail; uniform.
MODLE_DEV u32 draw_moves_step(Cell& c, f64 speed, f64 std, u32 tail) {
  const u32 lane = wave::lane();
  Rng& g = c.g;
  u32* q_move = c.lds.stage;
  u32* q_end = c.lds.stage + MOVQ_CAP;
  // (fed stream: the producer learns how far this consumer has come)
  if (g.feed != nullptr) wave::st_release_wg(&g.feed[FEED_POS], static_cast<u32>(g.pos));
  rng_ensure(g, 65);
  u32 bucket;
  const f64 u = int_float_pair8(rng_peek(g, g.pos + lane), bucket);
  const u32 layer = bucket >> 1;
  const f64 xi = c.lds.zig_norm_x[layer], xi1 = c.lds.zig_norm_x[layer + 1];
  const f64 x = u * xi;
  const bool fast = x < xi1;
  bool accept = fast, irregular = false;
  if (!fast) {
    if (layer == 0) {
      irregular = true;  // tail of the distribution
    } else {
      const f64 y01 = static_cast<f64>(rng_peek(g, g.pos + lane + 1)) * TWO_M64;
      if (!(y01 < 1.0)) {
        irregular = true;  // uniform_01 draws again
      } else {
        const f64 yi = c.lds.zig_norm_y[layer], yi1 = c.lds.zig_norm_y[layer + 1];
        const f64 y = yi + y01 * (yi1 - yi);
        const f64 chord = (xi - xi1) * y01 - (xi - x);
        const f64 tangent = y - (yi + (xi - x) * yi * xi);
        const f64 y_above_ubound = (xi >= 1) ? chord : tangent;
        const f64 y_above_lbound = (xi >= 1) ? tangent : chord;
        accept = y_above_ubound < 0 && (y_above_lbound < 0 || y < wave::f_exp(-(x * x / 2)));
      }
    }
  }
  const u32 mv = move_from_normal((bucket & 1u) ? x : -x, speed, std);
  // positions that start a two-output attempt: every other position of a run of slow positions
  u64 two = wave::ballot(!fast);
  u64 dbl = 0;
  while (two != 0) {
    const u32 b = static_cast<u32>(wave::ctz64(two));
    dbl |= u64(1) << b;
    two &= ~(u64(3) << b);
  }
  const u64 starts = ~(dbl << 1);
  const u64 irr = wave::ballot(irregular) & starts;
  const u32 stop = irr != 0 ? static_cast<u32>(wave::ctz64(irr)) : 64u;  // first irregular attempt
  const u64 below = stop < 64 ? lanemask_lt(stop) : ~u64(0);
  const u64 acc = wave::ballot(accept) & starts & below;
  wave::lockstep();  // queue slots read by the consumer of the previous step may be overwritten
  if ((acc >> lane) & 1u) {
    const u32 e = tail + static_cast<u32>(wave::popc64(acc & lanemask_lt(lane)));
    q_move[e % MOVQ_CAP] = mv;
    q_end[e % MOVQ_CAP] = static_cast<u32>(g.pos) + lane + 1 + static_cast<u32>((dbl >> lane) & 1u);
  }
  tail += static_cast<u32>(wave::popc64(acc));
  if (stop == 64) {
    rng_advance(g, 64 + static_cast<u32>(dbl >> 63));
  } else {
    rng_advance(g, stop);
    const f64 exact = unit_normal_exact(g, c.lds);
    if (lane == 0) {
      q_move[tail % MOVQ_CAP] = move_from_normal(exact, speed, std);
      q_end[tail % MOVQ_CAP] = static_cast<u32>(g.pos);
    }
    ++tail;
  }
  wave::sync_lds();
  return tail;
}

template <bool FWD>
MODLE_DEV_NOINLINE void generate_moves_dir(Cell& c, f64 speed, f64 std) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  move_t* moves = FWD ? ws.f_move : ws.r_move;
  ensure_inverse<FWD>(c);
  const u32* rank = FWD ? ws.f_rank : ws.r_rank;
  if (std == 0.0) {
    const u32 move_int = static_cast<u32>(static_cast<u64>(wave::f_round(speed)));
    if (NARROW_MOVES && move_int > MOVE_LIMIT) c.error = ERR_MOVE_RANGE;
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) moves[rank[i]] = ws.epoch[i] != UNBOUND ? move_int : 0;
    }
    return;
  }
  // Bound LEFs take the draws in id order.  The draws are produced 64 stream positions at a time
  // into a queue, independently of how the LEFs fall into batches; what the last step produced
  // beyond the draw of the last bound LEF is handed back by rewinding the stream position.
  const u32* q_move = c.lds.stage;
  const u32* q_end = c.lds.stage + MOVQ_CAP;
  u32 head = 0, tail = 0;  // entries consumed / produced
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct LefRegs {
    u32 E[UX], S[UX];
  };
  const auto load_lefs = [&](auto op, u32 group, LefRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 i = group + 64 * u + lane;
      r.E[u] = op(ws.epoch, i, i < n, UNBOUND, r.E[u]);
      r.S[u] = op(rank, i, i < n, 0, r.S[u]);
    }
  };
  LefRegs cur;
  load_lefs(wave::LdRaw{}, 0, cur);
  for (u32 group = 0; group < n; group += 64 * UX) {
    LefRegs g = cur;
    load_lefs(wave::LdMask{}, group, g);  // (defaults of the lanes outside the range)
    if (group + 64 * UX < n) load_lefs(wave::LdRaw{}, group + 64 * UX, cur);
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 base = group + 64 * u;
    if (base >= n) break;
    const u32 i = base + lane;
    const bool act = i < n;
    const bool bnd = act && g.E[u] != UNBOUND;
    const u32 slot = g.S[u];
    const u64 bm = wave::ballot(bnd);
    const u32 need = static_cast<u32>(wave::popc64(bm));
    while (tail - head < need) tail = draw_moves_step(c, speed, std, tail);
    u32 mv = 0;
    if (bnd) mv = q_move[(head + static_cast<u32>(wave::popc64(bm & lanemask_lt(lane)))) % MOVQ_CAP];
    head += need;
    if (NARROW_MOVES && wave::any(mv > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
    if (act) moves[slot] = mv;
    }
  }
  if (head != 0) {
    // the stream ends right after the draw of the last bound LEF: hand back what the last step
    // evaluated beyond it (queued draws, and rejected attempts that no draw followed)
    const u32 end_low = wave::uniform(q_end[(head - 1) % MOVQ_CAP]);
    c.g.pos = wave::known_uniform(c.g.pos - static_cast<u32>(static_cast<u32>(c.g.pos) - end_low));
  }
}

// The same when every active LEF is bound (always the case inside the epoch loop: generate_moves
// runs after select_and_bind_lefs): LEF i takes the i-th accepted draw of the direction, so the
// moves are a function of the stream alone.  They are stored in LEF-id order with coalesced
// stores -- no per-LEF state is read -- and the move adjustment, which walks the units in rank
// order, fetches each unit's move through its LEF id from this freshly written, compact array
// (instead of this pass scattering 4-byte stores over the rank-ordered array).  Measured on one
// box against the scattering form: 23 % fewer bytes written to the fabric and a kernel 0.7 %
// faster -- the adjustment pass itself takes twice as long (dependent gathers), every other pass
// gains from the lighter write traffic.
MODLE_DEV_NOINLINE void generate_moves_by_id(Cell& c, f64 speed_in, f64 std_in, move_t* mv_by_id) {
  const f64 speed = wave::own_regs(speed_in), std = wave::own_regs(std_in);
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  if (std == 0.0) {
    const u32 move_int = static_cast<u32>(static_cast<u64>(wave::f_round(speed)));
    if (NARROW_MOVES && move_int > MOVE_LIMIT) c.error = ERR_MOVE_RANGE;
    for (u32 base = 0; base < n; base += 64) {
      const u32 i = base + lane;
      if (i < n) mv_by_id[i] = move_int;
    }
    return;
  }
  const u32* q_move = c.lds.stage;
  const u32* q_end = c.lds.stage + MOVQ_CAP;
  u32 head = 0, tail = 0;  // entries consumed / produced
  u32 widest = 0;          // largest move this lane has stored
  for (u32 base = 0; base < n; base += 64) {
    const u32 need = umin(64u, n - base);
    while (tail - head < need) tail = draw_moves_step(c, speed, std, tail);
    const u32 mv = lane < need ? q_move[(head + lane) % MOVQ_CAP] : 0u;
    if (lane < need) mv_by_id[base + lane] = mv;
    widest = umax(widest, mv);
    head += need;
  }
  if (NARROW_MOVES && wave::any(widest > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
  if (head != 0) {
    // hand back what the last step evaluated beyond the draw of the last LEF
    const u32 end_low = wave::uniform(q_end[(head - 1) % MOVQ_CAP]);
    c.g.pos = wave::known_uniform(c.g.pos - static_cast<u32>(static_cast<u32>(c.g.pos) - end_low));
  }
}

// =============================================================================================
// adjust_moves_of_consecutive_extr_units (reference: simulation.cpp:350-407) as two segmented
// scans over rank order, fused with clamp_moves (reference: simulation.cpp:332-347).
//
// rev units, ranks high -> low:  land'[k] = min(land[k], land'[k+1] - 1) while both units are
// bound and neither reaches the 5'-end.  With d[k] = land[k] - k this is a segmented suffix
// minimum of d.  The reference tests "unit k+1 reaches the 5'-end" on the *updated* move of
// k+1; the scan uses the original move and the (rare, chromosome-end only) cases where the
// update changes the answer are replayed sequentially from the first affected rank.
// `do_adjust` / `do_clamp` exist for the phase-level test entry point.
// =============================================================================================
// The same two sweeps with FOUR consecutive ranks per lane (blocks of 256 ranks; used whenever the
// 32-bit scan applies, i.e. on every real chromosome): one 128-bit load per array and lane, the
// scan runs over the four units of a lane in registers, ONE cross-lane scan joins the 64 lanes,
// and the carries, the loop control and the violation test are paid once per 256 units instead
// of once per 64.  rev: lane 0 holds the highest ranks of a block and a lane walks its four units
// downwards, so that the suffix scan over ranks is again a prefix scan over (lane, unit).
// Returns the rank the sequential replay has to start from (adjust_moves_rev / _fwd), or -1.
template <bool FWD>
struct AdjustSweepX4 {
  struct Blk {
    wave::U32x4 P, M;
  };
  const u32* pos;
  const lefid_t* uid;
  const move_t *mv_in, *mv_by_id;
  move_t* mv_out;
  u32 n, lane, start, last, nblk;
  bool by_id, do_adjust, do_clamp;
  i32 carry_d;
  bool carry_ok, carry_cross;
  i64 viol_rank;
  u32 lane_max;  // largest move this lane has stored
  wave::U32x4 ids;
  Blk cur;

  // first rank of this lane in block t of the sweep (t = 0 is the block the sweep starts with)
  MODLE_DEV_MEMBER u32 word0(u32 t) const { return (FWD ? t : nblk - 1 - t) * 256 + 4 * (FWD ? lane : 63 - lane); }
  // the ids of a block are requested one block ahead of its positions and (gathered) moves
  MODLE_DEV_MEMBER void load_ids(u32 t) {
    const u32 w = word0(t);
    ids = wave::ld4(uid, (by_id && t < nblk && w < n) ? w : 0u);
  }
  MODLE_DEV_MEMBER void load_blk(u32 t) {
    const u32 w = word0(t);
    const bool in = t < nblk && w < n;
    cur.P = wave::ld4(pos, in ? w : 0u);
    if (by_id) {
#pragma unroll
      for (u32 q = 0; q < 4; ++q) cur.M.v[q] = wave::LdRaw{}(mv_by_id, ids.v[q], in && w + q < n, 0, 0u);
    } else {
      cur.M = wave::ld4(mv_in, in ? w : 0u);
    }
  }
  MODLE_DEV_MEMBER void init(Cell& c, bool adjust, bool clamp, const move_t* by_id_moves, move_t* out) {
    Workspace& ws = c.ws;
    n = wave::uniform(c.n_active);
    lane = wave::lane();
    // (registers of their own: as fields of the interval descriptor they come back from a spill
    // sixteen registers at a time, once per block)
    start = wave::own_regs(c.iv->start);
    last = wave::own_regs(c.iv->end - 1);
    pos = FWD ? ws.f_pos : ws.r_pos;
    uid = FWD ? ws.f_id : ws.r_id;
    mv_in = FWD ? ws.f_move : ws.r_move;
    mv_by_id = by_id_moves;
    mv_out = out;
    by_id = by_id_moves != nullptr;
    do_adjust = adjust;
    do_clamp = clamp;
    nblk = (n + 255) / 256;
    carry_d = 0;
    carry_ok = false;
    carry_cross = false;
    viol_rank = -1;
    lane_max = 0;
    load_ids(0);
    load_blk(0);
    if (1 < nblk) load_ids(1);
  }
  MODLE_DEV_MEMBER void step(u32 t) {
    const Blk g = cur;
    if (t + 1 < nblk) {
      load_blk(t + 1);
      if (t + 2 < nblk) load_ids(t + 2);
    }
    const u32 w = word0(t);
    u32 P[4], M[4], k[4];
    bool bnd[4], ok[4];
    i32 d[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {  // j: position in sweep order inside the lane
      const u32 q = FWD ? j : 3 - j;
      k[j] = w + q;
      const bool act = k[j] < n;
      P[j] = g.P.v[q];
      M[j] = g.M.v[q];
      bnd[j] = act && P[j] != UNBOUND;
      if (FWD) {
        ok[j] = do_adjust && bnd[j] && static_cast<u64>(P[j]) + M[j] <= last;
        d[j] = ok[j] ? static_cast<i32>(P[j] + M[j] - k[j]) : 0;
      } else {
        ok[j] = do_adjust && bnd[j] && static_cast<u64>(P[j]) > static_cast<u64>(start) + M[j];
        d[j] = ok[j] ? static_cast<i32>(P[j] - M[j] - k[j]) : 0;
      }
    }
    const auto pick = [](i32 a, i32 b) { return FWD ? (a > b ? a : b) : (a < b ? a : b); };
    const bool ok_in = wave::shfl_up1(ok[3]);
    bool link[4], open[4];  // link: to the unit before; open: the chain reaches the start of the lane
    i32 v[4];
    link[0] = ok[0] && (lane > 0 ? ok_in : carry_ok);
    open[0] = link[0];
    v[0] = d[0];
#pragma unroll
    for (u32 j = 1; j < 4; ++j) {
      link[j] = ok[j] && ok[j - 1];
      v[j] = link[j] ? pick(d[j], v[j - 1]) : d[j];
      open[j] = link[j] && open[j - 1];
    }
    const SegScan inc = wave_prefix_segscan32<FWD>(v[3], open[3]);
    const i32 inc_val = static_cast<i32>(inc.val);
    const i32 whole = inc.cont ? pick(inc_val, carry_d) : inc_val;  // scan value of the lane's last unit
    const i32 whole_in = static_cast<i32>(wave::shfl_up1(static_cast<u32>(whole)));
    const i32 before = lane > 0 ? whole_in : carry_d;
    bool cross[4];
    wave::U32x4 O;
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
      const i32 val = open[j] ? pick(v[j], before) : v[j];
      u32 Mnew = M[j];
      if (ok[j]) Mnew = FWD ? static_cast<u32>(val) + k[j] - P[j] : P[j] - (static_cast<u32>(val) + k[j]);
      cross[j] = ok[j] && (FWD ? static_cast<u64>(P[j]) + Mnew > last
                               : static_cast<u64>(P[j]) <= static_cast<u64>(start) + Mnew);
      O.v[FWD ? j : 3 - j] = (bnd[j] && do_clamp) ? umin(Mnew, FWD ? last - P[j] : P[j] - start) : Mnew;
      lane_max = umax(lane_max, k[j] < n ? O.v[FWD ? j : 3 - j] : 0u);
    }
    if (w + 3 < n) {
      wave::st4(mv_out, w, O);
    } else {
#pragma unroll
      for (u32 q = 0; q < 4; ++q) {
        if (w + q < n) mv_out[w + q] = O.v[q];
      }
    }
    // first unit in sweep order whose link leads to a unit that crosses the end with its updated move
    const bool cross_in = wave::shfl_up1(cross[3]);
    const bool viol0 = link[0] && (lane > 0 ? cross_in : carry_cross);
    const bool viol1 = link[1] && cross[0], viol2 = link[2] && cross[1], viol3 = link[3] && cross[2];
    const u64 vm = wave::ballot(viol0 || viol1 || viol2 || viol3);
    if (vm != 0 && viol_rank < 0) {
      const u32 fl = static_cast<u32>(wave::ctz64(vm));
      const u32 jf = wave::bcast(viol0 ? 0u : viol1 ? 1u : viol2 ? 2u : 3u, fl);
      const u32 s = 4 * fl + jf;
      const u32 b = FWD ? t : nblk - 1 - t;
      viol_rank = FWD ? static_cast<i64>(b) * 256 + s - 1 : static_cast<i64>(b) * 256 + (255 - s) + 1;
    }
    carry_d = wave::bcast(whole, 63);
    carry_ok = wave::bcast(ok[3], 63);
    carry_cross = wave::bcast(cross[3], 63);
  }
};

template <bool FWD>
MODLE_DEV_NOINLINE i64 adjust_moves_x4(Cell& c, bool do_adjust, bool do_clamp, const move_t* mv_by_id) {
  AdjustSweepX4<FWD> sw;
  sw.init(c, do_adjust, do_clamp, mv_by_id, as_moves(c.ws.tmp[0]));
  for (u32 t = 0; t < sw.nblk; ++t) sw.step(t);
  if (NARROW_MOVES && wave::any(sw.lane_max > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
  return sw.viol_rank;
}

// `mv_by_id`: moves in LEF-id order (generate_moves_by_id) or nullptr when they already sit in
// r_move in rank order (phase-level test entry point).
// `out_slot` / `swept`: the scratch array the sweep writes (ws.tmp[out_slot]) and, when the sweep
// has been done already (adjust_moves_both_x4), the rank its replay starts from
MODLE_DEV_NOINLINE void adjust_moves_rev(Cell& c, bool do_adjust, bool do_clamp,
                                         const move_t* mv_by_id = nullptr, u32 out_slot = 0,
                                         const i64* swept = nullptr) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u64 start = c.iv->start;
  // landing positions minus ranks fit 32 bits on every real chromosome: scans at half the cost
  const bool narrow = wave::uniform(c.iv->end) < 0x7F000000u;
  const move_t* mv_in = ws.r_move;
  move_t* mv_out = as_moves(ws.tmp[out_slot]);
  const u32 nbatch = (n + 63) / 64;
  const bool by_id = mv_by_id != nullptr;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  // lanes hold the ranks of a batch in DESCENDING order (lane 0 = highest rank), so that the
  // suffix scan over ranks is a prefix scan over lanes
  if (swept != nullptr) {
    viol_rank = *swept;
  } else if (narrow) {
    viol_rank = adjust_moves_x4<false>(c, do_adjust, do_clamp, mv_by_id);
  } else {
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], M[UX];
  };
  struct IdRegs {
    u32 I[UX];
  };
  // the ids of a group are requested one group ahead of its positions and (gathered) moves
  const auto load_ids = [&](u32 bg, IdRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const bool in = bg + u < nbatch;
      const u32 kq = (nbatch - 1 - (bg + u)) * 64 + (63 - lane);
      r.I[u] = wave::ld_sel(ws.r_id, kq, by_id && in && kq < n, 0);
    }
  };
  const auto load_units = [&](auto op, u32 bg, const IdRegs& ids, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const bool in = bg + u < nbatch;
      const u32 kq = (nbatch - 1 - (bg + u)) * 64 + (63 - lane);
      r.P[u] = op(ws.r_pos, kq, in && kq < n, UNBOUND, r.P[u]);
      r.M[u] = op(by_id ? mv_by_id : mv_in, by_id ? ids.I[u] : kq, in && kq < n, 0, r.M[u]);
    }
  };
  IdRegs ids;
  UnitRegs cur;
  load_ids(0, ids);
  load_units(wave::LdRaw{}, 0, ids, cur);
  if (UX < nbatch) load_ids(UX, ids);
  for (u32 bg = 0; bg < nbatch; bg += UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, bg, ids, g);  // (defaults of the lanes outside the range)
    if (bg + UX < nbatch) {
      load_units(wave::LdRaw{}, bg + UX, ids, cur);
      if (bg + 2 * UX < nbatch) load_ids(bg + 2 * UX, ids);
    }
    const u32* Pq = g.P;
    const u32* Mq = g.M;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    if (bg + u >= nbatch) break;
    const u32 bi = nbatch - 1 - (bg + u);
    const u32 k = bi * 64 + (63 - lane);
    const bool act = k < n;
    const u32 P = Pq[u];
    const u32 M = Mq[u];
    const bool bnd = act && P != UNBOUND;
    const bool okself = do_adjust && bnd && static_cast<u64>(P) > start + M;
    const i64 d = okself ? static_cast<i64>(P - M) - static_cast<i64>(k) : 0;
    const bool ok_next_in = wave::shfl_up1(okself);
    const bool ok_next = lane > 0 ? ok_next_in : carry_ok;
    const bool link = okself && ok_next;
    const SegScan sc = narrow ? wave_prefix_segscan32<false>(static_cast<i32>(d), link)
                              : wave_prefix_segscan<false>(SegScan{d, link});
    i64 val = sc.val;
    if (sc.cont) val = imin64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = P - static_cast<u32>(val + static_cast<i64>(k));
    const bool cross = okself && static_cast<u64>(P) <= start + Mnew;
    const u32 Mst = (bnd && do_clamp) ? umin(Mnew, P - static_cast<u32>(start)) : Mnew;
    if (act) wave::st_stream(&mv_out[k], Mst);
    if (NARROW_MOVES && wave::any(act && Mst > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
    const bool cross_next_in = wave::shfl_up1(cross);
    const bool cross_next = lane > 0 ? cross_next_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_next);
    // highest rank k whose link to k+1 the scan got wrong (lowest lane); the replay starts at k+1
    if (vm != 0 && viol_rank < 0) viol_rank = bi * 64 + (63 - wave::ctz64(vm)) + 1;
    carry_d = wave::bcast(val, 63);
    carry_ok = wave::bcast(okself, 63);
    carry_cross = wave::bcast(cross, 63);
    }
  }
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    // sequential replay (reference loop) from the first rank whose decision the scan got wrong.
    // The reference adjusts all moves first and clamps afterwards, so the replay carries the
    // UNCLAMPED updated move of the unit it has just left (mv_out holds clamped values); the
    // unit it starts from is the one whose updated move crosses the 5'-end.
    bool first = true;
    u32 M2u = 0;
    for (u32 i = static_cast<u32>(viol_rank); i > 0; --i) {
      u32 M1 = by_id ? mv_by_id[ws.r_id[i - 1]] : mv_in[i - 1];
      const u64 P1 = ws.r_pos[i - 1], P2 = ws.r_pos[i];
      const bool both = P1 != UNBOUND && P2 != UNBOUND;
      if (both) {
        const bool cross2 = first || P2 <= start + M2u;
        if (!(P1 <= start + M1 || cross2)) {
          const u64 pos1 = P1 - M1, pos2 = P2 - M2u;
          if (pos2 <= pos1) M1 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      M2u = M1;
      first = false;
      const u32 Mst = (P1 != UNBOUND && do_clamp) ? umin(M1, static_cast<u32>(P1 - start)) : M1;
      mv_out[i - 1] = Mst;
      if (NARROW_MOVES && Mst > MOVE_LIMIT) c.error = ERR_MOVE_RANGE;
    }
    wave::sync_mem();
  }
  swap_with_scratch(ws.r_move, ws.tmp[out_slot]);
}

// `out_slot` / `swept`: the scratch array the sweep writes (ws.tmp[out_slot]) and, when the sweep
// has been done already (adjust_moves_both_x4), the rank its replay starts from
MODLE_DEV_NOINLINE void adjust_moves_fwd(Cell& c, bool do_adjust, bool do_clamp,
                                         const move_t* mv_by_id = nullptr, u32 out_slot = 0,
                                         const i64* swept = nullptr) {
  Workspace& ws = c.ws;
  const u32 n = wave::uniform(c.n_active);
  const u32 lane = wave::lane();
  const u64 last = static_cast<u64>(c.iv->end) - 1;
  const bool narrow = wave::uniform(c.iv->end) < 0x7F000000u;  // see adjust_moves_rev
  const move_t* mv_in = ws.f_move;
  move_t* mv_out = as_moves(ws.tmp[out_slot]);
  const u32 nbatch = (n + 63) / 64;
  const bool by_id = mv_by_id != nullptr;
  i64 carry_d = 0;
  bool carry_ok = false, carry_cross = false;
  i64 viol_rank = -1;
  if (swept != nullptr) {
    viol_rank = *swept;
  } else if (narrow) {
    viol_rank = adjust_moves_x4<true>(c, do_adjust, do_clamp, mv_by_id);
  } else {
  constexpr u32 UX = 4;  // batches per group; the next group's loads go before this group's stores
  struct UnitRegs {
    u32 P[UX], M[UX];
  };
  struct IdRegs {
    u32 I[UX];
  };
  const auto load_ids = [&](u32 bg, IdRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 kq = (bg + u) * 64 + lane;
      r.I[u] = wave::ld_sel(ws.f_id, kq, by_id && kq < n, 0);
    }
  };
  const auto load_units = [&](auto op, u32 bg, const IdRegs& ids, UnitRegs& r) {
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
      const u32 kq = (bg + u) * 64 + lane;
      r.P[u] = op(ws.f_pos, kq, kq < n, UNBOUND, r.P[u]);
      r.M[u] = op(by_id ? mv_by_id : mv_in, by_id ? ids.I[u] : kq, kq < n, 0, r.M[u]);
    }
  };
  IdRegs ids;
  UnitRegs cur;
  load_ids(0, ids);
  load_units(wave::LdRaw{}, 0, ids, cur);
  if (UX < nbatch) load_ids(UX, ids);
  for (u32 bg = 0; bg < nbatch; bg += UX) {
    UnitRegs g = cur;
    load_units(wave::LdMask{}, bg, ids, g);  // (defaults of the lanes outside the range)
    if (bg + UX < nbatch) {
      load_units(wave::LdRaw{}, bg + UX, ids, cur);
      if (bg + 2 * UX < nbatch) load_ids(bg + 2 * UX, ids);
    }
    const u32* Pq = g.P;
    const u32* Mq = g.M;
#pragma unroll
    for (u32 u = 0; u < UX; ++u) {
    const u32 bi = bg + u;
    if (bi >= nbatch) break;
    const u32 k = bi * 64 + lane;
    const bool act = k < n;
    const u32 P = Pq[u];
    const u32 M = Mq[u];
    const bool bnd = act && P != UNBOUND;
    const bool okself = do_adjust && bnd && static_cast<u64>(P) + M <= last;
    const i64 d = okself ? static_cast<i64>(static_cast<u64>(P) + M) - static_cast<i64>(k) : 0;
    const bool ok_prev_in = wave::shfl_up1(okself);
    const bool ok_prev = lane > 0 ? ok_prev_in : carry_ok;
    const bool link = okself && ok_prev;  // link between k-1 and k
    const SegScan sc = narrow ? wave_prefix_segscan32<true>(static_cast<i32>(d), link)
                              : wave_prefix_segscan<true>(SegScan{d, link});
    i64 val = sc.val;
    if (sc.cont) val = imax64(val, carry_d);
    u32 Mnew = M;
    if (okself) Mnew = static_cast<u32>(val + static_cast<i64>(k) - static_cast<i64>(P));
    const bool cross = okself && static_cast<u64>(P) + Mnew > last;
    const u32 Mst = (bnd && do_clamp) ? umin(Mnew, static_cast<u32>(last - P)) : Mnew;
    if (act) wave::st_stream(&mv_out[k], Mst);
    if (NARROW_MOVES && wave::any(act && Mst > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
    const bool cross_prev_in = wave::shfl_up1(cross);
    const bool cross_prev = lane > 0 ? cross_prev_in : carry_cross;
    const u64 vm = wave::ballot(link && cross_prev);
    // lowest rank k-1 whose updated move crosses the 3'-end while the scan linked it to k
    if (vm != 0 && viol_rank < 0) viol_rank = static_cast<i64>(bi) * 64 + wave::ctz64(vm) - 1;
    carry_d = wave::bcast(val, 63);
    carry_ok = wave::bcast(okself, 63);
    carry_cross = wave::bcast(cross, 63);
    }
  }
  }
  wave::sync_mem();
  if (viol_rank >= 0) {
    // see adjust_moves_rev: the replay works on unclamped moves; unit viol_rank is the one whose
    // updated move crosses the 3'-end
    bool first = true;
    u32 M1u = 0;
    for (u32 i = static_cast<u32>(viol_rank) + 1; i < n; ++i) {
      u32 M2 = by_id ? mv_by_id[ws.f_id[i]] : mv_in[i];
      const u64 P1 = ws.f_pos[i - 1], P2 = ws.f_pos[i];
      const bool both = P1 != UNBOUND && P2 != UNBOUND;
      if (both) {
        const bool cross1 = first || P1 + M1u > last;
        if (!(cross1 || P2 + M2 > last)) {
          const u64 pos1 = P1 + M1u, pos2 = P2 + M2;
          if (pos1 >= pos2) M2 += static_cast<u32>(pos1 - pos2) + 1;
        }
      }
      M1u = M2;
      first = false;
      const u32 Mst = (P2 != UNBOUND && do_clamp) ? umin(M2, static_cast<u32>(last - P2)) : M2;
      mv_out[i] = Mst;
      if (NARROW_MOVES && Mst > MOVE_LIMIT) c.error = ERR_MOVE_RANGE;
    }
    wave::sync_mem();
  }
  swap_with_scratch(ws.f_move, ws.tmp[out_slot]);
}

// Both sweeps in one loop: they are independent of each other (rev walks the blocks downwards, fwd
// upwards), so every iteration carries two dependency chains instead of one.
MODLE_DEV_NOINLINE void adjust_moves_both_x4(Cell& c, const move_t* mv_rev, const move_t* mv_fwd, i64& viol_rev,
                                             i64& viol_fwd) {
  AdjustSweepX4<false> r;
  AdjustSweepX4<true> f;
  r.init(c, true, true, mv_rev, as_moves(c.ws.tmp[0]));
  f.init(c, true, true, mv_fwd, as_moves(c.ws.tmp[1]));
  for (u32 t = 0; t < r.nblk; ++t) {
    r.step(t);
    f.step(t);
  }
  viol_rev = r.viol_rank;
  viol_fwd = f.viol_rank;
  if (NARROW_MOVES && wave::any(r.lane_max > MOVE_LIMIT || f.lane_max > MOVE_LIMIT)) c.error = ERR_MOVE_RANGE;
  // the largest fwd move of the epoch bounds what a fwd unit can contribute to a primary collision
  // (detect_primary's filter pass); unknown when the sequential replay is going to change moves
  c.max_fwd_move = viol_fwd < 0 ? wave::bcast(wave_prefix_max_u32(f.lane_max), 63) : 0xFFFFFFFFu;
}

// `all_bound`: every active LEF is bound (the epoch loop's invariant at this point)
// the move adjustment on the id-ordered moves of ws.tmp[8] (rev) / ws.tmp[9] (fwd)
MODLE_DEV void phase_adjust_moves_by_id(Cell& c) {
  move_t* mv_rev = as_moves(c.ws.tmp[8]);
  move_t* mv_fwd = as_moves(c.ws.tmp[9]);
  if (wave::uniform(c.iv->end) < 0x7F000000u) {  // (the 32-bit scans apply: see adjust_moves_rev)
    PHASE(c, 6, i64 vr; i64 vf; adjust_moves_both_x4(c, mv_rev, mv_fwd, vr, vf);
          wave::sync_mem();
          adjust_moves_rev(c, true, true, mv_rev, 0, &vr); adjust_moves_fwd(c, true, true, mv_fwd, 1, &vf));
  } else {
    PHASE(c, 6, adjust_moves_rev(c, true, true, mv_rev); adjust_moves_fwd(c, true, true, mv_fwd));
  }
}

MODLE_DEV void phase_generate_moves(Cell& c, bool burnin_completed, bool all_bound = true) {
  const Params& p = *c.p;
  c.max_fwd_move = 0xFFFFFFFFu;  // (set by adjust_moves_both_x4 when it runs)
  if (!all_bound) {
    PHASE(c, 5, generate_moves_dir<false>(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std);
          generate_moves_dir<true>(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std);
          wave::sync_mem());
    PHASE(c, 6, adjust_moves_rev(c, true, true); adjust_moves_fwd(c, true, true));
    return;
  }
  // id-ordered moves: two scratch arrays of their own (the helper wave of sim_pair.h fills them
  // while the rank updates use the others)
  move_t* mv_rev = as_moves(c.ws.tmp[8]);
  move_t* mv_fwd = as_moves(c.ws.tmp[9]);
  PHASE(c, 5, generate_moves_by_id(c, burnin_completed ? p.rev_speed : p.rev_speed_burnin, p.rev_std, mv_rev);
        generate_moves_by_id(c, burnin_completed ? p.fwd_speed : p.fwd_speed_burnin, p.fwd_std, mv_fwd);
        wave::sync_mem());
  phase_adjust_moves_by_id(c);
}

}  // namespace modle_dev
