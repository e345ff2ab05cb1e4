"""Replays the reference's unit-test vectors (tests/golden/reference_kats.json) on a backend.

A backend exposes the phase-level entry points that mirror the reference's
`Simulation::test_*` hooks (src/libmodle/cpu/include/modle/simulation.hpp:413-567).  The same
runner drives the CPU oracle and the HIP path, so the parity tests read like the reference's own.
"""
import json
import os

import numpy as np

from modle_amd.params import DIR_FWD, DIR_REV

UNBOUND = np.uint64(0xFFFFFFFFFFFFFFFF)
EVENT_SHIFT = 56

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")


def load_cases():
    with open(GOLDEN) as fh:
        return json.load(fh)["cases"]


def word(idx, ev):
    return np.uint64(idx) | (np.uint64(ev) << np.uint64(EVENT_SHIFT))


class KatState:
    """Plain-array image of the reference test fixtures (Lef array, ranks, moves, collisions)."""

    def __init__(self, case, lefs_key="lefs"):
        lefs = case[lefs_key]
        n = len(lefs)
        self.n = n
        self.rev_pos = np.array([l[0] for l in lefs], dtype=np.uint64)
        self.fwd_pos = np.array([l[1] for l in lefs], dtype=np.uint64)
        self.epoch = np.array([l[2] for l in lefs], dtype=np.uint64)
        for i in case.get("released", []):
            self.rev_pos[i] = self.fwd_pos[i] = self.epoch[i] = UNBOUND
        self.rev_rank = np.array(case.get("rev_ranks") or list(range(n)), dtype=np.uint64)
        self.fwd_rank = np.array(case.get("fwd_ranks") or list(range(n)), dtype=np.uint64)
        self.rev_moves = np.array(case.get("rev_moves") or [0] * n, dtype=np.uint64)
        self.fwd_moves = np.array(case.get("fwd_moves") or [0] * n, dtype=np.uint64)
        self.rev_coll = np.zeros(n, dtype=np.uint64)
        self.fwd_coll = np.zeros(n, dtype=np.uint64)
        iv = case.get("interval", {"start": 0, "end": 1 << 40})
        self.start, self.end = iv["start"], iv["end"]
        bars = case.get("barriers", [])
        self.bar_pos = np.array([b["pos"] for b in bars], dtype=np.uint64)
        # motif '+' blocks units extruding in rev direction (extrusion_barriers_impl.hpp:61-72)
        self.bar_dir = np.array([DIR_REV if b["strand"] == "+" else DIR_FWD for b in bars],
                                dtype=np.uint8)
        self.bar_active = np.array([1 if b["active"] else 0 for b in bars], dtype=np.uint8)
        self.n5 = 0
        self.n3 = 0


def run_case(backend, case):
    """Executes the hook sequence of one reference test case and asserts its expectations."""
    name = case["name"]
    if name.startswith("LEFs ranking"):
        for k in ("1", "2"):
            st = KatState(case, "lefs" + k)
            backend.rank_lefs(st, init_buffers=True)
            assert st.rev_rank.tolist() == case["rev_ranks_expected" + k], name
            assert st.fwd_rank.tolist() == case["fwd_ranks_expected" + k], name
        return
    st = KatState(case)
    cfg = backend.make_config(case["config"])
    rng = backend.make_prng(case["seed"])
    for call in case["calls"]:
        if call == "test_adjust_and_clamp_moves":
            backend.adjust_and_clamp_moves(st)
        elif call == "test_detect_units_at_interval_boundaries":
            backend.detect_units_at_interval_boundaries(st)
        elif call == "test_detect_primary_lef_lef_collisions":
            backend.detect_primary_lef_lef_collisions(cfg, st, rng)
        elif call == "test_process_lef_lef_collisions":
            backend.process_lef_lef_collisions(cfg, st, rng)
        elif call == "test_detect_lef_bar_collisions":
            backend.detect_lef_bar_collisions(cfg, st, rng)
        elif call == "test_correct_moves_for_lef_bar_collisions":
            backend.correct_moves_for_lef_bar_collisions(st)
        elif call == "test_process_collisions":
            backend.process_collisions(cfg, st, rng)
        elif call == "test_fix_secondary_lef_lef_collisions":
            backend.fix_secondary_lef_lef_collisions(st)
        else:
            raise AssertionError(f"unknown hook {call}")
    for key in ("rev_moves", "fwd_moves"):
        exp = case.get(key + "_expected", case.get(key + "_adjusted"))
        if exp is not None:
            assert getattr(st, key).tolist() == exp, f"{name}: {key}"
    for key, arr in (("rev_collisions_expected", st.rev_coll),
                     ("fwd_collisions_expected", st.fwd_coll)):
        if key in case:
            exp = [int(word(i, e)) for i, e in case[key]]
            assert [int(x) for x in arr] == exp, f"{name}: {key}"
    for key, arr in (("rev_ranks_after", st.rev_rank), ("fwd_ranks_after", st.fwd_rank)):
        if key in case:
            assert arr.tolist() == case[key], f"{name}: {key}"
