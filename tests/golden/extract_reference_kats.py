#!/usr/bin/env python3
"""Extract the known-answer vectors held by the reference's own unit tests into JSON.

Runs in the authoring container only (needs /root/reference).  It reads the *data* of the
deterministic hot-path test cases -- LEF layouts, barrier layouts, rank arrays, input moves,
expected moves / collision words / ranks, and the ordered list of `Simulation::test_*` hooks each
case calls -- from

    test/units/simulation_cpu/simulation_simple_unit_test.cpp   (cases at :131-195, :241-846)
    test/units/simulation_cpu/simulation_complex_unit_test.cpp  (cases at :25-756)

and writes tests/golden/reference_kats.json.  No reference source text is stored: only numbers,
case names and hook names.
"""
import json
import os
import re
import sys

REF = "/root/reference/test/units/simulation_cpu"
FILES = ["simulation_simple_unit_test.cpp", "simulation_complex_unit_test.cpp"]
DEFAULT_SEED = 10556020843759504871  # test/units/simulation_cpu/common.hpp:21

EVENTS = {  # test/units/simulation_cpu/common.hpp:23-29 (all carry the COLLISION bit)
    "CHROM_BOUNDARY": 0x10 | 0x08,
    "LEF_BAR": 0x10 | 0x04,
    "LEF_LEF_PRIMARY": 0x10 | 0x02,
    "LEF_LEF_SECONDARY": 0x10 | 0x01,
}
HOOKS = [
    "test_adjust_and_clamp_moves", "test_adjust_moves", "test_rank_lefs",
    "test_detect_units_at_interval_boundaries", "test_detect_lef_bar_collisions",
    "test_correct_moves_for_lef_bar_collisions", "test_detect_primary_lef_lef_collisions",
    "test_process_lef_lef_collisions", "test_process_collisions",
    "test_fix_secondary_lef_lef_collisions",
]


def ints(body):
    return [int(x) for x in re.findall(r"-?\d+", body)]


def parse_case(name, line, body):
    case = {"name": name, "source_line": line}
    m = re.search(r"init_config\((\d+),\s*(\d+)\)", body)
    cfg = {"bypass": 0.0, "major_pblock": 1.0, "minor_pblock": 0.0}
    if m:
        cfg["rev_speed"], cfg["fwd_speed"] = int(m.group(1)), int(m.group(2))
    for key, field in (("bypass", "probability_of_extrusion_unit_bypass"),
                       ("major_pblock", "lef_bar_major_collision_pblock"),
                       ("minor_pblock", "lef_bar_minor_collision_pblock")):
        mm = re.search(r"c\." + field + r"\s*=\s*([\d.]+)\s*;", body)
        if mm:
            cfg[key] = float(mm.group(1))
    case["config"] = cfg
    m = re.search(r'init_interval\("(\w+)",\s*(\d+)(?:,\s*(\d+))?(?:,\s*(\d+))?\)', body)
    if m:
        size = int(m.group(2))
        start = int(m.group(3)) if m.group(3) else 0
        end = min(int(m.group(4)), size) if m.group(4) else size
        case["interval"] = {"name": m.group(1), "size": size, "start": start, "end": end}
    m = re.search(r"random::PRNG\((\d+)ULL\)", body)
    case["seed"] = int(m.group(1)) if m else DEFAULT_SEED

    # LEF collections (lefs / lefs1 / lefs2)
    for lm in re.finditer(r"(?:std::array<Lef,\s*\w+>|std::vector<Lef>)\s+(\w+)\s*\{(.*?)\};",
                          body, re.S):
        lefs = []
        for a, b, e in re.findall(r"construct_lef\((\d+),\s*(\d+)(?:,\s*(\d+))?\)", lm.group(2)):
            lefs.append([int(a), int(b), int(e) if e else 0])
        case[lm.group(1)] = lefs
    released = [int(x) for x in re.findall(r"lefs\[(\d+)\]\.release\(\)", body)]
    if released:
        case["released"] = released

    bars = re.findall(r"ExtrusionBarrier\{(\d+),\s*([\d.]+),\s*([\d.]+),\s*'([+-])'\}", body)
    if bars:
        inactive = "State::INACTIVE" in body
        case["barriers"] = [{"pos": int(p), "stp_active": float(a), "stp_inactive": float(i),
                             "strand": s, "active": not inactive} for p, a, i, s in bars]

    # plain integer arrays
    for am in re.finditer(r"std::array<(?:std::size_t|bp_t),\s*\w+>\s+(\w+)\s*\{([^;]*?)\};",
                          body, re.S):
        case[am.group(1)] = ints(am.group(2))
    # expected collisions
    for cm in re.finditer(r"std::array<CollisionT,\s*\w+>\s+(\w+)\s*\{(.*?)\};", body, re.S):
        words = []
        for item in re.findall(r"CollisionT\{([^}]*)\}", cm.group(2)):
            item = item.strip()
            if not item:
                words.append([0, 0])
            else:
                idx, ev = [x.strip() for x in item.split(",")]
                words.append([int(idx), EVENTS[ev]])
        case[cm.group(1)] = words
    # explicit rank checks after fix_secondary
    for which in ("rev_ranks", "fwd_ranks"):
        chk = re.findall(r"CHECK\(" + which + r"\[(\d+)\]\s*==\s*(\d+)\)", body)
        if chk:
            exp = [None] * len(chk)
            for i, v in chk:
                exp[int(i)] = int(v)
            case[which + "_after"] = exp
    # ordered hook calls
    calls = []
    for hm in re.finditer(r"\b(" + "|".join(HOOKS) + r")\s*\(", body):
        calls.append((hm.start(), hm.group(1)))
    case["calls"] = [c for _, c in sorted(calls)]
    return case


def main(out_path):
    cases = []
    for fn in FILES:
        text = open(os.path.join(REF, fn)).read()
        heads = list(re.finditer(r'TEST_CASE\("([^"]+)"', text))
        for k, h in enumerate(heads):
            end = heads[k + 1].start() if k + 1 < len(heads) else len(text)
            body = text[h.start():end]
            name = h.group(1)
            if name.startswith(("Bind LEFs", "Generate LEF moves")):
                continue  # property tests with RNG: their parameters are extracted by
                # extract_reference_unit_vectors.py and replayed by tests/unit_vector_runner.py
            line = text.count("\n", 0, h.start()) + 1
            c = parse_case(name, line, body)
            c["source_file"] = "test/units/simulation_cpu/" + fn
            cases.append(c)
    with open(out_path, "w") as fh:
        json.dump({"default_seed": DEFAULT_SEED, "cases": cases}, fh, indent=1)
    print(f"{len(cases)} cases -> {out_path}")
    for c in cases:
        print(" ", c["name"], c["calls"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else
         os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json"))
