#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's SMALL unit tests that touch the hot path
into JSON (tests/golden/reference_unit_vectors.json).  Companion of extract_reference_kats.py;
runs in the authoring container only (needs /root/reference).  Only numbers, case names and call
names are stored -- no reference source text.

    test/units/stats/descriptive_test.cpp:27-115            mean / SSD / variance / std
    test/units/contact_matrix/contact_matrix_internal_test.cpp:15-49   transpose / encode / decode
    test/units/contact_matrix/contact_matrix_dense_test.cpp:37-123     increment / missed updates
    test/units/simulation_cpu/collision_encoding_test.cpp:27-175       Collision<> words
    test/units/simulation_cpu/simulation_simple_unit_test.cpp:27-128, 198-238
                                                             Bind LEFs 001-003, Generate LEF moves 001
                                                             (property tests: their parameters)
    test/units/stats/correlation_test.cpp:124-222            Pearson / Spearman, ties, weights
                                                             (values and p-values)
    test/units/libmodle_io/bed_parser_test.cpp:71-122        BED "strip quotes" valid / invalid, CRLF
    test/units/libmodle_io/bigwig_test.cpp:100-135           bigwig::Writer (two chromosomes, iota)
"""
import json
import os
import re
import sys

REF = "/root/reference/test/units"
EV = {"COLLISION": 0x10, "CHROM_BOUNDARY": 0x08, "LEF_BAR": 0x04, "LEF_LEF_PRIMARY": 0x02,
      "LEF_LEF_SECONDARY": 0x01}


def cases_of(path):
    text = open(os.path.join(REF, path)).read()
    heads = list(re.finditer(r'TEST_CASE\("([^"]+)"', text))
    for k, h in enumerate(heads):
        end = heads[k + 1].start() if k + 1 < len(heads) else len(text)
        yield h.group(1), text.count("\n", 0, h.start()) + 1, text[h.start():end]


def event_value(expr):
    """'Collision<>::COLLISION | Collision<>::LEF_BAR' or 'COLLISION() | LEF_BAR()' -> int"""
    v = 0
    for name in re.findall(r"Collision<>::(\w+)", expr):
        v |= EV[name]
    return v


def stats_vectors():
    out = []
    path = "stats/descriptive_test.cpp"
    want = {"Mean": "mean", "Sum of squared deviations": "ssd", "Variance": "variance",
            "Standard Deviation": "std"}
    for name, line, body in cases_of(path):
        if name not in want:
            continue
        v = [int(x) for x in re.search(r"v1\{([^}]*)\}", body).group(1).split(",")]
        m = re.search(r"const auto result = ([\d.]+);", body)
        exp = float(m.group(1)) if m else float(re.search(r"WithinRel\(([\d.]+),", body).group(1))
        out.append({"name": name, "source": f"test/units/{path}:{line}", "what": want[name],
                    "values": v, "expected": exp,
                    # Catch2's default: float epsilon * 100 (descriptive_test.cpp:20-21)
                    "rel_tolerance": 1.1920928955078125e-07 * 100})
    return out


def matrix_internal_vectors():
    out = []
    path = "contact_matrix/contact_matrix_internal_test.cpp"
    for name, line, body in cases_of(path):
        src = f"test/units/{path}:{line}"
        if name.endswith("transpose_coords"):
            calls = re.findall(r"transpose_coords\((\d+),\s*(\d+)\)", body)
            exps = re.findall(r"== PixelCoordinates\{(\d+),\s*(\d+)\}", body)
            out.append({"name": name, "source": src, "what": "transpose",
                        "cases": [[int(a), int(b), int(c), int(d)] for (a, b), (c, d) in zip(calls, exps)]})
        elif name.endswith("encode_idx"):
            nrows = int(re.search(r"nrows = (\d+);", body).group(1))
            calls = [(0, 0) if "PixelCoordinates" in c else tuple(int(x) for x in re.findall(r"\d+", c)[:2])
                     for c in re.findall(r"encode_idx\(([^;]*)\);", body)]
            exps = [int(x) for x in re.findall(r"CHECK\(i\d == (\d+)\)", body)]
            out.append({"name": name, "source": src, "what": "encode", "nrows": nrows,
                        "cases": [[r, c, e] for (r, c), e in zip(calls, exps)]})
        elif name.endswith("decode_idx"):
            nrows = int(re.search(r"nrows = (\d+);", body).group(1))
            calls = [int(x) for x in re.findall(r"decode_idx\((\d+),", body)]
            exps = re.findall(r"== PixelCoordinates\{(\d+),\s*(\d+)\}", body)
            out.append({"name": name, "source": src, "what": "decode", "nrows": nrows,
                        "cases": [[i, int(r), int(c)] for i, (r, c) in zip(calls, exps)]})
    return out


def matrix_dense_vectors():
    """the increment / missed-update behaviour: ops in order with the state each CHECK expects"""
    out = []
    path = "contact_matrix/contact_matrix_dense_test.cpp"
    for name, line, body in cases_of(path):
        if name not in ("ContactMatrixDense simple", "ContactMatrixDense in/decrement"):
            continue
        body = body.split("if constexpr")[0]  # debug-build-only out-of-bound checks
        m = re.search(r"ContactMatrixDense<>\s+(\w)\((\d+),\s*(\d+)\)", body)
        var, nrows, ncols = m.group(1), int(m.group(2)), int(m.group(3))
        steps = []
        for tok in re.finditer(var + r"\.(increment|decrement|subtract)\((\d+),\s*(\d+)(?:,\s*(\d+))?\)|"
                               r"(?:CHECK|REQUIRE)\(" + var + r"\.(get|get_tot_contacts|get_n_of_missed_updates)"
                               r"\(([^)]*)\) == (\d+)\)", body):
            if tok.group(1):
                steps.append({"op": tok.group(1), "row": int(tok.group(2)), "col": int(tok.group(3)),
                              "n": int(tok.group(4)) if tok.group(4) else 1})
            else:
                args = [int(x) for x in re.findall(r"\d+", tok.group(6))]
                steps.append({"check": tok.group(5), "args": args, "expected": int(tok.group(7))})
        out.append({"name": name, "source": f"test/units/{path}:{line}", "nrows": nrows,
                    "ncols": ncols, "steps": steps})
    return out


def collision_vectors():
    out = []
    path = "simulation_cpu/collision_encoding_test.cpp"
    for name, line, body in cases_of(path):
        src = f"test/units/{path}:{line}"
        if name == "Collision encoding":
            idx = int(re.search(r"std::size_t idx = (\d+);", body).group(1))
            arr = re.search(r"events\{(.*?)\};", body, re.S).group(1)
            events = [event_value(e) for e in arr.split(",") if "Collision" in e]
            idx2 = int(re.search(r"idx = (\d+);\s*collision\.set_idx", body).group(1))
            out.append({"name": name, "source": src, "what": "roundtrip", "index": idx,
                        "events": events, "second_index": idx2,
                        # INDEX_MASK of the reference's 64-bit word (collision_encoding_test.cpp:20-24)
                        "max_index_bits": 55})
            continue
        # predicate cases: (index list, event with COLLISION, event without)
        m = re.search(r"std::array<std::size_t,\s*\d+>\{([^}]*)\}", body)
        idxs = [int(x) for x in m.group(1).split(",")] if m else \
            [int(re.search(r"std::size_t idx = (\d+);", body).group(1))]
        kind = [k for k in ("CHROM_BOUNDARY", "LEF_BAR", "LEF_LEF_PRIMARY", "LEF_LEF_SECONDARY")
                if ("Collision<>::" + k) in body][0]
        out.append({"name": name, "source": src, "what": "predicates", "indices": idxs,
                    "kind": EV[kind],
                    # with the COLLISION bit: occurred, occurred(kind); not avoided, not avoided(kind);
                    # without it: the reverse (the four CHECK / CHECK_FALSE groups of every case)
                    "n_checks": len(re.findall(r"CHECK(?:_FALSE)?\(", body))})
    return out


def property_tests():
    out = []
    path = "simulation_cpu/simulation_simple_unit_test.cpp"
    for name, line, body in cases_of(path):
        if not name.startswith(("Bind LEFs", "Generate LEF moves")):
            continue
        c = {"name": name, "source": f"test/units/{path}:{line}"}
        m = re.search(r'init_interval\("(\w+)",\s*(\d+)(?:,\s*(\d+))?(?:,\s*(\d+))?\)', body)
        size = int(m.group(2))
        c["interval"] = {"name": m.group(1), "size": size, "start": int(m.group(3) or 0),
                         "end": min(int(m.group(4)), size) if m.group(4) else size}
        m = re.search(r"nlefs = (\d+);", body)
        c["nlefs"] = int(m.group(1)) if m else 0
        m = re.search(r"iters = (\d+);", body)
        if m:
            c["iters"] = int(m.group(1))
        m = re.search(r"bernoulli_trial\{([\d.]+)\}", body)
        if m:
            c["mask_probability"] = float(m.group(1))
        m = re.search(r"c\.bin_size = (\d+);", body)
        if m:
            c["bin_size"] = int(m.group(1))
            c["speed_std_fraction"] = float(re.search(r"\* ([\d.]+);", body).group(1))
        c["seed"] = 10556020843759504871  # DEFAULT_PRNG, test/units/simulation_cpu/common.hpp:21
        out.append(c)
    return out


def _c_string(expr):
    """concatenation of adjacent C string literals -> the string they denote"""
    parts = re.findall(r'"((?:[^"\\]|\\.)*)"', expr)
    txt = "".join(parts)
    return txt.encode().decode("unicode_escape")


def correlation_vectors():
    out = []
    path = "stats/correlation_test.cpp"
    for name, line, body in cases_of(path):
        if "data_dir()" in body or "v1{" not in body:
            continue  # (the cases that read the Zenodo data files)
        v1 = [int(x) for x in re.search(r"v1\{([^}]*)\}", body).group(1).split(",")]
        v2 = [int(x) for x in re.search(r"v2\{([^}]*)\}", body).group(1).split(",")]
        w = re.search(r"\bw\{([^}]*)\}", body)
        checks = [float(x) for x in re.findall(r"WithinRel\((-?[\d.]+),", body)]
        c = {"name": name, "source": f"test/units/{path}:{line}",
             "method": "spearman" if "Spearman<>" in body else "pearson", "v1": v1, "v2": v2,
             "weights": [float(x) for x in w.group(1).split(",")] if w else None,
             "expected": checks[0],
             # weighted forms: CHECK(std::isnan(pv))
             "expected_pvalue": checks[1] if len(checks) > 1 else None,
             # DEFAULT_FP_TOLERANCE (correlation_test.cpp:28-30): float epsilon * 100
             "rel_tolerance": 1.1920928955078125e-07 * 100}
        out.append(c)
    return out


def bed_vectors():
    out = []
    path = "libmodle_io/bed_parser_test.cpp"
    for name, line, body in cases_of(path):
        if name == "BED: strip quotes":
            valid, invalid = body.split('SECTION("invalid")')
            m = re.search(r"line\{(.*?)\};", valid, re.S)
            rec = {"line": _c_string(m.group(1))}
            for field, pat in (("chrom", r'record\.chrom == "([^"]*)"'), ("chrom_start", r"chrom_start == (\d+)"),
                               ("chrom_end", r"chrom_end == (\d+)"), ("name", r'record\.name == "([^"]*)"'),
                               ("score", r"record\.score == ([\d.]+)"), ("strand", r"record\.strand == '(.)'"),
                               ("thick_start", r"thick_start == (\d+)"), ("thick_end", r"thick_end == (\d+)")):
                v = re.search(pat, valid).group(1)
                rec[field] = v if field in ("chrom", "name", "strand") else float(v) if field == "score" else int(v)
            m = re.search(r'bed::BED\((".*?")\)\.chrom == (".*?")\);', valid)
            out.append({"name": name + " / valid", "source": f"test/units/{path}:{line}", "record": rec,
                        "unbalanced_quote": {"line": _c_string(m.group(1)), "chrom": _c_string(m.group(2))},
                        "must_throw": [_c_string(x) for x in re.findall(r"CHECK_THROWS\(bed::BED\((.*?)\)\);", invalid)]})
        elif name == "BED Parser CRLF":
            n = int(re.search(r"num_records = (\d+);", body).group(1))
            fmt_w = _c_string(re.search(r'w\.write\(fmt::format\((".*?"), i\)\);', body).group(1))
            fmt_r = _c_string(re.search(r'bed::BED\(fmt::format\((".*?"), i\)\)', body).group(1))
            out.append({"name": name, "source": f"test/units/{path}:{line}",
                        "file_lines": [fmt_w.replace("{}", str(i)) for i in range(n)],
                        "expected_records": [fmt_r.replace("{}", str(i)) for i in range(n)]})
    return out


def bigwig_vectors():
    out = []
    path = "libmodle_io/bigwig_test.cpp"
    for name, line, body in cases_of(path):
        if name != "bigwig::Writer":
            continue
        names = re.findall(r'"(chr\w+)"', re.search(r"chrom_names\{([^}]*)\}", body).group(1))
        sizes = [int(x) for x in re.search(r"chrom_sizes\{([^}]*)\}", body).group(1).split(",")]
        bin_size = int(re.search(r"bin_size = (\d+);", body).group(1))
        # values: iota from 0 over size / bin_size bins, written with span = step = bin_size;
        # read back: interval i = [bin_size * i, bin_size * (i + 1)) with value float(i)
        out.append({"name": name, "source": f"test/units/{path}:{line}", "chrom_names": names,
                    "chrom_sizes": sizes, "bin_size": bin_size, "values": "iota"})
    return out


def main(out_path):
    data = {"stats": stats_vectors(), "matrix_internal": matrix_internal_vectors(),
            "matrix_dense": matrix_dense_vectors(), "collision_encoding": collision_vectors(),
            "property_tests": property_tests(), "correlation": correlation_vectors(),
            "bed_parser": bed_vectors(), "bigwig_writer": bigwig_vectors()}
    with open(out_path, "w") as fh:
        json.dump(data, fh, indent=1)
    for k, v in data.items():
        print(k, len(v), [c["name"] for c in v])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else
         os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_unit_vectors.json"))
