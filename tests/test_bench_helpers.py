"""bench.py's small helpers that do not need a GPU: the committed traffic figure is quoted only
for the kernel it was measured on, and a missing / malformed profiles/traffic.json costs the
`traffic` field, not the benchmark line."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tools.csrc_hash import csrc_sha256  # noqa: E402


@pytest.fixture
def fake_root(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # the hash is still the one of the real sources
    monkeypatch.setattr(bench, "csrc_hash", lambda: csrc_sha256(ROOT))
    return tmp_path


def test_traffic_file_absent_gives_a_pair_of_nones(fake_root):
    assert bench.measured_traffic("grch38:2048") == (None, None)


def test_traffic_file_malformed_gives_a_pair_of_nones(fake_root):
    (fake_root / "profiles" / "traffic.json").write_text("{ not json")
    assert bench.measured_traffic("grch38:2048") == (None, None)
    (fake_root / "profiles" / "traffic.json").write_text("[1, 2]")
    assert bench.measured_traffic("grch38:2048") == (None, None)


def test_traffic_of_another_kernel_is_not_quoted(fake_root):
    entry = {"bytes_per_launch": 1.0e12, "source": "profiles/rXX", "kernel": "k", "csrc_sha256": "0" * 64}
    (fake_root / "profiles" / "traffic.json").write_text(json.dumps({"grch38:2048": entry}))
    value, source = bench.measured_traffic("grch38:2048")
    assert value is None and source.startswith("stale")
    entry["csrc_sha256"] = csrc_sha256(ROOT)
    (fake_root / "profiles" / "traffic.json").write_text(json.dumps({"grch38:2048": entry}))
    value, source = bench.measured_traffic("grch38:2048")
    assert value == 1.0e12 and "profiles/rXX" in source
    assert bench.measured_traffic("chr1:512") == (None, None)


def test_launch_mode_words_follow_the_library_record():
    one = dict(helper_waves=0, prng_producer_waves=0, tail_helpers=1)
    assert bench.launch_mode(one) == "one wave per cell, idle waves help in the tail of the launch"
    assert bench.launch_mode(dict(one, tail_helpers=0)) == "one wave per cell"
    assert bench.launch_mode(dict(helper_waves=1, prng_producer_waves=1, tail_helpers=0)).startswith(
        "main wave + helper + PRNG producer")
    assert bench.launch_mode(dict(helper_waves=1, prng_producer_waves=0, tail_helpers=0)).startswith(
        "main wave + helper (")


def test_size_class_words_follow_the_library_record():
    assert bench.size_class({"size_class": 0}).startswith("narrow (16-bit")
    assert bench.size_class({"size_class": 1}).startswith("wide (32-bit")
    assert bench.size_class({}).startswith("narrow")  # (a library older than the field)


def test_workspace_placement_words_follow_the_library_record():
    """`config.workspace_placement` of the bench line: what `modle_hip_launch_info` says the placement search saw
    (include/modle_hip.h), microseconds turned into ms; a library record without a search reads as such"""
    assert bench.workspace_placement({"workspace_tries": 0}) == "first allocation (no search)"
    assert bench.workspace_placement({}) == "first allocation (no search)"
    got = bench.workspace_placement({"workspace_tries": 12, "workspace_probe_us": 604, "workspace_probe_worst_us": 699})
    assert got == {"candidates_probed": 12, "probe_ms_kept": 0.604, "probe_ms_slowest": 0.699}


def test_the_launch_info_record_mirrors_the_header():
    """the ctypes mirror of `modle_hip_launch_info` has the header's fields in the header's order"""
    import os
    import re

    from modle_amd.params import LaunchInfo

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "modle_hip.h")) as f:
        text = f.read()
    body = text[text.index("typedef struct modle_hip_launch_info {"):text.index("} modle_hip_launch_info;")]
    fields = re.findall(r"^\s*uint64_t\s+(\w+);", body, flags=re.M)
    assert fields == [name for name, _ in LaunchInfo._fields_]


def test_the_committed_traffic_figure_is_of_the_committed_kernel():
    """profiles/traffic.json carries the hash of the device sources it was measured on: a change to any file
    of modle_amd/csrc after the last profile round makes `roofline.traffic` null -- say so here, in the CPU
    suite, not only on the GPU box."""
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        table = json.load(f)
    stale = [key for key, entry in table.items() if entry["csrc_sha256"] != csrc_sha256(ROOT)]
    if stale:
        # (not a failure of the code: the line will say `traffic: null` until the profile round is run again)
        pytest.skip(f"profiles/traffic.json is stale for {stale}: re-run tools/profile_round.sh + tools/assemble_profile.py")
