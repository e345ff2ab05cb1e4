"""The device code under the lane emulator AND the sanitizers (the GPU has none on this pool).  Whole cells in
the regimes that use the most scratch -- a burn-in whose rank updates borrow the generator's ring (2 400 LEFs
at a processivity of 25 kb), BASELINE configs[4]'s parameters with Bernoulli trials, a small default cell --
under MemorySanitizer (tests/wave_emu: `make msan_emu`, clang: no word of LDS or workspace is read that nothing
has written; the harness poisons both), under gcc's Address + UndefinedBehavior sanitizers (`make asan_emu`), and
-- every lane a thread, every collective a barrier -- under ThreadSanitizer (`make tsan_emu`): a word that one
lane writes and another touches without a collective in between is a reported race (round 4: the rank update
seeded the counts inside its keys by a read-modify-write across lanes with no barrier behind the writes; the
GPU executes a wave's LDS operations in order and 16 000 fuzz seeds never differed, the lane emulator broke)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "wave_emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# size, barriers, cells, LEFs per Mb, processivity, skip burn-in, target contact density, minor pblock
REGIMES = (["120000000", "1", "1", "20", "25000", "0", "0.002"],
           ["60000000", "1", "1", "64", "0", "1", "0.01", "0.3"],
           ["8000000", "1", "1", "0", "0", "0", "0.1"])  # (defaults: burn-in + contact sampling, barriers)


@pytest.fixture(scope="module")
def builds():
    """the sanitizer builds, side by side (a minute each)"""
    procs = {t: subprocess.Popen(["make", "-C", EMU, t], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for t in ("msan_emu", "asan_emu", "tsan_emu", "tsan_emu_philox", "tsan_emu_w12")}
    return {t: (p.communicate()[0], p.returncode) for t, p in procs.items()}


def run_regimes(exe, marks):
    for args in REGIMES:
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
        assert not any(m in run.stderr for m in marks), run.stderr[:3000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[:1000])


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with MemorySanitizer")
def test_whole_cells_read_no_uninitialised_memory(builds):
    out, rc = builds["msan_emu"]
    if rc != 0 and "msan" in out.lower():
        pytest.skip("MemorySanitizer runtime not available")
    assert rc == 0, out[-2000:]
    run_regimes(os.path.join(EMU, "msan_emu"), ("MemorySanitizer",))


def test_whole_cells_under_address_and_undefined_behaviour_sanitizers(builds):
    out, rc = builds["asan_emu"]
    assert rc == 0, out[-2000:]
    run_regimes(os.path.join(EMU, "asan_emu"), ("Sanitizer", "runtime error"))


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with ThreadSanitizer")
def test_no_lane_touches_what_another_wrote_without_a_collective(builds):
    out, rc = builds["tsan_emu"]
    if rc != 0 and "tsan" in out.lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert rc == 0, out[-2000:]
    exe = os.path.join(EMU, "tsan_emu")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    # 400 LEFs at a processivity of 300 kb (more than 64 re-inserted units per rank update); trials; a burn-in
    for args in (["5000000", "1", "1", "80", "300000", "1", "0.02"],
                 ["30000000", "1", "1", "40", "0", "1", "0.004", "0.3"],
                 ["2000000", "1", "1", "0", "0", "0", "0.01", "0", "110"]):  # (burn-in cut at 110 epochs)
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=1200, env=env)
        assert "ThreadSanitizer" not in run.stderr, run.stderr[:4000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[:1000])


# Round 5: the regimes the race detector had not seen.  The build prints which form of a
# pass a cell takes (-DMODLE_EMU_TRACE_RANK); every regime must show its marker, or the test tests nothing.
#   LEF-BAR detection with Bernoulli trials, batches resolved in rounds (a barrier every ~100 bp, blocking
#     probabilities 0.7 / 0.4: tests/parity_cases.py "dense_barriers_trials")
#   the one-sweep rank update with its keys in the generator's ring (1 023 keys; "rebinds_beyond_sort_buffer")
#   more LEFs released than the LDS list holds: overflow of release_lefs, sweeping bind, general rank update
#     ("mass_release")
TSAN_REGIMES_R05 = (
    (["600000", "1", "1", "64", "0", "1", "0.1", "0", "0", "spacing=100", "major=0.7", "minor=0.4"],
     ("lef_bar_trials: rev", "lef_bar_trials: fwd", "(resolved in rounds)")),
    (["120000000", "1", "1", "20", "25000", "1", "0.003"], ("key_cap 1023",)),
    (["120000000", "1", "1", "20", "8000", "1", "0.0008"],
     ("the next bind sweeps", "rank_update: general")),
)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with ThreadSanitizer")
def test_race_detector_on_trials_ring_borrow_and_mass_release(builds):
    out, rc = builds["tsan_emu"]
    if rc != 0 and "tsan" in out.lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert rc == 0, out[-2000:]
    exe = os.path.join(EMU, "tsan_emu")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    for args, markers in TSAN_REGIMES_R05:
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=1500, env=env)
        assert "ThreadSanitizer" not in run.stderr, run.stderr[:4000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[-1000:])
        for m in markers:
            assert m in run.stderr, (args, m, run.stderr[-1500:])


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with ThreadSanitizer")
def test_race_detector_on_the_philox_policy(builds):
    """the counter-based block generator (sim_rng.h, -DMODLE_RNG_PHILOX) through a burn-in and a cell with trials"""
    out, rc = builds["tsan_emu_philox"]
    if rc != 0 and "tsan" in out.lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert rc == 0, out[-2000:]
    exe = os.path.join(EMU, "tsan_emu_philox")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    for args in (["2000000", "1", "1", "0", "0", "0", "0.01", "0", "60"],
                 ["30000000", "1", "1", "40", "0", "1", "0.002", "0.3"]):
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=1500, env=env)
        assert "ThreadSanitizer" not in run.stderr, run.stderr[:4000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[-1000:])


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with ThreadSanitizer")
def test_race_detector_on_the_geometry_of_the_12_wave_kernels(builds):
    """round 5: the kernels for 12 waves per workgroup halve the PRNG blocks and the LDS key buffers
    (-DMODLE_WAVES_PER_CU=12): a rank update that borrows the (smaller) ring for 511 keys, and a burn-in on the
    256-key sort buffer"""
    out, rc = builds["tsan_emu_w12"]
    if rc != 0 and "tsan" in out.lower():
        pytest.skip("ThreadSanitizer runtime not available")
    assert rc == 0, out[-2000:]
    exe = os.path.join(EMU, "tsan_emu_w12")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    for args, marker in ((["60000000", "1", "1", "20", "25000", "1", "0.004"], "key_cap 511"),
                         (["2000000", "1", "1", "0", "0", "0", "0.01", "0", "60"], "key_cap 255")):
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=1500, env=env)
        assert "ThreadSanitizer" not in run.stderr, run.stderr[:4000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[-1000:])
        assert marker in run.stderr, (args, marker, run.stderr[-1500:])
