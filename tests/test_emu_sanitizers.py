"""The device code under the lane emulator AND the sanitizers (the GPU has none on this pool): MemorySanitizer (tests/wave_emu: `make msan_emu`, clang):
whole cells in the regimes that use the most scratch -- a burn-in whose rank updates borrow the generator's
ring (2 400 LEFs at a processivity of 25 kb), and BASELINE configs[4]'s parameters with Bernoulli trials --
must not read a word of LDS or workspace that nothing has written (the harness poisons both)."""
import os
import shutil
import subprocess

import pytest

# size, barriers, cells, LEFs per Mb, processivity, skip burn-in, target contact density, minor pblock
REGIMES = (["120000000", "1", "1", "20", "25000", "0", "0.002"],
           ["60000000", "1", "1", "64", "0", "1", "0.01", "0.3"],
           ["5000000", "0", "1"])

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang++ with MemorySanitizer")
def test_whole_cells_read_no_uninitialised_memory():
    emu = os.path.join(HERE, "wave_emu")
    build = subprocess.run(["make", "-C", emu, "msan_emu"], capture_output=True, text=True)
    if build.returncode != 0 and "msan" in (build.stderr + build.stdout).lower():
        pytest.skip("MemorySanitizer runtime not available")
    assert build.returncode == 0, build.stderr[-2000:]
    exe = os.path.join(emu, "msan_emu")
    for args in REGIMES:
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
        assert "MemorySanitizer" not in run.stderr, run.stderr[:3000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[:1000])


def test_whole_cells_under_address_and_undefined_behaviour_sanitizers():
    """the same cells under gcc's AddressSanitizer + UndefinedBehaviorSanitizer (`make asan_emu`)"""
    emu = os.path.join(HERE, "wave_emu")
    build = subprocess.run(["make", "-C", emu, "asan_emu"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    exe = os.path.join(emu, "asan_emu")
    for args in REGIMES:
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
        assert "Sanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[:3000]
        assert run.returncode == 0 and "rc=0" in run.stdout, (run.stdout, run.stderr[:1000])
