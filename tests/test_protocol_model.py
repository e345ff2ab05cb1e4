"""The hand-over protocol of the helper-wave mode under ThreadSanitizer (tests/protocol_model).

The product's own protocol code -- modle_amd/csrc/sim_pair.h, sim_helper.h and the fed stream of
sim_rng.h -- compiled against a one-lane backend and run as main wave / helper / PRNG producer on
host threads, the passes replaced by stand-ins whose outputs the other side recomputes:

* `fixed`, `dynamic`: no data race, every hand-over delivers the values of ITS request, the
  generator comes back at the position the stream says, no hand-over is lost;
* `stuck`: a helper that withholds a signal (the test fault of the GPU test
  tests/test_gpu_wait_deadline.py) -- the abort word releases every spin loop, the cell reports
  ERR_CANCELLED;
* the regression build reads the request counter AFTER the claim (the race that hung a GPU box for
  15 minutes in round 3): the watchdog must catch it hanging.
Reference semantics being preserved: one cell per worker, no waiting between workers
(scheduler_simulate.cpp:190-271); `_ctx` polled every epoch (simulation.cpp:933).
"""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "protocol_model")


@pytest.fixture(scope="module")
def binaries():
    proc = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    return HERE


def _run(binaries, exe, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    return subprocess.run([os.path.join(binaries, exe), *args], capture_output=True, text=True, env=env, timeout=600)


@pytest.mark.parametrize("mode", [("fixed",), ("dynamic", "150")])
def test_protocol_has_no_race_and_loses_no_hand_over(binaries, mode):
    proc = _run(binaries, "handover_model", *mode)
    assert "ThreadSanitizer" not in proc.stderr, proc.stderr[:4000]
    assert proc.returncode == 0, proc.stdout + proc.stderr[:4000]
    assert "every hand-over checked" in proc.stdout


def test_abort_word_releases_a_wave_whose_helper_stopped_answering(binaries):
    proc = _run(binaries, "handover_model", "stuck")
    assert "ThreadSanitizer" not in proc.stderr, proc.stderr[:4000]
    assert proc.returncode == 0, proc.stdout + proc.stderr[:4000]
    assert "ERR_CANCELLED" in proc.stdout


def test_request_counter_read_after_the_claim_is_caught_hanging(binaries):
    proc = _run(binaries, "handover_model_race", "dynamic", "200")
    assert proc.returncode == 3, (proc.returncode, proc.stdout, proc.stderr[:2000])
    assert "hand-over was lost" in proc.stderr
