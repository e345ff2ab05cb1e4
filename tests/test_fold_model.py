"""The arithmetic of the device's burn-in fold (modle_amd/csrc/sim_burnin.h: fold_terms_exact), lane by
lane in scalar C++ (tests/fold_model/fold_model.cpp), against the plain sequential sum on 1.8 M batches of
every hard shape: ties of the rounding, binade crossings inside a batch, zero / huge terms."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parallel_exact_fold_equals_the_sequential_sum(tmp_path):
    exe = str(tmp_path / "fold_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                    os.path.join(HERE, "fold_model", "fold_model.cpp")], check=True)
    out = subprocess.run([exe, "300000"], check=True, capture_output=True, text=True).stdout
    assert " 0 mismatches" in out, out
    # the simulation's shape at a whole chromosome's size: few true additions per batch
    out = subprocess.run([exe, "400", "10000", "0"], check=True, capture_output=True, text=True).stdout
    assert " 0 mismatches" in out, out
    assert float(out.split(" batches, ")[1].split()[0]) < 2.0, out
