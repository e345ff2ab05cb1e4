#!/usr/bin/env python3
"""sha256 over the device sources of the simulation kernel, in a fixed order.

`profiles/traffic.json` records it next to every measured HBM-traffic figure, and `bench.py` quotes
the figure only while the sources still hash the same (`roofline.traffic` is a committed
measurement of another run: it must not outlive the kernel it describes).

    python tools/csrc_hash.py            prints the hash of the working tree
"""
import hashlib
import os
import sys

DEVICE_SOURCES = [
    "modle_hip.hip", "sim_device.h", "sim_types.h", "sim_cell.h", "sim_rng.h", "sim_pair.h",
    "sim_helper.h", "sim_bind_rank.h", "sim_moves.h", "sim_barriers.h", "sim_collisions.h",
    "sim_release.h", "sim_contacts.h", "sim_burnin.h", "sim_epoch.h", "wave_hip.h", "modle_math.h",
    "Makefile",
]


def csrc_sha256(root):
    h = hashlib.sha256()
    for name in DEVICE_SOURCES:
        with open(os.path.join(root, "modle_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    sys.exit(0)
