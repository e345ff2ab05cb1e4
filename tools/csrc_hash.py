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

def device_sources(root):
    """Every *.h / *.hpp / *.hip of modle_amd/csrc plus the Makefile, sorted by name: whatever is compiled into
    the kernel or decides the workspace layout (a hand-kept list missed four of them in round 4)."""
    d = os.path.join(root, "modle_amd", "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".h", ".hpp", ".hip")) or n == "Makefile")
    return names


def csrc_sha256(root):
    h = hashlib.sha256()
    for name in device_sources(root):
        with open(os.path.join(root, "modle_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    sys.exit(0)
