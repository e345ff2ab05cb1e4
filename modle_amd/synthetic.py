"""Deterministic synthetic inputs shaped like the reference's GRCh38 example data.

The reference bundles `examples/data/hg38.chrom.sizes` and
`examples/data/hg38_extrusion_barriers.bed.xz` (38 815 CTCF barriers, one per ~79.6 kb, BED score =
occupancy in [0.60, 1.00], mean 0.838, sd 0.115, strands ~50/50).  Those files cannot travel with
this repository, so benchmarks and parity tests use a generator that reproduces their
*statistics*: the 24 public GRCh38 primary-assembly lengths and, per chromosome, the NUMBER of
barriers the bundled file holds for it (facts like the lengths: chr1 3 518 ... chr21 427, chrY 26 --
the per-chromosome shape sets the ragged tail of a whole-genome launch; a uniform density would
give chrY 719) at distinct uniform positions, strand ~ Bernoulli(0.5) and occupancy
~ Normal(0.84, 0.115) clipped to [0.60, 1.0].  Randomness comes from a self-contained SplitMix64
stream so the inputs are bit-identical on every machine and Python/numpy version.
(`tests/test_reference_inputs.py` checks the table against the real files where they exist.)
"""
import math

import numpy as np

from .params import DIR_FWD, DIR_REV

# GRCh38 primary assembly chromosome lengths (public facts; same values as hg38.chrom.sizes)
GRCH38 = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
]

# barriers per chromosome of examples/data/hg38_extrusion_barriers.bed.xz (H1 CTCF sites; 38 815 in
# all, every record with a strand)
GRCH38_H1_BARRIERS = {
    "chr1": 3518, "chr2": 2974, "chr3": 2469, "chr4": 1905, "chr5": 2048, "chr6": 2109, "chr7": 1985,
    "chr8": 1772, "chr9": 1618, "chr10": 1899, "chr11": 2159, "chr12": 1943, "chr13": 943,
    "chr14": 1238, "chr15": 1278, "chr16": 1335, "chr17": 1719, "chr18": 866, "chr19": 1441,
    "chr20": 1132, "chr21": 427, "chr22": 793, "chrX": 1218, "chrY": 26,
}

BARRIER_SPACING_BP = 79564  # mean spacing of that file: what a chromosome without an entry above gets
_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next() >> 11) * (1.0 / (1 << 53))

    def normal(self):
        # Box-Muller on two uniforms (only the statistics matter here)
        u1 = max(self.uniform(), 1e-300)
        u2 = self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def synthetic_barriers(name, length, seed=42, spacing=BARRIER_SPACING_BP, count=None):
    """Returns (pos u64[B] sorted unique, dir u8[B], occupancy f64[B]) for one chromosome: `count`
    barriers, or one per `spacing` bp."""
    h = 0
    for ch in name.encode():
        h = (h * 131 + ch) & _MASK
    rng = SplitMix64(seed ^ h ^ (length << 1))
    n = int(round(length / spacing)) if count is None else int(count)
    picked = set()
    while len(picked) < n:
        picked.add(1 + rng.next() % (length - 2))
    pos = np.array(sorted(picked), dtype=np.uint64)
    dirs = np.empty(n, dtype=np.uint8)
    occ = np.empty(n, dtype=np.float64)
    for i in range(n):
        dirs[i] = DIR_REV if rng.uniform() < 0.5 else DIR_FWD
        occ[i] = min(1.0, max(0.60, 0.84 + 0.115 * rng.normal()))
    return pos, dirs, occ


def grch38_like(seed=42, chroms=None):
    """List of dicts {name, size, start, end, bar_pos, bar_dir, bar_occupancy} in genome order."""
    out = []
    for name, size in GRCH38:
        if chroms is not None and name not in chroms:
            continue
        pos, dirs, occ = synthetic_barriers(name, size, seed, count=GRCH38_H1_BARRIERS[name])
        out.append({"name": name, "size": size, "start": 0, "end": size, "bar_pos": pos,
                    "bar_dir": dirs, "bar_occupancy": occ})
    return out


def synthetic_chromosome(name, size, seed=42, with_barriers=True, spacing=BARRIER_SPACING_BP):
    if with_barriers:
        pos, dirs, occ = synthetic_barriers(name, size, seed, spacing)
    else:
        pos = np.zeros(0, dtype=np.uint64)
        dirs = np.zeros(0, dtype=np.uint8)
        occ = np.zeros(0, dtype=np.float64)
    return {"name": name, "size": size, "start": 0, "end": size, "bar_pos": pos, "bar_dir": dirs,
            "bar_occupancy": occ}
