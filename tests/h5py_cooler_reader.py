"""Independent reader of a cooler (v3) file: prints its tables as JSON.

Run by tests/test_cooler_writer.py under an interpreter that has h5py (the image's conda Python
3.9: /opt/conda/bin/python3.9; the suite's own interpreter has no HDF5 binding).  Pixels are
fetched chromosome by chromosome THROUGH the indexes (chrom_offset -> bin1_offset -> pixel rows),
the way the cooler / hictk readers the reference's validator relies on do
(test/integration/.../validators/cooler.py: per-chromosome `fetch`), so that a wrong index shows
up as wrong pixels and not only as a wrong index table."""
import json
import sys

import h5py


def main(path):
    out = {}
    with h5py.File(path, "r") as f:
        out["attrs"] = {k: (v.decode() if isinstance(v, bytes) else (v.item() if hasattr(v, "item") else v))
                        for k, v in f.attrs.items()}
        out["attr_dtypes"] = {k: str(f.attrs.get_id(k).dtype) for k in f.attrs}
        out["dtypes"] = {name: str(f[name].dtype) for name in
                         ("chroms/length", "bins/chrom", "bins/start", "bins/end", "pixels/bin1_id",
                          "pixels/bin2_id", "pixels/count", "indexes/bin1_offset",
                          "indexes/chrom_offset")}
        out["filters"] = {name: {"compression": f[name].compression, "opts": f[name].compression_opts,
                                 "chunks": list(f[name].chunks or [])}
                          for name in ("pixels/bin1_id", "pixels/count")}
        names = [n.decode().rstrip("\x00") for n in f["chroms/name"][:]]
        out["chroms"] = list(zip(names, [int(x) for x in f["chroms/length"][:]]))
        out["bins"] = [[int(a), int(b), int(c)] for a, b, c in
                       zip(f["bins/chrom"][:], f["bins/start"][:], f["bins/end"][:])]
        chrom_offset = f["indexes/chrom_offset"][:]
        bin1_offset = f["indexes/bin1_offset"][:]
        fetched = {}
        for k, name in enumerate(names):
            lo, hi = int(bin1_offset[chrom_offset[k]]), int(bin1_offset[chrom_offset[k + 1]])
            rows = []
            # row by row through bin1_offset, like a range query
            for b in range(int(chrom_offset[k]), int(chrom_offset[k + 1])):
                p0, p1 = int(bin1_offset[b]), int(bin1_offset[b + 1])
                b1 = f["pixels/bin1_id"][p0:p1]
                assert (b1 == b).all(), f"bin1_offset[{b}] does not delimit the pixels of bin {b}"
            b1 = f["pixels/bin1_id"][lo:hi]
            b2 = f["pixels/bin2_id"][lo:hi]
            cn = f["pixels/count"][lo:hi]
            rows = [[int(a), int(b), int(c)] for a, b, c in zip(b1, b2, cn)]
            fetched[name] = rows
        out["pixels_by_chrom"] = fetched
        out["n_pixels"] = int(f["pixels/count"].shape[0])
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main(sys.argv[1])
