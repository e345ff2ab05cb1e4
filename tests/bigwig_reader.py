"""Minimal bigWig reader for the tests: parses the file by the published layout (header, total
summary, chromosome B+ tree, R-tree index, zlib-compressed wig sections), independently of the
writer's code.  `query(chrom, start, end)` walks the R-tree like a genome browser would."""
import struct
import zlib


class BigWig:
    def __init__(self, path):
        self.d = open(path, "rb").read()
        d = self.d
        (magic, self.version, self.zoom_levels, self.chrom_tree_off, self.data_off, self.index_off,
         self.field_count, self.defined_fields, self.autosql_off, self.summary_off,
         self.uncompress_buf, self.ext_off) = struct.unpack_from("<IHHQQQHHQQIQ", d, 0)
        assert magic == 0x888FFC26, hex(magic)
        assert struct.unpack_from("<I", d, len(d) - 4)[0] == 0x888FFC26  # trailing magic
        self.summary = dict(zip(("bases", "min", "max", "sum", "sumsq"),
                                struct.unpack_from("<Qdddd", d, self.summary_off)))
        self.chroms = self._read_chrom_tree()
        self.n_sections = struct.unpack_from("<Q", d, self.data_off)[0]

    def _read_chrom_tree(self):
        d, off = self.d, self.chrom_tree_off
        magic, block, key_size, val_size, count, _ = struct.unpack_from("<IIIIQQ", d, off)
        assert magic == 0x78CA8C91 and val_size == 8
        out = []

        def node(o):
            leaf, _, n = struct.unpack_from("<BBH", d, o)
            o += 4
            for _ in range(n):
                key = d[o:o + key_size].rstrip(b"\0").decode()
                if leaf:
                    cid, size = struct.unpack_from("<II", d, o + key_size)
                    out.append((cid, key, size))
                    o += key_size + 8
                else:
                    child = struct.unpack_from("<Q", d, o + key_size)[0]
                    node(child)
                    o += key_size + 8

        node(off + 32)
        assert len(out) == count
        return [(name, size) for _, name, size in sorted(out)]

    def _leaves(self, chrom_id=None, start=0, end=2**32 - 1):
        d = self.d
        magic, block, count, c0, s0, c1, s1, end_off, per_slot, _ = struct.unpack_from(
            "<IIQIIIIQII", d, self.index_off)
        assert magic == 0x2468ACE0 and count == self.n_sections
        found = []

        def overlaps(a0, b0, a1, b1):
            if chrom_id is None:
                return True
            return (a0, b0) < (chrom_id, end) and (a1, b1) > (chrom_id, start)

        def node(o):
            leaf, _, n = struct.unpack_from("<BBH", d, o)
            o += 4
            for _ in range(n):
                if leaf:
                    a0, b0, a1, b1, off, size = struct.unpack_from("<IIIIQQ", d, o)
                    if overlaps(a0, b0, a1, b1):
                        found.append((off, size))
                    o += 32
                else:
                    a0, b0, a1, b1, child = struct.unpack_from("<IIIIQ", d, o)
                    if overlaps(a0, b0, a1, b1):
                        node(child)
                    o += 24

        node(self.index_off + 48)
        return found

    def _section(self, off, size):
        raw = zlib.decompress(self.d[off:off + size])
        assert len(raw) <= self.uncompress_buf
        cid, s0, s1, step, span, typ, _, n = struct.unpack_from("<IIIIIBBH", raw, 0)
        assert typ == 3 and len(raw) == 24 + 4 * n  # fixedStep
        vals = struct.unpack_from(f"<{n}f", raw, 24)
        return cid, s0, s1, step, span, list(vals)

    def sections(self):
        return [self._section(off, size) for off, size in self._leaves()]

    def query(self, chrom, start, end):
        """[(base start, base end, value)] of the items overlapping [start, end)"""
        cid = [n for n, _ in self.chroms].index(chrom)
        out = []
        for off, size in self._leaves(cid, start, end):
            c, s0, s1, step, span, vals = self._section(off, size)
            if c != cid:
                continue
            for i, v in enumerate(vals):
                a = s0 + i * step
                b = min(a + span, self.chroms[cid][1])
                if a < end and b > start:
                    out.append((a, b, v))
        return out
