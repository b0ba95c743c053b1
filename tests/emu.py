"""Pure-Python walk-through of the DEVICE data layout (rank blocks, SA samples, exception
list, segment table) used by the CPU tests to validate what the host builder hands to the
GPU.  Test helper only: it lives under tests/, is never imported by the product, and is far
too slow to be anything but a checker."""
import numpy as np

BLOCK_SYMS = 192


class IndexEmu:
    def __init__(self, pindex, graph, view=None):
        v = pindex.view if view is None else view
        self.n = v.text_len
        self.sa_rate = v.sa_rate
        self.C = [int(x) for x in v.C]
        nb = v.n_blocks
        raw = pindex._arr(v.bwt_blocks, nb * 16, np.uint32).reshape(nb, 16)
        self.hdr = raw[:, :4].astype(np.int64)
        self.sym = raw[:, 4:].copy().view(np.uint64).reshape(nb, 6)
        self.samples = pindex._arr(v.sa_samples, v.n_samples, np.uint32)
        self.exc_row = pindex._arr(v.exc_row, v.n_exc, np.uint32)
        self.exc_sa = pindex._arr(v.exc_sa, v.n_exc, np.uint32)
        self.seg_start = pindex._arr(v.seg_start, v.n_segs + 1, np.uint32)
        self.seg_node = pindex._arr(v.seg_node, v.n_segs, np.uint32)
        self.seg_noff = pindex._arr(v.seg_noff, v.n_segs, np.uint32)
        self.seg_dir = pindex._arr(v.seg_dir, v.n_dir, np.uint32)
        self.ftab_len = v.ftab_len
        self.ftab = pindex._arr(v.ftab, 2 << (2 * v.ftab_len), np.uint32) if v.ftab_len else None
        self.exc_shift = v.exc_shift
        self.exc_super = pindex._arr(v.exc_super, ((nb - 1) >> v.exc_shift) + 1, np.uint32)
        self.exc_pos = {int(r): i for i, r in enumerate(self.exc_row)}
        self.node_id = graph.node_id

    def bwt(self, i):
        b, o = divmod(i, BLOCK_SYMS)
        return self._sym(b, o)

    def _sym(self, b, o):
        g, j = divmod(o, 64)
        return ((int(self.sym[b, 2 * g]) >> j) & 1) | (((int(self.sym[b, 2 * g + 1]) >> j) & 1) << 1)

    def rank(self, c, i):
        b, o = divmod(i, BLOCK_SYMS)
        h = self.hdr[b]
        if c < 3:
            base = int(h[c])
        else:
            base = b * BLOCK_SYMS - int(h[0]) - int(h[1]) - int(h[2]) - (int(h[3]) >> 8) - int(self.exc_super[b >> self.exc_shift])
        cnt = 0
        for j in range(o):
            if self._sym(b, j) == c:
                cnt += 1
        if c == 0 and (int(h[3]) & 0xFF):
            e = (int(h[3]) >> 8) + int(self.exc_super[b >> self.exc_shift])
            while e < len(self.exc_row) and self.exc_row[e] < i and self.exc_row[e] < (b + 1) * BLOCK_SYMS:
                cnt -= 1
                e += 1
        return base + cnt

    def search(self, kmer):
        """kmer: string over ACGT.  Half-open SA interval of its occurrences."""
        l, r = 0, self.n
        q = self.ftab_len
        if q and len(kmer) >= q:
            code = 0
            for ch in kmer[-q:]:
                code = code * 4 + 'ACGT'.index(ch)
            l, r = int(self.ftab[2 * code]), int(self.ftab[2 * code + 1])
            if r <= l:
                return 0, 0
            kmer = kmer[:-q]
        for ch in reversed(kmer):
            c = 'ACGT'.index(ch)
            l = self.C[c] + self.rank(c, l)
            r = self.C[c] + self.rank(c, r)
            if r <= l:
                return 0, 0
        return l, r

    def locate(self, row):
        steps = 0
        while True:
            if row % self.sa_rate == 0:
                return int(self.samples[row // self.sa_rate]) + steps
            if row in self.exc_pos:
                return int(self.exc_sa[self.exc_pos[row]]) + steps
            c = self.bwt(row)
            row = self.C[c] + self.rank(c, row)
            steps += 1

    def map(self, pos):
        d = int(self.seg_dir[pos >> 6])
        while self.seg_start[d + 1] <= pos:
            d += 1
        return int(self.node_id[self.seg_node[d]]), int(self.seg_noff[d]) + pos - int(self.seg_start[d])

    def on_path_hits(self, seeds):
        """seeds: iterable of (read_id, read_off, kmer)."""
        out = []
        cache = {}
        for rid, roff, km in seeds:
            if 'N' in km:
                continue
            if km not in cache:
                l, r = self.search(km)
                cache[km] = [self.map(self.locate(i)) for i in range(l, r)]
            for nid, noff in cache[km]:
                out.append((nid, noff, rid, roff))
        return out


class PartEmu(IndexEmu):
    """One part of an index in several parts (every part is a complete FM index)."""

    def __init__(self, pindex, view, graph):
        super().__init__(pindex, graph, view)
