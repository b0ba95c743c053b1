"""ctypes face of the oracle (oracle/psi_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``psi_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libpsi_oracle.so')
_lib = None

u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'psi_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or \
            (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpsi_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get('PSI_ORACLE_LIB')         # (a sanitized build of the checker: tools/asan_full.sh)
        if not alt:
            build()
        L = C.CDLL(alt or _LIB_PATH)
        L.orc_graph_new.restype = C.c_void_p
        L.orc_graph_new.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p,
                                    C.c_void_p, C.c_void_p]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_pindex_build.restype = C.c_void_p
        L.orc_pindex_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_pindex_position.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, u64p, u64p]
        L.orc_strset_position.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.orc_pindex_free.argtypes = [C.c_void_p]
        L.orc_pindex_textlen.restype = C.c_uint64
        L.orc_pindex_textlen.argtypes = [C.c_void_p]
        L.orc_pindex_text.restype = C.c_void_p
        L.orc_pindex_text.argtypes = [C.c_void_p]
        L.orc_fm_from_text.restype = C.c_void_p
        L.orc_fm_from_text.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_fm_find.restype = C.c_uint64
        L.orc_fm_find.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_seeding.restype = C.c_void_p
        L.orc_seeding.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                  C.c_uint64]
        L.orc_seeds_free.argtypes = [C.c_void_p]
        L.orc_seeds_count.restype = C.c_uint64
        L.orc_seeds_count.argtypes = [C.c_void_p]
        L.orc_seeds_get.argtypes = [C.c_void_p, C.c_uint64, u64p, C.POINTER(C.c_int), u64p, u64p]
        L.orc_seeds_all.restype = C.c_int
        L.orc_seeds_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p), u64p, u64p, u64p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_reads(reads: Sequence[str]) -> Tuple[bytes, np.ndarray]:
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if len(reads):
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    return ''.join(reads).encode(), off


class OracleGraph:
    """CSR arrays -> orc_graph.  node ranks are positions in `node_id`."""

    def __init__(self, node_id, label_off, labels: bytes, edge_off, edge_to):
        self.node_id = np.ascontiguousarray(node_id, dtype=np.uint64)
        self.label_off = np.ascontiguousarray(label_off, dtype=np.uint64)
        self.labels = bytes(labels)
        self.edge_off = np.ascontiguousarray(edge_off, dtype=np.uint64)
        self.edge_to = np.ascontiguousarray(edge_to, dtype=np.uint64)
        self.h = lib().orc_graph_new(len(self.node_id), _ptr(self.node_id), _ptr(self.label_off),
                                     self.labels, _ptr(self.edge_off), _ptr(self.edge_to))

    @classmethod
    def from_brute(cls, g) -> 'OracleGraph':
        rank = {v: i for i, v in enumerate(g.ids)}
        label_off = [0]
        edge_off = [0]
        edge_to: List[int] = []
        for v in g.ids:
            label_off.append(label_off[-1] + len(g.seq[v]))
            edge_to.extend(rank[t] for t in g.out[v])
            edge_off.append(len(edge_to))
        labels = ''.join(g.seq[v] for v in g.ids).encode()
        o = cls(g.ids, label_off, labels, edge_off, edge_to if edge_to else np.zeros(0, np.uint64))
        o.rank = rank
        return o

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.orc_graph_free(self.h)
            self.h = None


class OraclePathIndex:
    def __init__(self, g: OracleGraph, paths: Sequence[Sequence[int]],
                 ext_sa: Optional[np.ndarray] = None, left: Optional[Sequence[int]] = None,
                 right: Optional[Sequence[int]] = None):
        """`paths`: lists of node RANKS; `left`/`right`: Path trims (0 = whole node)."""
        self.g = g
        off = np.zeros(len(paths) + 1, dtype=np.uint64)
        if len(paths):
            off[1:] = np.cumsum([len(p) for p in paths], dtype=np.uint64)
        nodes = np.ascontiguousarray(
            np.concatenate([np.asarray(p, dtype=np.uint64) for p in paths])
            if len(paths) else np.zeros(0, np.uint64), dtype=np.uint64)
        st = C.c_int(0)
        sa = None if ext_sa is None else np.ascontiguousarray(ext_sa, dtype=np.uint32)
        lf = None if left is None else np.ascontiguousarray(left, dtype=np.uint64)
        rt = None if right is None else np.ascontiguousarray(right, dtype=np.uint64)
        self.h = lib().orc_pindex_build(g.h, len(paths), _ptr(off), _ptr(nodes), _ptr(lf),
                                        _ptr(rt), _ptr(sa), C.byref(st))
        if st.value == -1:
            raise ValueError('supplied suffix array failed verification')
        if st.value != 0:
            raise ValueError('oracle path index build failed: %d' % st.value)

    def text(self) -> np.ndarray:
        """The reversed, '$'-joined, 0-terminated text as oracle symbol codes."""
        n = lib().orc_pindex_textlen(self.h)
        buf = (C.c_uint8 * n).from_address(lib().orc_pindex_text(self.h))
        return np.frombuffer(buf, dtype=np.uint8).copy()

    def position(self, sid: int, rev_off: int, k: int = 1) -> Tuple[int, int]:
        """(string id, offset of a length-k occurrence in the REVERSED string) ->
        (node id, node offset)."""
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_pindex_position(self.h, sid, rev_off, k, C.byref(a), C.byref(b))
        return a.value, b.value

    def strset_position(self, pos: int) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_strset_position(self.h, pos, C.byref(a), C.byref(b))
        return a.value, b.value

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.orc_pindex_free(self.h)
            self.h = None


def seeds_all(g: OracleGraph, pindex: Optional[OraclePathIndex], reads_bases: bytes,
              read_off: np.ndarray, k: int, step: int, loci_node: np.ndarray,
              loci_off: np.ndarray, rec_offset: int = 0, gocc_thr: int = 0, threads: int = 1,
              phases: int = 3, want_stats: bool = False):
    """Run the restated SeedFinder::seeds_all.  Returns hits as an (n,4) uint64 array in
    emission order: (node_id, node_offset, read_id, read_offset)."""
    L = lib()
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    s = L.orc_seeding(reads_bases, _ptr(read_off), len(read_off) - 1, k, step, rec_offset)
    if not s:
        raise ValueError('bad seed length')
    loci_node = np.ascontiguousarray(loci_node, dtype=np.uint64)
    loci_off = np.ascontiguousarray(loci_off, dtype=np.uint64)
    hits = C.c_void_p()
    n = C.c_uint64()
    n_on = C.c_uint64()
    gd = C.c_uint64()
    L.orc_seeds_all(g.h, pindex.h if pindex is not None else None, s, _ptr(loci_node),
                    _ptr(loci_off), len(loci_node), gocc_thr, threads, phases,
                    C.byref(hits), C.byref(n), C.byref(n_on), C.byref(gd))
    n_seeds = L.orc_seeds_count(s)
    if n.value:
        buf = (C.c_uint64 * (4 * n.value)).from_address(hits.value)
        out = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
    else:
        out = np.zeros((0, 4), dtype=np.uint64)
    L.orc_free(hits)
    L.orc_seeds_free(s)
    if want_stats:
        return out, dict(n_seeds=n_seeds, n_on=n_on.value, n_godown=gd.value)
    return out


def sort_unique(hits: np.ndarray) -> np.ndarray:
    if len(hits) == 0:
        return hits.reshape(0, 4)
    return np.unique(hits, axis=0)


def seeding(reads: Sequence[str], k: int, step: int, rec_offset: int = 0):
    """[(key, has_n, read_id, read_off)] from the C restatement of seeding()."""
    L = lib()
    bases, off = pack_reads(reads)
    s = L.orc_seeding(bases, _ptr(off), len(reads), k, step, rec_offset)
    out = []
    key, rid, ro, hn = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
    for i in range(L.orc_seeds_count(s)):
        L.orc_seeds_get(s, i, C.byref(key), C.byref(hn), C.byref(rid), C.byref(ro))
        out.append((key.value, hn.value, rid.value, ro.value))
    L.orc_seeds_free(s)
    return out


class FMText:
    """FM-index over a raw DNA/'$' text -- test hook for the reference's FM known answers."""

    def __init__(self, text: str):
        self.h = lib().orc_fm_from_text(text.encode(), len(text))

    def find(self, pat: str) -> List[int]:
        cap = 1 << 16
        out = (C.c_uint64 * cap)()
        n = lib().orc_fm_find(self.h, pat.encode(), len(pat), out, cap)
        return sorted(out[i] for i in range(min(n, cap)))

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.orc_pindex_free(self.h)
            self.h = None
