"""One host index for all the GPUs of a node, one process per GPU.

The C ABI takes the graph and the index as VIEWS -- plain host arrays and a few scalars (include/psi_gpu.h:
psigpu_graph_view, psigpu_index_view; psigpu_load_graph / psigpu_load_index copy them to the device).  A view does not
care who owns its arrays, so the rank that built the index writes every array once into a directory of a memory-backed
file system (/dev/shm) and the other ranks map the files read-only and hand the mappings to their own context: the
whole-genome index (tens of GB) exists once in host memory however many processes upload it, and no rank but the
builder runs the host-side index construction.  (The reference is one process with one index: SeedFinder,
include/psi/seed_finder.hpp:1747-1752; read batches shard by contiguous ranges, sequence.hpp:1277-1282.)

    rank 0:   export_views(dir, graph, pindex, extra={...})
    others:   g, px, extra = import_views(dir)        # objects with a .view, good for SeedFinder(g, k) / set_path_index(px)
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import GraphView, IndexView, NO_FTAB


def _dump(dirpath: str, name: str, ptr: Optional[int], n: int, dtype, man: Dict) -> None:
    if not ptr or n == 0:
        man[name] = None
        return
    dt = np.dtype(dtype)
    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dt, count=n)
    path = os.path.join(dirpath, name + '.bin')
    mm = np.lib.format.open_memmap(path, mode='w+', dtype=dt, shape=(n,))
    # (big arrays in pieces: a single 40-GB assignment holds the GIL and a temporary)
    step = 1 << 28
    for a in range(0, n, step):
        mm[a:a + step] = arr[a:a + step]
    mm.flush()
    del mm
    man[name] = {'n': int(n), 'dtype': dt.str}


def _index_arrays(v: IndexView):
    """(field, length, dtype) of every array a psigpu_index_view part points at (include/psi_gpu.h)."""
    has_ftab = v.ftab_len not in (0, NO_FTAB) and v.ftab
    return [('bwt_blocks', v.n_blocks * 16 if v.bwt_blocks else 0, np.uint32),       # 64-byte blocks
            ('sa_samples', v.n_samples, np.uint32), ('exc_row', v.n_exc, np.uint32), ('exc_sa', v.n_exc, np.uint32),
            ('ftab', 2 * (4 ** v.ftab_len) if has_ftab else 0, np.uint32),
            ('text4', v.text_len // 16 + 2 if v.text4 else 0, np.uint64),
            ('seg_start', v.n_segs + 1, np.uint32), ('seg_node', v.n_segs, np.uint32), ('seg_noff', v.n_segs, np.uint32),
            ('seg_dir', v.n_dir, np.uint32), ('loci_node', v.n_loci, np.uint32), ('loci_off', v.n_loci, np.uint32),
            ('exc_super', ((v.n_blocks - 1) >> v.exc_shift) + 1 if (v.exc_super and v.n_blocks) else 0, np.uint32)]


_INDEX_SCALARS = ['seed_len', 'sa_rate', 'context', 'n_paths', 'text_len', 'n_blocks', 'n_samples', 'n_exc', 'ftab_len',
                  'exc_shift', 'n_segs', 'n_dir', 'n_loci']


def export_views(dirpath: str, graph, pindex, extra: Optional[dict] = None) -> None:
    """Write the arrays behind graph.view and pindex.view (every part) into `dirpath` + a manifest, last."""
    os.makedirs(dirpath, exist_ok=True)
    man: Dict = {'extra': extra or {}}
    gv = graph.view
    n = int(gv.n_nodes)
    lo = np.frombuffer((C.c_uint64 * (n + 1)).from_address(gv.label_off), dtype=np.uint64) if n else np.zeros(1, np.uint64)
    eo = np.frombuffer((C.c_uint64 * (n + 1)).from_address(gv.edge_off), dtype=np.uint64) if n else np.zeros(1, np.uint64)
    gman: Dict = {'n_nodes': n}
    _dump(dirpath, 'g_node_id', gv.node_id, n, np.uint64, gman)
    _dump(dirpath, 'g_label_off', gv.label_off, n + 1, np.uint64, gman)
    _dump(dirpath, 'g_labels', gv.labels, int(lo[n]), np.uint8, gman)
    _dump(dirpath, 'g_edge_off', gv.edge_off, n + 1, np.uint64, gman)
    _dump(dirpath, 'g_edge_to', gv.edge_to, int(eo[n]), np.uint32, gman)
    man['graph'] = gman
    parts = [pindex.view] + list(pindex.more_parts())
    man['parts'] = []
    for i, v in enumerate(parts):
        pm: Dict = {s: int(getattr(v, s)) for s in _INDEX_SCALARS}
        pm['C'] = [int(x) for x in v.C]
        for name, ln, dt in _index_arrays(v):
            _dump(dirpath, 'x%d_%s' % (i, name), getattr(v, name), int(ln), dt, pm)
        man['parts'].append(pm)
    tmp = os.path.join(dirpath, 'manifest.json.tmp')
    with open(tmp, 'w') as fh:
        json.dump(man, fh)
    os.replace(tmp, os.path.join(dirpath, 'manifest.json'))          # (readers wait for this name)


class SharedGraph:
    """A psigpu_graph_view over mapped arrays (what SeedFinder needs of a Graph: .view, .n_nodes, .n_edges)."""

    def __init__(self, dirpath: str, man: Dict):
        self._keep = {}
        self.view = GraphView()
        self.view.n_nodes = man['n_nodes']
        for f in ('node_id', 'label_off', 'labels', 'edge_off', 'edge_to'):
            a = _map(dirpath, 'g_' + f, man.get('g_' + f))
            self._keep[f] = a
            setattr(self.view, f, a.ctypes.data if a is not None else None)
        self.n_nodes = int(man['n_nodes'])
        self.n_edges = int(man['g_edge_to']['n']) if man.get('g_edge_to') else 0

    def array(self, name: str) -> Optional[np.ndarray]:
        return self._keep.get(name)


class SharedIndex:
    """A psigpu_index_view (all parts) over mapped arrays: good for SeedFinder.set_path_index."""

    def __init__(self, dirpath: str, parts):
        self._keep = []
        views = []
        for i, pm in enumerate(parts):
            v = IndexView()
            for s in _INDEX_SCALARS:
                setattr(v, s, pm[s])
            for c in range(4):
                v.C[c] = pm['C'][c]
            for name in ('bwt_blocks', 'sa_samples', 'exc_row', 'exc_sa', 'ftab', 'text4', 'seg_start', 'seg_node', 'seg_noff',
                         'seg_dir', 'loci_node', 'loci_off', 'exc_super'):
                a = _map(dirpath, 'x%d_%s' % (i, name), pm.get('x%d_%s' % (i, name)))
                self._keep.append(a)
                setattr(v, name, a.ctypes.data if a is not None else None)
            v.n_more_parts = 0
            v.reserved2 = 0
            v.more_parts = None
            views.append(v)
        self.view = views[0]
        if len(views) > 1:
            self._more = (IndexView * (len(views) - 1))(*views[1:])
            self.view.n_more_parts = len(views) - 1
            self.view.more_parts = C.addressof(self._more)
        self.text_len = int(self.view.text_len)

    def more_parts(self):
        return [self._more[i] for i in range(int(self.view.n_more_parts))] if int(self.view.n_more_parts) else []


def _map(dirpath: str, name: str, ent) -> Optional[np.ndarray]:
    if not ent:
        return None
    return np.load(os.path.join(dirpath, name + '.bin'), mmap_mode='r')


def import_views(dirpath: str) -> Tuple[SharedGraph, SharedIndex, dict]:
    with open(os.path.join(dirpath, 'manifest.json')) as fh:
        man = json.load(fh)
    return SharedGraph(dirpath, man['graph']), SharedIndex(dirpath, man['parts']), man.get('extra', {})
