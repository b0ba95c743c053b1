#!/usr/bin/env python3
"""Host entry point (psigpu_find_seeds: H2D + kernels + device sort-unique + D2H) on the bench
workload for several sub-batch sizes; prints ms per 1 M-read chunk.  Needs a GPU."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import psi_amd
from psi_amd import synth

sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
px = psi_amd.PathIndex.build(g, 21, 1, rng_seed=1, device=0)
f = psi_amd.SeedFinder(g, 21, device=0)
f.set_path_index(px)
f.prepare()
batches = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in batches]
L = psi_amd.lib()
hits = psi_amd.Hits()
for flags, name in ((psi_amd.ALL | psi_amd.SORT_UNIQUE, 'sort-unique'), (psi_amd.ALL, 'raw')):
    for mb in (4, 8, 16, 32):
        os.environ['PSIGPU_SUB_BYTES'] = str(mb << 20)
        calls = [(f.ctx, psi_amd._ptr(p[0].array), psi_amd._ptr(p[1].array), 1_000_000, 21, 21, 0, flags, C.byref(hits)) for p in pin]
        for i in range(3):
            assert L.psigpu_find_seeds(*calls[i % 2]) == 0
            L.psigpu_free_hits(C.byref(hits))
        t = time.perf_counter()
        for i in range(10):
            assert L.psigpu_find_seeds(*calls[i % 2]) == 0
            n = hits.n
            L.psigpu_free_hits(C.byref(hits))
        dt = (time.perf_counter() - t) / 10
        if os.environ.get('E2E_TRACE'):
            os.environ['PSIGPU_TRACE'] = '1'
            L.psigpu_find_seeds(*calls[0]); L.psigpu_free_hits(C.byref(hits))
            del os.environ['PSIGPU_TRACE']
        c = f.counters()
        print('%-11s sub-batch %3d MiB: %.2f ms / chunk, %d hits, device %.2f ms (sort %.2f)' % (name, mb, dt * 1e3, n, c['ms_total'], c['ms_sort']), flush=True)
