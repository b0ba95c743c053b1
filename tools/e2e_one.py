#!/usr/bin/env python3
"""One traced call of the host entry on the bench workload after a few untraced ones (PSIGPU_TRACE timeline on stderr),
with the wall time of every call as the caller sees it."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psi_amd
from psi_amd import synth
sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
px = psi_amd.PathIndex.build(g, 21, 1, rng_seed=1, device=0)
f = psi_amd.SeedFinder(g, 21, device=0)
f.set_path_index(px)
f.prepare()
real = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in real]
L = psi_amd.lib()
hits = psi_amd.Hits()
calls = [(f.ctx, psi_amd._ptr(p[0].array), psi_amd._ptr(p[1].array), 1_000_000, 21, 21, 0, psi_amd.ALL | psi_amd.SORT_UNIQUE, C.byref(hits)) for p in pin]
for i in range(12):
    if i >= 8:
        os.environ['PSIGPU_TRACE'] = '1'
    t = time.perf_counter()
    assert L.psigpu_find_seeds(*calls[i % 2]) == 0
    t1 = time.perf_counter()
    L.psigpu_free_hits(C.byref(hits))
    t2 = time.perf_counter()
    print('call %d: %.3f ms, free %.3f ms' % (i, (t1 - t) * 1e3, (t2 - t1) * 1e3), file=sys.stderr, flush=True)
