#!/usr/bin/env python3
"""Host entry on the bench workload with reads that hit (haplotype walks) and reads that do not (uniform random
bases: nearly no records go back): how much of the chunk time is the link's two directions getting in each other's way."""
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
real = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
rng = np.random.RandomState(5)
rand = [(np.frombuffer(b'ACGT', np.uint8)[rng.randint(0, 4, size=len(b))].copy(), o) for b, o in real]
L = psi_amd.lib()
hits = psi_amd.Hits()
for name, src in (('reads that hit', real), ('random reads', rand), ('reads that hit', real), ('random reads', rand)):
    pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in src]
    calls = [(f.ctx, psi_amd._ptr(p[0].array), psi_amd._ptr(p[1].array), 1_000_000, 21, 21, 0, psi_amd.ALL | psi_amd.SORT_UNIQUE, C.byref(hits)) for p in pin]
    for i in range(3):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0
        L.psigpu_free_hits(C.byref(hits))
    t = time.perf_counter()
    for i in range(10):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0
        n = hits.n
        L.psigpu_free_hits(C.byref(hits))
    dt = (time.perf_counter() - t) / 10
    print('%-15s %.2f ms / chunk, %d records out, reads in at %.1f GB/s' % (name, dt * 1e3, n, 158e6 / dt / 1e9), flush=True)
    del pin
