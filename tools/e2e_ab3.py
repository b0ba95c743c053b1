#!/usr/bin/env python3
"""Host entry on the bench workload, variants alternated call by call in one process (the boxes are shared: ten-call
averages of one variant after the other differ by more than the variants do): median and minimum per variant."""
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
pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in real]
L = psi_amd.lib()
hits = psi_amd.Hits()
calls = [(f.ctx, psi_amd._ptr(p[0].array), psi_amd._ptr(p[1].array), 1_000_000, 21, 21, 0, psi_amd.ALL | psi_amd.SORT_UNIQUE, C.byref(hits)) for p in pin]
variants = [('run-ahead, tapered tail', {}), ('run-ahead', {'PSIGPU_NO_TAPER': '1'}), ('two slots', {'PSIGPU_NO_AHEAD': '1'})]
if len(sys.argv) > 1:
    # e.g.  "cached io,PSIGPU_CACHED_IO=1"  "default"
    variants = [(a.split(',')[0], dict(x.split('=') for x in a.split(',')[1:] if '=' in x)) for a in sys.argv[1:]]
times = {name: [] for name, _ in variants}
for rnd in range(int(os.environ.get('ROUNDS', '40')) + 3):
    for name, env in variants:
        os.environ.update(env)
        t = time.perf_counter()
        assert L.psigpu_find_seeds(*calls[rnd % 2]) == 0
        dt = time.perf_counter() - t
        L.psigpu_free_hits(C.byref(hits))
        for e in env:
            os.environ.pop(e)
        if rnd >= 3:
            times[name].append(dt * 1e3)
for name, _ in variants:
    a = np.array(times[name])
    print('%-28s median %.2f ms, min %.2f, mean %.2f, p90 %.2f  (%d calls)' % (name, np.median(a), a.min(), a.mean(), np.percentile(a, 90), len(a)), flush=True)
