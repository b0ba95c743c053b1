#!/usr/bin/env python3
"""One fuzz seed, one (k, step, npaths) configuration, every query mode and a few switches: which of them lose or add records
against the brute-force definition.  python tools/repro_seed.py SEED K STEP NPATHS"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import psi_amd
from oracle import brute
import test_gpu_parity as T

seed, k, step, npaths = (int(x) for x in sys.argv[1:5])
g, reads = T._random_graph(seed)
rank = {v: i for i, v in enumerate(g.ids)}
label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
labels = ''.join(g.seq[v] for v in g.ids).encode()
edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
edge_to = [rank[t] for v in g.ids for t in g.out[v]]
pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to, paths=[[rank[v] for v in g.paths[0][1]]])
print('nodes', len(g.ids), 'reads', [len(r) for r in reads])
for kk in (k - 1, k, k + 1):
    want = np.array(brute.hit_set(g, [r.upper() for r in reads], kk, step), dtype=np.uint64).reshape(-1, 4)
    for sa_rate, ftab in ((1, 0), (1, 4), (1, psi_amd.NO_FTAB), (2, 0)):
        for dev in (None, 0):
            px = psi_amd.PathIndex.build(pg, kk, npaths, rng_seed=seed, sa_rate=sa_rate, ftab_len=ftab, device=dev)
            for mode in ('kmer-table', 'locus-table', 'traverse'):
                for opt in ((), (('no_pfx_roots', 1),), (('wire', 32),), (('no_lookahead', 1),)):
                    f = psi_amd.SeedFinder(pg, kk, mode=mode)
                    for n, v in opt:
                        f.set_option(n, v)
                    f.set_path_index(px)
                    got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
                    on = psi_amd.sort_unique(f.seeds_on_paths(reads, step=step))
                    off = psi_amd.sort_unique(f.seeds_off_paths(reads, step=step))
                    ok = got.shape == want.shape and bool((got == want).all())
                    if not ok or (kk == k and opt == () and dev is None):
                        a, b = set(map(tuple, got.tolist())), set(map(tuple, want.tolist()))
                        print('k', kk, 'sa', sa_rate, 'ftab', ftab, 'dev', dev, mode, opt, 'OK' if ok else 'WRONG', len(got), len(want),
                              'on', len(on), 'off', len(off), 'missing', sorted(b - a)[:3], 'extra', sorted(a - b)[:3], flush=True)
                    f.close()
print('paths', [[g.ids[r] for r in p] for p in px.paths()][:2])
print('node 7:', g.seq.get(7), 'out', g.out.get(7))
print('read 4:', reads[4] if len(reads) > 4 else None)
