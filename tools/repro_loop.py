#!/usr/bin/env python3
"""One fuzz configuration again and again -- index built on the device, one finder, the brute-force answer -- for
processes that share a GPU: `python tools/repro_loop.py SEED K STEP NPATHS SECONDS [MODE]`.  On a wrong answer: which
array of the index differs from the host build, and whether a second build / a second finder repeats it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import psi_amd
from oracle import brute
import test_gpu_parity as T
from build_stress import arrays_of

seed, k, step, npaths = (int(x) for x in sys.argv[1:5])
limit = float(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else 'kmer-table'
g, reads = T._random_graph(seed)
rank = {v: i for i, v in enumerate(g.ids)}
label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
labels = ''.join(g.seq[v] for v in g.ids).encode()
edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
edge_to = [rank[t] for v in g.ids for t in g.out[v]]
pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to, paths=[[rank[v] for v in g.paths[0][1]]])
want = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
bargs = dict(rng_seed=seed, sa_rate=1, ftab_len=0)
host = arrays_of(psi_amd.PathIndex.build(pg, k, npaths, device=None, **bargs))


def differs(px):
    got = arrays_of(px)
    return [n for n in host if host[n].shape != got[n].shape or not bool((host[n] == got[n]).all())]


def answer(px, opts=()):
    f = psi_amd.SeedFinder(pg, k, mode=mode)
    for n, v in opts:
        f.set_option(n, v)
    f.set_path_index(px)
    r = psi_amd.sort_unique(f.seeds_all(reads, step=step))
    c = f.counters()
    f.close()
    return bool(r.shape == want.shape and (r == want).all()), r, c


t0 = time.time()
n = 0
while time.time() - t0 < limit:
    px = psi_amd.PathIndex.build(pg, k, npaths, device=0, **bargs)
    ok, r, c = answer(px)
    n += 1
    d = differs(px)
    if not ok or d:
        print('WRONG' if not ok else 'answer ok', 'iteration', n, 'arrays that differ from the host build:', d, flush=True)
        if not ok:
            a, b = set(map(tuple, r.tolist())), set(map(tuple, want.tolist()))
            print(' missing', sorted(b - a)[:4], 'extra', sorted(a - b)[:4], {x: c[x] for x in ('n_loci', 'n_kpaths', 'n_hits_off_path', 'n_hits_on_path')}, flush=True)
        for name in d[:4]:
            w = np.nonzero(host[name] != arrays_of(px)[name])[0][:6].tolist() if host[name].shape == arrays_of(px)[name].shape else None
            print('  ', name, 'at', w, 'host', host[name][w].tolist() if w else host[name].shape,
                  'device', arrays_of(px)[name][w].tolist() if w else arrays_of(px)[name].shape, flush=True)
        print(' same index, new finder:', answer(px)[0], '| without prefix roots:', answer(px, (('no_pfx_roots', 1),))[0], flush=True)
        px2 = psi_amd.PathIndex.build(pg, k, npaths, device=0, **bargs)
        print(' index built again on the device: differs', differs(px2), 'answer', answer(px2)[0], flush=True)
        pxh = psi_amd.PathIndex.build(pg, k, npaths, device=None, **bargs)
        print(' index built on the host: answer', answer(pxh)[0], flush=True)
        sys.exit(1)
print('ok: %d iterations in %.0f s' % (n, time.time() - t0))
