#!/usr/bin/env python3
"""The device index build against the host build, array by array, over the random graphs of the fuzz campaign --
meant to run in several processes that share one GPU (tools/build_par.sh): what the device builder hands back must
not depend on who else is on the device.  Each index is built on the device `REPS` times.
`python tools/build_stress.py FIRST LAST [SECONDS]`."""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

REPS = int(os.environ.get('BUILD_STRESS_REPS', '3'))


def arrays_of(px):
    from psi_amd import shared
    out = {}
    for i, v in enumerate([px.view] + list(px.more_parts())):
        for s in shared._INDEX_SCALARS:
            out['%d.%s' % (i, s)] = np.array([int(getattr(v, s))])
        out['%d.C' % i] = np.array([int(x) for x in v.C])
        for name, n, dt in shared._index_arrays(v):
            ptr = getattr(v, name)
            if not ptr or not n:
                out['%d.%s' % (i, name)] = np.zeros(0, dt)
                continue
            dt = np.dtype(dt)
            buf = (C.c_uint8 * (int(n) * dt.itemsize)).from_address(ptr)
            out['%d.%s' % (i, name)] = np.frombuffer(buf, dtype=dt, count=int(n)).copy()
    return out


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    limit = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    import psi_amd
    import test_gpu_parity as T
    t0 = time.time()
    n_built = 0
    for seed in range(first, last):
        if time.time() - t0 > limit:
            break
        g, _reads = T._random_graph(seed)
        rng = random.Random(seed)
        rank = {v: i for i, v in enumerate(g.ids)}
        label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
        labels = ''.join(g.seq[v] for v in g.ids).encode()
        edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
        edge_to = [rank[t] for v in g.ids for t in g.out[v]]
        pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to, paths=[[rank[v] for v in g.paths[0][1]]])
        for _ in range(2):
            k = rng.choice([3, 8, 12, 13, 16, 21, 25, 31, 31, 32, 40])
            npaths = rng.choice([0, 1, 1, 2, 3, 5])
            patched = npaths > 1 and rng.random() < 0.6
            bargs = dict(rng_seed=seed, sa_rate=rng.choice([1, 1, 1, 2, 8]), ftab_len=rng.choice([0, 0, 4, psi_amd.NO_FTAB]),
                         patched=patched, context=rng.choice([0, k, k + 1, k + 7]) if patched else 0,
                         step=rng.choice([1, 1, 2, 3]))
            want = arrays_of(psi_amd.PathIndex.build(pg, k, npaths, device=None, **bargs))
            for rep in range(REPS):
                got = arrays_of(psi_amd.PathIndex.build(pg, k, npaths, device=0, **bargs))
                n_built += 1
                for name in want:
                    a, b = want[name], got[name]
                    if a.shape != b.shape or not bool((a == b).all()):
                        where = np.nonzero(a != b)[0][:8].tolist() if a.shape == b.shape else None
                        print('BUILD MISMATCH seed', seed, 'k', k, 'npaths', npaths, bargs, 'rep', rep, 'array', name,
                              'shapes', a.shape, b.shape, 'at', where,
                              'host', a[where].tolist() if where else None, 'device', b[where].tolist() if where else None, flush=True)
                        # once more, alone in time: was it the moment or the input?
                        again = arrays_of(psi_amd.PathIndex.build(pg, k, npaths, device=0, **bargs))
                        print(' built again:', 'equal to the host' if all(
                            want[n].shape == again[n].shape and bool((want[n] == again[n]).all()) for n in want) else 'WRONG AGAIN',
                            flush=True)
                        sys.exit(1)
    print('ok: seeds %d..%d, %d device builds in %.0f s' % (first, seed, n_built, time.time() - t0))


if __name__ == '__main__':
    main()
