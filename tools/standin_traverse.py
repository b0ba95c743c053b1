#!/usr/bin/env python3
"""Traverse mode (the reference's scheme) on the clustered-variation stand-in against the uniform one: per-kernel device times of a
1 M-read step, the number of prefix walks the per-chunk sweep streams, k-walks completed.  Needs a GPU.
    python tools/standin_traverse.py [uniform|clustered] [mode]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import psi_amd
from psi_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else 'clustered'
mode = sys.argv[2] if len(sys.argv) > 2 else 'traverse'
k = 21
sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11, cluster_frac=0.2 if which == 'clustered' else 0.0)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
bases, off = synth.sim_reads_snv(sg, 1_000_000, 150, seed=13)
px = psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0)
os.environ['PSIGPU_TRACE'] = '1'
f = psi_amd.SeedFinder(g, k, mode=mode)
f.set_path_index(px)
f.prepare()
os.environ.pop('PSIGPU_TRACE')
d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
stream = torch.cuda.current_stream().cuda_stream
call = lambda: f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), 1_000_000, len(bases), step=k, stream=stream, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)   # noqa: E731
for _ in range(3):
    call()
ms = {}
for _ in range(5):
    call()
    c = f.counters()
    for n in ('ms_total', 'ms_probe', 'ms_table', 'ms_traverse', 'ms_pack', 'ms_search', 'ms_locate'):
        ms[n] = ms.get(n, 0.0) + c[n] / 5
import time
pk = psi_amd.PackedReads(bases, off, pinned=True, threads=8)
ts = []
for i in range(8):
    t = time.perf_counter()
    h = f.seeds_all_packed(pk, step=k, sort_unique=True)
    ts.append((time.perf_counter() - t) * 1e3)
host_ms = float(np.median(ts[3:]))
host_dev_ms = float(f.counters()['ms_total'])
print(json.dumps({'host_entry_ms_incl_python_copy': host_ms, 'host_entry_device_ms': host_dev_ms, 'host_entry_hits': int(len(h))}))
print(json.dumps({'stand_in': which, 'mode': mode, 'starting_loci': int(px.view.n_loci), 'ms': {a: round(b, 4) for a, b in ms.items()},
                  'n_kpaths': int(c['n_kpaths']), 'n_spilled': int(c['n_spilled']), 'traverse_launches': int(c['traverse_launches']),
                  'n_loci_traversed': int(c['n_loci_traversed']), 'hits': int(c['n_hits'])}))
