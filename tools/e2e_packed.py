#!/usr/bin/env python3
"""The packed host entry (psigpu_find_seeds_packed: H2D of 2-bit reads + kernels + device sort-unique + D2H of 8-byte wire
records, widened on the host) on the bench workload: median / min ms per 1 M-read chunk for several sub-batch sizes and
wire formats, and one traced call (E2E_TRACE=1: per-sub-batch host timeline on stderr).  Needs a GPU."""
import ctypes as C
import json
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
if os.environ.get('E2E_NO_NUMA'):
    f.set_option('no_numa', 1)
if os.environ.get('E2E_ONE_OUT_ENGINE'):
    f.set_option('one_out_engine', 1)
if os.environ.get('E2E_WIDEN_THREADS'):
    f.set_option('widen_threads', int(os.environ['E2E_WIDEN_THREADS']))
f.set_path_index(px)
f.prepare()
batches = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
packed = [psi_amd.PackedReads(b, o, pinned=True, threads=8) for b, o in batches]
L = psi_amd.lib()
hits = psi_amd.Hits()
flags = psi_amd.ALL | psi_amd.SORT_UNIQUE
calls = [(f.ctx, psi_amd._ptr(p.words), psi_amd._ptr(p.mask), psi_amd._ptr(p.off), 1_000_000, 21, 21, 0, flags, C.byref(hits)) for p in packed]
out = []


def run(label, reps=15):
    for i in range(3):
        assert L.psigpu_find_seeds_packed(*calls[i % 2]) == 0
        L.psigpu_free_hits(C.byref(hits))
    ts = []
    for i in range(reps):
        t = time.perf_counter()
        assert L.psigpu_find_seeds_packed(*calls[i % 2]) == 0
        n = hits.n
        L.psigpu_free_hits(C.byref(hits))
        ts.append((time.perf_counter() - t) * 1e3)
    c = f.counters()
    r = {'label': label, 'median_ms': float(np.median(ts)), 'min_ms': min(ts), 'max_ms': max(ts), 'hits': int(n),
         'device_ms': float(c['ms_total']), 'wire': int(c['wire_bytes_per_hit']), 'lookahead_subbatches': int(c['lookahead_subbatches']),
         'lookahead_fallbacks': int(c['lookahead_fallbacks'])}
    out.append(r)
    print(json.dumps(r), flush=True)


for mb in (() if os.environ.get('E2E_QUICK') else (4, 8, 16, 32, 64, 160)):
    f.set_option('sub_bytes', mb << 20)
    run('sub %d Mi bases' % mb)
f.set_option('sub_bytes', 0)
for w in (() if os.environ.get('E2E_QUICK') else (16, 32)):
    f.set_option('wire', w)
    run('wire %d' % w)
f.set_option('wire', 0)
f.set_option('no_lookahead', 1)
run('one sub-batch at a time (no lookahead)')
f.set_option('no_lookahead', 0)
run('two sub-batches in flight (default)')
f.set_option('no_ahead', 1)
run('two slots (no transfers queued ahead)')
f.set_option('no_ahead', 0)
# the two loops ALTERNATED call by call (the boxes are shared: a variant measured a second later meets another machine)
ab = {0: [], 1: []}
for i in range(240):
    v = i & 1
    f.set_option('no_lookahead', v)
    t = time.perf_counter()
    assert L.psigpu_find_seeds_packed(*calls[(i >> 1) % 2]) == 0
    L.psigpu_free_hits(C.byref(hits))
    if i >= 40:
        ab[v].append((time.perf_counter() - t) * 1e3)
f.set_option('no_lookahead', 0)
print(json.dumps({'label': 'alternated, 100 calls each', 'two_in_flight_median_ms': float(np.median(ab[0])), 'two_in_flight_min_ms': min(ab[0]),
                  'one_at_a_time_median_ms': float(np.median(ab[1])), 'one_at_a_time_min_ms': min(ab[1])}), flush=True)
# the wire formats alternated the same way: packed records (5-7 bytes, round 5) against 8-byte keys
abw = {0: [], 8: []}
for i in range(240):
    w = 8 if i & 1 else 0
    f.set_option('wire', w)
    t = time.perf_counter()
    assert L.psigpu_find_seeds_packed(*calls[(i >> 1) % 2]) == 0
    L.psigpu_free_hits(C.byref(hits))
    if i >= 40:
        abw[w].append((time.perf_counter() - t) * 1e3)
    if i in (238, 239):
        print(json.dumps({'label': 'wire option %d' % w, 'wire_bytes_per_hit': int(f.counters()['wire_bytes_per_hit']),
                          'lookahead_fallbacks': int(f.counters()['lookahead_fallbacks'])}), flush=True)
f.set_option('wire', 0)
print(json.dumps({'label': 'wire formats alternated, 100 calls each', 'packed_median_ms': float(np.median(abw[0])), 'packed_min_ms': min(abw[0]),
                  'keys8_median_ms': float(np.median(abw[8])), 'keys8_min_ms': min(abw[8])}), flush=True)
if os.environ.get('E2E_ENGINE_AB'):
    # (round 6, review item 1c) the transfers by raw HSA engine copies (default) against HIP stream copies (option no_engine_copy,
    # read when a context makes its pipeline): two finders over one index, alternated call by call
    f2 = psi_amd.SeedFinder(g, 21, device=0)
    f2.set_option('no_engine_copy', 1)
    f2.set_path_index(px)
    f2.prepare()
    calls2 = [(f2.ctx,) + c[1:] for c in calls]
    abe = {0: [], 1: []}
    for i in range(440):
        v = i & 1
        cs = (calls2 if v else calls)[(i >> 1) % 2]
        t = time.perf_counter()
        assert L.psigpu_find_seeds_packed(*cs) == 0
        L.psigpu_free_hits(C.byref(hits))
        if i >= 240:
            abe[v].append((time.perf_counter() - t) * 1e3)
    print(json.dumps({'label': 'engine copies against HIP stream copies, alternated, 100 calls each (after 120 warm-up calls each)',
                      'engine_copy_median_ms': float(np.median(abe[0])), 'engine_copy_min_ms': min(abe[0]),
                      'hip_stream_copy_median_ms': float(np.median(abe[1])), 'hip_stream_copy_min_ms': min(abe[1])}), flush=True)
    f2.close()
if os.environ.get('E2E_TRACE'):
    # PSIGPU_TRACE=2: the timeline of the default (two sub-batches in flight) path; =1: the synchronous loop's
    for tr in ('2', '2', '1'):
        os.environ['PSIGPU_TRACE'] = tr
        t = time.perf_counter()
        L.psigpu_find_seeds_packed(*calls[0]); L.psigpu_free_hits(C.byref(hits))
        print('traced call (PSIGPU_TRACE=%s): %.3f ms' % (tr, (time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)
    del os.environ['PSIGPU_TRACE']
