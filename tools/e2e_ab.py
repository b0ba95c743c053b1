"""Host entry (psigpu_find_seeds, sort-unique) on the bench workload, alternating A/B settings on ONE box (the boxes of
the pool differ by 10-15 % in link rate): 16-byte wire records against 32-byte records, pinned and pageable reads."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, psi_amd
from psi_amd import synth
sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
px = psi_amd.PathIndex.build(g, 21, 1, rng_seed=1, device=0)
f = psi_amd.SeedFinder(g, 21, device=0); f.set_path_index(px); f.prepare()
batches = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in batches]
page = [(np.ascontiguousarray(b), np.ascontiguousarray(o.astype(np.uint64))) for b, o in batches]
L = psi_amd.lib(); hits = psi_amd.Hits()
def run(src, reps=10):
    calls = [(f.ctx, psi_amd._ptr(a), psi_amd._ptr(b), 1_000_000, 21, 21, 0, psi_amd.ALL | psi_amd.SORT_UNIQUE, C.byref(hits)) for a, b in src]
    for i in range(3):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0; L.psigpu_free_hits(C.byref(hits))
    t = time.perf_counter()
    for i in range(reps):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0; L.psigpu_free_hits(C.byref(hits))
    return (time.perf_counter() - t) / reps * 1e3
for rnd in range(3):
    for wire in ('1', '0'):          # 16-byte wire records widened on the host / 32-byte records over the link
            if wire == '0': os.environ['PSIGPU_NO_WIRE16'] = '1'
            else: os.environ.pop('PSIGPU_NO_WIRE16', None)
            print('round', rnd, 'wire16', wire, 'pinned %.2f ms' % run([(p[0].array, p[1].array) for p in pin]), 'pageable %.2f ms' % run(page, 6), flush=True)
