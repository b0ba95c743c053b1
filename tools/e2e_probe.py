import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psi_amd
from psi_amd import synth
sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
px = psi_amd.PathIndex.build(g, 21, 1, rng_seed=1, device=0)
f = psi_amd.SeedFinder(g, 21, device=0); f.set_path_index(px); f.prepare()
batches = [synth.sim_reads_snv(sg, 1_000_000, 150, seed=13 + 100 * b) for b in range(2)]
L = psi_amd.lib(); hits = psi_amd.Hits()
def measure(tag, src, flags, reps=10):
    calls = [(f.ctx, psi_amd._ptr(b), psi_amd._ptr(o), 1_000_000, 21, 21, 0, flags, C.byref(hits)) for b, o in src]
    for i in range(2):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0; L.psigpu_free_hits(C.byref(hits))
    t = time.perf_counter()
    for i in range(reps):
        assert L.psigpu_find_seeds(*calls[i % 2]) == 0; L.psigpu_free_hits(C.byref(hits))
    print('%-40s %.2f ms' % (tag, (time.perf_counter() - t) / reps * 1e3), flush=True)
pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in batches]
ps = [(p[0].array, p[1].array) for p in pin]
pg = [(np.ascontiguousarray(b), np.ascontiguousarray(o.astype(np.uint64))) for b, o in batches]
SU = psi_amd.ALL | psi_amd.SORT_UNIQUE
which = sys.argv[1] if len(sys.argv) > 1 else 'a'
if which == 'b':   # what bench does before: device-resident loop with torch tensors + other finders
    dev = [(torch.from_numpy(b).cuda(), torch.from_numpy(o.astype(np.int64)).cuda(), len(b)) for b, o in batches]
    for i in range(50):
        f.seeds_all_device(dev[i%2][0].data_ptr(), dev[i%2][1].data_ptr(), 1_000_000, dev[i%2][2], step=21, stream=torch.cuda.current_stream().cuda_stream)
    for m in ('locus-table', 'traverse'):
        f2 = psi_amd.SeedFinder(g, 21, device=0, mode=m); f2.set_path_index(px); f2.prepare()
        for i in range(5):
            f2.seeds_all_device(dev[0][0].data_ptr(), dev[0][1].data_ptr(), 1_000_000, dev[0][2], step=21, stream=torch.cuda.current_stream().cuda_stream)
        f2.close()
measure('pinned sort-unique', ps, SU)
measure('pinned raw', ps, psi_amd.ALL)
measure('pageable sort-unique', pg, SU)
measure('pinned sort-unique again', ps, SU)
os.environ['PSIGPU_TRACE'] = '1'
L.psigpu_find_seeds(f.ctx, psi_amd._ptr(ps[0][0]), psi_amd._ptr(ps[0][1]), 1_000_000, 21, 21, 0, SU, C.byref(hits)); L.psigpu_free_hits(C.byref(hits))
