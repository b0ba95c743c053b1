#!/usr/bin/env python3
"""Patched paths (psikt's default indexing mode) against full paths at BASELINE.json configs[1] size:
the chr22-like graph, n walks per region, 1 M x 150 bp reads, k = 21.  Per variant: index build time,
indexed text, paths / patches, starting loci, ms per step (device-resident), and that the sort-unique
hit sets of all variants are identical.  One JSON line per variant.  Needs a GPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import psi_amd
    from psi_amd import synth
    sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 1_000_000, 150, seed=13)
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ref = None
    for n, patched, ctx in ((1, False, 0), (4, False, 0), (4, True, 21), (8, False, 0), (8, True, 21), (8, True, 42)):
        t = time.time()
        px = psi_amd.PathIndex.build(g, 21, n, rng_seed=1, device=0, patched=patched, context=ctx)
        t_ix = time.time() - t
        f = psi_amd.SeedFinder(g, 21, device=0)
        f.set_path_index(px)
        t = time.time()
        f.prepare()
        t_prep = time.time() - t
        for _ in range(3):
            f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 1_000_000, len(bases), stream=stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 1_000_000, len(bases), stream=stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        c = f.counters()
        ptr, nh = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 1_000_000, len(bases),
                                     flags=psi_amd.ALL | psi_amd.SORT_UNIQUE, stream=stream)
        su = f.copy_hits(ptr, nh)
        if ref is None:
            ref = su
        same = bool(su.shape == ref.shape and (su == ref).all())
        tr = px.trims()
        print(json.dumps({'walks_per_region': n, 'patched': patched, 'context': ctx or (21 if patched else 0),
                          'index_build_s': t_ix, 'tables_s': t_prep, 'paths_in_index': len(tr),
                          'trimmed_paths': sum(1 for a in tr if a != (0, 0)), 'text_len': int(px.text_len),
                          'starting_loci': int(px.view.n_loci), 'ms_per_step': dt * 1e3, 'raw_hits_per_step': int(c['n_hits']),
                          'sort_unique_hits': int(nh), 'same_hit_set_as_first_variant': same}), flush=True)
        f.close()
        del px


if __name__ == '__main__':
    main()
