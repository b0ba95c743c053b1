#!/usr/bin/env python3
"""SURVEY §8(d) series beside the headline bench line, one JSON line each (profiles/r02_series.jsonl):

  config 2 (chr22-like SNV graph, 1 M x 150 bp, k = 21):  P in {1, 8} x d in {21, 1} x error in {0, 1 %}
  config 5 (HLA-like bubble graph, k = 31, no path index: every locus goes through the traverser)

`python tools/series.py [--quick]`; needs a GPU.  Not bench lines: bench.py measures the headline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(f, bases, off, n_reads, step, steps, torch, np):
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, len(bases), step=step, stream=stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, len(bases), step=step, stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    c = f.counters()
    return {'ms_per_step': dt * 1e3, 'seeds_per_s': c['n_seeds'] / dt, 'hits_per_s': c['n_hits'] / dt,
            'n_seeds': c['n_seeds'], 'n_seeds_on_path': c['n_seeds_on_path'], 'n_hits_on_path': c['n_hits_on_path'],
            'n_hits_off_path': c['n_hits_off_path'], 'n_hits': c['n_hits'], 'n_kpaths': c['n_kpaths'], 'n_loci': c['n_loci'],
            'n_spilled': c['n_spilled'], 'traverse_launches': c['traverse_launches'],
            'kernel_ms': {k: round(v, 3) for k, v in c.items() if k.startswith('ms_')}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true', help='1/10 size (plumbing check)')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--skip-hla', action='store_true')
    args = ap.parse_args()
    import numpy as np
    import torch
    import psi_amd
    from psi_amd import synth

    s = 10 if args.quick else 1
    sg = synth.snv_graph(51_000_000 // s, 1_100_000 // s, n_block=11_000_000 // s, seed=11)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    n_reads = 1_000_000 // s
    reads = {0.0: synth.sim_reads_snv(sg, n_reads, 150, seed=13), 0.01: synth.sim_reads_snv(sg, n_reads, 150, seed=13, sub_rate=0.01)}
    for P in (1, 8):
        t = time.time()
        px = psi_amd.PathIndex.build(g, 21, P, rng_seed=1, device=0)
        build_s = time.time() - t
        f = psi_amd.SeedFinder(g, 21, device=0)
        f.set_path_index(px)
        for d in (21, 1):
            for err in (0.0, 0.01):
                bases, off = reads[err]
                r = run(f, bases, off, n_reads, d, args.steps, torch, np)
                r.update(series='config2', paths=P, step=d, sub_rate=err, k=21, reads=n_reads, text_len=int(px.text_len),
                         index_build_s=round(build_s, 2))
                print(json.dumps(r), flush=True)
        f.close()
    if args.skip_hla:
        return
    t = time.time()
    L = 5_000_000 // s
    node_id, label_off, labels, edge_off, edge_to, ref_path = synth.bubble_graph(L, seed=31)
    gen_s = time.time() - t
    g = psi_amd.Graph.from_csr(node_id, label_off, labels, edge_off, edge_to, paths=[ref_path])
    n_reads = 1_000_000 // s
    t = time.time()
    bases, off = synth.sim_reads_walk(node_id, label_off, labels, edge_off, edge_to, n_reads, 150, seed=33)
    reads_s = time.time() - t
    px = psi_amd.PathIndex.build(g, 31, 0, rng_seed=1)
    f = psi_amd.SeedFinder(g, 31, device=0)
    f.set_path_index(px)
    for d in (31, 1):
        r = run(f, bases, off, n_reads, d, args.steps, torch, np)
        r.update(series='config5', paths=0, step=d, sub_rate=0.0, k=31, reads=n_reads, nodes=len(node_id), edges=len(edge_to),
                 graph_s=round(gen_s, 1), reads_s=round(reads_s, 1))
        print(json.dumps(r), flush=True)
    f.close()


if __name__ == '__main__':
    main()
