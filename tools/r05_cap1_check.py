#!/usr/bin/env python3
"""kmer-table mode with walk_cap 1 at a reduced configs[1] size: counters and records, this build against PSI_AMD_LIB (round 4's)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import psi_amd
from psi_amd import synth
k = 21
sg = synth.snv_graph(6_000_000, 130_000, n_block=1_000_000, seed=11)
g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
bases, off = synth.sim_reads_snv(sg, 100_000, 150, seed=13)
px = psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0)
res = {}
for mode, cap in (('kmer-table', 0), ('kmer-table', 1), ('traverse', 0)):
    f = psi_amd.SeedFinder(g, k, mode=mode, walk_cap=cap)
    f.set_path_index(px)
    r = f.seeds_all((bases, off), step=k, sort_unique=True)
    c = f.counters()
    res[(mode, cap)] = r
    print(mode, cap, 'records', len(r), {x: c[x] for x in ('n_loci', 'n_loci_traversed', 'n_locus_kmers', 'n_path_kmers', 'traverse_launches', 'n_kpaths')}, flush=True)
    f.close()
a = res[('kmer-table', 0)]
print('equal', [bool(v.shape == a.shape and (v == a).all()) for v in res.values()])
