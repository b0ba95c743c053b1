#!/usr/bin/env python3
"""A/B of the FM kernels on the chr22-like workload, locus-table mode, on three indexes (interval table 13,
no interval table, SA sampled at 32): lane-per-seed K1 / quad LF kernel with and without text verification;
decoupled walk + resolve kernels against the lock-step locate kernel of rounds 1-2.  Prints one JSON line per
(index, variant): K1 / K2 / step times from the library's HIP events, LF steps, hits (must agree).
`python tools/lf_ab.py [reads] [backbone] [snvs]`."""
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
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    backbone = int(sys.argv[2]) if len(sys.argv) > 2 else 51_000_000
    snvs = int(sys.argv[3]) if len(sys.argv) > 3 else 1_100_000
    k = 21
    sg = synth.snv_graph(backbone, snvs, n_block=min(11_000_000, backbone // 5), seed=11)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, reads, 150, seed=13)
    eb, eo = synth.sim_reads_snv(sg, reads, 150, seed=14, sub_rate=0.01)
    d = [(torch.from_numpy(b).cuda(), torch.from_numpy(o.astype(np.int64)).cuda(), len(b)) for b, o in ((bases, off), (eb, eo))]
    T = psi_amd
    NO_DV = T.TUNE_NO_DIRECT | T.TUNE_NO_VERIFY
    f0 = psi_amd.SeedFinder(g, k, mode='locus-table')
    print(json.dumps({'random_16B_lane_loads_per_s': f0.measure_random_loads(4 << 30, 7_000_000, False),
                      'random_64B_quad_sectors_per_s': f0.measure_random_loads(4 << 30, 56_000_000, True),
                      'random_64B_quad_sectors_per_s_64MB': f0.measure_random_loads(64 << 20, 56_000_000, True),
                      'random_64B_quad_sectors_per_s_16MB': f0.measure_random_loads(16 << 20, 56_000_000, True),
                      'random_64B_quad_sectors_per_s_2MB': f0.measure_random_loads(2 << 20, 56_000_000, True)}), flush=True)
    f0.close()
    for label, kw in (('ftab13', {}), ('no_ftab', dict(ftab_len=psi_amd.NO_FTAB)), ('sa32', dict(sa_rate=32))):
        ix = psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0, **kw)
        want = {}
        variants = [('default', 0), ('quad K1 + verify', T.TUNE_NO_DIRECT), ('quad K1, LF only', NO_DV)]
        if label == 'sa32':
            variants = [('walk + resolve', 0)]
        for vname, tune in variants:
            f = psi_amd.SeedFinder(g, k, mode='locus-table')
            f.set_tuning(tune)
            f.set_path_index(ix)
            f.prepare()
            for bi, (db, do, nb) in enumerate(d):
                acc = {}
                for it in range(8):
                    ptr, n = f.seeds_all_device(db.data_ptr(), do.data_ptr(), reads, nb, step=k)
                    c = f.counters()
                    if it >= 3:
                        for key in ('ms_search', 'ms_locate', 'ms_probe', 'ms_total'):
                            acc[key] = acc.get(key, 0.0) + c[key] / 5
                hits = psi_amd.sort_unique(f.copy_hits(ptr, n))
                sig = (len(hits), int(hits.sum() % (1 << 61)))
                ok = want.setdefault(bi, sig) == sig
                print(json.dumps({'index': label, 'variant': vname, 'reads': 'error-free' if bi == 0 else '1% substitutions',
                                  **{a: round(b, 4) for a, b in acc.items()}, 'n_lf_steps': c['n_lf_steps'],
                                  'n_rows_verified': c['n_rows_verified'], 'n_locate_steps': c['n_locate_steps'],
                                  'hits': len(hits), 'same_hits': ok}), flush=True)
            f.close()
        del ix


if __name__ == '__main__':
    main()
