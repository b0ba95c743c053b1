#!/usr/bin/env python3
"""BASELINE.json configs[2] at full size on ONE MI355X: whole-genome-like synthetic graph
(3.1 Gbp backbone, 80 M SNV bubbles, 24 components are not modelled: one component), one indexed
path (text < 2^32 symbols: the 32-bit index layout), 10 M x 150 bp reads, k = 21.

Not a bench line (bench.py measures configs[1]); a capability + property run whose output goes
to profiles/.  Needs a host with a few hundred GB of RAM (the GPU box has it).

    python tools/wg_scale.py [--backbone 3100000000 --snvs 80000000 --reads 10000000 --steps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', type=int, default=3_100_000_000)
    ap.add_argument('--snvs', type=int, default=80_000_000)
    ap.add_argument('--nblock', type=int, default=150_000_000)
    ap.add_argument('--reads', type=int, default=10_000_000)
    ap.add_argument('--k', type=int, default=21)
    ap.add_argument('--host-entry', action='store_true', help='also run the chunk through psigpu_find_seeds and compare')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--paths', type=int, default=1, help='walks per region (psikt -n)')
    ap.add_argument('--patched', action='store_true', help="psikt's default indexing mode")
    ap.add_argument('--context', type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import psi_amd
    from psi_amd import synth

    log = lambda *a: print(*a, file=sys.stderr, flush=True)   # noqa: E731
    out = {'config': vars(args)}
    t = time.time()
    sg = synth.snv_graph(args.backbone, args.snvs, n_block=args.nblock, seed=11)
    out['graph_s'] = time.time() - t
    log('graph %.0f s: %d nodes, %d edges' % (out['graph_s'], sg.n_nodes, len(sg.edge_to)))
    t = time.time()
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, args.reads, 150, seed=13)
    out['reads_s'] = time.time() - t
    nodes, edges = int(g.n_nodes), int(g.n_edges)
    backbone, alt = sg.backbone, sg.alt
    del sg
    t = time.time()
    px = psi_amd.PathIndex.build(g, args.k, args.paths, rng_seed=1, device=0, patched=args.patched, context=args.context)
    out['index_build_s'] = time.time() - t
    out.update(nodes=nodes, edges=edges, text_len=int(px.text_len), starting_loci=int(px.view.n_loci),
               ftab_len=int(px.view.ftab_len), index_parts=1 + int(px.view.n_more_parts), paths_in_index=len(px.trims()))
    log('index %.0f s: text %d, %d loci' % (out['index_build_s'], px.text_len, px.view.n_loci))
    t = time.time()
    f = psi_amd.SeedFinder(g, args.k, device=0)
    f.set_path_index(px)
    out['upload_s'] = time.time() - t
    t = time.time()
    f.prepare()
    out['prepare_s'] = time.time() - t
    log('upload %.0f s' % out['upload_s'])
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ptr, n_hits = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), args.reads, len(bases), stream=stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.steps):
        ptr, n_hits = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), args.reads, len(bases), stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / args.steps
    c = f.counters()
    out.update(ms_per_step=dt * 1e3, seeds_per_s=c['n_seeds'] / dt, hits_per_s=c['n_hits'] / dt,
               counters={k: v for k, v in c.items() if not k.startswith('ms_')},
               kernel_ms={k: v for k, v in c.items() if k.startswith('ms_')})
    log('step %.1f ms, %.3g seeds/s' % (dt * 1e3, out['seeds_per_s']))
    # properties (no oracle at this size): every seed of every error-free read is found where it was
    # sampled from, and sampled hits spell their seed
    hits = f.copy_hits(ptr, n_hits)
    per_read = (150 - args.k) // args.k + 1
    found = np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])
    out['all_seeds_found'] = bool(len(found) == args.reads * per_read)
    lo = g.label_off.astype(np.int64)
    labels = g.labels
    rank = hits[:200000, 0].astype(np.int64) - 1
    first = labels[lo[rank] + hits[:200000, 1].astype(np.int64)]
    seed_first = bases[(hits[:200000, 2] * np.uint64(150) + hits[:200000, 3]).astype(np.int64)]
    out['first_base_agrees'] = bool((first == seed_first).all())
    if args.host_entry:
        # the same chunk through the host entry point (pageable reads staged by the helper thread, ~100
        # sub-batches, sort-unique on the device, 2+ GB of records into pinned host memory): same records
        want = psi_amd.sort_unique(hits)
        want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
        del hits, found
        t = time.perf_counter()
        su = f.seeds_all((bases, off), step=args.k, sort_unique=True)
        out['host_entry_first_call_ms'] = (time.perf_counter() - t) * 1e3       # (cold pinned pool: 2+ GB of hipHostMalloc)
        out['host_entry_records'] = int(len(su))
        import ctypes as C
        pin = (psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off.astype(np.uint64)))
        h = psi_amd.Hits()
        L = psi_amd.lib()
        call = (f.ctx, psi_amd._ptr(pin[0].array), psi_amd._ptr(pin[1].array), args.reads, args.k, args.k, 0,
                psi_amd.ALL | psi_amd.SORT_UNIQUE, C.byref(h))
        ts = []
        for _ in range(3):                        # reads in pinned memory, records left in the library's pinned buffer
            t = time.perf_counter()
            assert L.psigpu_find_seeds(*call) == 0
            ts.append((time.perf_counter() - t) * 1e3)
            n_rec = h.n
            L.psigpu_free_hits(C.byref(h))
        out['host_entry_ms'] = min(ts)
        out['host_entry_ms_all'] = ts
        out['host_entry_seeds_per_s'] = args.reads * per_read / (min(ts) * 1e-3)
        assert n_rec == len(su)
        out['host_entry_equals_device_entry'] = bool(su.shape == want.shape and (su == want).all())
        out['host_entry_sub_batches_sorted_in_place'] = int(f.counters()['sorted_in_place'])
        log('host entry %.0f ms (first call %.0f), %d records, equal: %s' % (out['host_entry_ms'], out['host_entry_first_call_ms'], len(su), out['host_entry_equals_device_entry']))
    f.close()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
