#!/usr/bin/env python3
"""BASELINE.json configs[2] / configs[3] at full size: whole-genome-like synthetic graph (3.1 Gbp backbone,
80 M SNV bubbles; the 24 components are not modelled: one component), k = 21, 150 bp reads.

  configs[2]  one MI355X, 10 M reads:          python tools/wg_scale.py
  configs[3]  100 M reads over the GPUs of a node (north_star: "read batches shard embarrassingly across the
              8 GPUs"):                         python tools/wg_scale.py --devices 0-7 --reads 100000000
              ONE process, ONE host index, one psigpu context + host thread per GPU, contiguous read ranges
              (read ids stay global through rec_offset); every GPU holds the whole graph + index + tables.
              Reports per-GPU and aggregate seeds/s (device-resident entry, barrier to barrier) and, with
              --host-entry, the end-to-end rate through every GPU's own host link.

--paths 3 --patched is psikt's default indexing (44.6 M patches, 5.74 G symbols: an index in two parts);
--mode locus-table / traverse answers the on-path phase from the FM index of every part; --compare-modes
runs the chunk in the k-mer table mode as well and compares the record sets; --mems N runs MEM mode on N reads.

Not a bench line (bench.py measures configs[1]); a capability + property run whose output goes to profiles/.
Needs a host with a few hundred GB of RAM (the GPU box has it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_devices(spec):
    out = []
    for part in spec.split(','):
        if '-' in part:
            a, b = part.split('-')
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


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
    ap.add_argument('--mode', choices=('kmer-table', 'locus-table', 'traverse'), default='kmer-table')
    ap.add_argument('--compare-modes', action='store_true', help='the same chunk in k-mer table mode too: same records?')
    ap.add_argument('--mems', type=int, default=0, help='MEM mode (find_mems) on this many reads')
    ap.add_argument('--devices', default='0', help='GPUs to shard the reads over, e.g. 0-7 or 0,0,0 (one context + thread each)')
    ap.add_argument('--max-part-text', type=int, default=0, help='text symbols per index part (tests: several parts at small size)')
    ap.add_argument('--general-reads', action='store_true', help='do not tell the library that all reads have one length (they have: 150 bp)')
    args = ap.parse_args()
    import threading
    import numpy as np
    import torch
    import psi_amd
    from psi_amd import synth
    uni = 0 if args.general_reads else psi_amd.UNIFORM_READS       # (synth.sim_reads_*: every read has the same length)

    log = lambda *a: print(*a, file=sys.stderr, flush=True)   # noqa: E731
    devices = parse_devices(args.devices)
    nd = len(devices)
    out = {'config': vars(args), 'n_devices': nd}
    t = time.time()
    sg = synth.snv_graph(args.backbone, args.snvs, n_block=args.nblock, seed=11)
    out['graph_s'] = time.time() - t
    log('graph %.0f s: %d nodes, %d edges' % (out['graph_s'], sg.n_nodes, len(sg.edge_to)))
    t = time.time()
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, args.reads, 150, seed=13)
    out['reads_s'] = time.time() - t
    nodes, edges = int(g.n_nodes), int(g.n_edges)
    del sg
    t = time.time()
    px = psi_amd.PathIndex.build(g, args.k, args.paths, rng_seed=1, device=devices[0], patched=args.patched, context=args.context,
                                 max_part_text=args.max_part_text)
    out['index_build_s'] = time.time() - t
    out.update(nodes=nodes, edges=edges, text_len=int(px.text_len), starting_loci=int(px.view.n_loci),
               ftab_len=int(px.view.ftab_len), index_parts=1 + int(px.view.n_more_parts), paths_in_index=len(px.trims()),
               separators=int(px.view.n_exc) + sum(int(v.n_exc) for v in px.more_parts()))
    log('index %.0f s: text %d in %d part(s), %d loci' % (out['index_build_s'], px.text_len, out['index_parts'], px.view.n_loci))
    try:
        import resource
        out['host_peak_rss_gb_after_index'] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    except Exception:
        pass

    per_read = (150 - args.k) // args.k + 1
    cuts = [args.reads * r // nd for r in range(nd + 1)]

    class Shard:
        pass

    shards = []
    for r, d in enumerate(devices):
        sh = Shard()
        sh.dev, sh.r0, sh.r1 = d, cuts[r], cuts[r + 1]
        shards.append(sh)

    def setup(sh):
        t0 = time.time()
        sh.f = psi_amd.SeedFinder(g, args.k, device=sh.dev, mode=args.mode)
        sh.f.set_path_index(px)
        sh.upload_s = time.time() - t0
        t0 = time.time()
        sh.f.prepare()
        sh.prepare_s = time.time() - t0
        b0, b1 = int(off[sh.r0]), int(off[sh.r1])
        sh.n_bases = b1 - b0
        with torch.cuda.device(sh.dev):
            sh.d_bases = torch.from_numpy(bases[b0:b1]).to('cuda:%d' % sh.dev)
            sh.d_off = torch.from_numpy((off[sh.r0:sh.r1 + 1] - off[sh.r0]).astype(np.int64)).to('cuda:%d' % sh.dev)
            torch.cuda.synchronize(sh.dev)

    def in_threads(fn):
        err = []

        def run(sh):
            try:
                fn(sh)
            except Exception as ex:          # noqa: BLE001
                err.append('device %d: %s' % (sh.dev, ex))
        th = [threading.Thread(target=run, args=(sh,)) for sh in shards]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if err:
            raise RuntimeError('; '.join(err))

    t = time.time()
    in_threads(setup)
    out['setup_wall_s'] = time.time() - t
    out['upload_s'] = max(sh.upload_s for sh in shards)
    out['prepare_s'] = max(sh.prepare_s for sh in shards)
    log('upload %.0f s, tables %.0f s (max over %d device(s))' % (out['upload_s'], out['prepare_s'], nd))

    def step(sh):
        sh.ptr, sh.n_hits = sh.f.seeds_all_device(sh.d_bases.data_ptr(), sh.d_off.data_ptr(), sh.r1 - sh.r0, sh.n_bases,
                                                   rec_offset=sh.r0, flags=psi_amd.ALL | uni)

    in_threads(step)                              # warm-up (buffers sized)
    start = threading.Barrier(nd + 1)
    done = threading.Barrier(nd + 1)

    def timed(sh):
        start.wait()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(sh)
        sh.dt = (time.perf_counter() - t0) / args.steps
        done.wait()

    th = [threading.Thread(target=timed, args=(sh,)) for sh in shards]
    for x in th:
        x.start()
    start.wait()
    t = time.perf_counter()
    done.wait()
    dt = (time.perf_counter() - t) / args.steps
    for x in th:
        x.join()
    cs = [sh.f.counters() for sh in shards]
    n_seeds = sum(c['n_seeds'] for c in cs)
    n_hits_all = sum(c['n_hits'] for c in cs)
    c = cs[0]
    out.update(ms_per_step=dt * 1e3, seeds_per_s=n_seeds / dt, hits_per_s=n_hits_all / dt, query_mode=args.mode,
               per_device=[{'device': sh.dev, 'reads': sh.r1 - sh.r0, 'ms_per_step': sh.dt * 1e3,
                            'seeds_per_s': cc['n_seeds'] / sh.dt} for sh, cc in zip(shards, cs)],
               counters={k: v for k, v in c.items() if not k.startswith('ms_')},
               kernel_ms={k: v for k, v in c.items() if k.startswith('ms_')})
    log('step %.1f ms, %.3g seeds/s over %d device(s)' % (dt * 1e3, out['seeds_per_s'], nd))
    # properties (no oracle at this size): every seed of every error-free read is found where it was
    # sampled from, and sampled hits spell their seed
    lo = g.label_off.astype(np.int64)
    labels = g.labels
    all_found, first_ok, hits0 = True, True, None
    for sh in shards:
        hits = sh.f.copy_hits(sh.ptr, sh.n_hits)
        found = np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])
        all_found = all_found and len(found) == (sh.r1 - sh.r0) * per_read and \
            (len(hits) == 0 or (int(hits[:, 2].min()) >= sh.r0 and int(hits[:, 2].max()) < sh.r1))
        rank = hits[:200000, 0].astype(np.int64) - 1
        first = labels[lo[rank] + hits[:200000, 1].astype(np.int64)]
        seed_first = bases[(hits[:200000, 2] * np.uint64(150) + hits[:200000, 3]).astype(np.int64)]
        first_ok = first_ok and bool((first == seed_first).all())
        if sh is shards[0]:
            hits0 = hits
        del found
    out['all_seeds_found'] = bool(all_found)
    out['first_base_agrees'] = first_ok
    hits = hits0
    sh0 = shards[0]
    if args.compare_modes and args.mode != 'kmer-table':
        # the first shard's chunk in the k-mer table mode: the same record set?
        a = psi_amd.sort_unique(hits)
        sh0.f.set_query_mode('kmer-table')
        t = time.time()
        sh0.f.prepare()
        out['kmer_table_prepare_s'] = time.time() - t
        step(sh0)
        b = psi_amd.sort_unique(sh0.f.copy_hits(sh0.ptr, sh0.n_hits))
        out['same_records_as_kmer_table_mode'] = bool(a.shape == b.shape and (a == b).all())
        out['records_compared'] = int(len(a))
        log('records equal to the k-mer table mode: %s (%d)' % (out['same_records_as_kmer_table_mode'], len(a)))
        del a, b
        sh0.f.set_query_mode(args.mode)
    if args.mems:
        # MEM mode (find_mems, minimum length k) on the first reads: a read's first pattern is reported the moment it is
        # k bases long, with every occurrence ON THE INDEXED PATHS -- exactly the on-path hits of the read's seed at
        # offset 0, hence a subset of that seed's hits (which also hold what the starting loci contribute)
        nm = min(args.mems, sh0.r1 - sh0.r0)
        t = time.perf_counter()
        mems = sh0.f.find_mems((bases[:int(off[nm])], off[:nm + 1]))
        out['mems_ms'] = (time.perf_counter() - t) * 1e3
        out['mems_records'] = int(len(mems))
        first_mem = mems[(mems[:, 3] == 0) & (mems[:, 4] == args.k)]
        seed0 = hits[(hits[:, 3] == 0) & (hits[:, 2] < nm)]
        key = lambda x: set(map(tuple, x[:, :3].tolist()))      # noqa: E731
        out['mems_first_patterns'] = int(len(first_mem))
        out['mems_first_pattern_subset_of_seed_hits'] = bool(len(first_mem) > 0 and key(first_mem) <= key(seed0))
        out['mems_min_len'] = int(mems[:, 4].min()) if len(mems) else 0
        log('MEM mode: %d records for %d reads in %.0f ms' % (len(mems), nm, out['mems_ms']))
    if args.host_entry:
        # the same chunk through the host entry point (pageable reads staged by the helper thread, ~100
        # sub-batches, sort-unique on the device, 2+ GB of records into pinned host memory): same records
        want = psi_amd.sort_unique(hits)
        want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
        del hits
        b0, b1 = int(off[sh0.r0]), int(off[sh0.r1])
        sub = (bases[b0:b1], (off[sh0.r0:sh0.r1 + 1] - off[sh0.r0]).astype(np.uint64))
        t = time.perf_counter()
        su = sh0.f.seeds_all(sub, step=args.k, sort_unique=True, rec_offset=sh0.r0)
        out['host_entry_first_call_ms'] = (time.perf_counter() - t) * 1e3       # (cold pinned pool: 2+ GB of hipHostMalloc)
        out['host_entry_records'] = int(len(su))
        out['host_entry_equals_device_entry'] = bool(su.shape == want.shape and (su == want).all())
        del su, want
        import ctypes as C
        L = psi_amd.lib()

        def host_loop(sh):
            b0, b1 = int(off[sh.r0]), int(off[sh.r1])
            pin = (psi_amd.pinned_copy(bases[b0:b1]), psi_amd.pinned_copy((off[sh.r0:sh.r1 + 1] - off[sh.r0]).astype(np.uint64)))
            h = psi_amd.Hits()
            call = (sh.f.ctx, psi_amd._ptr(pin[0].array), psi_amd._ptr(pin[1].array), sh.r1 - sh.r0, args.k, args.k, sh.r0,
                    psi_amd.ALL | psi_amd.SORT_UNIQUE | uni, C.byref(h))
            assert L.psigpu_find_seeds(*call) == 0          # warm: pinned pool, slot buffers
            L.psigpu_free_hits(C.byref(h))
            start.wait()
            ts = []
            for _ in range(3):                        # reads in pinned memory, records left in the library's pinned buffer
                t0 = time.perf_counter()
                assert L.psigpu_find_seeds(*call) == 0
                ts.append((time.perf_counter() - t0) * 1e3)
                sh.n_rec = h.n
                L.psigpu_free_hits(C.byref(h))
            sh.host_ms = ts
            done.wait()

        th = [threading.Thread(target=host_loop, args=(sh,)) for sh in shards]
        for x in th:
            x.start()
        start.wait()
        t = time.perf_counter()
        done.wait()
        wall = (time.perf_counter() - t) / 3
        for x in th:
            x.join()
        out['host_entry_ms'] = wall * 1e3
        out['host_entry_ms_per_device'] = [min(sh.host_ms) for sh in shards]
        out['host_entry_seeds_per_s'] = args.reads * per_read / wall
        out['host_entry_sub_batches_sorted_in_place'] = int(sh0.f.counters()['sorted_in_place'])
        log('host entry %.0f ms per chunk over %d device(s), equal: %s' % (out['host_entry_ms'], nd, out['host_entry_equals_device_entry']))
    try:
        import resource
        out['host_peak_rss_gb'] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    except Exception:
        pass
    for sh in shards:
        sh.f.close()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
