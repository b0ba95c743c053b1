#!/usr/bin/env python3
"""One-off campaign beyond the committed seeds of tests/test_gpu_parity.py::test_random_graphs_vs_brute:
random graphs (cycles, N runs, empty-ish nodes, out-degree up to 6) x random k / seed distance /
indexed paths -- full or PATCHED with a random context -- / SA rate / interval table, every query mode
and walk caps 0 / 1 / 3, device index build on and off, against the brute-force definition; the host
entry point with a random sub-batch size and sort-unique on the device; the starting loci against the
brute-force definition over the (trimmed) paths; MEM mode against brute.find_mems; a gocc threshold over
both phases against the oracle's seeds_all.
`python tools/fuzz_modes.py FIRST LAST [low]`."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    low_complexity = len(sys.argv) > 3 and sys.argv[3] == 'low'      # third argument "low": low-complexity sequences
    import psi_amd
    from oracle import brute
    os.environ.setdefault('PSI_AMD_MODE', 'kmer-table')
    import test_gpu_parity as T
    n_cases = 0
    for seed in range(first, last):
        if os.environ.get('FUZZ_TRACE'):
            print('seed', seed, flush=True)
        g, reads = T._random_graph(seed)
        rng = random.Random(seed)
        if low_complexity and rng.random() < 0.7:
            # homopolymers, dinucleotide repeats, k-mers that occur everywhere: sentinel values,
            # long runs in the tables, duplicate chains
            alpha = rng.choice(['T', 'A', 'AT', 'TG', 'TTTTA'])
            g.seq = {v: ''.join(rng.choice(alpha) if c != 'N' else 'N' for c in s) for v, s in g.seq.items()}
            reads = [''.join(rng.choice(alpha) for _ in r) for r in reads]
        rank = {v: i for i, v in enumerate(g.ids)}
        label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
        labels = ''.join(g.seq[v] for v in g.ids).encode()
        edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
        edge_to = [rank[t] for v in g.ids for t in g.out[v]]
        labels_at_top = bytes(bytearray(labels))          # (a copy of its own: campaign c found the ORIGINAL changed later on)

        def labels_intact(where):
            if labels == labels_at_top:
                return
            a_, b_ = np.frombuffer(labels, np.uint8), np.frombuffer(labels_at_top, np.uint8)
            d_ = np.flatnonzero(a_ != b_)
            print('LABELS CHANGED', seed, where, '%d of %d bytes differ, first at %s; is %s, was %s; buffer at 0x%x; bytes around the first (is): %s'
                  % (len(d_), a_.size, d_[:8].tolist(), a_[d_[:16]].tolist(), b_[d_[:16]].tolist(), a_.ctypes.data,
                     a_[max(0, int(d_[0]) - 32):int(d_[0]) + 96].tobytes().hex()), flush=True)
            sys.exit(1)
        pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to,
                                    paths=[[rank[v] for v in g.paths[0][1]]])
        labels_intact('after the graph object was made')
        for _ in range(2):
            k = rng.choice([3, 8, 12, 13, 16, 21, 25, 31, 31, 32, 40])      # (32, 40: two-word seeds)
            step = rng.choice([1, 2, k, k + 3])
            npaths = rng.choice([0, 1, 1, 2, 3, 5])
            patched = npaths > 1 and rng.random() < 0.6
            want = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
            bargs = dict(rng_seed=seed, sa_rate=rng.choice([1, 1, 1, 2, 8]), ftab_len=rng.choice([0, 0, 4, psi_amd.NO_FTAB]),
                         device=rng.choice([None, 0]), patched=patched,
                         context=rng.choice([0, k, k + 1, k + 7]) if patched else 0)
            labels_intact('before the index build (k %d)' % k)
            px = psi_amd.PathIndex.build(pg, k, npaths, **bargs)
            labels_intact('after the index build (k %d, npaths %d, on %s)' % (k, npaths, bargs['device']))
            # (round 5) canaries: what the HOST holds of this index and of the inputs, as it is now -- compared again after every
            # finder run below.  The oracle-side disagreements of rounds 4-5 say that something a loaded process holds in host
            # memory changes under it; if it does, what the new bytes look like says who wrote them.
            canary = {'loci_node': px.loci[0], 'loci_off': px.loci[1], 'paths': [p.copy() for p in px.paths()],
                      'labels': np.frombuffer(labels, np.uint8).copy(), 'edge_to': np.array(edge_to, dtype=np.uint64),
                      'reads': np.frombuffer(''.join(reads).encode(), np.uint8).copy()}

            def check_canaries(where):
                now = {'loci_node': px.loci[0], 'loci_off': px.loci[1], 'paths': [p for p in px.paths()],
                       'labels': np.frombuffer(labels, np.uint8), 'edge_to': np.array(edge_to, dtype=np.uint64),
                       'reads': np.frombuffer(''.join(reads).encode(), np.uint8)}
                for name, was in canary.items():
                    cur = now[name]
                    pairs = list(zip(was, cur)) if name == 'paths' else [(was, cur)]
                    for a_, b_ in pairs:
                        if a_.shape != b_.shape or not np.array_equal(a_, b_):
                            d = np.flatnonzero(a_ != b_) if a_.shape == b_.shape else np.zeros(0, np.int64)
                            print('HOST CANARY CHANGED', seed, where, name, 'shape', a_.shape, b_.shape, 'first differing indices', d[:12].tolist(),
                                  'was', a_[d[:12]].tolist() if len(d) else None, 'is', b_[d[:12]].tolist() if len(d) else None,
                                  'bytes around the first (is):', b_.view(np.uint8)[max(0, int(d[0]) * b_.itemsize - 32):int(d[0]) * b_.itemsize + 64].tobytes().hex() if len(d) else None,
                                  flush=True)
                            sys.exit(1)
            # starting loci = brute-force definition over the trimmed paths (acyclic graphs only: the
            # brute force lists walks, the product's candidate sets handle repeats on their own)
            paths_ids = [[g.ids[r] for r in p] for p in px.paths()]
            if all(len(set(p)) == len(p) for p in paths_ids) and len(g.ids) <= 40 and k <= 16:
                ln, lo = px.loci
                got_loci = [(g.ids[v], int(o)) for v, o in zip(ln.tolist(), lo.tolist())]
                if got_loci != brute.uncovered_loci(g, paths_ids, k, px.trims()):
                    print('LOCI MISMATCH', seed, k, npaths, patched, flush=True)
                    sys.exit(1)
            if npaths >= 2 and rng.random() < 0.5:
                # the same paths indexed in several parts (k-mer table only): hit set, raw multiplicities,
                # gocc threshold over all parts
                lens = [sum(len(g.seq[v]) for v in p) for p in paths_ids]
                cut = max(max(lens) + 40, int(px.text_len) // rng.choice([2, 3, 4]))      # at most 8 parts
                try:
                    pp = psi_amd.PathIndex.build(pg, k, npaths, rng_seed=seed, patched=patched, context=px.view.context,
                                                 device=rng.choice([None, 0]), max_part_text=cut)
                except psi_amd.PsiGpuError as e:          # more than 8 parts (one long path sets the cut): not a case
                    if 'too many parts' not in str(e):
                        raise
                    pp = None
                if pp is not None and pp.view.n_more_parts:
                    gocc = rng.choice([0, 0, 1, 3])
                    res = []
                    pmode = rng.choice(['kmer-table', 'locus-table', 'traverse'])      # every mode answers an index in parts
                    for ix in (px, pp):
                        f = psi_amd.SeedFinder(pg, k, mode='kmer-table' if ix is px else pmode, gocc_threshold=gocc)
                        f.set_path_index(ix)
                        raw = f.seeds_all(reads, step=step)
                        res.append(raw[np.lexsort(raw.T[::-1])])
                        f.close()
                    if px.view.sa_rate != 1 or pmode != 'kmer-table':       # (the FM modes emit every occurrence, the traverser
                        res = [psi_amd.sort_unique(r) for r in res]           # what it finds again: compare the sets)
                    if not (res[0].shape == res[1].shape and (res[0] == res[1]).all()):
                        print('PARTS MISMATCH', seed, k, step, npaths, patched, cut, gocc, flush=True)
                        sys.exit(1)
                    if gocc == 0 and not (psi_amd.sort_unique(res[1]).shape == want.shape and (psi_amd.sort_unique(res[1]) == want).all()):
                        print('PARTS vs BRUTE MISMATCH', seed, k, step, npaths, pmode, flush=True)
                        sys.exit(1)
                    if px.view.sa_rate == 1:
                        # MEM mode over the parts = over the one-part index
                        fa, fb = psi_amd.SeedFinder(pg, k, gocc_threshold=gocc), psi_amd.SeedFinder(pg, k, gocc_threshold=gocc)
                        fa.set_path_index(px); fb.set_path_index(pp)
                        ma, mb = fa.find_mems(reads), fb.find_mems(reads)
                        fa.close(); fb.close()
                        if not (ma.shape == mb.shape and (ma == mb).all()):
                            print('PARTS MEM MISMATCH', seed, k, npaths, gocc, flush=True)
                            sys.exit(1)
                    n_cases += 1
            for mode in ('kmer-table', 'locus-table', 'traverse'):
                for cap in ((0,) if mode == 'traverse' else (0, 1, 3)):
                    labels_intact('before the finder of mode %s cap %d' % (mode, cap))
                    f = psi_amd.SeedFinder(pg, k, mode=mode, walk_cap=cap)
                    if mode == 'traverse' and rng.random() < 0.5:
                        f.set_tuning(psi_amd.TUNE_NO_PATH_TABLE)        # the FM index on the paths instead of their k-mer table
                    f.set_path_index(px)
                    labels_intact('after the finder of mode %s cap %d was loaded' % (mode, cap))
                    got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
                    labels_intact('after seeds_all of mode %s cap %d' % (mode, cap))
                    if not (got.shape == want.shape and (got == want).all()):
                        print('MISMATCH', seed, k, step, npaths, mode, cap, got.shape, want.shape, 'sa_rate', px.view.sa_rate, 'ftab', px.view.ftab_len,
                              'patched', patched, f.counters(), flush=True)
                        a, b = set(map(tuple, got.tolist())), set(map(tuple, want.tolist()))
                        print(' extra', sorted(a - b)[:6], 'missing', sorted(b - a)[:6], flush=True)
                        # is what the finder reads on the device still what was loaded?
                        print(' arrays on the device that changed since they were loaded:', repr(f.verify_resident()), flush=True)
                        # the same finder again (a glitch of that call, or of what the finder built?), then a new finder
                        again = psi_amd.sort_unique(f.seeds_all(reads, step=step))
                        print(' same finder again:', 'ok' if (again.shape == want.shape and (again == want).all()) else 'WRONG AGAIN', flush=True)
                        f3 = psi_amd.SeedFinder(pg, k, mode=mode, walk_cap=cap)
                        f3.set_path_index(px)
                        fresh = psi_amd.sort_unique(f3.seeds_all(reads, step=step))
                        print(' new finder:', 'ok' if (fresh.shape == want.shape and (fresh == want).all()) else 'WRONG TOO', flush=True)
                        # what the finder was given: the index built again (same place, then on the host), its loci compared
                        for dev in (bargs['device'], None):
                            px2 = psi_amd.PathIndex.build(pg, k, npaths, **dict(bargs, device=dev))
                            same_loci = all(bool((a == b).all()) if a.shape == b.shape else False for a, b in zip(px.loci, px2.loci))
                            for o in ((), (('no_pfx_roots', 1),)):
                                f4 = psi_amd.SeedFinder(pg, k, mode=mode, walk_cap=cap)
                                for n_, v_ in o:
                                    f4.set_option(n_, v_)
                                f4.set_path_index(px2)
                                r4 = psi_amd.sort_unique(f4.seeds_all(reads, step=step))
                                print(' index rebuilt on', dev, o, '| loci equal to the first:', same_loci, '| result:',
                                      'ok' if (r4.shape == want.shape and (r4 == want).all()) else 'WRONG', flush=True)
                                f4.close()
                        # (round 5, after campaign b: wrong in every variant above) what all those finders share is the HOST graph
                        # object: is what it holds still what it was made from?  and does a second object made from the same
                        # Python data answer right?
                        up = np.frombuffer(labels.upper(), np.uint8)
                        print(' graph object unchanged: labels', bool(np.array_equal(pg.labels, up)),
                              'label_off', bool(np.array_equal(pg.label_off, label_off)), 'edge_off', bool(np.array_equal(pg.edge_off, edge_off)),
                              'edge_to', bool(np.array_equal(pg.edge_to, np.array(edge_to, dtype=np.uint32))), flush=True)
                        if not np.array_equal(pg.labels, up):
                            d = np.flatnonzero(pg.labels != up)
                            print('  labels differ at', d[:16].tolist(), 'is', bytes(pg.labels[d[:16]]), 'was', bytes(up[d[:16]]), flush=True)
                        pg2 = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to, paths=[[rank[v] for v in g.paths[0][1]]])
                        px3 = psi_amd.PathIndex.build(pg2, k, npaths, **dict(bargs, device=None))
                        f7 = psi_amd.SeedFinder(pg2, k, mode=mode, walk_cap=cap)
                        f7.set_path_index(px3)
                        r7 = psi_amd.sort_unique(f7.seeds_all(reads, step=step))
                        print(' a second graph object from the same data:', 'ok' if (r7.shape == want.shape and (r7 == want).all()) else 'WRONG', flush=True)
                        want2 = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
                        print(' the brute-force set computed again == the first:', bool(want2.shape == want.shape and (want2 == want).all()), flush=True)
                        f5 = psi_amd.SeedFinder(pg, k, mode=mode, walk_cap=cap)
                        f5.set_option('no_pfx_roots', 1)
                        f5.set_path_index(px)
                        r5 = psi_amd.sort_unique(f5.seeds_all(reads, step=step))
                        print(' first index, no prefix roots:', 'ok' if (r5.shape == want.shape and (r5 == want).all()) else 'WRONG', flush=True)
                        sys.exit(1)
                    if rng.random() < 0.5 and k <= 31:      # (the oracle's k-mers are one word)
                        # psikt -r T over BOTH phases against the oracle's seeds_all( gocc_thr = T ): on-path k-mers
                        # over the threshold skipped, the traverser not thresholded (index_iter.hpp:826-847);
                        # set on the live finder, i.e. after its tables were made without one
                        thr = rng.choice([1, 1, 2, 3])
                        rb, ro = psi_amd.pack_reads([r.upper() for r in reads])
                        arrays = (np.array(g.ids), label_off, np.frombuffer(labels, np.uint8), edge_off,
                                  np.array(edge_to, dtype=np.uint64))
                        wg = T._oracle_hits(arrays, f, rb, ro, k, step, gocc=thr)
                        f.set_gocc_threshold(thr)
                        gg = psi_amd.sort_unique(f.seeds_all(reads, step=step))
                        f.set_gocc_threshold(0)
                        if not (gg.shape == wg.shape and (gg == wg).all()):
                            print('GOCC MISMATCH', seed, k, step, npaths, patched, mode, cap, thr, gg.shape, wg.shape, flush=True)
                            # who is it: the product or the checker?  both again -- the oracle on one thread too
                            f.set_gocc_threshold(thr)
                            g2 = psi_amd.sort_unique(f.seeds_all(reads, step=step))
                            w2 = T._oracle_hits(arrays, f, rb, ro, k, step, gocc=thr)
                            w1 = T._oracle_hits(arrays, f, rb, ro, k, step, gocc=thr, threads=1)
                            print(' again: product', g2.shape, 'oracle', w2.shape, 'oracle on one thread', w1.shape,
                                  '| first product run == second:', bool(gg.shape == g2.shape and (gg == g2).all()),
                                  '| first oracle run == one-thread run:', bool(wg.shape == w1.shape and (wg == w1).all()), flush=True)
                            # (round 5) which INPUT of the side that is wrong has changed?  The oracle takes the graph arrays made above,
                            # the reads, and paths / trims / loci from the finder's HOST index object; the product took the same loci
                            # onto the device when the finder was made.
                            arrays2 = (np.array(g.ids), np.cumsum([0] + [len(g.seq[v]) for v in g.ids]),
                                       np.frombuffer(''.join(g.seq[v] for v in g.ids).encode(), np.uint8),
                                       np.cumsum([0] + [len(g.out[v]) for v in g.ids]),
                                       np.array([rank[t] for v in g.ids for t in g.out[v]], dtype=np.uint64))
                            print(' graph arrays unchanged:', [bool(np.array_equal(np.asarray(a), np.asarray(b))) for a, b in zip(arrays, arrays2)], flush=True)
                            for nm_, a_, b_ in zip(('ids', 'label_off', 'labels', 'edge_off', 'edge_to'), arrays, arrays2):
                                a_, b_ = np.asarray(a_), np.asarray(b_)
                                if a_.shape == b_.shape and not np.array_equal(a_, b_):
                                    d_ = np.flatnonzero(a_ != b_)
                                    # (campaign c: the labels' bytes object had changed -- WHAT was written there says who wrote it)
                                    print('  %s: %d of %d elements differ, first at %s; is %s, was %s; address of the buffer 0x%x; bytes around the first (is): %s'
                                          % (nm_, len(d_), a_.size, d_[:8].tolist(), a_[d_[:16]].tolist(), b_[d_[:16]].tolist(), a_.ctypes.data,
                                             a_.view(np.uint8)[max(0, int(d_[0]) * a_.itemsize - 32):int(d_[0]) * a_.itemsize + 96].tobytes().hex()), flush=True)
                            pxh = psi_amd.PathIndex.build(pg, k, npaths, **dict(bargs, device=None))
                            l1, l2 = px.loci, pxh.loci
                            same = bool(np.array_equal(l1[0], l2[0]) and np.array_equal(l1[1], l2[1]))
                            print(' loci of the index object == loci of a fresh host build:', same, len(l1[0]), len(l2[0]), flush=True)
                            if not same and len(l1[0]) == len(l2[0]):
                                d = np.flatnonzero((l1[0] != l2[0]) | (l1[1] != l2[1]))
                                print('  differing loci (index, object, fresh):', [(int(i), (int(l1[0][i]), int(l1[1][i])), (int(l2[0][i]), int(l2[1][i]))) for i in d[:8]], flush=True)
                            print(' paths of the index object == fresh:', [a.tolist() == b.tolist() for a, b in zip(px.paths(), pxh.paths())],
                                  'trims', px.trims() == pxh.trims(), flush=True)
                            if npaths == 0:
                                print(' (no paths: both sides should equal the brute-force set) product == brute:', bool(gg.shape == want.shape and (gg == want).all()),
                                      '| oracle == brute:', bool(wg.shape == want.shape and (wg == want).all()), flush=True)
                            f9 = psi_amd.SeedFinder(pg, k, mode=mode, walk_cap=cap)
                            f9.set_path_index(pxh)
                            w9 = T._oracle_hits(arrays2, f9, rb, ro, k, step, gocc=thr)
                            f9.set_gocc_threshold(thr)
                            g9 = psi_amd.sort_unique(f9.seeds_all(reads, step=step))
                            print(' with the fresh index and fresh arrays: product', g9.shape, 'oracle', w9.shape, flush=True)
                            sys.exit(1)
                        n_cases += 1
                    if cap == 0:
                        # host entry point in pieces, sorted on the device
                        os.environ['PSIGPU_SUB_BYTES'] = sub_b = str(rng.choice([16, 200, 3000, 1 << 30]))
                        su = f.seeds_all(reads, step=step, sort_unique=True, rec_offset=5)
                        os.environ.pop('PSIGPU_SUB_BYTES')
                        w2 = want.copy(); w2[:, 2] += 5
                        w2 = w2[np.lexsort((w2[:, 1], w2[:, 0], w2[:, 3], w2[:, 2]))]
                        if not (su.shape == w2.shape and (su == w2).all()):
                            print('SORT-UNIQUE MISMATCH', seed, k, step, npaths, mode, 'cap', cap, 'sa_rate', px.view.sa_rate, 'ftab', px.view.ftab_len,
                                  'patched', patched, 'sub', sub_b, su.shape, w2.shape, f.counters(), flush=True)
                            a, b = set(map(tuple, su.tolist())), set(map(tuple, w2.tolist()))
                            print(' extra', sorted(a - b)[:6], 'missing', sorted(b - a)[:6], 'duplicates', len(su) - len(a),
                                  'sorted', bool((np.lexsort((su[:, 1], su[:, 0], su[:, 3], su[:, 2])) == np.arange(len(su))).all()), flush=True)
                            sys.exit(1)
                    if mode == 'kmer-table' and cap == 0 and npaths and px.view.sa_rate == 1:
                        gocc, mm = rng.choice([0, 0, 2]), rng.choice([0, 0, 3])
                        f2 = psi_amd.SeedFinder(pg, k, mode=mode, gocc_threshold=gocc)
                        f2.set_path_index(px)
                        gm = f2.find_mems(reads, max_mem=mm)
                        wm = np.array(brute.find_mems(g, paths_ids, reads, k, px.trims(), gocc, mm), dtype=np.uint64).reshape(-1, 6)
                        if not (gm.shape == wm.shape and (gm == wm).all()):
                            print('MEM MISMATCH', seed, k, npaths, patched, gocc, mm, gm.shape, wm.shape,
                                  '| arrays of THIS finder that changed on the device since they were loaded:', repr(f2.verify_resident()), flush=True)
                            f6 = psi_amd.SeedFinder(pg, k, mode=mode, gocc_threshold=gocc)
                            f6.set_path_index(px)
                            g6 = f6.find_mems(reads, max_mem=mm)
                            a6, b6 = set(map(tuple, gm.tolist())), set(map(tuple, wm.tolist()))
                            print(' extra', sorted(a6 - b6)[:4], 'missing', sorted(b6 - a6)[:4], 'duplicates', len(gm) - len(a6),
                                  '| a new finder:', 'ok' if (g6.shape == wm.shape and (g6 == wm).all()) else 'WRONG TOO',
                                  '| arrays on the device that changed since they were loaded:', repr(f6.verify_resident()), flush=True)
                            sys.exit(1)
                        f2.close()
                    f.close()
                    check_canaries('after mode %s cap %d' % (mode, cap))
                    n_cases += 1
        if (seed - first) % 25 == 24:
            print('.. through seed %d, %d finder runs' % (seed, n_cases), flush=True)
    print('ok: seeds %d..%d, %d finder runs' % (first, last, n_cases))


if __name__ == '__main__':
    main()
