#!/usr/bin/env python3
"""Load campaign for the one open correctness question of round 3: hit sets under GPU SHARING -- several contexts and
host threads in one process (the psikt --devices / wg_scale --devices arrangement) and several processes on one GPU.

    python tools/stress.py --procs 4 --threads 4 --graphs 2000 [--first 500000] [--opt no_engine_copy=1 ...]

Every thread owns its contexts.  Per random graph (tests/test_gpu_parity.py::_random_graph: cycles, N runs, out-degree
up to 6) one brute-force hit set (oracle/brute.py: the definition), then a handful of finder LIFETIMES (create, load,
tables, queries, destroy) in random query modes, each queried through
  D  the device entry (reads copied in with the runtime's memcpy, records copied out with psigpu_copy_hits),
  H  the host entry: raw and sorted, pageable and pinned reads, ASCII and packed, sub-batches of random sizes
and compared with the definition.  The two are counted apart: a fault that only H shows is in the host entry's
transfers (named SDMA engines, read-ahead ring, wire records, host-side widening); one that D shows too is in what
`prepare` built or in the kernels.  Exit code 0 only when no comparison failed and every process ended normally.
Prints one JSON line."""
import argparse
import json
import os
import random
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def graph_arrays(g):
    import numpy as np
    rank = {v: i for i, v in enumerate(g.ids)}
    label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
    labels = ''.join(g.seq[v] for v in g.ids).encode()
    edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
    edge_to = [rank[t] for v in g.ids for t in g.out[v]]
    return rank, label_off, labels, edge_off, edge_to


def worker(args, proc, tid, seeds, stats, lock):
    import numpy as np
    import torch
    import psi_amd
    from oracle import brute
    import test_gpu_parity as T
    opts = dict(o.split('=') for o in args.opt)

    def eq(a, b):
        return a.shape == b.shape and bool((a == b).all())

    def bad(kind, seed, detail):
        with lock:
            stats['mismatch_' + kind] = stats.get('mismatch_' + kind, 0) + 1
            stats.setdefault('first_mismatches', [])
            if len(stats['first_mismatches']) < 12:
                stats['first_mismatches'].append({'kind': kind, 'seed': seed, 'proc': proc, 'thread': tid, **detail})
        print('MISMATCH', kind, seed, detail, file=sys.stderr, flush=True)

    n_calls = n_life = n_stale = 0
    for seed in seeds:
        g, reads = T._random_graph(seed)
        rng = random.Random(seed * 7919 + 13)
        rank, label_off, labels, edge_off, edge_to = graph_arrays(g)
        pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to, paths=[[rank[v] for v in g.paths[0][1]]])
        up = [r.upper() for r in reads]
        k = rng.choice([3, 8, 12, 16, 21, 25, 31, 31, 40])
        step = rng.choice([1, 2, k, k + 3])
        npaths = rng.choice([0, 1, 1, 2, 3])
        patched = npaths > 1 and rng.random() < 0.5
        want = np.array(brute.hit_set(g, up, k, step), dtype=np.uint64).reshape(-1, 4)
        rec0 = rng.choice([0, 5, 1 << 33])
        w2 = want.copy(); w2[:, 2] += np.uint64(rec0)
        w_sorted = w2[np.lexsort((w2[:, 1], w2[:, 0], w2[:, 3], w2[:, 2]))]
        px = psi_amd.PathIndex.build(pg, k, npaths, rng_seed=seed, sa_rate=rng.choice([1, 1, 1, 4]),
                                     ftab_len=rng.choice([0, 0, 4, psi_amd.NO_FTAB]), device=rng.choice([None, args.device]),
                                     patched=patched, context=rng.choice([0, k, k + 5]) if patched else 0)
        bases, off = psi_amd.pack_reads(reads)
        pin = (psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off))
        pk = psi_amd.PackedReads(bases, off, pinned=rng.random() < 0.5)
        d_b = torch.from_numpy(bases).cuda(args.device) if len(bases) else torch.zeros(1, dtype=torch.uint8, device='cuda:%d' % args.device)
        d_o = torch.from_numpy(off.astype(np.int64)).cuda(args.device)
        for life in range(args.lifetimes):
            mode = rng.choice(['kmer-table', 'kmer-table', 'kmer-table', 'locus-table', 'traverse'])
            f = psi_amd.SeedFinder(pg, k, device=args.device, mode=mode, walk_cap=rng.choice([0, 0, 1, 3]) if mode != 'traverse' else 0)
            for name, v in opts.items():
                f.set_option(name, int(v))
            f.set_path_index(px)
            if rng.random() < 0.5:
                f.prepare()
            n_life += 1
            # D: device entry
            ptr, n = f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), step=step, rec_offset=rec0)
            got = psi_amd.sort_unique(f.copy_hits(ptr, n))
            n_calls += 1
            if not eq(got, psi_amd.sort_unique(w2)):
                bad('device_entry', seed, {'k': k, 'step': step, 'npaths': npaths, 'mode': mode, 'life': life,
                                           'got': int(len(got)), 'want': int(len(want))})
            # D2: the same chunk twice through begin / end, the second begun while the first is in flight (ABI 6)
            if rng.random() < 0.5:
                f.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), step=step, rec_offset=rec0)
                f.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(bases), step=step, rec_offset=rec0)
                for _half in range(2):
                    ptr, n = f.seeds_all_device_end()
                    got = psi_amd.sort_unique(f.copy_hits(ptr, n))
                    n_calls += 1
                    if not eq(got, psi_amd.sort_unique(w2)):
                        bad('device_entry_two_in_flight', seed, {'k': k, 'step': step, 'npaths': npaths, 'mode': mode, 'life': life,
                                                                 'half': _half, 'got': int(len(got)), 'want': int(len(want))})
            # H: host entry, a few shapes per lifetime
            for rep in range(args.calls):
                f.set_option('sub_bytes', rng.choice([16, 64, 200, 3000, 1 << 30]))
                if 'no_ahead' not in opts:
                    f.set_option('no_ahead', rng.choice([0, 0, 1]))
                if 'wire' not in opts:
                    f.set_option('wire', rng.choice([0, 0, 0, 16, 32]))
                shape = rng.choice(['pageable', 'pinned', 'packed'])
                su = rng.random() < 0.5
                if shape == 'packed':
                    r = f.seeds_all_packed(pk, step=step, rec_offset=rec0, sort_unique=su)
                else:
                    src = (bases, off) if shape == 'pageable' else (pin[0].array, pin[1].array)
                    r = f.seeds_all(src, step=step, rec_offset=rec0, sort_unique=su)
                n_calls += 1
                ok = eq(r, w_sorted) if su else eq(psi_amd.sort_unique(r), psi_amd.sort_unique(w2))
                if not ok:
                    a, b = set(map(tuple, r.tolist())), set(map(tuple, w2.tolist()))
                    bad('host_entry', seed, {'k': k, 'step': step, 'npaths': npaths, 'mode': mode, 'life': life, 'shape': shape,
                                             'sorted': su, 'got': int(len(r)), 'want': int(len(want)),
                                             'extra': sorted(a - b)[:4], 'missing': sorted(b - a)[:4], 'counters': f.counters()})
            n_stale += int(f.counters()['stale_handbacks'])
            f.close()
        del pin, pk, px, pg
    with lock:
        stats['calls'] = stats.get('calls', 0) + n_calls
        stats['lifetimes'] = stats.get('lifetimes', 0) + n_life
        stats['graphs'] = stats.get('graphs', 0) + len(seeds)
        stats['stale_handbacks'] = stats.get('stale_handbacks', 0) + n_stale


def run_process(args):
    """One process: --threads threads, each with contexts of its own, over this process's share of the graphs."""
    stats, lock = {}, threading.Lock()
    per = (args.graphs + args.procs - 1) // args.procs
    mine = list(range(args.first + args.proc * per, args.first + min(args.graphs, (args.proc + 1) * per)))
    th, errs = [], []

    def body(t):
        try:
            worker(args, args.proc, t, mine[t::args.threads], stats, lock)
        except Exception as ex:       # noqa: BLE001
            import traceback
            traceback.print_exc()
            errs.append('%s: %s' % (type(ex).__name__, ex))
    for t in range(args.threads):
        x = threading.Thread(target=body, args=(t,))
        x.start()
        th.append(x)
    for x in th:
        x.join()
    stats['errors'] = errs
    print('STATS ' + json.dumps(stats), flush=True)
    bad = sum(v for kk, v in stats.items() if kk.startswith('mismatch_')) + len(errs)
    os._exit(1 if bad else 0)                     # (no interpreter teardown under library threads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--procs', type=int, default=4)
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--graphs', type=int, default=2000)
    ap.add_argument('--first', type=int, default=500000)
    ap.add_argument('--lifetimes', type=int, default=3, help='finders made and destroyed per graph')
    ap.add_argument('--calls', type=int, default=4, help='host-entry calls per finder')
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--opt', action='append', default=[], help='psigpu_set_option name=value on every finder')
    ap.add_argument('--proc', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--timeout', type=int, default=1500)
    ap.add_argument('--gdb', action='store_true', help='every second process under rocgdb (backtraces of all threads on a fatal signal)')
    args = ap.parse_args()
    if args.proc >= 0:
        return run_process(args)
    t0 = time.time()
    children = []
    for p in range(args.procs):
        cmd = [sys.executable, os.path.abspath(__file__), '--proc', str(p)] + [a for a in sys.argv[1:] if a != '--gdb']
        if args.gdb and p % 2 == 1 and os.path.exists('/opt/rocm/bin/rocgdb'):
            cmd = ['/opt/rocm/bin/rocgdb', '-q', '-batch', '-ex', 'handle SIGUSR1 SIGUSR2 SIGPIPE SIGALRM nostop noprint pass',
                   '-ex', 'run', '-ex', 'echo \n==== ALL THREADS ====\n', '-ex', 'thread apply all bt 24', '--args'] + cmd
        children.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    total, abnormal, tails = {}, [], []
    for p, ch in enumerate(children):
        try:
            out, err = ch.communicate(timeout=max(10, args.timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            ch.kill()
            out, err = ch.communicate()
            abnormal.append({'proc': p, 'why': 'timeout'})
        line = next((ln for ln in out.splitlines() if ln.startswith('STATS ')), None)
        if line is None:
            abnormal.append({'proc': p, 'rc': ch.returncode, 'why': 'no result'})
            tails.append({'proc': p, 'stdout': out[-3000:], 'stderr': err[-3000:]})
            continue
        st = json.loads(line[6:])
        for kk, v in st.items():
            if isinstance(v, int):
                total[kk] = total.get(kk, 0) + v
            elif isinstance(v, list):
                total.setdefault(kk, [])
                total[kk] += v
        if ch.returncode not in (0, 1) and not (args.gdb and p % 2 == 1):
            abnormal.append({'proc': p, 'rc': ch.returncode})
            tails.append({'proc': p, 'stderr': err[-3000:]})
        elif 'ALL THREADS' in out and 'exited normally' not in out and 'exited with code' not in out:
            abnormal.append({'proc': p, 'why': 'fatal signal under gdb'})
            tails.append({'proc': p, 'stdout': out[-6000:]})
    total.update(procs=args.procs, threads=args.threads, opts=args.opt, wall_s=round(time.time() - t0, 1),
                 abnormal_exits=abnormal, tails=tails)
    total['mismatches'] = sum(v for kk, v in total.items() if kk.startswith('mismatch_') and isinstance(v, int))
    print(json.dumps(total), flush=True)
    sys.exit(1 if (total['mismatches'] or abnormal or total.get('errors')) else 0)


if __name__ == '__main__':
    main()
