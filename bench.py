#!/usr/bin/env python3
"""bench.py -- seeds queried / s on the chr22-like configuration (BASELINE.json configs[1]).

A "step" is one pass of the hot path (SeedFinder::seeds_all, reference
include/psi/seed_finder.hpp:1724-1732, as driven per chunk by src/psikt.cpp:195-204) over one
batch of 1 M synthetic 150-bp reads; two batches (different reads) alternate in the timed loop.
In the default query mode a step is: seeding (k_seed_scan_*, k_seed_pack) -> one probe of the
k-mer table per seed (k_kmer_probe: the table holds what the FM backward search and the traverser
would return, tabulated once per index) -> emission (k_kmer_emit).  `roofline_by_mode` carries the
same workload through the FM-index kernels (k_fm_search*, k_fm_locate*) and the query-time
traverser (k_traverse), the kernels BASELINE.json's north_star names.

`value` follows the bench contract: reads resident in HBM when the timed region starts, hits left
in HBM.  The SURVEY 8(d) number -- H2D of the reads + kernels + device sort-unique + D2H of the hits
through psigpu_find_seeds -- is the `end_to_end` object of the same line (with a PCIe roofline).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one process per GPU (started here when not already under torchrun), every rank holds the
whole index and its own read batches (weak scaling, no data-path collective); the hit lists are
gathered on rank 0 over RCCL once, outside the timed region, and that time is reported.
Prints ONE JSON line on rank 0 (`roofline`, `cpu_baseline`, `parity_vs_cpu_sample` included).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0           # PCIe Gen5 x16, one direction (64 GT/s x 16 lanes, 128b/130b)
BLOCK = 64                     # bytes per rank block / HBM sector
PROFILE_ROUND = os.environ.get('PSI_PROFILE_ROUND', 'r06')
TRAFFIC_FILES = {'kmer-table': '%s_k_traffic.json', 'locus-table': '%s_l_traffic.json', 'traverse': '%s_t_traffic.json',
                 # the fm-lf series (tools/profile.sh f1 / f2 / f3): locus-table mode with the LF kernels doing the work
                 'fm-lf/after_ftab': '%s_f1_traffic.json', 'fm-lf/no_ftab': '%s_f2_traffic.json',
                 'fm-lf/sa32': '%s_f3_traffic.json', 'traverse/fm': '%s_t2_traffic.json'}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(kernel, c, k, sa_rate, ftab_len=0, implicit_info=False):
    """Algorithmic bytes one launch needs: SURVEY.md 8(d)'s per-unit figures;
    for K1 the work the interval table / text verification replace is priced as what replaces it."""
    if kernel == 'k_fm_search':
        # per N-free seed one 8-byte interval-table entry; per LF step actually needed two rank
        # probes (one 64-byte block each, not discounted when both ends share a block); per SA row
        # finished against its record 16 bytes (row record) -- 12 when finished against the text
        # (round 5: ONE block per LF step.  SURVEY 8(d) says 2 k B per seed -- one block per interval end -- but the two
        # ends share their block once the interval is small, which is most steps: priced at two, the no-ftab series came
        # out at 1.01 of the 8 TB/s peak, which measures the formula, not the kernel.  One per step is what a step cannot
        # do without; the counter-based figure is `traffic`.)
        if ftab_len and ftab_len != 0xFFFFFFFF and k >= ftab_len:
            return 8.0 * c['n_seeds_valid'] + 1.0 * BLOCK * c['n_lf_steps'] + 12.0 * c['n_rows_verified']
        return 1.0 * BLOCK * max(c['n_lf_steps'], k * c['n_seeds_valid'] // 2)
    if kernel == 'k_table_insert':
        # per N-free seed one 16-byte table slot and one 4-byte bitmap word, both read-modify-write
        return 2.0 * (16 + 4) * c['n_seeds_valid']
    if kernel == 'k_fm_locate':
        if c['n_path_kmers']:
            # k-mer table mode: K2 is a stream -- 8 bytes of probe results (round 4; 16 before) + 8 bytes of (read, offset)
            # in per seed, one 32-byte record out per hit (positions were inline in the slots)
            # (equal read lengths answered from the table alone: no (read, offset) array -- 8 bytes in per seed)
            return (8.0 if implicit_info else 16.0) * c['n_seeds'] + 32.0 * (c['n_hits_on_path'] + c.get('n_hits_table', 0))      # (the traverser writes its own records)
        # SA-order sampling at rate s: expected s-1 LF steps (one block each) + the 4-byte
        # sample, two 64-byte segment-table probes, one 32-byte record out; hits that come from the
        # locus k-mer table: one 16-byte entry in, one 32-byte record out
        if sa_rate > 1:
            # SURVEY 8(d) verbatim: locate (s/2) B + 8, map 2 B, emit 32 per on-path hit (the walk itself is longer:
            # SA-order sampling, as sdsl's default, ends a walk with probability 1/s per step -- n_locate_steps)
            return (sa_rate / 2.0 * BLOCK + 8 + 2 * BLOCK + 32) * c['n_hits_on_path'] + (16 + 32.0) * c['n_hits_table']
        # whole suffix array resident (sa_rate 1): the kernel that runs is k_fm_locate_direct over the LOCATED suffix array --
        # 20 bytes per seed in (interval, count, flags, (read, offset)), one 8-byte (node, offset) entry per on-path hit,
        # one 32-byte record out per hit; 16 + 32 per hit that comes from the locus table.  (Rounds 1-3 priced a hit as
        # SURVEY 8(d) prices locate + map -- sample + two segment-table sectors, 164 B -- which this layout does not read:
        # the fraction came out above 1.)
        return 20.0 * c['n_seeds'] + (8 + 32.0) * c['n_hits_on_path'] + (16 + 32.0) * c['n_hits_table']
    if kernel == 'k_kmer_step':
        # the default step as ONE kernel (round 5): per seed its k bases in (ASCII, one byte each), per read its two offsets (the
        # equal-length claim is checked against them), per N-free seed one 16-byte table slot, per hit one 32-byte record out;
        # keys and probe results never leave the registers
        return float(k) * c['n_seeds'] + 16.0 * c['n_reads'] + 16.0 * c['n_seeds_valid'] + 32.0 * (c['n_hits_on_path'] + c.get('n_hits_table', 0))
    if kernel in ('k_kmer_probe', 'k_lkt_probe'):
        # per seed its 8-byte key in and 8 bytes of results out to K2 (round 4; 16 before: the fraction is of FEWER bytes now);
        # per N-free seed one 16-byte slot in
        return (8 + (8.0 if kernel == 'k_kmer_probe' else 16.0)) * c['n_seeds'] + 16.0 * c['n_seeds_valid']
    if kernel == 'k_seed_pack':
        # each seed's k bytes of bases in, 8-byte key + 8-byte (read, offset) out (no (read, offset) when it is implicit)
        return (k + (8.0 if implicit_info else 16.0)) * c['n_seeds']
    if kernel == 'k_traverse':
        # per k-walk from a starting locus (a launch resolves all of them, most by pruning):
        # ceil(k/4) label bytes + 4 per edge list touched + 16-byte seed-table probe (32 B at
        # k = 21, 40 B at k = 31); 32-byte record per hit
        ck = 32 if k <= 21 else 40
        return float(ck) * c['n_kwalks_all'] + 32.0 * c['n_hits_off_path']
    raise KeyError(kernel)


def oracle_objects(sg, px):
    """OracleGraph + OraclePathIndex over the product's paths (the oracle's own text; its suffix array
    is supplied by the product's SA-IS and VERIFIED by the oracle before use)."""
    import numpy as np
    import oracle
    import psi_amd
    og = oracle.OracleGraph(sg.node_id, sg.label_off, bytes(sg.labels), sg.edge_off, sg.edge_to.astype(np.uint64))
    paths = [p for p in px.paths()]
    pidx = None
    if paths:
        code = np.full(256, 5, np.uint8)
        for ch, v in ((65, 2), (67, 3), (71, 4), (84, 6)):
            code[ch] = v
        lo = sg.label_off.astype(np.int64)
        parts = []
        for i, p in enumerate(paths):
            p = p.astype(np.int64)
            lens = lo[p + 1] - lo[p]
            idx = np.repeat(lo[p] - (np.cumsum(lens) - lens), lens) + np.arange(int(lens.sum()))
            if i:
                parts.append(np.array([1], np.uint8))
            parts.append(code[sg.labels[idx]][::-1])
        parts.append(np.array([0], np.uint8))
        text = np.concatenate(parts)
        sa = psi_amd.suffix_array(text, 7)
        pidx = oracle.OraclePathIndex(og, [p.tolist() for p in paths], ext_sa=sa.astype(np.uint32))
    return og, pidx


def cpu_baseline(og, pidx, px, bases, off, k, step, n_reads_sample, budget_1t_s=8.0):
    """The oracle (C restatement of the reference path) timed on this host: 'port'.  All cores for the
    reported value; one thread (what psikt's loop is) as two measured rates on bounded samples."""
    import numpy as np
    import oracle
    threads = oracle.lib().orc_max_threads()
    ln, lo_ = px.loci
    nb = int(off[n_reads_sample])
    t0 = time.perf_counter()
    hits, st = oracle.seeds_all(og, pidx, bytes(bases[:nb]), off[:n_reads_sample + 1], k, step, ln, lo_,
                                threads=threads, want_stats=True)
    dt = time.perf_counter() - t0
    out = {'value': st['n_seeds'] / dt, 'unit': 'seeds/s', 'cores': threads, 'kind': 'port',
           'sample': '%d of the %d reads of one step (%d seeds), whole index and all %d starting '
                     'loci, %.1f s wall' % (n_reads_sample, len(off) - 1, st['n_seeds'], len(ln), dt),
           'hits_per_s': len(hits) / dt}
    # cpu-1t: the reference loop is single-threaded.  One thread cannot walk
    # all starting loci within the bench's time, so two rates are MEASURED on bounded samples -- the
    # on-path phase on r1 reads, the traverser on the first l1 loci against the same reads -- and the
    # chunk rate they imply is reported as derived.
    try:
        r1 = max(1000, min(n_reads_sample, 20_000))
        nb1 = int(off[r1])
        e = np.zeros(0, np.uint32)
        t0 = time.perf_counter()
        _, s_on = oracle.seeds_all(og, pidx, bytes(bases[:nb1]), off[:r1 + 1], k, step, e, e, threads=1, want_stats=True)
        t_on = time.perf_counter() - t0
        l1 = min(len(ln), 50_000)
        t0 = time.perf_counter()
        oracle.seeds_all(og, None, bytes(bases[:nb1]), off[:r1 + 1], k, step, ln[:l1], lo_[:l1], threads=1)
        t_off = time.perf_counter() - t0
        while t_off < budget_1t_s / 8 and l1 < len(ln):
            l1 = min(len(ln), l1 * 4)
            t0 = time.perf_counter()
            oracle.seeds_all(og, None, bytes(bases[:nb1]), off[:r1 + 1], k, step, ln[:l1], lo_[:l1], threads=1)
            t_off = time.perf_counter() - t0
        seeds_rate_on = s_on['n_seeds'] / t_on
        loci_rate = l1 / t_off
        n_seeds_chunk = st['n_seeds'] * (len(off) - 1) / n_reads_sample
        t_chunk = n_seeds_chunk / seeds_rate_on + len(ln) / loci_rate
        out['cpu_1t'] = {'on_path_seeds_per_s': seeds_rate_on, 'on_path_sample': '%d reads, %.2f s' % (r1, t_on),
                         'traverser_loci_per_s': loci_rate,
                         'traverser_sample': 'first %d of %d starting loci against %d reads, %.2f s' % (l1, len(ln), r1, t_off),
                         'derived_chunk_seeds_per_s': n_seeds_chunk / t_chunk,
                         'note': 'derived = seeds of one step / (seeds / on-path rate + loci / traverser rate): '
                                 'the reference traverses every starting locus for every chunk'}
    except Exception as ex:            # the 1-thread series is informative only
        out['cpu_1t'] = {'error': str(ex)}
    return out, hits


# ------------------------------------------------------------------------------------------------------
# The ONE stdout line.  Round 4's line had grown to 20.8 KB and the driver, which keeps a bounded tail of stdout, parsed
# nothing (BENCH_r04.json: parsed null).  The line is now the contract's keys + roofline + cpu_baseline + the SURVEY 8(d)
# end-to-end rate and a handful of scalars (< 6 KB, tests/test_host.py::test_bench_line_is_small checks the bound);
# everything else -- roofline_by_mode, timing blocks, break-even, series -- goes to a side file (--full-out, default
# gpurun_out/bench_full.json or ./bench_full.json) and to stderr.
# ------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000


def _r(x, nd=6):
    """floats to nd significant digits (the line is for a parser and a reader, not for bit-exact replay)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float('%.*g' % (nd, x)) if x == x and abs(x) != float('inf') else None
    if isinstance(x, dict):
        return {k_: _r(v, nd) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    try:
        return _r(float(x), nd)
    except Exception:
        return str(x)


def slim_line(out, full_path=None):
    """The bench line proper from the full report (pure function: the CPU suite feeds it a recorded report)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'hits_per_s')
    line = {k_: out.get(k_) for k_ in keep}
    cfg = out.get('config', {})
    wl = cfg.get('workload', '')
    line['config'] = {'workload': wl if len(wl) <= 420 else wl[:417] + '...'}
    for k_ in ('reads_per_gpu', 'read_len', 'k', 'seed_step', 'indexed_paths', 'nodes', 'text_len', 'starting_loci', 'sa_rate',
               'query_mode', 'seeds_per_step_per_gpu', 'hits_per_step_per_gpu', 'locus_kmers', 'path_kmers', 'table_build_ms',
               'index_build_s', 'index_built_on', 'parallelism', 'whole_genome', 'record_order', 'shared_index_cached', 'shared_index_dir'):
        if k_ in cfg:
            line['config'][k_] = cfg[k_]
    rf = out.get('roofline') or {}
    line['roofline'] = {k_: rf.get(k_) for k_ in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                   'avg_launch_ms', 'algorithmic_bytes_per_launch', 'traffic_gbs') if k_ in rf} or None
    if rf.get('kernel_ms_per_step'):
        line['roofline']['kernel_ms_per_step'] = {k_: v for k_, v in rf['kernel_ms_per_step'].items() if v}
    cb = out.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = {k_: cb.get(k_) for k_ in ('value', 'unit', 'cores', 'kind', 'sample', 'hits_per_s')}
        one = cb.get('cpu_1t') or {}
        if 'derived_chunk_seeds_per_s' in one:
            line['cpu_baseline']['one_thread_derived_seeds_per_s'] = one['derived_chunk_seeds_per_s']
        line['gpu_over_cpu'] = (out['value'] / cb['value']) if cb.get('value') else None
    else:
        line['cpu_baseline'] = None
    # SURVEY 8(d)'s own definition (H2D + kernels + device sort-unique + D2H), beside the device-resident `value`
    e2e = out.get('end_to_end')
    if e2e:
        line['value_end_to_end'] = e2e.get('value')
        line['end_to_end'] = {'ms_per_step': e2e.get('ms_per_step'), 'value': e2e.get('value'), 'unit': 'seeds/s',
                              'hits_per_step_sort_unique': e2e.get('hits_per_step_sort_unique'),
                              'pcie_frac': (e2e.get('roofline') or {}).get('frac'),
                              'pcie_gbs': (e2e.get('roofline') or {}).get('achieved'),
                              'device_ms_per_step': e2e.get('device_ms_per_step'),
                              'wire_bytes_per_hit': e2e.get('wire_bytes_per_hit'),
                              'ascii_ms_per_step': e2e.get('ascii_ms_per_step'),
                              'first_calls_ms_per_step': e2e.get('first_calls_ms_per_step'),
                              'what': 'psigpu_find_seeds_packed: H2D of packed reads + kernels + device sort-unique + D2H of the hits; '
                                      'median of 100 calls after 400 untimed ones (first_calls: calls 3-22 of the series)'}
    mg = out.get('multi_gpu')
    if mg and mg.get('end_to_end_all_links'):
        line['value_end_to_end'] = mg['end_to_end_all_links'].get('value')
        line['end_to_end'] = {'ms_per_chunk': mg['end_to_end_all_links'].get('ms_per_chunk'), 'value': line['value_end_to_end'],
                              'unit': 'seeds/s', 'what': 'psigpu_find_seeds_packed on every rank at once, barrier to barrier'}
    if mg:
        line['multi_gpu'] = {'n_ranks': mg.get('n_ranks'), 'backend': mg.get('backend'),
                             'per_gpu_ms_per_step': [g_.get('ms_per_step') for g_ in mg.get('per_gpu', [])],
                             'properties_ok': (mg.get('properties') or {}).get('every_seed_of_the_first_reads_found_on_every_rank')}
    gh = out.get('gather_hits')
    if gh:
        line['gather_hits'] = {k_: gh.get(k_) for k_ in ('ms', 'records', 'gb_per_s', 'backend', 'sorted_by_read_id')}
        if isinstance(gh.get('cxx'), dict):
            line['gather_hits']['cxx'] = {k_: gh['cxx'].get(k_) for k_ in ('ms', 'gb_per_s', 'same_records_as_torch_gather', 'error')
                                          if k_ in gh['cxx']}
    if 'parity_vs_cpu_sample' in out:
        line['parity_vs_cpu_sample'] = out['parity_vs_cpu_sample']
        line['parity_detail'] = out.get('parity_detail')
    if 'psikt_wall' in out:
        line['psikt_wall'] = out['psikt_wall']
    # the kernels north_star names, one scalar set per route (details: the full report)
    rbm = out.get('roofline_by_mode') or {}
    routes = {}

    def brief(e, kernels):
        b = {'ms_per_step': e.get('ms_per_step'), 'seeds_per_s': e.get('seeds_per_s')}
        for kn in kernels:
            r_ = e.get(kn)
            if isinstance(r_, dict):
                tb = r_.get('traffic')
                b[kn] = {'ms': r_.get('avg_launch_ms'), 'frac': r_.get('frac'),
                         'frac_by_traffic': (tb / (r_['avg_launch_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tb and r_.get('avg_launch_ms') else None}
        return b
    if 'traverse' in rbm:
        routes['traverse'] = brief(rbm['traverse'], ('k_traverse', 'k_table_insert', 'k_kmer_step'))
        if rbm['traverse'].get('by_chunk_reads'):      # ms per 1 M reads at 1 / 2 / 4 M reads per chunk (psikt -c)
            routes['traverse']['ms_per_1M_reads_by_chunk'] = {n_: v_.get('ms_per_1M_reads') for n_, v_ in rbm['traverse']['by_chunk_reads'].items()}
        if 'fm_route' in rbm['traverse']:
            routes['traverse_fm_route'] = brief(rbm['traverse']['fm_route'], ('k_fm_search', 'k_fm_locate', 'k_traverse'))
    if 'locus-table' in rbm:
        routes['locus-table'] = brief(rbm['locus-table'], ('k_fm_search', 'k_fm_locate'))
    for lab, e in (rbm.get('fm-lf') or {}).items():
        routes['fm-lf/' + lab] = brief(e, ('k_fm_search', 'k_fm_locate'))
    if routes:
        line['routes'] = routes
    tc = out.get('two_chunks_in_flight')
    if tc:
        line['two_chunks_in_flight_ms_per_step'] = tc.get('ms_per_step')
    if full_path:
        line['full_report'] = full_path
    line = _r(line)
    # the bound is part of the contract with the driver: shed the optional parts rather than print a line it cannot keep
    for victim in ('routes', 'parity_detail', 'gather_hits', 'multi_gpu', 'psikt_wall'):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(victim, None)
    return line


def emit(out, args):
    """rank 0: the full report to the side file and stderr, the slim line (alone) to stdout."""
    path = getattr(args, 'full_out', '') or ''
    if not path:
        d = os.path.join(ROOT, 'gpurun_out')
        path = os.path.join(d if os.path.isdir(d) and os.access(d, os.W_OK) else os.getcwd(), 'bench_full.json')
    try:
        with open(path, 'w') as fh:
            json.dump(out, fh)
            fh.write('\n')
    except OSError as ex:
        log('full report not written (%s)' % ex)
        path = None
    log('FULL_REPORT ' + json.dumps(out))
    sys.stderr.flush()
    print(json.dumps(slim_line(out, os.path.relpath(path, ROOT) if path else None), allow_nan=False), flush=True)


# ------------------------------------------------------------------------------------------------------
# N > 1: BASELINE.json configs[3] -- the whole-genome graph, reads sharded over the GPUs of the node.
# One process per GPU as the driver launches them, ONE host index: rank 0 synthesises the graph and builds
# the index, writes the arrays behind the two C-ABI views into a memory-backed directory (psi_amd/shared.py),
# every rank (rank 0 too) maps them and uploads from the mapping.  The directory is keyed by the workload and
# kept, so the driver's N = 2, 4, 8 runs back to back build it once.
# ------------------------------------------------------------------------------------------------------
WG_FULL = {'backbone': 3_100_000_000, 'snvs': 80_000_000, 'nblock': 150_000_000, 'reads_per_gpu': 12_500_000}


def wg_decide(args, world):
    """rank 0: the N > 1 workload.  Whole genome when the host has room for it (the builder's peak is ~100 GB of RSS for
    one indexed walk, the shared arrays ~60 GB, every rank's reads ~4 GB), else None = chr22 weak scaling as at N = 1."""
    mode = os.environ.get('PSI_BENCH_WG', args.wg)
    if mode == 'off':
        return None, 'whole-genome workload switched off (PSI_BENCH_WG=off)'
    plan = dict(WG_FULL)
    for key, v in (('backbone', args.wg_backbone), ('snvs', args.wg_snvs), ('nblock', args.wg_nblock), ('reads_per_gpu', args.wg_reads)):
        if v:
            plan[key] = v
    scale = plan['backbone'] / WG_FULL['backbone']
    need_ram = (170e9 * scale + 6e9 * world * plan['reads_per_gpu'] / WG_FULL['reads_per_gpu'])
    need_dir = 65e9 * scale + (1 << 26)
    avail = 0
    try:
        for ln in open('/proc/meminfo'):
            if ln.startswith('MemAvailable:'):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    # (everything that changes the exported arrays is in the name: a later run with another --sa-rate / --ftab / builder, or of
    # another round's layout, must not map an older directory -- round-4 advisor.  The directory is kept so that the
    # driver's N = 2, 4, 8 runs build it once; its path and size are logged when it is made.)
    tag = 'psi_bench_wg_abi%d_%d_%d_%d_k%d_p%d_s%d_f%d_%s' % (7, plan['backbone'], plan['snvs'], plan['nblock'], args.k, args.paths,
                                                            args.sa_rate, args.ftab, 'host' if args.host_build else 'dev')
    where = None
    for base in [os.environ.get('PSI_BENCH_SHARE_DIR'), '/dev/shm', os.environ.get('TMPDIR'), '/tmp']:
        if not base or not os.path.isdir(base):
            continue
        d = os.path.join(base, tag)
        if os.path.exists(os.path.join(d, 'manifest.json')):
            where = d                    # built by an earlier run
            break
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize >= need_dir and os.access(base, os.W_OK):
                where = d
                break
        except OSError:
            continue
    cached = bool(where and os.path.exists(os.path.join(where, 'manifest.json')))
    if mode != 'force':
        if where is None:
            return None, 'no directory with %.0f GB free for the shared index' % (need_dir / 1e9)
        if avail and avail < (need_ram - (need_dir if cached else 0)):
            return None, 'host has %.0f GB available, the whole-genome set-up wants %.0f' % (avail / 1e9, need_ram / 1e9)
    elif where is None:
        where = os.path.join('/tmp', tag)
    plan.update(dir=where, cached=cached, host_mem_available_gb=avail / 1e9)
    return plan, 'ok'


def wg_build_and_export(plan, args, local_rank):
    """rank 0: graph, index (on its GPU), the read simulator's arrays -> plan['dir']; returns the set-up times."""
    import numpy as np
    import psi_amd
    from psi_amd import shared, synth
    t = {}
    t0 = time.time()
    sg = synth.snv_graph(plan['backbone'], plan['snvs'], n_block=plan['nblock'], seed=11)
    t['graph_s'] = time.time() - t0
    t0 = time.time()
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    os.makedirs(plan['dir'], exist_ok=True)
    np.save(os.path.join(plan['dir'], 'sim_backbone.npy'), sg.backbone)
    np.save(os.path.join(plan['dir'], 'sim_alt.npy'), sg.alt)
    n_block = int(sg.n_block)
    del sg
    px = psi_amd.PathIndex.build(g, args.k, args.paths, sa_rate=args.sa_rate, rng_seed=1,
                                 ftab_len=psi_amd.NO_FTAB if args.ftab < 0 else args.ftab,
                                 device=None if args.host_build else local_rank)
    t['index_build_s'] = time.time() - t0
    t0 = time.time()
    shared.export_views(plan['dir'], g, px, extra={'n_block': n_block, 'index_build_s': t['index_build_s'], 'graph_s': t['graph_s']})
    t['export_s'] = time.time() - t0
    try:
        size = sum(os.path.getsize(os.path.join(plan['dir'], f)) for f in os.listdir(plan['dir']))
        log('shared index written to %s (%.1f GB, kept for the next run of this workload: remove it by hand when done)' % (plan['dir'], size / 1e9))
    except OSError:
        pass
    del px, g
    import gc
    gc.collect()
    return t


class SimArrays:
    """what synth.sim_reads_snv needs of a SnvGraph, over the mapped arrays"""

    def __init__(self, d, n_block):
        import numpy as np
        self.backbone = np.load(os.path.join(d, 'sim_backbone.npy'), mmap_mode='r')
        self.alt = np.load(os.path.join(d, 'sim_alt.npy'), mmap_mode='r')
        self.n_block = n_block


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks as CHILD processes (before anything
    here has touched the GPU) and relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--reads', type=int, default=1_000_000)
    ap.add_argument('--read-len', type=int, default=150)
    ap.add_argument('--k', type=int, default=21)
    ap.add_argument('--step', type=int, default=0, help='seed distance (psikt -d); 0 = k')
    ap.add_argument('--paths', type=int, default=1, help='indexed paths per region (psikt -n)')
    ap.add_argument('--sa-rate', type=int, default=1)
    ap.add_argument('--ftab', type=int, default=0, help='interval-table length (0 = auto, -1 = none)')
    ap.add_argument('--tune', type=int, default=0, help='psigpu_set_tuning flags of the main finder (1 no direct K1, 2 no text '
                                                        'verification, 4 no row records)')
    ap.add_argument('--sweep-tail', type=int, default=-1, help="k_fm_sweep: steps at the end of a seed that go to memory directly (option 'sweep_tail'; -1 = the library's default)")
    ap.add_argument('--series', default='', help="label of a roofline_by_mode['fm-lf'] point this run reproduces (traffic lookup)")
    ap.add_argument('--host-build', action='store_true', help='build the index on the host (SA-IS) instead of the GPU')
    ap.add_argument('--backbone', type=int, default=51_000_000)
    ap.add_argument('--snvs', type=int, default=1_100_000)
    ap.add_argument('--nblock', type=int, default=11_000_000)
    ap.add_argument('--batches', type=int, default=2, help='read batches alternating in the timed loop')
    ap.add_argument('--cpu-reads', type=int, default=-1, help='reads in the CPU-baseline sample (0 = skip baseline and check)')
    ap.add_argument('--no-check', action='store_true', help='skip the comparison of the GPU hit set with the CPU sample')
    ap.add_argument('--lean', action='store_true', help='headline only: no other modes, no end-to-end, no CPU baseline (profiling runs)')
    ap.add_argument('--general-reads', action='store_true', help='do not tell the library that all reads have one length')
    ap.add_argument('--wg', choices=('auto', 'force', 'off'), default='auto',
                    help='N > 1: the whole-genome workload of BASELINE.json configs[3] (auto: when the host has the memory)')
    ap.add_argument('--wg-backbone', type=int, default=0, help='(tests) backbone of the N > 1 graph instead of 3.1 Gbp')
    ap.add_argument('--wg-snvs', type=int, default=0)
    ap.add_argument('--wg-nblock', type=int, default=0)
    ap.add_argument('--wg-reads', type=int, default=0, help='reads per GPU of the N > 1 workload (default 12.5 M: 100 M over 8)')
    ap.add_argument('--ordered', action='store_true', help='the timed steps without PSIGPU_ANY_ORDER: raw records seed by seed in read order')
    ap.add_argument('--full-out', default='', help='where the full report goes (default gpurun_out/bench_full.json, else ./bench_full.json)')
    ap.add_argument('--mode', choices=('kmer-table', 'locus-table', 'traverse'), default='kmer-table',
                    help="kmer-table: path k-mers and the starting loci's k-walks tabulated once in HBM, one probe "
                         "per seed; locus-table: FM index on the paths, table for the loci; traverse: FM index + "
                         "every starting locus traversed per chunk, as the reference does")
    ap.add_argument('--psikt-reads', type=int, default=10_000_000,
                    help='reads of the FASTQ the live psikt run answers (0 = no psikt run); chunks of 1 M reads (psikt -c 1000000)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # ---- psikt as a process, LIVE (round 6): the drop-in CLI on this configuration -- a 10 M-read FASTQ, GFA graph,
    # `psikt -l 21 -n 1 -c 1000000 -I ix` -- run to its end as a child process BEFORE this process makes its first GPU call (a
    # child started later would be a fork of a process that holds the device; run beside the timed steps it would disturb them).
    psikt_live = None
    if (args.gpus == 1 and not args.lean and args.psikt_reads > 0 and args.reads == 1_000_000 and args.k == 21
            and os.path.exists(os.path.join(ROOT, 'psi_amd', 'bin', 'psikt'))):
        t_pk = time.time()
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'psikt_config1.py'), '--live', '--reads', str(args.psikt_reads),
                                 '--chunk', '1000000', '--backbone', str(args.backbone), '--snvs', str(args.snvs), '--nblock', str(args.nblock)],
                                capture_output=True, text=True, timeout=900)
            psikt_live = json.loads(pr.stdout.strip().splitlines()[-1])
            psikt_live['child_wall_s'] = time.time() - t_pk
        except Exception as ex:
            log('live psikt run failed (%s: %s)' % (type(ex).__name__, ex))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log('error: --gpus %d but WORLD_SIZE %d' % (args.gpus, world))
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    import psi_amd
    from psi_amd import synth
    from psi_amd import dist as pdist

    # PSI_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than
    # ranks (ranks share devices; the collectives run on the host).  The driver's runs use RCCL.
    backend = os.environ.get('PSI_BENCH_BACKEND', 'nccl')
    n_dev = max(1, torch.cuda.device_count())
    if backend != 'nccl':
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    red_dev = 'cuda' if backend == 'nccl' else 'cpu'
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    k = args.k
    step = args.step or k
    lean = args.lean
    t0 = time.time()
    # N > 1: the whole-genome workload (configs[3]) when the host has room for it -- rank 0 decides, everybody follows
    wg, wg_note, wg_times = None, '', {}
    if world > 1:
        box = [None, '']
        if rank == 0:
            box = list(wg_decide(args, world))
        dist.broadcast_object_list(box, src=0, device=torch.device('cuda', local_rank) if backend == 'nccl' else None)
        wg, wg_note = box
        if rank == 0:
            log('N = %d workload: %s' % (world, ('whole genome, shared index in %s%s' % (wg['dir'], ' (cached)' if wg['cached'] else ''))
                                         if wg else 'chr22 weak scaling (%s)' % wg_note))
    if wg:
        from psi_amd import shared
        failed_mark = os.path.join(wg['dir'], 'build_failed')
        if rank == 0 and not wg['cached']:
            try:
                if os.path.exists(failed_mark):
                    os.remove(failed_mark)
                wg_times = wg_build_and_export(wg, args, local_rank)
                log('whole-genome set-up on rank 0: graph %.0f s, index %.0f s, export %.0f s' %
                    (wg_times['graph_s'], wg_times['index_build_s'], wg_times['export_s']))
            except BaseException as ex:           # (memory, disk: every rank then takes the chr22 workload instead of waiting for nothing)
                log('whole-genome set-up failed on rank 0 (%s: %s): falling back to configs[1] weak scaling' % (type(ex).__name__, ex))
                try:
                    os.makedirs(wg['dir'], exist_ok=True)
                    open(failed_mark, 'w').write(repr(ex))
                except OSError:
                    pass
        # (the others wait for the manifest, not inside a collective: the build takes minutes)
        t_wait = time.time()
        while not os.path.exists(os.path.join(wg['dir'], 'manifest.json')) and not os.path.exists(failed_mark):
            if time.time() - t_wait > 1500:
                break
            time.sleep(0.5)
        ok_here = os.path.exists(os.path.join(wg['dir'], 'manifest.json')) and not os.path.exists(failed_mark)
        okt = torch.tensor([1 if ok_here else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if not int(okt.item()):
            wg_note = 'the whole-genome set-up failed or timed out on this host'
            wg = None
    if wg:
        from psi_amd import shared
        g, px, extra = shared.import_views(wg['dir'])
        sg = SimArrays(wg['dir'], int(extra.get('n_block', 0)))
        args.reads = wg['reads_per_gpu']
        args.batches = 1                         # (87 M seeds per step: nothing of a batch stays in any cache)
        nb = 1
        batches = [synth.sim_reads_snv(sg, args.reads, args.read_len, seed=13 + rank)]
        t_ix = float(extra.get('index_build_s', 0.0))
    else:
        sg = synth.snv_graph(args.backbone, args.snvs, n_block=args.nblock, seed=11)
        g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                                   paths=[sg.ref_path])
        nb = max(1, args.batches)
        batches = [synth.sim_reads_snv(sg, args.reads, args.read_len, seed=13 + 100 * b + rank) for b in range(nb)]
        t_ix = time.time()
        px = psi_amd.PathIndex.build(g, k, args.paths, sa_rate=args.sa_rate, rng_seed=1,
                                     ftab_len=psi_amd.NO_FTAB if args.ftab < 0 else args.ftab,
                                     device=None if args.host_build else local_rank)
        t_ix = time.time() - t_ix
    t_up = time.time()
    finder = psi_amd.SeedFinder(g, k, device=local_rank, mode=args.mode)
    if args.tune:
        finder.set_tuning(args.tune)
    if args.sweep_tail >= 0:
        finder.set_option('sweep_tail', args.sweep_tail)
    finder.set_path_index(px)
    t_up = time.time() - t_up
    t_prep = time.time()
    finder.prepare()                      # the tables of the query mode: index load time, not query time
    t_prep = time.time() - t_prep
    time.sleep(0.3)     # the driver clears the gigabytes of temporaries prepare() freed on the copy engines: let it finish
    if rank == 0:
        log('setup %.1f s (index %.1f s on the %s, tables %.2f s): %d nodes, %d edges, text %d, %d starting loci' %
            (time.time() - t0, t_ix, 'host' if args.host_build else 'device', t_prep, g.n_nodes, g.n_edges,
             px.text_len, px.view.n_loci))

    # the synthetic reads all have --read-len bases: the caller says so (PSIGPU_UNIFORM_READS, checked on the device) as
    # psikt does for a chunk of equal-length reads; --general-reads times the path that assumes nothing
    uni = 0 if args.general_reads else psi_amd.UNIFORM_READS
    stream = torch.cuda.current_stream().cuda_stream
    rec_offset = rank * args.reads
    L = psi_amd.lib()
    dev = [(torch.from_numpy(b).cuda(), torch.from_numpy(o.astype(np.int64)).cuda(), len(b)) for b, o in batches]

    def time_mode(f, steps, warmup, mode, sync_ranks, blocks=False, call_flags=None):
        if call_flags is None:
            call_flags = uni
        """K timed steps of the device-resident entry over the alternating batches; per-kernel times from
        the library's HIP events (recorded on the streams the kernels run on)."""
        # untimed: every k-walk from the starting loci (the unit SURVEY 8(d) prices the traverser by).
        # Table modes enumerate them once into the locus k-mer table; traverse mode counts them
        # with one pass that has the pruning switched off (the timed steps prune)
        if mode == 'traverse':
            os.environ['PSIGPU_NO_PFX'] = '1'
        f.seeds_all_device(dev[0][0].data_ptr(), dev[0][1].data_ptr(), args.reads, dev[0][2], step=step,
                           rec_offset=rec_offset, stream=stream)
        c0 = f.counters()
        kwalks_all = c0['n_kpaths'] if mode == 'traverse' else c0['n_locus_kmers']
        os.environ.pop('PSIGPU_NO_PFX', None)
        cs = psi_amd.Counters()
        d_hits, n_out = C.c_void_p(), C.c_uint64()
        # the timed loop goes through the C ABI with prebuilt arguments: the binding's conveniences
        # (argument objects, a dict of counters) cost tens of microseconds per call
        # (PSIGPU_ANY_ORDER: raw records in any order, as the reference's callback stream is -- the default mode's kernel then
        # takes a tile's output range with one atomic add instead of a look-back; --ordered times the default without the flag)
        order_flag = 0 if args.ordered else psi_amd.ANY_ORDER
        calls = [(f.ctx, d[0].data_ptr(), d[1].data_ptr(), args.reads, d[2], k, step, rec_offset, psi_amd.ALL | call_flags | order_flag, stream,
                  C.byref(d_hits), C.byref(n_out)) for d in dev]
        for i in range(warmup):
            if L.psigpu_find_seeds_device(*calls[i % nb]):
                raise RuntimeError(L.psigpu_last_error(f.ctx).decode())
        # (the on-path phase is a probe of a k-mer table in the default mode, and in traverse mode unless the FM route is asked for)
        probe_name = 'k_kmer_probe' if (mode == 'kmer-table' or (mode == 'traverse' and c0['n_path_kmers'])) else 'k_lkt_probe'
        if c0.get('fused_step'):
            probe_name = 'k_kmer_step'          # seeding + probe + emission in one kernel: ms_probe is all of it
        kern = {'k_fm_search': 0.0, 'k_fm_locate': 0.0, 'k_traverse': 0.0, 'k_table_insert': 0.0, probe_name: 0.0,
                'k_seed_pack': 0.0}
        # BLOCKS of exactly `steps` steps, each bracketed by barrier + synchronize on both sides and each giving one
        # ms_per_step; enough blocks that the timed region is >= ~0.6 s whatever --steps is (a 20-step block of the
        # default mode is 9 ms: one scheduling hiccup on a shared box moved round 3's `value` by 10 %).  The line
        # reports the MEDIAN block (value, ms_per_step) and the spread (min / max / all blocks).
        n_blocks = 1
        if blocks:
            t_est = time.perf_counter()
            for i in range(3):
                if L.psigpu_find_seeds_device(*calls[i % nb]):
                    raise RuntimeError(L.psigpu_last_error(f.ctx).decode())
            torch.cuda.synchronize()
            est = (time.perf_counter() - t_est) / 3
            n_blocks = int(min(400, max(3, -(-0.6 // (steps * max(est, 1e-6))))))
            if sync_ranks and world > 1:
                nbt = torch.tensor([n_blocks], dtype=torch.int64, device=red_dev)
                dist.all_reduce(nbt, op=dist.ReduceOp.MAX)
                n_blocks = int(nbt.item())
        seeds = hits = 0
        block_s, own_s = [], []
        for _ in range(n_blocks):
            torch.cuda.synchronize()
            if sync_ranks and world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t_begin = time.perf_counter()
            for i in range(steps):
                if L.psigpu_find_seeds_device(*calls[i % nb]):
                    raise RuntimeError(L.psigpu_last_error(f.ctx).decode())
                L.psigpu_get_counters(f.ctx, C.byref(cs))
                kern['k_fm_search'] += cs.ms_search; kern['k_fm_locate'] += cs.ms_locate; kern['k_traverse'] += cs.ms_traverse
                kern['k_table_insert'] += cs.ms_table; kern[probe_name] += cs.ms_probe; kern['k_seed_pack'] += cs.ms_pack
                seeds += cs.n_seeds; hits += cs.n_hits
            torch.cuda.synchronize()
            own_s.append(time.perf_counter() - t_begin)       # this rank's own steps, before it waits for the others
            if sync_ranks and world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            block_s.append(time.perf_counter() - t_begin)
        if sync_ranks and world > 1:              # every block's time = the slowest rank's
            bt = torch.tensor(block_s, dtype=torch.float64, device=red_dev)
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
            block_s = [float(x) for x in bt.tolist()]
        elapsed = float(np.median(block_s))       # of ONE block of `steps` steps
        seeds //= n_blocks; hits //= n_blocks
        for n_ in kern:
            kern[n_] /= n_blocks
        c = f.counters()
        c['n_kwalks_all'] = kwalks_all
        c['n_hits_table'] = c['n_hits_off_path'] if c['n_locus_kmers'] and not c['n_loci_traversed'] else 0
        return {'elapsed': elapsed, 'kern': kern, 'seeds': seeds, 'hits': hits, 'c': c, 'steps': steps,
                'block_ms_per_step': [b / steps * 1e3 for b in block_s], 'own_ms_per_step': float(np.median(own_s)) / steps * 1e3}

    rand_peak = {}

    def random_load_peak(quad):
        """Independent random loads / s this device retires now -- 16 bytes per lane (the table probe) or one
        64-byte sector per quad (a rank block) -- on a 4 GiB scratch table, as many loads as a step has seeds
        (x 8 for the sector series: the LF steps of a seed); measured once per run, before the timed loops."""
        if quad not in rand_peak:
            n = args.reads * max(1, (args.read_len - k) // step + 1)
            rand_peak[quad] = finder.measure_random_loads(4 << 30, n * (8 if quad else 1), quad)
        return rand_peak[quad]

    def roofline_of(res, mode, kernel=None, ix=None, traffic_key=None):
        """Roofline object of one kernel of a mode (default: the one with the largest summed event time)."""
        kern, c, steps = res['kern'], res['c'], res['steps']
        ix = ix or px
        dom = kernel or max(kern, key=lambda n: kern[n])
        if dom not in kern:
            return None                          # (this mode did not run that kernel)
        avg_ms = kern[dom] / steps
        abytes = algorithmic_bytes(dom, c, k, int(ix.view.sa_rate), int(ix.view.ftab_len), implicit_info=bool(uni) and mode == 'kmer-table')
        achieved = abytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side traffic per launch from separate rocprofv3 --pmc passes over this same command
        # (tools/profile.sh -> profiles/<round>_*_traffic.json, committed); null when not collected
        traffic, src = None, None
        tname = TRAFFIC_FILES[traffic_key or mode] % PROFILE_ROUND
        tpath = os.path.join(ROOT, 'profiles', tname)
        if os.path.exists(tpath) and world == 1 and args.reads == 1_000_000 and k == 21 and step == 21 and args.paths == 1:
            tj = json.load(open(tpath))
            pl = tj.get('per_launch', {})
            # (a bench "kernel" may be several launches: the walk and the per-hit resolve of a sampled suffix array; the
            # partition and the per-bucket build of the chunk's seed table)
            names = {'k_fm_locate': [['void k_kmer_emit<true'], ['void k_kmer_emit<false'], ['k_kmer_emit'], ['k_fm_locate_direct'], ['k_fm_walk', 'k_hits_resolve']],
                     'k_fm_search': [['k_fm_search_direct'], ['void k_fm_search<false']],
                     'k_traverse': [['void k_traverse<false']],
                     'k_kmer_probe': [['void k_kmer_probe<true'], ['void k_kmer_probe<false'], ['k_kmer_probe']],
                     'k_kmer_step': [['void k_kmer_step<false, true'], ['void k_kmer_step<true, true'], ['void k_kmer_step<false, false'], ['k_kmer_step']],
                     'k_seed_pack': [['void k_seed_pack<false, false, true'], ['void k_seed_pack<false, false, false'], ['k_seed_pack']],
                     'k_table_insert': [['k_sb_count', 'k_sb_scatter', 'k_sb_build']]}.get(dom, [[dom]])
            tot = lambda t: t.get('fetch_size_bytes', 0.0) + t.get('write_size_bytes', 0.0)      # noqa: E731
            find = lambda pre: next((kn for kn in pl if kn == pre or kn.startswith(pre + ',') or kn.startswith(pre + '>')), None)   # noqa: E731
            sweep_kernels = [kn for kn in pl if 'k_fm_sweep' in kn or 'k_sweep_' in kn]
            if dom == 'k_fm_search' and sweep_kernels and tj.get('mode') == mode and tj.get('series', '') == (traffic_key or '').partition('/')[2]:
                # (round 6) K1 is the level-synchronous search: rounds of k_sweep_count / k_sweep_scatter / k_fm_sweep + k_sweep_totals
                traffic = sum(tot(pl[n]) for n in sweep_kernels)
                src = 'profiles/' + tname + ':' + '+'.join(sorted(n.split('(')[0] for n in sweep_kernels))
                names = []
            for group in names:
                real = [find(n) for n in group]
                if all(real) and tot(pl[real[0]]) > 1e6 and tj.get('mode') == mode and \
                        tj.get('series', '') == (traffic_key or '').partition('/')[2]:
                    traffic = sum(tot(pl[n]) for n in real)
                    src = 'profiles/' + tname + ':' + '+'.join(real)
                    break
        out = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': src,
               'avg_launch_ms': avg_ms, 'algorithmic_bytes_per_launch': abytes,
               'traffic_gbs': (traffic / (avg_ms * 1e-3) / 1e9) if traffic and avg_ms > 0 else None,
               'kernel_ms_per_step': {n: v / steps for n, v in kern.items()}}
        if dom == 'k_fm_search' and avg_ms > 0:
            # the same secondary bound for K1: its divergent loads per launch -- interval-table entry per
            # N-free seed, the locus-table slot probed beside it (locus-table mode), one 16-byte row record
            # per row verified, two rank blocks per LF step still executed
            has_ftab = int(ix.view.ftab_len) not in (0, 0xFFFFFFFF)
            # (an LF step needs one rank block when both interval ends share it -- the rule once the interval is small --
            # and two otherwise: one per step is the lower bound counted here)
            loads = c['n_seeds_valid'] * ((1 if has_ftab else 0) + (1 if mode == 'locus-table' and res.get('probe_in_k1', True) else 0)) + \
                c['n_rows_verified'] + c['n_lf_steps']
            out['random_loads_per_launch'] = float(loads)
            out['random_loads_per_s'] = loads / (avg_ms * 1e-3)
            out['random_load_peak_per_s'] = random_load_peak(True)
            out['random_load_peak_source'] = 'psigpu_measure_random_loads (64-byte sector per quad, 4 GiB table), this run'
            out['lf_steps_per_launch'] = int(c['n_lf_steps'])
            out['lf_steps_per_s'] = c['n_lf_steps'] / (avg_ms * 1e-3)
        if dom in ('k_kmer_probe', 'k_kmer_step'):
            # secondary bound: divergent 16-byte loads per second against the rate
            # tools/rand_sector2.hip measures on this part for a table of this size
            out['random_loads_per_s'] = c['n_seeds_valid'] / (avg_ms * 1e-3) if avg_ms > 0 else None
            out['random_load_peak_per_s'] = random_load_peak(False)
            out['random_load_peak_source'] = 'psigpu_measure_random_loads (16 bytes per lane, 4 GiB table), this run'
            # SURVEY's price for the search this probe replaces (2*k*64 B per seed)
            out['survey_8d_bytes_per_launch'] = algorithmic_bytes('k_fm_search', c, k, args.sa_rate, 0)
        if dom == 'k_fm_locate' and int(ix.view.sa_rate) > 1:
            out['lf_steps_per_launch'] = int(c['n_locate_steps'])
            out['random_loads_per_s'] = (c['n_locate_steps'] + 2.0 * c['n_hits_on_path']) / (avg_ms * 1e-3) if avg_ms > 0 else None
            out['random_load_peak_per_s'] = random_load_peak(True)
        return out

    if rank == 0 and not lean:
        random_load_peak(False); random_load_peak(True)
        time.sleep(0.3)                   # (4 GiB of scratch just freed: see above)
    main_res = time_mode(finder, args.steps, args.warmup, args.mode, True, blocks=True)
    elapsed = main_res['elapsed']
    seeds_total, hits_total = float(main_res['seeds']), float(main_res['hits'])
    if world > 1:                 # (elapsed is already the max over the ranks, block by block)
        tot = torch.tensor([seeds_total, hits_total], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        seeds_total, hits_total = float(tot[0].item()), float(tot[1].item())

    # the one exchange the path has (BASELINE north_star: "RCCL over xGMI only to gather hit lists"):
    # every rank's sort-unique hits of one batch to rank 0, once, outside the timed region
    gather = None
    if world > 1:
        # (whole genome: the hit list of a rank's 12.5 M reads is 2.8 GB and eight of them, twice -- torch's gather and the
        # library's -- do not fit beside a 150-GB k-mer table on the root: the hit lists of the first 2 M reads of every rank)
        n_g = min(args.reads, 2_000_000) if wg else args.reads
        nb_g = int(batches[0][1][n_g])
        ptr, n = finder.seeds_all_device(dev[0][0].data_ptr(), dev[0][1].data_ptr(), n_g, nb_g, step=step,
                                         rec_offset=rec_offset, flags=psi_amd.ALL | psi_amd.SORT_UNIQUE, stream=stream)
        mine = None
        if backend == 'nccl' and n:
            # the library's device buffer itself (no trip through the host), copied once because the gather outlives the call
            try:
                mine = torch.as_tensor(psi_amd.DeviceHits(ptr, n), device='cuda').clone()
            except Exception as ex:            # (a torch build without the array interface: through the host, as before)
                log('DeviceHits view failed (%s): gathering through a host copy' % ex)
        if mine is None:
            mine = torch.from_numpy(finder.copy_hits(ptr, n).view(np.int64))
            if backend == 'nccl':
                mine = mine.cuda()
        dist.barrier()
        t1 = time.perf_counter()
        allh = pdist.gather_hits(mine, dst=0)
        if backend == 'nccl':
            torch.cuda.synchronize()
        dist.barrier()
        t_g = time.perf_counter() - t1
        if rank == 0:
            ids = allh[:, 2]
            gather = {'reads_per_rank': int(n_g), 'ms': t_g * 1e3, 'records': int(allh.shape[0]), 'bytes': int(allh.shape[0]) * 32,
                      'gb_per_s': allh.shape[0] * 32 / t_g / 1e9, 'backend': backend,
                      'sorted_by_read_id': bool((ids[1:] >= ids[:-1]).all().item()) if allh.shape[0] > 1 else True}

    # the same gather through the library's own C++ entry (psigpu_gather_hits: RCCL loaded by the library, counts
    # all-gathered, one send/recv per rank in one group).  It runs last and under a watchdog: no torch collective
    # follows it, so if RCCL's second communicator could not come up on some rank the line is still printed.
    cxx_stuck = False
    if world > 1 and backend == 'nccl' and not os.environ.get('PSIGPU_BENCH_NO_CXX_GATHER'):
        res_cxx = {}
        ok = torch.tensor([1 if psi_amd.HitGather.available() else 0], device='cuda')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()):
            idt = torch.zeros(128, dtype=torch.uint8, device='cuda')
            if rank == 0:
                idt = torch.frombuffer(bytearray(psi_amd.HitGather.unique_id()), dtype=torch.uint8).cuda()
            dist.broadcast(idt, src=0)
            id_bytes = bytes(idt.cpu().numpy().tobytes())
            dist.barrier()

            def cxx_leg():
                try:
                    hg = psi_amd.HitGather(local_rank, id_bytes, rank, world)
                    hg.gather(ptr, n, 0)                       # warm: the first transfer sets the links up
                    t2 = time.perf_counter()
                    d_all, n_all, counts = hg.gather(ptr, n, 0)
                    res_cxx.update(ms=(time.perf_counter() - t2) * 1e3, records=int(n_all), counts=[int(x) for x in counts])
                    if rank == 0 and n_all:
                        ids = torch.as_tensor(psi_amd.DeviceHits(d_all, n_all), device='cuda')[:, 2]
                        res_cxx['sorted_by_read_id'] = bool((ids[1:] >= ids[:-1]).all().item())
                    hg.close()
                except Exception as ex:
                    res_cxx['error'] = str(ex)

            import threading
            th = threading.Thread(target=cxx_leg, daemon=True)
            th.start()
            th.join(timeout=90)
            cxx_stuck = th.is_alive()
            if cxx_stuck:
                res_cxx = {'error': 'timed out'}
        else:
            res_cxx['error'] = 'RCCL could not be loaded by the library on some rank'
        if gather is not None:
            if 'ms' in res_cxx:
                res_cxx['gb_per_s'] = res_cxx['records'] * 32 / (res_cxx['ms'] * 1e-3) / 1e9
                res_cxx['same_records_as_torch_gather'] = res_cxx['records'] == gather['records']
            gather['cxx'] = res_cxx

    multi = None
    if world > 1:
        # ---- per-GPU rates (each rank's own steps, before it waits at the barrier) --------------------------------
        c_m = main_res['c']
        pk_name = 'k_kmer_step' if 'k_kmer_step' in main_res['kern'] else 'k_kmer_probe'
        c_m['n_hits_table'] = c_m.get('n_hits_table', 0)
        probe_ms = main_res['kern'].get(pk_name, 0.0) / max(1, main_res['steps'])
        mine_t = torch.tensor([main_res['own_ms_per_step'], float(c_m['n_seeds']), probe_ms,
                               algorithmic_bytes(pk_name, c_m, k, args.sa_rate) if probe_ms else 0.0, t_up, t_prep],
                              dtype=torch.float64, device=red_dev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        per_gpu = [{'rank': r, 'ms_per_step': float(x[0]), 'seeds_per_s': float(x[1]) / (float(x[0]) * 1e-3) if float(x[0]) else 0.0,
                    'k_kmer_probe_ms': float(x[2]),          # (k_kmer_step when the default step is one kernel)
                    'roofline_frac': (float(x[3]) / (float(x[2]) * 1e-3) / 1e9 / HBM_PEAK_GBS) if float(x[2]) else None,
                    'index_upload_s': float(x[4]), 'tables_s': float(x[5])} for r, x in enumerate(allt)]
        # ---- SURVEY 8(d) through every GPU's own host link at once: psigpu_find_seeds_packed on every rank ----------
        e2e = None
        if not lean:
            pr = psi_amd.PackedReads(batches[0][0], batches[0][1], pinned=True, threads=4)
            hh = psi_amd.Hits()
            call = (finder.ctx, psi_amd._ptr(pr.words), psi_amd._ptr(pr.mask), psi_amd._ptr(pr.off), args.reads, k, step, rec_offset,
                    psi_amd.ALL | psi_amd.SORT_UNIQUE | uni, C.byref(hh))
            if L.psigpu_find_seeds_packed(*call):                 # warm: pinned pool, slot buffers
                raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
            n_rec = hh.n
            L.psigpu_free_hits(C.byref(hh))
            reps = 3
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                if L.psigpu_find_seeds_packed(*call):
                    raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                L.psigpu_free_hits(C.byref(hh))
            t_own = (time.perf_counter() - t1) / reps
            dist.barrier()
            t_all = (time.perf_counter() - t1) / reps
            tt = torch.tensor([t_own, t_all, float(n_rec)], dtype=torch.float64, device=red_dev)
            alle = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(alle, tt)
            t_wall = max(float(x[1]) for x in alle)
            e2e = {'what': 'psigpu_find_seeds_packed on every rank at once (its own reads, its own host link): H2D + kernels + '
                           'device sort-unique + D2H, barrier to barrier, %d calls' % reps,
                   'ms_per_chunk': t_wall * 1e3, 'value': float(c_m['n_seeds']) * world / t_wall, 'unit': 'seeds/s',
                   'per_gpu_ms': [float(x[0]) * 1e3 for x in alle], 'records_per_gpu': [int(x[2]) for x in alle],
                   'wire_bytes_per_hit': int(finder.counters()['wire_bytes_per_hit'])}
            del pr
        # ---- properties (no oracle at this size): every seed of the first reads found where it was sampled ---------
        n_chk = min(args.reads, 1_000_000)
        nb_chk = int(batches[0][1][n_chk])
        ptr_c, n_c = finder.seeds_all_device(dev[0][0].data_ptr(), dev[0][1].data_ptr(), n_chk, nb_chk, step=step,
                                             rec_offset=rec_offset, stream=stream)
        hc = finder.copy_hits(ptr_c, n_c)
        per_read = (args.read_len - k) // step + 1
        found = np.unique((hc[:, 2] - np.uint64(rec_offset)) * np.uint64(100000) + hc[:, 3])
        seeds_found = bool(len(found) == n_chk * per_read)
        ok_props = torch.tensor([1 if seeds_found else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok_props, op=dist.ReduceOp.MIN)
        multi = {'per_gpu': per_gpu, 'end_to_end_all_links': e2e,
                 'properties': {'every_seed_of_the_first_reads_found_on_every_rank': bool(int(ok_props.item())),
                                'reads_checked_per_rank': int(n_chk)},
                 'n_ranks': world, 'rccl_ranks': int(dist.get_world_size()) if backend == 'nccl' else 0, 'backend': backend}

    if rank == 0:
        c = main_res['c']
        steps = args.steps
        out = {
            'metric': 'seeds queried/sec (and hits located/sec), 150bp reads k=21, chr22 1000G graph',
            'value': seeds_total / elapsed,
            'unit': 'seeds/s',
            'n_gpus': world,
            'steps': steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / steps * 1e3,
            'timing': {'what': 'blocks of exactly `steps` steps, each bracketed by barrier + synchronize on both sides (max over '
                               'ranks per block); value and ms_per_step are the MEDIAN block',
                       'blocks': len(main_res['block_ms_per_step']), 'timed_region_s': sum(main_res['block_ms_per_step']) * steps / 1e3,
                       'ms_per_step_min': min(main_res['block_ms_per_step']), 'ms_per_step_max': max(main_res['block_ms_per_step']),
                       'ms_per_step_median': elapsed / steps * 1e3,
                       'ms_per_step_blocks': [round(x, 4) for x in main_res['block_ms_per_step']][:64]},
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u64',
            'data': 'synthetic',
            'value_definition': 'reads resident in HBM, hits left in HBM (bench contract); the SURVEY 8(d) rate '
                                '(H2D + kernels + device sort-unique + D2H) is end_to_end.value',
            'hits_per_s': hits_total / elapsed,
            'config': {
                'workload': ('whole-genome-like synthetic stand-in (BASELINE.json configs[3]): %d bp backbone incl. %d bp leading N, '
                             '%d bi-allelic SNV bubbles, nodes <= 32 bp; %d x %d bp error-free haplotype-walk reads per GPU = %d '
                             'reads over the %d GPUs (contiguous ranges, global read ids), k=%d, seed distance %d, %d indexed '
                             'path(s), SA sampling %d; ONE host index built by rank 0 and mapped by every rank (psi_amd/shared.py)'
                             % (wg['backbone'], wg['nblock'], wg['snvs'], args.reads, args.read_len, args.reads * world, world, k,
                                step, args.paths, args.sa_rate)) if wg else
                            (('[N > 1 fell back to configs[1] weak scaling: %s] ' % wg_note if world > 1 else '') +
                            'chr22-like synthetic stand-in (BASELINE.json configs[1]): %d bp backbone '
                            'incl. %d bp leading N, %d bi-allelic SNV bubbles, nodes <= 32 bp; %d batches of %d x %d bp '
                            'error-free haplotype-walk reads per GPU alternating, k=%d, seed distance %d, %d indexed '
                            'path(s), SA sampling %d' % (args.backbone, args.nblock, args.snvs, nb, args.reads,
                                                         args.read_len, k, step, args.paths, args.sa_rate)),
                'reads_per_gpu': args.reads, 'read_len': args.read_len, 'k': k, 'seed_step': step, 'read_batches': nb,
                'indexed_paths': args.paths, 'nodes': int(g.n_nodes), 'edges': int(g.n_edges),
                'text_len': int(px.text_len), 'starting_loci': int(px.view.n_loci),
                'ftab_len': int(px.view.ftab_len), 'sa_rate': int(px.view.sa_rate),
                'index_build_s': t_ix, 'index_built_on': 'host' if args.host_build else 'device',
                'seeds_per_step_per_gpu': int(c['n_seeds']), 'hits_per_step_per_gpu': int(c['n_hits']),
                'hits_on_path': int(c['n_hits_on_path']), 'hits_off_path': int(c['n_hits_off_path']),
                'uniform_reads_flag': not args.general_reads, 'record_order': 'read order' if args.ordered else 'any (PSIGPU_ANY_ORDER)',
                'query_mode': args.mode, 'locus_kmers': int(c['n_locus_kmers']), 'path_kmers': int(c['n_path_kmers']),
                'table_build_ms': float(c['ms_locus_table_build']), 'prepare_wall_s': t_prep,
                'loci_traversed_per_step': int(c['n_loci_traversed']),
                'kwalks_from_loci': int(c['n_kwalks_all']), 'kwalks_completed_per_step': int(c['n_kpaths']),
                'lf_steps_per_step': int(c['n_lf_steps']), 'rows_verified_per_step': int(c['n_rows_verified']),
                'parallelism': 'reads sharded x%d, index replicated' % world,
            },
            'roofline': roofline_of(main_res, args.mode, traffic_key=(('traverse/' if args.mode == 'traverse' else 'fm-lf/') + args.series) if args.series else None),
        }
        if args.tune or args.series:
            out['config']['tune'] = args.tune
            out['config']['series'] = args.series
        if gather:
            out['gather_hits'] = gather
        if multi:
            out['multi_gpu'] = multi
            out['config']['whole_genome'] = bool(wg)
            if wg:
                out['config']['shared_index_dir'] = wg['dir']
                out['config']['shared_index_cached'] = bool(wg['cached'])
                out['config']['setup_s'] = dict(wg_times, upload_s=t_up, tables_s=t_prep)
                out['config']['host_mem_available_gb'] = wg.get('host_mem_available_gb')
    if world == 1 and not lean:
        # ---- the host entry point, PCIe included ------------------------------------
        hits = psi_amd.Hits()
        pin = [(psi_amd.pinned_copy(b), psi_amd.pinned_copy(o)) for b, o in batches]
        time.sleep(0.3)

        def host_entry(src, flags, reps):
            calls = [(finder.ctx, psi_amd._ptr(b), psi_amd._ptr(o), args.reads, k, step, rec_offset, flags, C.byref(hits))
                     for b, o in src]
            n_h = 0
            for i in range(2):                       # warm: pinned pool, slot buffers
                if L.psigpu_find_seeds(*calls[i % nb]):
                    raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                L.psigpu_free_hits(C.byref(hits))
            t1 = time.perf_counter()
            for i in range(reps):
                if L.psigpu_find_seeds(*calls[i % nb]):
                    raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                n_h = hits.n
                L.psigpu_free_hits(C.byref(hits))
            return (time.perf_counter() - t1) / reps, n_h

        def host_entry_packed(src, flags, reps, warm=400):
            """psigpu_find_seeds_packed: reads as 2-bit words (+ a "not ACGT" bit per base when the chunk has any)"""
            calls = [(finder.ctx, psi_amd._ptr(pr.words), psi_amd._ptr(pr.mask), psi_amd._ptr(pr.off), args.reads, k, step,
                      rec_offset, flags, C.byref(hits)) for pr in src]
            n_h = 0
            # The call is ~1.5 ms of a dozen host threads, two copy engines and the link working together, and the box takes a
            # second or two of back-to-back calls to get there (tools/e2e_packed.py, job 20: 2.7 ms over the first 15 calls of
            # a process, 2.1 over the next 200, 1.5 from then on -- core clocks, link power states): a chunk stream is the
            # steady state, so the calls are run until they have settled and THEN timed; what the first calls took is kept.
            ts = []
            for i in range(warm + reps):
                t1 = time.perf_counter()
                if L.psigpu_find_seeds_packed(*calls[i % nb]):
                    raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                n_h = hits.n
                L.psigpu_free_hits(C.byref(hits))
                ts.append(time.perf_counter() - t1)
            first = ts[2:22]
            ts = ts[warm:]
            return float(np.median(ts)), n_h, min(ts), max(ts), float(np.median(first))

        pinned_src = [(p[0].array, p[1].array) for p in pin]
        pageable_src = [(np.ascontiguousarray(b), np.ascontiguousarray(o.astype(np.uint64))) for b, o in batches]
        # what psikt's reader does while it parses (psi::Records): ASCII -> 2-bit words, measured here on 1 and 8 threads
        t1 = time.perf_counter()
        packed_src = [psi_amd.PackedReads(b, o, pinned=True, threads=8) for b, o in batches]
        t_pack8 = (time.perf_counter() - t1) / nb
        t1 = time.perf_counter()
        psi_amd.PackedReads(batches[0][0], batches[0][1], pinned=False, threads=1)
        t_pack1 = time.perf_counter() - t1
        t_pk, n_pk, t_pk_min, t_pk_max, t_pk_first = host_entry_packed(packed_src, psi_amd.ALL | psi_amd.SORT_UNIQUE | uni, 100)
        c_e = finder.counters()
        t_su, n_su = host_entry(pinned_src, psi_amd.ALL | psi_amd.SORT_UNIQUE | uni, 10)
        c_a = finder.counters()
        t_raw, n_raw = host_entry(pinned_src, psi_amd.ALL | uni, 10)
        t_pg, _ = host_entry(pageable_src, psi_amd.ALL | psi_amd.SORT_UNIQUE | uni, 6)
        pr0 = packed_src[0]
        bytes_in = float(pr0.words.nbytes + (pr0.mask.nbytes if pr0.mask is not None else 0) + 8 * (args.reads + 1))
        bytes_out = float(c_e['wire_bytes_per_hit'] or 32) * n_pk       # (8-byte wire records, widened on the host)
        pcie_bound_ms = max(bytes_in, bytes_out) / (PCIE_PEAK_GBS * 1e9) * 1e3
        bytes_in_ascii = float(len(batches[0][0]) + 8 * (args.reads + 1))
        bytes_out_ascii = float(c_a['wire_bytes_per_hit'] or 32) * n_su
        out['end_to_end'] = {
            'what': 'psigpu_find_seeds_packed: H2D of the reads (2-bit words, in pinned host memory) + kernels + sort-unique on '
                    'the device + D2H of the hits (packed wire records of 5-8 bytes, widened to the 32-byte records by host threads inside '
                    'the call): SURVEY 8(d) timed region; sub-batches pipelined over the two named copy engines and one compute '
                    'stream; median of 100 calls after 400 untimed ones (the first calls of a process: first_calls_ms_per_step)',
            'reads_format': '2-bit packed (psigpu_find_seeds_packed); the ASCII entry (psigpu_find_seeds) is ascii_*',
            'value': c_e['n_seeds'] / t_pk, 'unit': 'seeds/s', 'ms_per_step': t_pk * 1e3,
            'ms_per_step_min': t_pk_min * 1e3, 'ms_per_step_max': t_pk_max * 1e3, 'first_calls_ms_per_step': t_pk_first * 1e3,
            'hits_per_step_sort_unique': int(n_pk), 'hits_per_s': n_pk / t_pk,
            'same_record_count_as_ascii_entry': bool(n_pk == n_su),
            'ascii_ms_per_step': t_su * 1e3, 'ascii_seeds_per_s': c_a['n_seeds'] / t_su,
            'ascii_wire_bytes_per_hit': int(c_a['wire_bytes_per_hit']),
            'ascii_pcie_frac': (max(bytes_in_ascii, bytes_out_ascii) / (PCIE_PEAK_GBS * 1e9)) / t_su,
            'raw_hits_ms_per_step': t_raw * 1e3, 'hits_per_step_raw': int(n_raw),
            'pageable_reads_ms_per_step': t_pg * 1e3,
            'host_pack_ms_per_chunk': {'threads_8': t_pack8 * 1e3, 'threads_1': t_pack1 * 1e3,
                                       'note': 'ASCII -> 2-bit words on the host (psigpu_pack_reads), outside the timed '
                                               'region: psikt packs while it parses the FASTQ'},
            'device_ms_per_step': float(c_e['ms_total']), 'device_sort_ms_per_step': float(c_e['ms_sort']),
            'sub_batches_sorted_in_place': int(c_e['sorted_in_place']),      # (of the last call: no radix sort needed)
            'wire_bytes_per_hit': int(c_e['wire_bytes_per_hit']),
            'roofline': {'bound': 'pcie', 'achieved': max(bytes_in, bytes_out) / t_pk / 1e9, 'peak': PCIE_PEAK_GBS,
                         'unit': 'GB/s', 'frac': pcie_bound_ms / (t_pk * 1e3), 'bytes_in': bytes_in, 'bytes_out': bytes_out,
                         'note': 'full duplex: the bound is max(bytes in, bytes out) / one-direction rate'},
        }
        out['host_entry_ms_per_step'] = t_pk * 1e3

        # ---- the same steps with two chunks in flight (psigpu_find_seeds_device_begin / _end, ABI 6) ----------------------
        # chunk i + 1 is queued before chunk i is ended: the device does not wait for the host between two steps.  NOT
        # `value` (whose steps are synchronous calls, as in every round): what a loop that double-buffers its batches gets.
        if args.mode == 'kmer-table':
            bcalls = [(finder.ctx, d[0].data_ptr(), d[1].data_ptr(), args.reads, d[2], k, step, rec_offset,
                       psi_amd.ALL | uni | (0 if args.ordered else psi_amd.ANY_ORDER), stream) for d in dev]
            dh, nh = C.c_void_p(), C.c_uint64()

            def pipelined(n_steps):
                tot = 0
                if L.psigpu_find_seeds_device_begin(*bcalls[0]):
                    raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                for i in range(n_steps):
                    if i + 1 < n_steps and L.psigpu_find_seeds_device_begin(*bcalls[(i + 1) % nb]):
                        raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                    if L.psigpu_find_seeds_device_end(finder.ctx, C.byref(dh), C.byref(nh)):
                        raise RuntimeError(L.psigpu_last_error(finder.ctx).decode())
                    tot += nh.value
                return tot
            pipelined(6)
            torch.cuda.synchronize()
            pb = []
            for _ in range(max(3, min(60, int(0.3 / max(1e-6, args.steps * main_res['elapsed'] / main_res['steps']))))):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ph = pipelined(args.steps)
                torch.cuda.synchronize()
                pb.append((time.perf_counter() - t1) / args.steps)
            pm = float(np.median(pb))
            cp = finder.counters()
            out['two_chunks_in_flight'] = {
                'ms_per_step': pm * 1e3, 'ms_per_step_min': min(pb) * 1e3, 'ms_per_step_max': max(pb) * 1e3, 'blocks': len(pb),
                'seeds_per_s': main_res['seeds'] / main_res['steps'] / pm, 'hits_per_step': ph // args.steps,
                'queued_chunks': bool(cp['lookahead_subbatches']), 'fallbacks': int(cp['lookahead_fallbacks']),
                'what': 'the steps of `value` through psigpu_find_seeds_device_begin / _end: the next chunk queued before the '
                        'current one is ended (same kernels, same records; the host round trip between two steps hidden)'}

        # ---- the same steps without the equal-length claim (reads of any lengths: scan + search for a seed's read) --------
        if uni:
            res_g = time_mode(finder, 10, 3, args.mode, False, call_flags=0)
            out['general_reads_path'] = {'ms_per_step': res_g['elapsed'] / res_g['steps'] * 1e3, 'seeds_per_s': res_g['seeds'] / res_g['elapsed'],
                                         'what': 'the same steps without PSIGPU_UNIFORM_READS'}
        # ---- second series: 1 % substitution errors ------------------------------------
        eb, eo = synth.sim_reads_snv(sg, args.reads, args.read_len, seed=13 + rank, sub_rate=0.01)
        d_eb, d_eo = torch.from_numpy(eb).cuda(), torch.from_numpy(eo.astype(np.int64)).cuda()
        for _ in range(3):
            finder.seeds_all_device(d_eb.data_ptr(), d_eo.data_ptr(), args.reads, len(eb), step=step, stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            finder.seeds_all_device(d_eb.data_ptr(), d_eo.data_ptr(), args.reads, len(eb), step=step, stream=stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 10
        ce = finder.counters()
        out['series_1pct_error'] = {'ms_per_step': dt * 1e3, 'seeds_per_s': ce['n_seeds'] / dt, 'hits_per_s': ce['n_hits'] / dt,
                                    'hits_per_step': int(ce['n_hits'])}
        del d_eb, d_eo

        # ---- the same workload in the other query modes: the kernels north_star names --------------
        # (after the end-to-end measurement: closing a finder frees ~10 GB of tables, and the driver's
        # clearing of freed VRAM keeps the copy engines busy for a few hundred milliseconds)
        by_mode = {args.mode: main_res}
        for m in ('kmer-table', 'locus-table', 'traverse'):
            if m in by_mode:
                continue
            f2 = psi_amd.SeedFinder(g, k, device=local_rank, mode=m)
            f2.set_path_index(px)
            f2.prepare()
            by_mode[m] = time_mode(f2, 10, 3, m, False)
            if m == 'traverse' and args.reads == 1_000_000:
                # the reference's scheme as a function of the chunk size (psikt -c, src/psikt.cpp:190-208): the sweep over the
                # loci's prefix walks and the traverser's walk over the survivors are paid per CHUNK, the seed table per seed
                sweep = {}
                d_hits2, n_out2 = C.c_void_p(), C.c_uint64()
                for m_reads in (2_000_000, 4_000_000):
                    try:
                        b_, o_ = synth.sim_reads_snv(sg, m_reads, args.read_len, seed=4242)
                        db_, do_ = torch.from_numpy(b_).cuda(), torch.from_numpy(o_.astype(np.int64)).cuda()
                        call = (f2.ctx, db_.data_ptr(), do_.data_ptr(), m_reads, len(b_), k, step, 0,
                                psi_amd.ALL | uni | (0 if args.ordered else psi_amd.ANY_ORDER), stream, C.byref(d_hits2), C.byref(n_out2))
                        for _ in range(2):
                            if L.psigpu_find_seeds_device(*call):
                                raise RuntimeError(L.psigpu_last_error(f2.ctx).decode())
                        torch.cuda.synchronize()
                        t_ = time.perf_counter()
                        for _ in range(5):
                            if L.psigpu_find_seeds_device(*call):
                                raise RuntimeError(L.psigpu_last_error(f2.ctx).decode())
                        torch.cuda.synchronize()
                        ms_ = (time.perf_counter() - t_) / 5 * 1e3
                        c_ = f2.counters()
                        sweep[str(m_reads)] = {'ms_per_chunk': ms_, 'ms_per_1M_reads': ms_ / (m_reads / 1e6), 'hits': int(c_['n_hits']),
                                               'ms_table': float(c_['ms_table']), 'ms_traverse': float(c_['ms_traverse']),
                                               'ms_probe': float(c_['ms_probe']), 'n_kpaths': int(c_['n_kpaths'])}
                        del db_, do_, b_, o_
                    except Exception as ex:
                        sweep[str(m_reads)] = {'error': '%s: %s' % (type(ex).__name__, ex)}
                by_mode[m]['chunk_sweep'] = sweep
            f2.close()
        rbm = {}
        for m, res in by_mode.items():
            e = {'ms_per_step': res['elapsed'] / res['steps'] * 1e3, 'seeds_per_s': res['seeds'] / res['elapsed'],
                 'hits_per_step': int(res['c']['n_hits']), 'dominant': roofline_of(res, m)}
            if m == 'locus-table':
                e['k_fm_search'] = roofline_of(res, m, 'k_fm_search')
                e['k_fm_locate'] = roofline_of(res, m, 'k_fm_locate')
            if m == 'traverse':
                # paths: one probe of the table of their k-mers + the emit stream; loci: the chunk's seed table, the traverser
                # (round 5: the on-path phase is k_kmer_step -- seeding, probe and emission in one kernel -- unless PSIGPU_NO_FUSED)
                pk = 'k_kmer_step' if 'k_kmer_step' in res['kern'] else 'k_kmer_probe'
                e[pk] = roofline_of(res, m, pk)
                e['k_fm_locate'] = roofline_of(res, m, 'k_fm_locate')
                e['k_table_insert'] = roofline_of(res, m, 'k_table_insert')
                e['k_traverse'] = roofline_of(res, m, 'k_traverse')
                if res.get('chunk_sweep'):
                    e['by_chunk_reads'] = dict({str(args.reads): {'ms_per_chunk': e['ms_per_step'], 'ms_per_1M_reads': e['ms_per_step'] / (args.reads / 1e6)}},
                                               **res['chunk_sweep'])
            rbm[m] = e
        # traverse mode with the FM index answering the on-path phase: the reference's scheme as written
        f2 = psi_amd.SeedFinder(g, k, device=local_rank, mode='traverse')
        f2.set_tuning(psi_amd.TUNE_NO_PATH_TABLE)
        f2.set_path_index(px)
        f2.prepare()
        res = time_mode(f2, 10, 3, 'traverse', False)
        f2.close()
        rbm['traverse']['fm_route'] = {
            'what': 'traverse mode with the FM index of the paths instead of their k-mer table (PSIGPU_TUNE_NO_PATH_TABLE): K1 here is '
                    'k_fm_search_direct = interval-table entry + per-row records, NO LF step is executed (lf_steps_per_launch 0); the '
                    "LF / rank kernel doing the work is the series roofline_by_mode['fm-lf']",
            'ms_per_step': res['elapsed'] / res['steps'] * 1e3, 'seeds_per_s': res['seeds'] / res['elapsed'],
            'hits_per_step': int(res['c']['n_hits']), 'tune': psi_amd.TUNE_NO_PATH_TABLE,
            'k_fm_search': roofline_of(res, 'traverse', 'k_fm_search', traffic_key='traverse/fm'),
            'k_fm_locate': roofline_of(res, 'traverse', 'k_fm_locate', traffic_key='traverse/fm'),
            'k_traverse': roofline_of(res, 'traverse', 'k_traverse', traffic_key='traverse/fm')}
        # ---- the LF / rank kernels doing the work (north_star: "FM-index backward-search (LF-mapping via rank
        # over the path-set BWT)"): in the modes above the interval table + the rows' records answer nearly every
        # seed without an LF step.  Three points in locus-table mode: (after_ftab) the interval table, then every
        # remaining base by an LF step; (no_ftab) all k bases by LF steps -- fmindex.hpp:851-869 as written;
        # (sa32) SA sampled at 32 in SA order, as sdsl's csa_wt< wt_huff<>, 32, 64 > is: K2 walks to a sample.
        lf = {}
        NO_DV = psi_amd.TUNE_NO_DIRECT | psi_amd.TUNE_NO_VERIFY
        for label, kw, tune in (('after_ftab', None, NO_DV),
                                ('no_ftab', dict(ftab_len=psi_amd.NO_FTAB), NO_DV),
                                ('sa32', dict(sa_rate=32), 0)):
            ix = px if kw is None else psi_amd.PathIndex.build(g, k, args.paths, rng_seed=1, device=local_rank, **kw)
            f2 = psi_amd.SeedFinder(g, k, device=local_rank, mode='locus-table')
            f2.set_tuning(tune)
            f2.set_path_index(ix)
            f2.prepare()
            res = time_mode(f2, 10, 3, 'locus-table', False)
            res['probe_in_k1'] = False
            f2.close()
            key = 'fm-lf/' + label
            lf[label] = {'ms_per_step': res['elapsed'] / res['steps'] * 1e3, 'seeds_per_s': res['seeds'] / res['elapsed'],
                         'hits_per_step': int(res['c']['n_hits']), 'ftab_len': int(ix.view.ftab_len) if int(ix.view.ftab_len) != psi_amd.NO_FTAB else 0,
                         'sa_rate': int(ix.view.sa_rate), 'tune': tune,
                         'k_fm_search': roofline_of(res, 'locus-table', 'k_fm_search', ix, key),
                         'k_fm_locate': roofline_of(res, 'locus-table', 'k_fm_locate', ix, key)}
            del ix
        rbm['fm-lf'] = lf
        out['roofline_by_mode'] = rbm
        # ---- what the tables of the default mode cost against what they save (review item: the headline's kernel looks up what
        # `prepare` computed): chunks of this size after which the k-mer table mode has paid for its tables, against traverse mode
        # (nothing about the loci tabulated) -- PSIGPU_MODE_AUTO / psikt --query-mode auto decide by the same arithmetic
        if 'traverse' in rbm and args.mode == 'kmer-table':
            t_tab = float(c['ms_locus_table_build'])
            saved = rbm['traverse']['ms_per_step'] - out['ms_per_step']
            out['break_even'] = {'table_build_ms': t_tab, 'prepare_wall_s': t_prep, 'ms_per_step_default': out['ms_per_step'],
                                 'ms_per_step_traverse': rbm['traverse']['ms_per_step'],
                                 'chunks': (t_tab / saved) if saved > 0 else None,
                                 'reads': (t_tab / saved * args.reads) if saved > 0 else None,
                                 'note': 'steady state is the metric (index load excluded, SURVEY 8d); a finder that answers fewer '
                                         'chunks than this is better off in traverse mode: psigpu_set_query_mode( PSIGPU_MODE_AUTO )'}

        # ---- psikt as a process on this configuration: the chunk loop a drop-in user runs (src/psikt.cpp:190-208), wall clock,
        # measured by THIS run (the child process at the top of main)
        if psikt_live and psikt_live.get('runs'):
            best = psikt_live['runs'].get('index from file, 2nd run') or {}
            first = psikt_live['runs'].get('index from file') or {}
            out['psikt_wall'] = {'source': 'live: tools/psikt_config1.py --live run by this bench process before its first GPU call '
                                           '(%d-read FASTQ, GFA graph, psikt -l 21 -n 1 -c %d -I ix; inputs in %s, records written to %s)'
                                           % (psikt_live.get('reads', 0), psikt_live.get('chunk', 0), psikt_live.get('dir', '?'), best.get('out_path', '?')),
                                 'reads': psikt_live.get('reads'), 'chunk': psikt_live.get('chunk'),
                                 'find_seeds_s': best.get('find_s'), 'find_seeds_s_per_1M_reads': best.get('find_s_per_1M_reads'),
                                 'reads_per_s': best.get('reads_per_s'), 'device_s': best.get('device_s'),
                                 'parse_pack_s_per_chunk': best.get('parse_pack_s_per_chunk'), 'breakdown_s': best.get('breakdown_s'),
                                 'call_ms_per_chunk': best.get('call_ms_per_chunk'),
                                 'process_wall_s_index_from_file': best.get('wall_s'), 'process_wall_s_first_run': first.get('wall_s'),
                                 'index_s_first_run': first.get('index_s'), 'hits': best.get('hits'), 'rc': best.get('rc'),
                                 'child_wall_s': psikt_live.get('child_wall_s'), 'inputs_s': psikt_live.get('inputs_s')}

        # ---- the other stand-ins (round 6): clustered variation and an HLA graph whose hot regions explode -- rows recorded by the
        # -m gpu tests of the same names (tests/test_gpu_parity.py::test_clustered_variation_stand_in / test_hla_hot_region_stand_in
        # write them; tools copy them to profiles/): what the tables cost there, what the walk cap leaves to the traverser, spills,
        # AUTO's decision, device time per mode
        rows = []
        for name in ('clustered', 'hla_hot'):
            sp = os.path.join(ROOT, 'profiles', '%s_standin_%s.json' % (PROFILE_ROUND, name))
            if os.path.exists(sp):
                try:
                    rows.append(json.load(open(sp)))
                except Exception as ex:
                    log('stand-in row %s not readable (%s)' % (sp, ex))
        if rows:
            out['stand_ins_recorded'] = rows

        # ---- CPU baseline + parity gate -------------------------------------------------------------
        if args.cpu_reads != 0:
            import oracle
            cores = oracle.lib().orc_max_threads()
            sample = args.cpu_reads if args.cpu_reads > 0 else min(args.reads, max(20_000, 125_000 * cores))
            og, pidx = oracle_objects(sg, px)
            base, cpu_hits = cpu_baseline(og, pidx, px, batches[0][0], batches[0][1], k, step, sample)
            out['cpu_baseline'] = base
            if not args.no_check:
                want = oracle.sort_unique(cpu_hits)
                ptr, n = finder.seeds_all_device(dev[0][0].data_ptr(), dev[0][1].data_ptr(), args.reads, dev[0][2], step=step,
                                                 rec_offset=0, stream=stream)
                got = psi_amd.sort_unique(finder.copy_hits(ptr, n))
                got = got[got[:, 2] < sample]
                ok_dev = bool(got.shape == want.shape and (got == want).all())
                # and the host entry point (pipeline + device sort-unique) returns the same set, in order
                su = finder.seeds_all(pinned_src[0], step=step, sort_unique=True)
                su = su[su[:, 2] < sample]
                ordered = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
                ok_host = bool(su.shape == ordered.shape and (su == ordered).all())
                # ... and the packed host entry (the end_to_end series)
                sp = finder.seeds_all_packed(packed_src[0], step=step, sort_unique=True)
                sp = sp[sp[:, 2] < sample]
                ok_packed = bool(sp.shape == ordered.shape and (sp == ordered).all())
                out['parity_vs_cpu_sample'] = ok_dev and ok_host and ok_packed
                out['parity_detail'] = {'device_entry': ok_dev, 'host_entry_sorted': ok_host, 'host_entry_packed_sorted': ok_packed,
                                        'reads_checked': int(sample), 'hits_checked': int(len(want))}
        else:
            out['cpu_baseline'] = None
    if rank == 0:
        if 'cpu_baseline' not in out:
            out['cpu_baseline'] = None
        emit(out, args)
    if cxx_stuck:
        # a thread is still inside RCCL: leave without tearing anything down under it -- and NOT as a success: the line
        # above is printed (it carries 'error': 'timed out'), the launcher and the driver see a failed rank, and the
        # peers that may still be blocked in ncclSend / ncclRecv are taken down by the launcher instead of lingering
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)
    finder.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
