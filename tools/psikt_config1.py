#!/usr/bin/env python3
"""psikt end to end on BASELINE.json configs[1]: the chr22-like synthetic graph written as GFA, 1 M x 150 bp
reads written as FASTQ, then `psikt graph.gfa -f reads.fq -l 21 -n 1 [-P]` timed as a process
(graph parsing, FASTQ parsing, index + tables, seed finding, writing 32-byte records), with the
per-phase times psikt logs.  Output: one JSON line.  Needs a GPU.

    python tools/psikt_config1.py [--reads 1000000] [--chunk 0] [--dir /tmp/psikt_c1]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_gfa(sg, path):
    import numpy as np
    lo = sg.label_off.astype(np.int64)
    lab = sg.labels.tobytes().decode()
    eo = sg.edge_off.astype(np.int64)
    with open(path, 'w') as f:
        f.write('H\tVN:Z:1.0\n')
        f.write(''.join('S\t%d\t%s\n' % (v + 1, lab[lo[v]:lo[v + 1]]) for v in range(sg.n_nodes)))
        src = np.repeat(np.arange(sg.n_nodes), np.diff(eo)) + 1
        dst = sg.edge_to.astype(np.int64) + 1
        f.write(''.join('L\t%d\t+\t%d\t+\t0M\n' % (a, b) for a, b in zip(src.tolist(), dst.tolist())))
        f.write('P\tref\t' + ','.join('%d+' % (v + 1) for v in sg.ref_path.tolist()) + '\t*\n')


def write_fastq(bases, off, path, first=0, mode='wb'):
    """Reads of ONE length as four-line records '@r<9 digits>' (vectorised: ten million reads are 3 GB of text)."""
    import numpy as np
    n = len(off) - 1
    L = int(off[1] - off[0]) if n else 0
    assert n == 0 or (np.diff(np.asarray(off, dtype=np.int64)) == L).all()
    with open(path, mode) as f:
        for a in range(0, n, 1 << 20):
            b = min(n, a + (1 << 20))
            m = b - a
            rec = np.empty((m, 11 + 1 + L + 3 + L + 1), np.uint8)
            ids = np.arange(first + a, first + b, dtype=np.int64)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r')
            for d in range(9):
                rec[:, 10 - d] = (ids // 10 ** d % 10 + 48).astype(np.uint8)
            rec[:, 11] = 10
            rec[:, 12:12 + L] = bases[int(off[a]):int(off[b])].reshape(m, L)
            rec[:, 12 + L:15 + L] = np.frombuffer(b'\n+\n', np.uint8)
            rec[:, 15 + L:15 + 2 * L] = ord('I')
            rec[:, -1] = 10
            f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=1_000_000)
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--dir', default='/tmp/psikt_c1')
    ap.add_argument('--backbone', type=int, default=51_000_000)
    ap.add_argument('--snvs', type=int, default=1_100_000)
    ap.add_argument('--nblock', type=int, default=11_000_000)
    ap.add_argument('--live', action='store_true',
                    help="bench.py's child process: two runs only (index made and saved, then loaded from file); inputs in /dev/shm when "
                         "there is room (the program is measured, not the disk)")
    args = ap.parse_args()
    if args.live and args.dir == '/tmp/psikt_c1':
        try:
            st = os.statvfs('/dev/shm')
            need = args.reads * 560 + (2 << 30)
            args.dir = '/dev/shm/psikt_live' if st.f_bavail * st.f_frsize > need else '/tmp/psikt_live'
        except OSError:
            args.dir = '/tmp/psikt_live'
    from psi_amd import synth
    os.makedirs(args.dir, exist_ok=True)
    gfa, fq = os.path.join(args.dir, 'graph.gfa'), os.path.join(args.dir, 'reads.fq')
    t = time.time()
    sg = synth.snv_graph(args.backbone, args.snvs, n_block=args.nblock, seed=11)
    write_gfa(sg, gfa)
    for a in range(0, args.reads, 2_000_000):          # (two million reads at a time: 10 M reads are 1.5 GB of bases + 3 GB of text)
        m = min(2_000_000, args.reads - a)
        bases, off = synth.sim_reads_snv(sg, m, 150, seed=13 + a // 2_000_000)
        write_fastq(bases, off.astype('int64'), fq, first=a, mode='wb' if a == 0 else 'ab')
    del bases, off
    out = {'inputs_s': time.time() - t, 'gfa_bytes': os.path.getsize(gfa), 'fastq_bytes': os.path.getsize(fq),
           'nodes': sg.n_nodes, 'reads': args.reads}
    psikt = os.path.join(ROOT, 'psi_amd', 'bin', 'psikt')
    runs = {}
    plan = (('patched (default)', ['-n', '1']), ('full paths (-P)', ['-n', '1', '-P']),
            ('patched, 2nd run', ['-n', '1']), ('index from file', ['-n', '1', '-I', os.path.join(args.dir, 'ix')]),
            ('index from file, 2nd run', ['-n', '1', '-I', os.path.join(args.dir, 'ix')]))
    if args.live:
        plan = (plan[3], plan[4])
    out['chunk'] = args.chunk
    out['dir'] = args.dir
    for name, extra in plan:
        log = os.path.join(args.dir, 'psi.log')
        if os.path.exists(log):
            os.remove(log)
        if args.live:             # (the records of the run before are 2 GB of dirty pages: their write-back is not this run's business)
            try:
                os.remove('/tmp/psikt_live_out.gam')
            except OSError:
                pass
            os.sync()
        t = time.time()
        # (--live: the records go to /tmp -- the box's disk-backed file system, i.e. the page cache -- as a user's out.gam would; the
        # inputs are read from /dev/shm.  tools/r06_psikt_writer_sweep.sh measured both: 0.33 s per 10 M reads' records on /tmp,
        # 0.56 s on /dev/shm, whose writes go at ~4 GB/s on these boxes whatever the number of writer threads)
        out_path = os.path.join('/tmp', 'psikt_live_out.gam') if args.live else os.path.join(args.dir, 'out.gam')
        p = subprocess.run([psikt, gfa, '-f', fq, '-l', '21', '-o', out_path, '-L', log,
                            '-c', str(args.chunk)] + extra, capture_output=True, text=True)
        wall = time.time() - t
        text = open(log).read() if os.path.exists(log) else ''
        r = {'rc': p.returncode, 'wall_s': wall, 'out_bytes': os.path.getsize(out_path), 'out_path': out_path}
        for key, pat in (('index_s', r'Created path index in ([0-9.]+) s'), ('find_s', r'Found seed in ([0-9.]+) s'),
                         ('device_s', r'\(([0-9.]+) s on the device\)'), ('hits', r'Total number of seeds found: (\d+)'),
                         ('reads_covered', r'Number of reads covered: (\d+)')):
            m = re.findall(pat, text)
            if m:
                r[key] = float(m[-1]) if '.' in m[-1] else int(m[-1])
        # (round 6) the loop's own breakdown, and what the reader's thread spent per chunk
        m = re.search(r'Seed loop breakdown: wait_reads ([0-9.]+) s, call ([0-9.]+) s, count ([0-9.]+) s, push ([0-9.]+) s, '
                      r'finish_write ([0-9.]+) s; parse\+pack \(reader thread\) ([0-9.]+) s', text)
        if m:
            r['breakdown_s'] = dict(zip(('wait_reads', 'call', 'count', 'push', 'finish_write', 'parse_pack_reader_thread'), map(float, m.groups())))
        m = re.findall(r'bp in ([0-9.]+) s \(waited ([0-9.]+) s\)', text)
        if m:
            r['chunks'] = len(m)
            r['parse_pack_s_per_chunk'] = sum(float(a) for a, _ in m) / len(m)
        m = re.findall(r'device time ([0-9.]+) ms, call ([0-9.]+) ms', text)
        if m:
            r['call_ms_per_chunk'] = [round(float(b), 2) for _, b in m]
        if r.get('find_s') and args.reads:
            r['find_s_per_1M_reads'] = r['find_s'] / (args.reads / 1e6)
            r['reads_per_s'] = args.reads / r['find_s']
        # seconds since process start at which each phase was reached (the log lines carry them)
        for key, pat in (('t_graph_loaded', r'\[\s*([0-9.]+)\] \[info\] Number of nodes'),
                         ('t_index_ready', r'\[\s*([0-9.]+)\] \[info\] Number of starting loci'),
                         ('t_reads_loaded', r'\[\s*([0-9.]+)\] \[info\] Fetched'),
                         ('t_done', r'\[\s*([0-9.]+)\] \[info\] Number of reads covered')):
            m = re.findall(pat, text)
            if m:
                r[key] = float(m[-1])
        if p.returncode:
            r['stderr'] = p.stderr[-500:]
        runs[name] = r
    out['runs'] = runs
    if args.live:
        import shutil
        shutil.rmtree(args.dir, ignore_errors=True)
        try:
            os.remove('/tmp/psikt_live_out.gam')
        except OSError:
            pass
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
