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


def write_fastq(bases, off, path):
    s = bases.tobytes().decode()
    q = 'I' * 150
    with open(path, 'w') as f:
        f.write(''.join('@r%d\n%s\n+\n%s\n' % (i, s[off[i]:off[i + 1]], q[:off[i + 1] - off[i]]) for i in range(len(off) - 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=1_000_000)
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--dir', default='/tmp/psikt_c1')
    ap.add_argument('--backbone', type=int, default=51_000_000)
    ap.add_argument('--snvs', type=int, default=1_100_000)
    ap.add_argument('--nblock', type=int, default=11_000_000)
    args = ap.parse_args()
    from psi_amd import synth
    os.makedirs(args.dir, exist_ok=True)
    gfa, fq = os.path.join(args.dir, 'graph.gfa'), os.path.join(args.dir, 'reads.fq')
    t = time.time()
    sg = synth.snv_graph(args.backbone, args.snvs, n_block=args.nblock, seed=11)
    bases, off = synth.sim_reads_snv(sg, args.reads, 150, seed=13)
    write_gfa(sg, gfa)
    write_fastq(bases, off.astype('int64').tolist(), fq)
    out = {'inputs_s': time.time() - t, 'gfa_bytes': os.path.getsize(gfa), 'fastq_bytes': os.path.getsize(fq),
           'nodes': sg.n_nodes, 'reads': args.reads}
    psikt = os.path.join(ROOT, 'psi_amd', 'bin', 'psikt')
    runs = {}
    for name, extra in (('patched (default)', ['-n', '1']), ('full paths (-P)', ['-n', '1', '-P']),
                        ('patched, 2nd run', ['-n', '1']), ('index from file', ['-n', '1', '-I', os.path.join(args.dir, 'ix')]),
                        ('index from file, 2nd run', ['-n', '1', '-I', os.path.join(args.dir, 'ix')])):
        log = os.path.join(args.dir, 'psi.log')
        if os.path.exists(log):
            os.remove(log)
        t = time.time()
        p = subprocess.run([psikt, gfa, '-f', fq, '-l', '21', '-o', os.path.join(args.dir, 'out.gam'), '-L', log,
                            '-c', str(args.chunk)] + extra, capture_output=True, text=True)
        wall = time.time() - t
        text = open(log).read() if os.path.exists(log) else ''
        r = {'rc': p.returncode, 'wall_s': wall, 'out_bytes': os.path.getsize(os.path.join(args.dir, 'out.gam'))}
        for key, pat in (('index_s', r'Created path index in ([0-9.]+) s'), ('find_s', r'Found seed in ([0-9.]+) s'),
                         ('device_s', r'\(([0-9.]+) s on the device\)'), ('hits', r'Total number of seeds found: (\d+)'),
                         ('reads_covered', r'Number of reads covered: (\d+)'), ('load_reads_s', r'bp in ([0-9.]+) s')):
            m = re.findall(pat, text)
            if m:
                r[key] = float(m[-1]) if '.' in m[-1] else int(m[-1])
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
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
