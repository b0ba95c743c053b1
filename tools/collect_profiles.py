#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile.sh) -> the summaries kept under profiles/:
kernel stats, PMC summary and per-launch traffic of the query kernels, per query mode."""
import csv, glob, json, sys

def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    for tag, mode in (('k', 'kmer_table'), ('l', 'locus_table'), ('t', 'traverse'), ('t2', 'traverse_fm_route'),
                      ('f1', 'fm_lf_after_ftab'), ('f2', 'fm_lf_no_ftab'), ('f3', 'fm_lf_sa32')):
        d = 'gpurun_out/prof_%s' % tag
        try:
            t = json.load(open(d + '/traffic.json'))
        except OSError:
            continue
        keep = lambda k: k.startswith('k_') or k.startswith('void k_') or 'rocclr' in k
        t['per_launch'] = {k: v for k, v in t['per_launch'].items() if keep(k)}
        json.dump(t, open('profiles/%s_%s_traffic.json' % (rnd, tag), 'w'), indent=1)
        rows = [l for l in open(d + '/pmc_summary.csv') if l.startswith('kernel,') or keep(l)]
        open('profiles/%s_pmc_summary_%s.csv' % (rnd, mode), 'w').writelines(rows)
        import os
        f = max(glob.glob(d + '/stats/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)   # (gpurun_out keeps earlier runs)
        out = []
        for r in csv.reader(open(f)):
            if r[0] == 'Name' or int(r[1]) >= 3:
                r[0] = r[0].replace('(anonymous namespace)::', '')
                if len(r[0]) > 160:
                    r[0] = r[0][:160] + '...'
                out.append(r)
        csv.writer(open('profiles/%s_kernel_stats_%s.csv' % (rnd, mode), 'w')).writerows(out)
        print(tag, {k: {a: round(b / 1e6, 1) for a, b in v.items()} for k, v in t['per_launch'].items() if not 'rocclr' in k})

main()
