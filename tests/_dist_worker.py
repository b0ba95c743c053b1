"""Worker for tests/test_dist.py: world_size-2 gloo run of the read-sharding logic
(psi_amd/dist.py): contiguous read ranges, rec_offset = range start, variable-length gather of the
hit lists to rank 0.  Without a GPU (the CPU suite) each rank's chunk goes through the oracle (the
checker); with PSI_DIST_HIP=1 (the `-m gpu` test: both ranks on the one GPU of the box) it goes
through the HIP path, sort-unique on the device, and the gathered array must be the sorted chunk."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import brute  # noqa: E402
from psi_amd.dist import gather_hits, shard_range  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    data = os.path.join(ROOT, 'tests', 'golden')
    z = np.load(os.path.join(data, 'hits_x_reads_n1000l100e0i0_k20_d20.npz'))
    reads = [str(r) for r in z['reads']]
    k, step = int(z['k']), int(z['step'])
    g = brute.parse_gfa(os.path.join(data, 'ref_data', 'x.gfa'))
    og = oracle.OracleGraph.from_brute(g)
    pidx = oracle.OraclePathIndex(og, [[og.rank[v] for v in p] for _, p in g.paths])
    loci = brute.uncovered_loci(g, [p for _, p in g.paths], k)
    ln = np.array([og.rank[v] for v, _ in loci], np.uint64)
    lo = np.array([o for _, o in loci], np.uint64)
    b, e = shard_range(len(reads), rank, world)
    hip = os.environ.get('PSI_DIST_HIP') == '1'
    if hip:
        import psi_amd
        pg = psi_amd.Graph.load(os.path.join(data, 'ref_data', 'x.gfa'))
        f = psi_amd.SeedFinder(pg, k, device=0)
        f.create_path_index(2, patched=True)
        mine = f.seeds_all(reads[b:e], step=step, rec_offset=b, sort_unique=True)
        f.close()
    else:
        bases, off = oracle.pack_reads(reads[b:e])
        mine = oracle.seeds_all(og, pidx, bases, off, k, step, ln, lo, rec_offset=b)
    # rank 1 also checks the empty-shard case of the gather
    allhits = gather_hits(torch.from_numpy(mine.astype(np.int64)), dst=0)
    empty = gather_hits(torch.zeros((0, 4), dtype=torch.int64) if rank == 1 else torch.from_numpy(
        mine[:3].astype(np.int64)), dst=0)
    ok = True
    if rank == 0:
        raw = allhits.numpy().astype(np.uint64)
        got = np.unique(raw, axis=0)
        ok = got.shape == z['hits'].shape and bool((got == z['hits']).all())
        if hip:     # per-rank sorted shards of contiguous read ranges, in rank order = the sorted chunk
            want = z['hits'][np.lexsort((z['hits'][:, 1], z['hits'][:, 0], z['hits'][:, 3], z['hits'][:, 2]))]
            ok = ok and raw.shape == want.shape and bool((raw == want).all())
        ok = ok and empty.shape[0] == 3
        ranges = [shard_range(len(reads), r, world) for r in range(world)]
        ok = ok and ranges[0][0] == 0 and ranges[-1][1] == len(reads) and \
            all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        with open(out_path, 'w') as fh:
            fh.write('OK %d\n' % len(got) if ok else 'FAIL\n')
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
