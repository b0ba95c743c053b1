"""Parity tests proper: the HIP path (through the C ABI) against the oracle and the committed
golden fixtures.  Bit-exact: hit sets are integer records and must be identical.
Run with `-m gpu` on an MI355X."""
import os
import zlib

import numpy as np
import pytest

import psi_amd
from psi_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
REF = os.path.join(GOLDEN, 'ref_data')


def _golden_files():
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith('hits_') and f.endswith('.npz'))


def _eq(a, b):
    return a.shape == b.shape and bool((a == b).all())


# Every test runs in every query mode: seeds answered from the k-mer table (the default), FM index
# + locus table, query-time traverser (the reference's scheme; paths by their k-mer table or the FM index), and the k-mer table with
# a walk cap of 1 so that every branching locus is left to the traverser and all three
# mechanisms contribute to one hit set.
@pytest.fixture(autouse=True, params=['kmer-table', 'locus-table', 'traverse', 'kmer-table-cap1'])
def query_mode(request, monkeypatch):
    mode = request.param
    monkeypatch.setenv('PSI_AMD_MODE', mode.replace('-cap1', ''))
    monkeypatch.setenv('PSI_AMD_WALK_CAP', '1' if mode.endswith('-cap1') else '0')
    # traverse mode answers the on-path phase from a table of the paths' k-mers, or (TUNE_NO_PATH_TABLE) from the
    # FM index as the reference does: every other test takes the second route
    if mode == 'traverse' and zlib.crc32(request.node.nodeid.encode()) & 1:
        monkeypatch.setenv('PSI_AMD_TUNE', str(psi_amd.TUNE_NO_PATH_TABLE))
    else:
        monkeypatch.delenv('PSI_AMD_TUNE', raising=False)
    return mode


_graphs = {}


def _graph(name):
    if name not in _graphs:
        _graphs[name] = psi_amd.Graph.load(os.path.join(REF, name))
    return _graphs[name]


# ---------------------------------------------------------------------------------------
# committed golden vectors (brute-force definition on the reference's own fixtures)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('fname', _golden_files())
@pytest.mark.parametrize('npaths', [0, 1, 3, -4])
def test_golden(fname, npaths):
    """npaths < 0: that many PATCHED paths (psikt's default indexing mode), context = k or k + 5."""
    z = np.load(os.path.join(GOLDEN, fname))
    reads = [str(r) for r in z['reads']]
    k, step = int(z['k']), int(z['step'])
    g = _graph(str(z['graph']))
    f = psi_amd.SeedFinder(g, k)
    patched, npaths = npaths < 0, abs(npaths)
    f.create_path_index(npaths, sa_rate=[1, 4, 32][npaths % 3], rng_seed=npaths, patched=patched,
                        context=(k + 5) * (k % 2) if patched else 0,
                        ftab_len=[0, psi_amd.NO_FTAB, 5, 12][(npaths + k) % 4])
    raw = f.seeds_all(reads, step=step)
    assert _eq(psi_amd.sort_unique(raw), z['hits'])
    c = f.counters()
    assert c['n_hits'] == len(raw) == c['n_hits_on_path'] + c['n_hits_off_path']
    if npaths == 0:
        assert c['n_hits_on_path'] == 0
    # the library's own sort-unique agrees (order: read_id, read_offset, node_id, node_offset)
    su = f.seeds_all(reads, step=step, sort_unique=True)
    want = z['hits'][np.lexsort((z['hits'][:, 1], z['hits'][:, 0], z['hits'][:, 3], z['hits'][:, 2]))]
    assert _eq(su, want)
    # ... by either route: ordering the hits of each seed in place, or the radix sort
    os.environ['PSIGPU_NO_GROUPED_SORT'] = '1'
    try:
        su2 = f.seeds_all(reads, step=step, sort_unique=True)
    finally:
        os.environ.pop('PSIGPU_NO_GROUPED_SORT')
    assert _eq(su2, want) and f.counters()['sorted_in_place'] == 0
    f.close()


def test_sort_unique_in_place_and_fallback(query_mode):
    """Hits come out seed by seed, so sort-unique is usually only the ordering of each seed's hits
    (HitSorter::fix_grouped, counted in sorted_in_place); a seed with more than 32 hits, a duplicate, or
    the traverser's unordered hits take the radix sort.  Same records either way."""
    g, reads = _x_case()
    k = 12
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3, patched=True)
    raw = f.seeds_all(reads[:500], step=5)
    want = psi_amd.sort_unique(raw)
    want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    su = f.seeds_all(reads[:500], step=5, sort_unique=True)
    c = f.counters()
    assert _eq(su, want)
    if query_mode.startswith('traverse'):
        assert c['sorted_in_place'] == 0
    elif len(raw) == len(want):                       # no duplicate in the raw stream: nothing but group order to fix
        assert c['sorted_in_place'] >= 1
    f.close()
    # a k-mer with more than 32 occurrences: the group is left to the radix sort
    labels = [('ACGT' * 60), 'TTGACCA', ('ACGT' * 50)]
    label_off = np.cumsum([0] + [len(x) for x in labels])
    rg = psi_amd.Graph.from_csr([5, 6, 9], label_off, ''.join(labels).encode(), [0, 1, 2, 2], [1, 2], paths=[[0, 1, 2]])
    f = psi_amd.SeedFinder(rg, k)
    f.create_path_index(1)
    rr = ['ACGTACGTACGTACGTAC', 'GACCAACGTACGTACGT', 'TTGACCAACGTAC']
    raw = f.seeds_all(rr, step=2)
    want = psi_amd.sort_unique(raw)
    want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    su = f.seeds_all(rr, step=2, sort_unique=True)
    assert len(raw) > 100 and _eq(su, want) and f.counters()['sorted_in_place'] == 0
    # the last read alone has one hit per seed: in place
    su = f.seeds_all(rr[2:], step=1, sort_unique=True)
    in_place = f.counters()['sorted_in_place']
    w1 = psi_amd.sort_unique(f.seeds_all(rr[2:], step=1))
    assert _eq(su, w1[np.lexsort((w1[:, 1], w1[:, 0], w1[:, 3], w1[:, 2]))])
    if not query_mode.startswith('traverse'):
        assert in_place == 1
    f.close()
@pytest.mark.parametrize('query_mode', ['traverse'], indirect=True)
def test_seed_table_builds_agree(monkeypatch, query_mode):
    """The traverser's per-chunk seed table is built bucket by bucket in LDS (seeds partitioned by their first six
    bases) for chunks of up to ~9 M seeds and in one region with atomics beyond; PSIGPU_SB_MAX=0 forces the second
    build on a small chunk: same records (golden), short seeds (k = 6: one bit of the prefix map per bucket; k = 9)
    included."""
    z = np.load(os.path.join(GOLDEN, 'hits_x_reads_n1000l100e0i0_k21_d1.npz'))
    reads = [str(r) for r in z['reads']]
    g = _graph(str(z['graph']))
    for sb_max in (None, '0', 'pb7'):
        if sb_max == 'pb7':               # the partition by SEVEN leading bases (chunks of 9 M .. 36 M seeds) on a small chunk
            monkeypatch.delenv('PSIGPU_SB_MAX')
            monkeypatch.setenv('PSIGPU_SB_PB', '7')
        elif sb_max is not None:
            monkeypatch.setenv('PSIGPU_SB_MAX', sb_max)
        f = psi_amd.SeedFinder(g, 21)
        f.create_path_index(1)
        assert _eq(psi_amd.sort_unique(f.seeds_all(reads, step=1)), z['hits'])
        f.close()
        from oracle import brute
        bg = brute.parse_gfa(os.path.join(REF, 'x.gfa'))
        for k in (4, 6, 9, 13):
            f = psi_amd.SeedFinder(g, k)
            f.create_path_index(0)
            got = psi_amd.sort_unique(f.seeds_all(reads[:60], step=k))
            want = np.array(brute.hit_set(bg, reads[:60], k, k), dtype=np.uint64).reshape(-1, 4)
            assert _eq(got, want)
            f.close()
    monkeypatch.delenv('PSIGPU_SB_PB')


def test_traverser_truth_table():
    """test/src/test_traverser.cpp:81-82 through the GPU traverser (no path index)."""
    truth = [(1, 0), (1, 1), (9, 4), (9, 17), (16, 0), (17, 0), (20, 0), (20, 31), (20, 38), (20, 38)]
    from oracle import brute
    reads = brute.read_seqs(os.path.join(REF, 'reads_n10l10e0i0.fastq'))
    for ext in ('x.gfa', 'x.vg'):
        f = psi_amd.SeedFinder(_graph(ext), 10)
        f.create_path_index(0)
        got = psi_amd.sort_unique(f.seeds_off_paths(reads, step=10))
        assert [tuple(h) for h in got.tolist()] == [(v, o, i, 0) for i, (v, o) in enumerate(truth)]
        f.close()


# ---------------------------------------------------------------------------------------
# phases, chunking, record offsets, edge cases
# ---------------------------------------------------------------------------------------
def _x_case():
    from oracle import brute
    reads = brute.read_seqs(os.path.join(REF, 'reads_n1000l100e0i0.seq'))
    return _graph('x.gfa'), reads


def test_phases_partition_and_chunking():
    g, reads = _x_case()
    k, step = 21, 7
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=4)
    full = psi_amd.sort_unique(f.seeds_all(reads, step=step))
    on = f.seeds_on_paths(reads, step=step)
    off = f.seeds_off_paths(reads, step=step)
    assert len(on) and len(off)
    assert _eq(psi_amd.sort_unique(np.concatenate([on, off])), full)
    # psikt's chunk loop: read ids are global across chunks (sequence.hpp:1616)
    parts = []
    for s in range(0, len(reads), 300):
        parts.append(f.seeds_all(reads[s:s + 300], step=step, rec_offset=s))
    assert _eq(psi_amd.sort_unique(np.concatenate(parts)), full)
    # step 0 means step = k (psikt.cpp:469)
    assert _eq(psi_amd.sort_unique(f.seeds_all(reads, step=0)),
               psi_amd.sort_unique(f.seeds_all(reads, step=k)))
    f.close()


def test_empty_ragged_and_n_reads():
    g, reads = _x_case()
    k = 12
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1)
    assert f.seeds_all([], step=k).shape == (0, 4)
    assert f.seeds_all(['', 'ACG', 'N' * 40], step=1).shape == (0, 4)
    c = f.counters()
    assert c['n_seeds'] == 29 and c['n_seeds_valid'] == 0
    base = reads[:50]
    ragged = []
    for i, r in enumerate(base):
        ragged += [r, '', r[:k - 1], r[:k], 'N' + r[1:], r[:30] + 'N' + r[31:]]
    from oracle import brute
    bg = brute.parse_gfa(os.path.join(REF, 'x.gfa'))
    want = np.array(brute.hit_set(bg, ragged, k, 5), dtype=np.uint64).reshape(-1, 4)
    assert _eq(psi_amd.sort_unique(f.seeds_all(ragged, step=5)), want)
    f.close()


def test_argument_errors():
    g, reads = _x_case()
    f = psi_amd.SeedFinder(g, 12)
    with pytest.raises(psi_amd.PsiGpuError):
        f.seeds_all(reads[:3])                 # no index loaded
    f.create_path_index(1)
    f2 = psi_amd.SeedFinder(g, 14)
    f2.set_path_index(f.pindex)                # loci were computed for k = 12
    with pytest.raises(psi_amd.PsiGpuError):
        f2.seeds_all(reads[:3])
    assert len(f2.seeds_on_paths(reads[:3]))   # the FM-index itself is k-independent
    with pytest.raises(psi_amd.PsiGpuError):
        psi_amd.SeedFinder(g, 64)
    f.close()
    f2.close()


# ---------------------------------------------------------------------------------------
# against the oracle (C restatement of the reference path) on seeded synthetic inputs
# ---------------------------------------------------------------------------------------
def _oracle_hits(graph_arrays, finder, bases, off, k, step, gocc=0, threads=2):
    import oracle
    node_id, label_off, labels, edge_off, edge_to = graph_arrays
    og = oracle.OracleGraph(node_id, label_off, bytes(labels), edge_off, edge_to.astype(np.uint64))
    paths = [p.tolist() for p in finder.pindex.paths()]
    # (head offset, tail length) -> the reference's Path::left / right: lengths of the first / last
    # node that belong to the path, 0 = all
    lo64 = np.asarray(label_off, dtype=np.int64)
    left = [int(lo64[p[0] + 1] - lo64[p[0]]) - h if h else 0 for p, (h, _) in zip(paths, finder.pindex.trims())]
    right = [t for _, t in finder.pindex.trims()]
    pidx = oracle.OraclePathIndex(og, paths, left=left, right=right) if paths else None
    ln, lo = finder.get_starting_loci()
    h = oracle.seeds_all(og, pidx, bytes(bases), off, k, step, ln, lo, gocc_thr=gocc, threads=threads)
    return oracle.sort_unique(h)


@pytest.mark.parametrize('k,step,npaths,err', [(21, 21, 1, 0.0), (21, 1, 2, 0.01), (31, 31, 4, 0.0),
                                               (11, 11, 1, 0.0), (16, 5, 0, 0.0), (21, 21, -3, 0.0), (25, 7, -6, 0.01)])
def test_snv_graph_vs_oracle(k, step, npaths, err):
    """npaths < 0: patched paths -- the oracle indexes the same trimmed paths (head offsets and all)."""
    sg = synth.snv_graph(300_000, 9_000, n_block=20_000, seed=k)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    n_reads = 3000 if npaths else 400
    bases, off = synth.sim_reads_snv(sg, n_reads, 150, seed=k + 1, sub_rate=err)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(abs(npaths), rng_seed=2, patched=npaths < 0, context=k + 3 if npaths < 0 else 0)
    if npaths < 0:
        assert len(f.pindex.paths()) > -npaths and any(h for h, _ in f.pindex.trims())
    got = psi_amd.sort_unique(f.seeds_all((bases, off), step=step))
    want = _oracle_hits((sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to), f, bases, off,
                        k, step)
    assert len(want) > n_reads
    assert _eq(got, want)
    f.close()


@pytest.mark.parametrize('k,step,ftab_len,sa_rate,tail', [
    (21, 21, 0, 1, 3), (21, 21, psi_amd.NO_FTAB, 1, 3), (21, 7, 9, 4, 0), (31, 31, psi_amd.NO_FTAB, 32, 3), (31, 9, 13, 1, 1),
    (13, 13, 13, 1, 3), (9, 4, psi_amd.NO_FTAB, 1, 3), (5, 5, psi_amd.NO_FTAB, 4, 3), (25, 25, 4, 32, 2), (30, 30, 6, 1, 0)])
def test_level_synchronous_search_equals_the_quad_search(query_mode, k, step, ftab_len, sa_rate, tail):
    """Every LF step of every seed (index_iter.hpp:835-841 -> fmindex.hpp:851-869) by k_fm_sweep -- seeds bucketed by interval,
    rank blocks staged in LDS, rounds of five steps -- equals the quad-per-seed search of rounds 1-5 (TUNE_NO_SWEEP) and the
    oracle, with and without an interval table, whole and sampled suffix array, seeds longer than a record's sixteen
    characters (refill), seeds no longer than the table's q-mers (no step at all), and with a gocc threshold."""
    if query_mode != 'locus-table':
        pytest.skip('the FM search answers the on-path phase in locus-table mode')
    sg = synth.snv_graph(200_000, 6_000, n_block=20_000, seed=k + 100)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 2500 if k >= 9 else 120, 150, seed=k + 1, sub_rate=0.005)      # (a 5-mer occurs everywhere: few reads)
    ix = psi_amd.PathIndex.build(g, k, 2, rng_seed=5, ftab_len=ftab_len, sa_rate=sa_rate)
    res = {}
    for name, tune in (('sweep', psi_amd.TUNE_NO_DIRECT | psi_amd.TUNE_NO_VERIFY),
                       ('quad', psi_amd.TUNE_NO_DIRECT | psi_amd.TUNE_NO_VERIFY | psi_amd.TUNE_NO_SWEEP)):
        f = psi_amd.SeedFinder(g, k, mode='locus-table')
        f.set_tuning(tune)
        f.set_option('sweep_tail', tail)
        f.set_path_index(ix)
        raw = f.seeds_all((bases, off), step=step)
        c = f.counters()
        thr_hits = None
        f.set_gocc_threshold(2)
        thr_hits = psi_amd.sort_unique(f.seeds_on_paths((bases, off), step=step))
        f.set_gocc_threshold(0)
        res[name] = (psi_amd.sort_unique(raw), len(raw), c, thr_hits)
        if name == 'sweep':
            want = _oracle_hits((sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to), f, bases, off, k, step)
        f.close()
    assert len(want) > 2000 and _eq(res['sweep'][0], want) and _eq(res['quad'][0], want)
    assert res['sweep'][1] == res['quad'][1]                               # the raw streams have one length
    assert _eq(res['sweep'][3], res['quad'][3]) and len(res['sweep'][3]) <= len(want)
    cs, cq = res['sweep'][2], res['quad'][2]
    assert cs['n_seeds_on_path'] == cq['n_seeds_on_path'] and cs['n_hits_on_path'] == cq['n_hits_on_path']
    assert cs['search_launches'] >= 6 and cq['search_launches'] == 1       # rounds of five launches + the totals, against one kernel
    if k > 13 or (ftab_len == psi_amd.NO_FTAB and k > 6):
        assert cs['n_lf_steps'] > 0


@pytest.mark.parametrize('seed', range(4))
def test_layered_graph_vs_brute(seed):
    """dense random DAGs with N bases, out-degree up to 4 and short nodes: stresses the
    traverser's forks and the LDS stack spill path."""
    from oracle import brute
    nid, lo, lab, eo, et, ref = synth.layered_graph(120, max_width=4, max_len=5, seed=seed, p_n=0.01)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    bg = brute.Graph()
    for i in range(len(nid)):
        bg.add_node(int(nid[i]), bytes(lab[int(lo[i]):int(lo[i + 1])]).decode())
    for i in range(len(nid)):
        for e in range(int(eo[i]), int(eo[i + 1])):
            bg.add_edge(int(nid[i]), int(nid[et[e]]))
    import random
    rng = random.Random(seed)
    reads = []
    for _ in range(200):
        v = rng.choice(bg.ids)
        s = bg.seq[v][rng.randrange(len(bg.seq[v])):]
        while len(s) < 40 and bg.out[v]:
            v = rng.choice(bg.out[v])
            s += bg.seq[v]
        reads.append(s[:40])
    for k, npaths in ((12, 0), (12, 2), (17, 1)):
        want = np.array(brute.hit_set(bg, reads, k, 3), dtype=np.uint64).reshape(-1, 4)
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(npaths, rng_seed=seed)
        got = psi_amd.sort_unique(f.seeds_all(reads, step=3))
        assert _eq(got, want)
        f.close()


def test_spill_path_is_exercised(query_mode):
    """k = 31 on a wide, short-node DAG floods the per-wave LDS stack; results must not change."""
    from oracle import brute
    nid, lo, lab, eo, et, ref = synth.layered_graph(400, max_width=4, max_len=2, seed=9, p_edge=1.0)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    bg = brute.Graph()
    for i in range(len(nid)):
        bg.add_node(int(nid[i]), bytes(lab[int(lo[i]):int(lo[i + 1])]).decode())
    for i in range(len(nid)):
        for e in range(int(eo[i]), int(eo[i + 1])):
            bg.add_edge(int(nid[i]), int(nid[et[e]]))
    import random
    rng = random.Random(1)
    reads = []
    for _ in range(60):
        v = bg.ids[rng.randrange(len(bg.ids) // 2)]
        s = bg.seq[v]
        while len(s) < 50 and bg.out[v]:
            v = rng.choice(bg.out[v])
            s += bg.seq[v]
        reads.append(s[:50])
    k = 14
    want = np.array(brute.hit_set(bg, reads, k, 6), dtype=np.uint64).reshape(-1, 4)
    f = psi_amd.SeedFinder(g, k)
    f.set_option('no_pfx_roots', 1)           # from the loci themselves: every walk enters the loop and the stack floods
    f.create_path_index(0)
    got = psi_amd.sort_unique(f.seeds_all(reads, step=6))
    c = f.counters()
    assert _eq(got, want)
    if query_mode == 'traverse':
        assert c['n_spilled'] > 0 and c['traverse_launches'] > 1
        assert c['n_locus_kmers'] == 0 and c['n_loci_traversed'] == c['n_loci']
    # ... and from the loci's prefix walks (the default since round 4: the chunk's 12-mer map prunes them while they are staged)
    f.set_option('no_pfx_roots', 0)
    assert _eq(psi_amd.sort_unique(f.seeds_all(reads, step=6)), want)
    f.close()
    # k = 31: hundreds of millions of k-walks per locus, far over every cap and every budget of the
    # table construction -- in every mode these loci are traversed per chunk, pruned by the seeds
    # (the brute-force definition enumerates them all; the reference here is the oracle's pruned traverser)
    k = 31
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=9)
    got = psi_amd.sort_unique(f.seeds_all(reads, step=6))
    c = f.counters()
    want = _oracle_hits((nid, lo, lab, eo, et), f, np.frombuffer(''.join(reads).encode(), np.uint8),
                        np.arange(0, 50 * len(reads) + 1, 50, dtype=np.uint64), k, 6)
    assert _eq(got, want) and len(want) > 0
    assert c['n_loci_traversed'] > 0.9 * c['n_loci'] and c['traverse_launches'] >= 1
    f.close()


def test_query_modes_split_the_work(query_mode):
    """The tables answer without traversing; a walk cap of 1 leaves exactly the branching loci to
    the traverser; the k-mer table replaces the FM search; the hit sets are the same."""
    sg = synth.snv_graph(80_000, 2500, seed=21)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 500, 100, seed=22)
    k = 21
    px = psi_amd.PathIndex.build(g, k, 1, rng_seed=1)
    ref = psi_amd.SeedFinder(g, k, mode='traverse')
    ref.set_path_index(px)
    want = psi_amd.sort_unique(ref.seeds_all((bases, off), step=5))
    want_on = psi_amd.sort_unique(ref.seeds_on_paths((bases, off), step=5))
    n_walks = None
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(px)
    f.set_tuning(0)
    for rep in range(2):                        # second call reuses the tables
        got = psi_amd.sort_unique(f.seeds_all((bases, off), step=5))
        assert _eq(got, want)
        c = f.counters()
        if query_mode == 'traverse':
            assert c['n_locus_kmers'] == 0 and c['n_loci_traversed'] == c['n_loci'] and c['traverse_launches'] >= 1
        elif query_mode.endswith('cap1'):
            assert 0 < c['n_loci_traversed'] < c['n_loci'] and c['n_locus_kmers'] == c['n_loci'] - c['n_loci_traversed']
        else:
            assert c['n_locus_kmers'] >= c['n_loci'] and c['n_loci_traversed'] == 0 and c['traverse_launches'] == 0
        if query_mode.startswith('kmer-table') or query_mode == 'traverse':      # (traverse mode: a table of the paths' k-mers only)
            assert c['n_path_kmers'] > 0 and c['search_launches'] == 0 and c['n_lf_steps'] == 0
        else:
            assert c['n_path_kmers'] == 0 and c['search_launches'] >= 1
        if n_walks is not None:
            assert c['n_locus_kmers'] == n_walks
        n_walks = c['n_locus_kmers']
    if query_mode == 'traverse':
        # the reference's scheme as written -- FM index for the on-path phase -- on request; and back
        for flags, table in ((psi_amd.TUNE_NO_PATH_TABLE, False), (0, True)):
            f.set_tuning(flags)
            assert _eq(psi_amd.sort_unique(f.seeds_all((bases, off), step=5)), want)
            c = f.counters()
            assert (c['n_path_kmers'] > 0) == table and (c['search_launches'] == 0) == table and c['n_loci_traversed'] == c['n_loci']
    # the phases alone, and switching the mode on a live finder
    assert _eq(psi_amd.sort_unique(f.seeds_on_paths((bases, off), step=5)), want_on)
    a = psi_amd.sort_unique(f.seeds_off_paths((bases, off), step=5))
    f.set_query_mode('traverse')
    b = psi_amd.sort_unique(f.seeds_off_paths((bases, off), step=5))
    assert _eq(a, b)
    f.close()
    ref.close()


def test_dense_sites_get_a_second_enumeration_pass(query_mode):
    """SNVs every ~2 bp: hundreds to thousands of k-walks per locus, more than the default walk cap.
    Few such loci are enumerated again with a larger cap instead of bringing the per-chunk traverser
    back; the hit set is the oracle's either way."""
    sg = synth.snv_graph(3_000, 1_300, seed=77)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    k = 21
    bases, off = synth.sim_reads_snv(sg, 300, 60, seed=78)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=3)
    got = psi_amd.sort_unique(f.seeds_all((bases, off), step=3))
    c = f.counters()
    want = _oracle_hits((sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to), f, bases, off, k, 3)
    assert _eq(got, want)
    if query_mode in ('kmer-table', 'locus-table'):
        assert c['n_locus_kmers'] > 256 * 100 and c['n_loci_traversed'] == 0
    elif query_mode == 'kmer-table-cap1':
        assert c['n_loci_traversed'] > 0          # an explicit cap is honoured: no second pass
    f.close()


def test_very_dense_sites_get_a_third_pass(query_mode):
    """Two one-base alleles at every one of 48 positions, all connected: 2^20 k-walks from a locus
    at k = 21, more than the second pass' cap; such loci (a few dozen) are enumerated a third
    time (cap 2^22) rather than left to the per-chunk traverser."""
    from oracle import brute
    import random
    rng = random.Random(4)
    n_layers = 48
    bg = brute.Graph()
    for layer in range(n_layers):
        a, b = rng.sample('ACGT', 2)
        bg.add_node(2 * layer + 1, a)
        bg.add_node(2 * layer + 2, b)
    for layer in range(n_layers - 1):
        for u in (2 * layer + 1, 2 * layer + 2):
            for v in (2 * layer + 3, 2 * layer + 4):
                bg.add_edge(u, v)
    rank = {v: i for i, v in enumerate(bg.ids)}
    label_off = np.arange(len(bg.ids) + 1)
    labels = ''.join(bg.seq[v] for v in bg.ids).encode()
    edge_off = np.cumsum([0] + [len(bg.out[v]) for v in bg.ids])
    edge_to = [rank[t] for v in bg.ids for t in bg.out[v]]
    ref = [rank[2 * layer + 1] for layer in range(n_layers)]
    g = psi_amd.Graph.from_csr(bg.ids, label_off, labels, edge_off, edge_to, paths=[ref])
    reads = []
    for _ in range(40):
        layer = rng.randrange(n_layers - 30)
        reads.append(''.join(bg.seq[2 * (layer + i) + rng.choice((1, 2))] for i in range(30)))
    k = 21
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=1)
    got = psi_amd.sort_unique(f.seeds_all(reads, step=2))
    c = f.counters()
    # (the brute-force definition walks all 2^20 k-walks per seed start: the oracle's pruned traverser instead)
    want = _oracle_hits((np.array(bg.ids), label_off, np.frombuffer(labels, np.uint8), edge_off, np.array(edge_to)), f,
                        np.frombuffer(''.join(reads).encode(), np.uint8),
                        np.arange(0, 30 * len(reads) + 1, 30, dtype=np.uint64), k, 2)
    assert _eq(got, want) and len(want) > 0
    if query_mode in ('kmer-table', 'locus-table'):
        assert c['n_loci_traversed'] == 0 and c['n_locus_kmers'] > 20 * (1 << 20)
    f.close()


def test_gocc_threshold_vs_oracle():
    """-r T: on-path k-mers with more than T path occurrences are skipped, the traverser is
    not thresholded (index_iter.hpp:843-847)."""
    sg = synth.snv_graph(60_000, 1500, seed=3)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    k = 8                                            # short seeds: many repeats on the path
    bases, off = synth.sim_reads_snv(sg, 300, 60, seed=5)
    for thr in (1, 3):
        f = psi_amd.SeedFinder(g, k, gocc_threshold=thr)
        f.create_path_index(1)
        got = psi_amd.sort_unique(f.seeds_on_paths((bases, off), step=k))
        import oracle
        og = oracle.OracleGraph(sg.node_id, sg.label_off, bytes(sg.labels), sg.edge_off,
                                sg.edge_to.astype(np.uint64))
        pidx = oracle.OraclePathIndex(og, [p.tolist() for p in f.pindex.paths()])
        none = np.zeros(0, np.uint64)
        want = oracle.sort_unique(oracle.seeds_all(og, pidx, bytes(bases), off, k, k, none, none,
                                                   gocc_thr=thr, phases=1))
        assert _eq(got, want)
        f.close()
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1)
    assert len(f.seeds_on_paths((bases, off), step=k)) > len(got)
    f.close()


def _graph_arrays(g):
    return (np.asarray(g.node_id), np.asarray(g.label_off), np.frombuffer(bytes(g.labels), np.uint8),
            np.asarray(g.edge_off), np.asarray(g.edge_to))


@pytest.mark.parametrize('k,npaths', [(8, 1), (10, 4), (12, -4), (9, -4)])
@pytest.mark.parametrize('thr', [1, 2, 3])
def test_gocc_threshold_all_phases_vs_oracle_x(k, npaths, thr):
    """psikt -r T with seeds_all (BOTH phases): an on-path k-mer with more than T occurrences in the
    path text is skipped by seeds_on_paths (index_iter.hpp:826-847) while the traverser is not
    thresholded -- so a starting locus at an on-path position still gives that hit.  The product
    leaves such loci out of the tables when the on-path occurrences are emitted and must keep them
    for the k-mers over the threshold (KT_OFFDUP): compared with the oracle's seeds_all( gocc_thr = T )
    on the reference's graph x, 1 / 4 full / 4 patched paths (npaths < 0), threshold given to the
    constructor (before the tables exist) and set after prepare() (tables rebuilt)."""
    g, reads = _x_case()
    reads = reads[:400]
    bases, off = psi_amd.pack_reads(reads)
    step = 3
    base = psi_amd.SeedFinder(g, k)
    base.create_path_index(abs(npaths), rng_seed=7, patched=npaths < 0, context=k + 2 if npaths < 0 else 0)
    px = base.pindex
    want0 = _oracle_hits(_graph_arrays(g), base, bases, off, k, step)
    want = _oracle_hits(_graph_arrays(g), base, bases, off, k, step, gocc=thr)
    assert len(want) <= len(want0) and (thr > 1 or len(want) < len(want0))      # the threshold bites
    base.close()
    # threshold known before the tables are made
    f = psi_amd.SeedFinder(g, k, gocc_threshold=thr)
    f.set_path_index(px)
    f.prepare()
    assert _eq(psi_amd.sort_unique(f.seeds_all((bases, off), step=step)), want)
    f.close()
    # threshold set after prepare(): the tables were made without one
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(px)
    f.prepare()
    assert _eq(psi_amd.sort_unique(f.seeds_all((bases, off), step=step)), want0)
    f.set_gocc_threshold(thr)
    assert _eq(psi_amd.sort_unique(f.seeds_all((bases, off), step=step)), want)
    su = f.seeds_all((bases, off), step=step, sort_unique=True)
    assert _eq(su, want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))])
    f.set_gocc_threshold(0)
    assert _eq(psi_amd.sort_unique(f.seeds_all((bases, off), step=step)), want0)
    f.close()


@pytest.mark.parametrize('npaths', [1, 3, -3])
def test_gocc_threshold_all_phases_vs_oracle_snv(npaths):
    """The same on the synthetic SNV graph (60 kbp, 1500 bubbles), k = 8: nearly every k-mer repeats on the
    path, and the uncovered loci around the bubbles sit at on-path positions."""
    sg = synth.snv_graph(60_000, 1500, seed=3)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    arrays = (sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to)
    k = 8
    bases, off = synth.sim_reads_snv(sg, 300, 60, seed=5)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(abs(npaths), rng_seed=5, patched=npaths < 0, context=k + 4 if npaths < 0 else 0)
    f.prepare()
    prev = None
    for thr in (1, 2, 3, 0):
        f.set_gocc_threshold(thr)
        got = psi_amd.sort_unique(f.seeds_all((bases, off), step=k))
        want = _oracle_hits(arrays, f, bases, off, k, k, gocc=thr)
        assert _eq(got, want)
        if prev is not None:
            assert len(want) >= prev
        prev = len(want)
    f.close()


# ---------------------------------------------------------------------------------------
# device-resident entry point, and size-independent properties at a larger size
# ---------------------------------------------------------------------------------------
def test_device_hits_as_a_torch_tensor():
    """psi_amd.DeviceHits: the records left in HBM seen by torch through the CUDA array interface, no copy."""
    import torch
    g, reads = _x_case()
    f = psi_amd.SeedFinder(g, 12)
    f.create_path_index(1)
    bases, off = psi_amd.pack_reads(reads[:200])
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    ptr, n = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 200, len(bases), step=5, flags=psi_amd.ALL | psi_amd.SORT_UNIQUE)
    t = torch.as_tensor(psi_amd.DeviceHits(ptr, n), device='cuda')
    assert t.shape == (n, 4) and t.dtype == torch.int64 and t.data_ptr() == ptr
    assert _eq(t.cpu().numpy().view(np.uint64), f.copy_hits(ptr, n))
    f.close()


def test_cxx_gather_of_hit_lists_one_rank():
    """psigpu_gather_hits (C++ over RCCL, loaded by the library): a world of one rank -- the communicator comes up,
    the counts are all-gathered, the root's own records land in the gathered buffer unchanged, twice (the buffer is
    reused) and with an empty list."""
    import torch
    g, reads = _x_case()
    f = psi_amd.SeedFinder(g, 12)
    f.create_path_index(1)
    bases, off = psi_amd.pack_reads(reads[:300])
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    ptr, n = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 300, len(bases), step=5, flags=psi_amd.ALL | psi_amd.SORT_UNIQUE)
    want = f.copy_hits(ptr, n)
    assert psi_amd.HitGather.available()
    hg = psi_amd.HitGather(0, psi_amd.HitGather.unique_id(), 0, 1)
    for _ in range(2):
        d_all, n_all, counts = hg.gather(ptr, n, 0)
        assert n_all == n and list(counts) == [n] and d_all not in (0, ptr)
        assert _eq(f.copy_hits(d_all, n_all), want)
    d_all, n_all, counts = hg.gather(0, 0, 0)
    assert n_all == 0 and list(counts) == [0]
    with pytest.raises(psi_amd.PsiGpuError):
        hg.gather(ptr, n, 3)                        # no such root
    hg.close()
    f.close()


def test_device_resident_entry_matches_host_entry():
    import torch
    sg = synth.snv_graph(200_000, 6000, seed=8)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 5000, 150, seed=9)
    k = 21
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2)
    host = psi_amd.sort_unique(f.seeds_all((bases, off), step=k, rec_offset=77))
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    ptr, n = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), len(off) - 1, len(bases),
                                step=k, rec_offset=77, stream=stream)
    c = f.counters()
    assert n == c['n_hits'] and c['ms_total'] > 0
    assert _eq(psi_amd.sort_unique(f.copy_hits(ptr, n)), host)
    f.close()


def test_properties_at_scale():
    """Checks that need no oracle run: every error-free read is found at its own origin
    (sensitivity), every reported hit spells its seed in the graph (specificity), and the
    answer does not depend on how many paths are indexed."""
    L, nsnv, n_reads, k = 5_000_000, 110_000, 100_000, 21
    sg = synth.snv_graph(L, nsnv, n_block=500_000, seed=21)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                               paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, n_reads, 150, seed=22)
    res = []
    for npaths in (1, 4):
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(npaths, rng_seed=1)
        res.append(psi_amd.sort_unique(f.seeds_all((bases, off), step=k)))
        f.close()
    assert _eq(res[0], res[1])
    hits = res[0]
    # sensitivity: all 7 seeds of every read have at least one hit
    seeds_hit = np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])
    assert len(seeds_hit) == n_reads * 7
    # specificity on a sample: the first base of the hit's node offset equals the seed's first base,
    # and single-node hits spell the whole seed
    lo = sg.label_off.astype(np.int64)
    rank = hits[:, 0].astype(np.int64) - 1
    first = sg.labels[lo[rank] + hits[:, 1].astype(np.int64)]
    seed_first = bases[(hits[:, 2] * np.uint64(150) + hits[:, 3]).astype(np.int64)]
    assert (first == seed_first).all()
    inside = (lo[rank + 1] - lo[rank] - hits[:, 1].astype(np.int64)) >= k
    idx = np.flatnonzero(inside)[:5000]
    for i in idx:
        a = lo[rank[i]] + int(hits[i, 1])
        b = int(hits[i, 2]) * 150 + int(hits[i, 3])
        assert bytes(sg.labels[a:a + k]) == bytes(bases[b:b + k])


@pytest.mark.parametrize('query_mode', ['kmer-table'], indirect=True)
def _independent_windows(sg, px, finder, k, n_windows, wlen, lo_pos, hi_pos, seed=7, reads_per_window=300):
    """A checker that takes NOTHING from the product (round-4 review: every oracle run fed paths, trims, loci and the
    suffix array of the finder under test, so a locus the product fails to detect was missing on both sides).  On
    `n_windows` random windows of `wlen` backbone bases: the window's subgraph as a brute-force Graph; its starting loci by
    the brute-force DEFINITION (oracle/brute.py uncovered_loci: every k-walk that is not a run of the path -- the intent of
    seed_finder.hpp:1481-1541, pinned by test_seedfinder.cpp:98-163 in tests/test_oracle_golden.py); the oracle's path
    index over the window with its OWN suffix array (ext_sa = None); reads drawn from the window.  Compared, away from
    the window's cut ends: (a) the product's loci inside the window with the definition's, (b) the product's records for
    those reads -- the finder over the WHOLE graph -- with the oracle's over the window."""
    import oracle
    from oracle import brute
    rng = np.random.default_rng(seed)
    snv_pos = np.flatnonzero(sg.alt)
    lab = lambda pos: int(pos) + int(np.searchsorted(snv_pos, pos, side='left'))           # noqa: E731  label coordinate of a backbone position
    lo = sg.label_off.astype(np.int64)
    eo = sg.edge_off.astype(np.int64)
    ln, lc = px.loci
    path = px.paths()[0].astype(np.int64)
    margin = 64
    n_loci_checked = n_hits_checked = 0
    for w0 in rng.integers(lo_pos, hi_pos - wlen, size=n_windows):
        w0 = int(w0)
        v0 = int(np.searchsorted(sg.label_off, np.uint64(lab(w0)), side='left'))
        v1 = int(np.searchsorted(sg.label_off, np.uint64(lab(w0 + wlen)), side='left'))
        labels = bytes(sg.labels[int(lo[v0]):int(lo[v1])]).decode()
        bg = brute.Graph()
        for v in range(v0, v1):
            bg.add_node(int(sg.node_id[v]), labels[int(lo[v] - lo[v0]):int(lo[v + 1] - lo[v0])])
        for v in range(v0, v1):
            bg.out[int(sg.node_id[v])] = [int(sg.node_id[t]) for t in sg.edge_to[eo[v]:eo[v + 1]].tolist() if v0 <= t < v1]
        sub_path = path[(path >= v0) & (path < v1)]
        # (a) the loci
        want_loci = brute.uncovered_loci(bg, [[int(sg.node_id[r]) for r in sub_path]], k)
        id_lo, id_hi = int(sg.node_id[v0 + margin]), int(sg.node_id[v1 - margin])
        want_in = [(v, o) for v, o in want_loci if id_lo <= v < id_hi]
        a, b = np.searchsorted(ln, [v0 + margin, v1 - margin])
        got_in = [(int(sg.node_id[v]), int(o)) for v, o in zip(ln[a:b].tolist(), lc[a:b].tolist())]
        assert got_in == want_in, 'starting loci differ from the brute-force definition in the window at %d' % w0
        n_loci_checked += len(want_in)
        # (b) the records: reads from inside the window, each SNV's allele at random
        start = rng.integers(w0 + 2500, w0 + wlen - 2700, size=reads_per_window)      # (inside the margins: 64 nodes of <= 32 bases)
        idx = start[:, None] + np.arange(150)[None, :]
        r, al = sg.backbone[idx], sg.alt[idx]
        wb = np.ascontiguousarray(np.where((al != 0) & (rng.random(size=idx.shape) < 0.5), al, r).reshape(-1))
        woff = np.arange(reads_per_window + 1, dtype=np.uint64) * np.uint64(150)
        og = oracle.OracleGraph.from_brute(bg)
        pidx = oracle.OraclePathIndex(og, [(sub_path - v0).tolist()])                 # (its own suffix array)
        rank = {v: i for i, v in enumerate(bg.ids)}
        o_ln = np.array([rank[v] for v, _ in want_loci], np.uint32)
        o_lo = np.array([o for _, o in want_loci], np.uint32)
        want = oracle.sort_unique(oracle.seeds_all(og, pidx, bytes(wb), woff, k, k, o_ln, o_lo, threads=4))
        got = psi_amd.sort_unique(finder.seeds_all((wb, woff), step=k))
        inner = lambda h: h[(h[:, 0] >= np.uint64(id_lo)) & (h[:, 0] < np.uint64(id_hi))]      # noqa: E731
        got, want = inner(got), inner(want)
        assert len(want) >= reads_per_window * 7 and _eq(got, want), 'records differ from the independent oracle in the window at %d' % w0
        n_hits_checked += len(want)
    return n_loci_checked, n_hits_checked


_config1_cache = {}


def test_query_modes_agree_at_full_size(query_mode):
    """BASELINE.json configs[1] at full size (51 Mbp, 1.1 M SNVs, 1 M x 150 bp reads, k = 21): the
    three query modes return the same 7.0 M sort-unique records, every seed is found where it was sampled."""
    k = 21
    if query_mode == 'locus-table':
        pytest.skip('the same three finders as the kmer-table parametrisation (the fixture changes nothing for it)')
    # graph, reads, index and the oracle's records: once per session (round-5 review: each of the four parametrisations rebuilt them)
    if 'sg' not in _config1_cache:
        sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
        g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to,
                                   paths=[sg.ref_path])
        bases, off = synth.sim_reads_snv(sg, 1_000_000, 150, seed=13)
        _config1_cache.update(sg=sg, g=g, bases=bases, off=off, px=psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0))
    sg, g, bases, off, px = (_config1_cache[n] for n in ('sg', 'g', 'bases', 'off', 'px'))
    res = {}
    for mode in ('kmer-table', 'locus-table', 'traverse'):
        f = psi_amd.SeedFinder(g, k, mode=mode)
        f.set_path_index(px)
        res[mode] = f.seeds_all((bases, off), step=k, sort_unique=True)
        c = f.counters()
        assert c['n_seeds'] == 7_000_000
        assert (c['n_path_kmers'] > 0) == (mode != 'locus-table')      # (traverse mode: the paths' k-mers, not the loci's)
        # (a walk cap of 1 leaves nearly every locus -- any that is uncovered has a second walk -- to the traverser in the table modes too)
        assert (c['n_loci_traversed'] > 0) == (mode == 'traverse' or query_mode.endswith('cap1'))
        f.close()
    assert _eq(res['kmer-table'], res['traverse']) and _eq(res['locus-table'], res['traverse'])
    hits = res['traverse']
    assert len(np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])) == 7_000_000
    # ... and they are the ORACLE's records: the C restatement of the reference path over the same index
    # paths and starting loci, all 1 M reads (a few seconds on the box's host cores)
    if 'want' not in _config1_cache:
        import oracle
        from bench import oracle_objects
        og, pidx = oracle_objects(sg, px)
        ln, lo = px.loci
        want = oracle.sort_unique(oracle.seeds_all(og, pidx, bytes(bases), off, k, k, ln, lo,
                                                   threads=oracle.lib().orc_max_threads()))
        _config1_cache['want'] = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    assert _eq(res['kmer-table'], _config1_cache['want'])
    # ... and a checker that takes nothing from the product: 20 random 50-kbp windows, loci by the brute-force definition,
    # the oracle's own suffix array (once per run of the suite: the modes agree, asserted above)
    if query_mode == 'kmer-table':
        f = psi_amd.SeedFinder(g, k)
        f.set_path_index(px)
        n_l, n_h = _independent_windows(sg, px, f, k, 20, 50_000, sg.n_block + 1000, 51_000_000 - 1000)
        f.close()
        assert n_l > 20 * 15_000 and n_h > 20 * 300 * 7 - 200


@pytest.mark.parametrize('query_mode', ['kmer-table'], indirect=True)
def test_config2_whole_genome(query_mode):
    """BASELINE.json configs[2]: whole-genome-like graph (3.1 Gbp backbone, 80 M SNV bubbles, 298 M nodes),
    10 M x 150 bp reads, k = 21, one MI355X, index and tables resident in HBM.  No oracle run is possible at
    this size, so: size-independent properties over all 70 M seeds, and oracle parity on a 50 Mbp window of
    the same graph (reads drawn from the window; hits inside it compared with the oracle run on the window
    alone).  Set PSI_TEST_WG=0 to skip (needs ~250 GB of host memory and ~4 minutes)."""
    if os.environ.get('PSI_TEST_WG', '1') == '0':
        pytest.skip('PSI_TEST_WG=0')
    import torch
    import oracle
    k = 21
    L, n_reads = 3_100_000_000, 10_000_000
    sg = synth.snv_graph(L, 80_000_000, n_block=150_000_000, seed=11)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, n_reads, 150, seed=13)
    # the last 200 k reads come from one 50 Mbp window: [w0, w0 + 50 M) of the backbone
    w0, wlen, n_win = 1_000_000_000, 50_000_000, 200_000
    rng = np.random.default_rng(5)
    start = rng.integers(w0 + 1000, w0 + wlen - 1150, size=n_win)
    idx = start[:, None] + np.arange(150)[None, :]
    r = sg.backbone[idx]
    a = sg.alt[idx]
    r = np.where((a != 0) & (rng.random(size=idx.shape) < 0.5), a, r)
    bases[(n_reads - n_win) * 150:] = r.reshape(-1)
    px = psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0)
    assert px.text_len > 2_900_000_000 and px.view.n_loci > 1_000_000_000
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(px)
    f.prepare()
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    ptr, n_hits = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, len(bases),
                                     stream=torch.cuda.current_stream().cuda_stream)
    c = f.counters()
    assert c['n_seeds'] == 70_000_000 and c['n_hits'] == n_hits
    print('whole genome: %d path k-mers + %d locus k-mers tabulated, %.1f ms on the device for 10 M reads' %
          (c['n_path_kmers'], c['n_locus_kmers'], c['ms_total']))
    hits = f.copy_hits(ptr, n_hits)
    # sensitivity: every seed of every (error-free) read is found; specificity: first bases agree
    assert len(np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])) == 70_000_000
    lo = sg.label_off.astype(np.int64)
    rank = hits[:500_000, 0].astype(np.int64) - 1
    first = sg.labels[lo[rank] + hits[:500_000, 1].astype(np.int64)]
    assert (first == bases[(hits[:500_000, 2] * np.uint64(150) + hits[:500_000, 3]).astype(np.int64)]).all()
    # ---- oracle parity on the window -----------------------------------------------------------------
    # node ranks of the window: labels follow the backbone in rank order with every alt base right behind
    # its ref base, so backbone position p sits at label coordinate p + (SNVs in front of p)
    lab = lambda pos: pos + int(np.count_nonzero(sg.alt[:pos]))                               # noqa: E731
    v0 = int(np.searchsorted(sg.label_off, np.uint64(lab(w0)), side='left'))
    v1 = int(np.searchsorted(sg.label_off, np.uint64(lab(w0 + wlen)), side='left'))
    eo = sg.edge_off.astype(np.int64)
    et = sg.edge_to[eo[v0]:eo[v1]].astype(np.int64)
    src = np.repeat(np.arange(v0, v1), np.diff(eo[v0:v1 + 1]))
    keep = (et >= v0) & (et < v1)
    sub_edge_off = np.concatenate([[0], np.cumsum(np.bincount(src[keep] - v0, minlength=v1 - v0))]).astype(np.uint64)
    og = oracle.OracleGraph(sg.node_id[v0:v1], (sg.label_off[v0:v1 + 1] - sg.label_off[v0]).astype(np.uint64),
                            bytes(sg.labels[int(lo[v0]):int(lo[v1])]), sub_edge_off, (et[keep] - v0).astype(np.uint64))
    path = px.paths()[0].astype(np.int64)
    sub_path = path[(path >= v0) & (path < v1)] - v0
    pidx = oracle.OraclePathIndex(og, [sub_path.tolist()])
    ln, lc = px.loci
    m = (ln >= v0) & (ln < v1)
    wb = bytes(bases[(n_reads - n_win) * 150:])
    woff = (np.arange(n_win + 1, dtype=np.uint64) * np.uint64(150))
    want = oracle.sort_unique(oracle.seeds_all(og, pidx, wb, woff, k, k, (ln[m] - v0).astype(np.uint32), lc[m],
                                               rec_offset=n_reads - n_win, threads=oracle.lib().orc_max_threads()))
    got = hits[hits[:, 2] >= n_reads - n_win]
    # compare away from the window's cut ends (walks that leave the window exist only in the whole graph)
    margin = 64
    inner = lambda h: h[(h[:, 0] >= sg.node_id[v0 + margin]) & (h[:, 0] < sg.node_id[v1 - margin])]     # noqa: E731
    got, want = psi_amd.sort_unique(inner(got)), inner(want)
    assert len(want) >= n_win * 7
    assert _eq(got, want)
    # ---- and the checker that takes nothing from the product (20 random 50-kbp windows of the 3.1 Gbp) ------------------
    n_l, n_h = _independent_windows(sg, px, f, k, 20, 50_000, sg.n_block + 1000, L - 1000)
    assert n_l > 20 * 15_000 and n_h > 20 * 300 * 7 - 200
    f.close()


# ---------------------------------------------------------------------------------------
# BASELINE.json configs[0] and configs[4], scaled to what the oracle finishes in seconds
# ---------------------------------------------------------------------------------------
def test_config1_linear_plumbing():
    """10-node linear graph (no variants), 1k x 50 bp reads, k = 11, distance 11 and 1."""
    rng = np.random.default_rng(1)
    lab = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, size=1000)]
    g = psi_amd.Graph.from_csr(np.arange(1, 11), np.arange(0, 1001, 100), lab, list(range(10)) + [9],
                               list(range(1, 10)), paths=[list(range(10))])
    rng = np.random.default_rng(2)
    st = rng.integers(0, 951, size=1000)
    reads = [bytes(lab[s:s + 50]).decode() for s in st]
    for npaths in (1, 0):
        for step in (11, 1):
            f = psi_amd.SeedFinder(g, 11)
            f.create_path_index(npaths)
            got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
            # linear graph: every seed occurs where it was sampled (and wherever its 11-mer repeats)
            text = bytes(lab).decode()
            want = set()
            for r, s in enumerate(reads):
                for o in range(0, 50 - 11 + 1, step):
                    km, p = s[o:o + 11], -1
                    while True:
                        p = text.find(km, p + 1)
                        if p < 0:
                            break
                        want.add((p // 100 + 1, p % 100, r, o))
            assert {tuple(h) for h in got.tolist()} == want
            if npaths:
                assert f.counters()['n_loci'] == 0          # one path covers a linear graph
            f.close()


@pytest.mark.parametrize('k,step,npaths', [(31, 31, 0), (31, 7, 1), (21, 21, 2)])
def test_config5_high_branching_vs_oracle(k, step, npaths):
    """multi-allelic SNVs + indels every ~20 bp: the traverser's forks, LDS stack and spill."""
    arrs = synth.bubble_graph(60_000, seed=31)
    nid, lo, lab, eo, et, ref = arrs
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    bases, off = synth.sim_reads_walk(nid, lo, lab, eo, et, 1500, 150, seed=k)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(npaths, rng_seed=3)
    got = psi_amd.sort_unique(f.seeds_all((bases, off), step=step))
    want = _oracle_hits((nid, lo, lab, eo, et), f, bases, off, k, step)
    assert len(want) >= 1500
    assert _eq(got, want)
    f.close()


@pytest.mark.parametrize('query_mode', ['kmer-table', 'traverse'], indirect=True)
def test_config4_hla_full_size(query_mode):
    """BASELINE.json configs[4] at FULL size: HLA-like high-branching graph (5 Mbp backbone, a multi-allelic
    SNV or an indel bubble every ~20 bp), k = 31, 1 M x 150 bp reads, no path index -- every locus is a
    starting locus and every hit comes from the traverser: enumerated once into the tables (default mode) or
    walked per chunk, pruned by the chunk's seeds (traverse mode: k_traverse, its LDS stack and spill queue).
    Properties over all 4 M seeds, the two modes' record sets against each other through a digest, and the
    oracle (whole graph, all loci) on the first 20 000 reads.  The traverse-mode run leaves a k_traverse
    roofline object in gpurun_out/config4_traverse.json.  PSI_TEST_HLA=0 skips."""
    if os.environ.get('PSI_TEST_HLA', '1') == '0':
        pytest.skip('PSI_TEST_HLA=0')
    import json
    import torch
    k, n_reads = 31, 1_000_000
    nid, lo, lab, eo, et, ref = synth.bubble_graph(5_000_000, seed=31)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    bases, off = synth.sim_reads_walk(nid, lo, lab, eo, et, n_reads, 150, seed=33)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(0)
    assert f.pindex.view.n_loci > 5_000_000 and f.pindex.view.n_paths == 0
    f.prepare()
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    call = lambda: f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, len(bases), step=k, stream=stream)   # noqa: E731
    kwalks_all = 0
    if query_mode == 'traverse':
        os.environ['PSIGPU_NO_PFX'] = '1'          # one pass without pruning: every k-walk from every locus (the roofline's unit)
        try:
            call()
            kwalks_all = f.counters()['n_kpaths']
        finally:
            os.environ.pop('PSIGPU_NO_PFX')
    call()
    ms = {}
    for _ in range(5):
        ptr, n_hits = call()
        c = f.counters()
        for name in ('ms_traverse', 'ms_table', 'ms_probe', 'ms_locate', 'ms_pack', 'ms_total'):
            ms[name] = ms.get(name, 0.0) + c[name] / 5
    per_read = (150 - k) // k + 1
    assert c['n_seeds'] == n_reads * per_read and c['n_hits'] == n_hits and c['n_hits_on_path'] == 0
    if query_mode == 'kmer-table':
        # (round-5 review: the one-kernel step did not engage here -- an index without paths got the 16-byte locus table and three
        # kernels; round 6: the k-mer table is built for the loci's k-mers alone)
        assert c['fused_step'] == 1 and c['n_locus_kmers'] > 0
    hits = f.copy_hits(ptr, n_hits)
    # sensitivity: every seed of every error-free read (a walk of the graph) is found; specificity: first bases agree
    assert len(np.unique(hits[:, 2] * np.uint64(1000) + hits[:, 3])) == n_reads * per_read
    rank = np.searchsorted(nid, hits[:300_000, 0])
    assert (nid[rank] == hits[:300_000, 0]).all()
    first = lab[lo.astype(np.int64)[rank] + hits[:300_000, 1].astype(np.int64)]
    assert (first == bases[(hits[:300_000, 2] * np.uint64(150) + hits[:300_000, 3]).astype(np.int64)]).all()
    su = psi_amd.sort_unique(hits)
    digest = (len(su), int((su * np.array([3, 5, 7, 11], np.uint64)).sum(dtype=np.uint64)))
    prev = getattr(test_config4_hla_full_size, 'digest', None)
    assert prev is None or prev == digest              # both modes: the same record set
    test_config4_hla_full_size.digest = digest
    # the oracle -- C restatement of the reference's traverser over ALL loci -- on the first reads
    n_chk = 20_000
    want = _oracle_hits((nid, lo, lab, eo, et), f, bases[:n_chk * 150], off[:n_chk + 1], k, k, threads=8)
    assert len(want) >= n_chk * per_read
    assert _eq(su[su[:, 2] < n_chk], want)
    rec = {'config': 'configs[4]: bubble_graph(5 Mbp, seed 31), 1 M x 150 bp walk reads (seed 33), k = 31, no path index',
           'query_mode': query_mode, 'nodes': int(len(nid)), 'edges': int(len(et)), 'starting_loci': int(f.pindex.view.n_loci),
           'seeds_per_step': int(c['n_seeds']), 'hits_per_step': int(n_hits), 'records_sort_unique': int(len(su)),
           'kernel_ms': {a: round(b, 4) for a, b in ms.items()}, 'seeds_per_s': c['n_seeds'] / (ms['ms_total'] * 1e-3),
           'n_loci_traversed': int(c['n_loci_traversed']), 'n_locus_kmers': int(c['n_locus_kmers']),
           'kwalks_completed_per_step': int(c['n_kpaths']), 'n_spilled': int(c['n_spilled']),
           'traverse_launches': int(c['traverse_launches']), 'oracle_reads_checked': n_chk, 'oracle_hits_checked': int(len(want))}
    if query_mode == 'traverse':
        # per k-walk from a starting locus 40 B at k = 31 (labels + edge lists + seed-table probe), 32 B per hit
        abytes = 40.0 * kwalks_all + 32.0 * n_hits
        t = ms['ms_traverse'] * 1e-3
        rec['k_traverse'] = {'bound': 'hbm', 'kernel': 'k_traverse', 'avg_launch_ms': ms['ms_traverse'], 'kwalks_from_loci': int(kwalks_all),
                             'algorithmic_bytes_per_launch': abytes, 'achieved': abytes / t / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': abytes / t / 8e12, 'traffic': None}
    print('config4 ' + json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        json.dump(rec, open(os.path.join(out_dir, 'config4_%s.json' % query_mode.replace('-', '_')), 'w'), indent=1)
    f.close()


def _standin_row(name, what, g, make_index, reads, k, n_reads, extra=None):
    """One stand-in through the three query modes + AUTO (round 6): what the tables cost on it (k-mers tabulated, device bytes,
    build time), what the walk cap leaves to the traverser, whether the traverser spills, what AUTO decides, the step's device
    time per mode -- and that the modes return ONE record set.  Returns (row, finder of the default mode, its sorted records)."""
    import json
    import torch
    bases, off = reads
    row = {'stand_in': name, 'what': what, 'k': k, 'reads': n_reads, 'nodes': int(g.n_nodes), 'modes': {}}
    px = make_index()
    row['starting_loci'] = int(px.view.n_loci)
    digest, keep, keep_su = None, None, None
    for mode in ('kmer-table', 'locus-table', 'traverse', 'auto'):
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        f = psi_amd.SeedFinder(g, k, mode=mode, walk_cap=0)
        if mode == 'auto':
            f.set_option('expected_calls', 1)          # (one chunk to answer: psikt on this FASTQ)
        f.set_path_index(px)
        f.prepare()
        torch.cuda.synchronize()
        dev_bytes = free0 - torch.cuda.mem_get_info()[0]
        su = f.seeds_all((bases, off), step=k, sort_unique=True)
        # device time of a step with the chunk resident in HBM (the host entry cuts a chunk into ~10 sub-batches, and the traverser's
        # sweep over all loci is paid per sub-batch: not the step's cost)
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        ms = []
        for _ in range(4):
            f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), n_reads, len(bases), step=k, stream=torch.cuda.current_stream().cuda_stream)
            ms.append(f.counters()['ms_total'])
        ms = ms[1:]
        c = f.counters()
        del d_b, d_o
        d = (len(su), int((su * np.array([3, 5, 7, 11], np.uint64)).sum(dtype=np.uint64)))
        assert digest is None or d == digest, (name, mode)
        digest = d
        row['modes'][mode] = {'resolved_to': f.query_mode(), 'device_bytes_index_and_tables': int(dev_bytes),
                              'locus_kmers_tabulated': int(c['n_locus_kmers']), 'path_kmers_tabulated': int(c['n_path_kmers']),
                              'table_build_ms': float(c['ms_locus_table_build']),
                              'loci_left_to_the_traverser': int(c['n_loci_traversed']),
                              'share_of_loci_left_to_the_traverser': float(c['n_loci_traversed']) / max(1, int(c['n_loci'])),
                              'n_spilled': int(c['n_spilled']), 'traverse_launches': int(c['traverse_launches']),
                              'device_ms_per_step': float(np.median(ms)), 'hits': int(c['n_hits'])}
        if mode == 'kmer-table':
            keep, keep_su = f, su
        else:
            f.close()
    row['records_sort_unique'] = digest[0]
    if extra:
        row.update(extra)
    print('standin ' + json.dumps(row))
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        json.dump(row, open(os.path.join(out_dir, 'standin_%s.json' % name), 'w'), indent=1)
    return row, keep, keep_su, px


@pytest.mark.parametrize('query_mode', ['kmer-table'], indirect=True)
def test_clustered_variation_stand_in(query_mode):
    """configs[1] with CLUSTERED variation (round-5 review: uniform bi-allelic SNVs are kind to the tabulating modes -- 2.9
    k-walks per locus): 51 Mbp, 1.1 M SNVs of which a fifth sit in 200-bp windows at one site per ~8 bp (up to ten sites in one
    21-mer window), 1 M x 150 bp reads, k = 21.  The three modes and AUTO return one record set; the checker that takes
    nothing from the product (_independent_windows: loci by the brute-force definition, the oracle's own suffix array) agrees
    on windows drawn over the graph; the row of numbers goes to gpurun_out/standin_clustered.json.
    Ref: seed_finder.hpp:1481-1585 (the loci), traverser_bfs.hpp:72-161."""
    k = 21
    sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11, cluster_frac=0.2)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 1_000_000, 150, seed=13)
    c = np.zeros(len(sg.alt) + 1, np.int64)
    c[1:] = np.cumsum(sg.alt != 0)
    w = c[k:] - c[:-k]
    row, f, su, px = _standin_row('clustered', 'snv_graph(51 Mbp, 1.1 M SNVs, cluster_frac 0.2), 1 M x 150 bp reads, k = 21, one path', g,
                                  lambda: psi_amd.PathIndex.build(g, k, 1, rng_seed=1, device=0), (bases, off), k, 1_000_000,
                                  {'max_sites_in_a_k_window': int(w.max()), 'share_of_k_windows_with_3_or_more_sites': float((w >= 3).mean())})
    assert row['max_sites_in_a_k_window'] >= 6 and len(su) >= 7_000_000
    # windows that hold a cluster are among them: the positions of the densest windows + random ones
    n_loci, n_hits = _independent_windows(sg, px, f, k, 6, 50_000, 11_000_100, 50_900_000)
    dense = np.flatnonzero(w >= max(6, int(w.max()) - 2))
    for pos in dense[:: max(1, len(dense) // 3)][:3]:
        lo_p = int(max(11_000_100, min(int(pos) - 25_000, 50_900_000 - 50_001)))
        a, b = _independent_windows(sg, px, f, k, 1, 50_000, lo_p, lo_p + 50_001, seed=int(pos) % 1000)
        n_loci += a; n_hits += b
    assert n_loci > 10_000 and n_hits >= 9 * 300 * 7 - 200
    f.close()


@pytest.mark.parametrize('query_mode', ['kmer-table'], indirect=True)
def test_hla_hot_region_stand_in(query_mode):
    """configs[4] with regions that EXPLODE (round-5 review: the old stand-in averages 2.9 k-walks per locus and never
    spills): bubble_graph(5 Mbp) with 5 % of the backbone in hot regions of 120 bp with a substitution site every ~3 bp
    -- ten sites of 2-4 alleles inside one 31-mer window, 2^10 .. 7 x 10^4 walks from the loci in front of it -- 1 M x 150 bp
    walk reads, k = 31, no path index.  The walk cap leaves those loci to the per-chunk traverser in every mode, whose LDS
    stack overflows into the spill queue (n_spilled > 0); same record set in all modes and AUTO; the oracle (the reference's
    traverser over ALL loci, traverser_bfs.hpp:72-161) on the first reads."""
    import json
    k, n_reads = 31, 1_000_000
    nid, lo, lab, eo, et, ref = synth.bubble_graph(5_000_000, seed=31, hot_frac=0.05)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    bases, off = synth.sim_reads_walk(nid, lo, lab, eo, et, n_reads, 150, seed=33)
    row, f, su, px = _standin_row('hla_hot', 'bubble_graph(5 Mbp, hot_frac 0.05: a site every ~3 bp in 120-bp regions), 1 M x 150 bp walk reads, '
                                  'k = 31, no path index', g, lambda: psi_amd.PathIndex.build(g, k, 0, device=0), (bases, off), k, n_reads)
    per_read = (150 - k) // k + 1
    assert len(np.unique(su[:, 2] * np.uint64(1000) + su[:, 3])) == n_reads * per_read      # every seed of every walk read is found
    assert row['modes']['kmer-table']['loci_left_to_the_traverser'] > 0                     # the walk cap bites
    assert row['modes']['traverse']['n_spilled'] > 0 or row['modes']['kmer-table']['n_spilled'] > 0
    n_chk = 4_000
    want = _oracle_hits((nid, lo, lab, eo, et), f, bases[:n_chk * 150], off[:n_chk + 1], k, k, threads=8)
    assert len(want) >= n_chk * per_read and _eq(psi_amd.sort_unique(su[su[:, 2] < n_chk]), want)      # (su is in read order, the oracle's in node order)
    f.close()


# ---------------------------------------------------------------------------------------
# randomised differential test: arbitrary small graphs (long nodes, N runs, out-degree up to 5,
# back edges / cycles, reads with N and ragged lengths) against the brute-force definition
# ---------------------------------------------------------------------------------------
def _random_graph(seed):
    import random
    from oracle import brute
    rng = random.Random(seed)
    n = rng.randint(3, 60)
    g = brute.Graph()
    for i in range(n):
        ln = rng.choice([1, 1, 2, 3, 5, 8, 13, 32, 33, 40, 80]) if rng.random() < 0.9 else rng.randint(1, 120)
        s = ''.join(rng.choice('ACGT') for _ in range(ln))
        if rng.random() < 0.15:
            p = rng.randrange(ln)
            q = min(ln, p + rng.randint(1, 4))
            s = s[:p] + 'N' * (q - p) + s[q:]
        g.add_node(i + 1, s)
    for i in range(n):
        if i + 1 < n and rng.random() < 0.9:
            g.add_edge(i + 1, i + 2)
        for _ in range(rng.choice([0, 0, 1, 1, 2, 4])):
            j = rng.randrange(n)
            if rng.random() < 0.85:
                j = min(n - 1, i + rng.randint(1, 4))        # mostly forward, nearby
            g.add_edge(i + 1, j + 1)
    # an embedded path: a greedy walk from node 1 that never repeats a node
    path, seen, v = [], set(), 1
    while v not in seen:
        path.append(v)
        seen.add(v)
        nxt = [t for t in g.out[v] if t not in seen]
        if not nxt:
            break
        v = rng.choice(nxt)
    g.paths.append(('p', path))
    reads = []
    for _ in range(rng.randint(1, 80)):
        v = rng.choice(g.ids)
        s = g.seq[v][rng.randrange(len(g.seq[v])):]
        while len(s) < 70 and g.out[v]:
            v = rng.choice(g.out[v])
            s += g.seq[v]
        s = s[:rng.randint(0, 70)]
        if rng.random() < 0.1:
            s = s.lower()
        reads.append(s)
    return g, reads


@pytest.mark.parametrize('seed', range(48))
def test_random_graphs_vs_brute(seed):
    from oracle import brute
    import random
    g, reads = _random_graph(1000 + seed)
    rng = random.Random(seed)
    rank = {v: i for i, v in enumerate(g.ids)}
    label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
    labels = ''.join(g.seq[v] for v in g.ids).encode()
    edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
    edge_to = [rank[t] for v in g.ids for t in g.out[v]]
    pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to,
                                paths=[[rank[v] for v in g.paths[0][1]]])
    for _ in range(3):
        k = rng.choice([3, 8, 12, 13, 16, 21, 25, 31])
        step = rng.choice([1, 2, k, k + 3])
        npaths = rng.choice([0, 1, 1, 2])
        want = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
        f = psi_amd.SeedFinder(pg, k)
        f.create_path_index(npaths, rng_seed=seed, sa_rate=rng.choice([1, 1, 2, 8]),
                            ftab_len=rng.choice([0, 0, 4, psi_amd.NO_FTAB]))
        got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
        assert _eq(got, want), (seed, k, step, npaths)
        f.close()


@pytest.mark.parametrize('seed', range(24))
def test_long_seeds_random_graphs_vs_brute(seed):
    """Seeds of 32..63 bases (psikt takes any -l, src/psikt.cpp:327): two-word k-mers through the FM search, the
    traverser and its seed table (keyed by a fingerprint, every candidate's k-mer compared); the tabulating modes
    answer them the same way.  Random graphs (cycles, N runs, out-degree up to 5) against the brute-force definition."""
    from oracle import brute
    import random
    g, reads = _random_graph(5000 + seed)
    rng = random.Random(seed)
    rank = {v: i for i, v in enumerate(g.ids)}
    label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
    labels = ''.join(g.seq[v] for v in g.ids).encode()
    edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
    edge_to = [rank[t] for v in g.ids for t in g.out[v]]
    pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to,
                                paths=[[rank[v] for v in g.paths[0][1]]])
    for _ in range(2):
        k = rng.choice([32, 33, 40, 47, 56, 63])
        step = rng.choice([1, 2, 5])
        npaths = rng.choice([0, 1, 2])
        want = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
        f = psi_amd.SeedFinder(pg, k)
        f.create_path_index(npaths, rng_seed=seed, sa_rate=rng.choice([1, 1, 4]), ftab_len=rng.choice([0, 4, psi_amd.NO_FTAB]),
                            build_on_device=rng.random() < 0.5)
        got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
        assert _eq(got, want), (seed, k, step, npaths)
        f.close()


@pytest.mark.parametrize('k,npaths', [(32, 1), (45, 0), (45, -3), (63, 2)])
def test_long_seeds_reference_graph(k, npaths):
    """The same on the reference's graph x with its 100-bp reads (npaths < 0: patched paths, context k + 2): brute-force
    hit set; the library's sort-unique; phases partition the set; a gocc threshold; an index in two parts."""
    from oracle import brute
    g, reads = _x_case()
    reads = reads[:150]
    bg = brute.parse_gfa(os.path.join(REF, 'x.gfa'))
    want = np.array(brute.hit_set(bg, reads, k, 7), dtype=np.uint64).reshape(-1, 4)
    assert len(want) > 150
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(abs(npaths), rng_seed=5, patched=npaths < 0, context=k + 2 if npaths < 0 else 0)
    assert _eq(psi_amd.sort_unique(f.seeds_all(reads, step=7)), want)
    su = f.seeds_all(reads, step=7, sort_unique=True)
    assert _eq(su, want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))])
    on, off_ = f.seeds_on_paths(reads, step=7), f.seeds_off_paths(reads, step=7)
    assert _eq(psi_amd.sort_unique(np.concatenate([on, off_])), want) and (len(on) > 0) == (npaths != 0)
    if npaths:
        bases, off = psi_amd.pack_reads(reads)
        f.set_gocc_threshold(1)
        got1 = psi_amd.sort_unique(f.seeds_all(reads, step=7))
        assert len(got1) <= len(want)
        f.set_gocc_threshold(0)
        lens = [sum(int(g.label_off[v + 1] - g.label_off[v]) for v in p) for p in f.pindex.paths()]
        if abs(npaths) >= 2:
            pp = psi_amd.PathIndex.build(g, k, abs(npaths), rng_seed=5, patched=npaths < 0, context=k + 2 if npaths < 0 else 0,
                                         max_part_text=max(max(lens) + 40, f.pindex.text_len // 2))
            assert pp.view.n_more_parts >= 1
            f2 = psi_amd.SeedFinder(g, k, gocc_threshold=1)
            f2.set_path_index(pp)
            assert _eq(psi_amd.sort_unique(f2.seeds_all(reads, step=7)), got1)
            f2.close()
    f.close()
    with pytest.raises(psi_amd.PsiGpuError):
        psi_amd.SeedFinder(g, 64)


def test_regression_all_t_31mer_with_an_ext_record(query_mode):
    """Found by tools/fuzz_modes.py (graph 31536): the all-T 31-mer is 62 one bits; with the EXT type
    (two more) its slot's key word was all ones -- what an empty slot looked like to the table
    builder, so a later insertion took the slot and the k-mer's hits were lost.  Empty is now
    told by the payload."""
    from oracle import brute
    g, reads = _random_graph(31536)
    rank = {v: i for i, v in enumerate(g.ids)}
    label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
    labels = ''.join(g.seq[v] for v in g.ids).encode()
    edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
    edge_to = [rank[t] for v in g.ids for t in g.out[v]]
    pg = psi_amd.Graph.from_csr(g.ids, label_off, labels, edge_off, edge_to,
                                paths=[[rank[v] for v in g.paths[0][1]]])
    k, step = 31, 1
    want = np.array(brute.hit_set(g, [r.upper() for r in reads], k, step), dtype=np.uint64).reshape(-1, 4)
    px = psi_amd.PathIndex.build(pg, k, 2, rng_seed=31536, sa_rate=1)
    for cap in (0, 1, 3):
        f = psi_amd.SeedFinder(pg, k, walk_cap=cap)
        f.set_path_index(px)
        got = psi_amd.sort_unique(f.seeds_all(reads, step=step))
        assert _eq(got, want), cap
        f.close()


# ---------------------------------------------------------------------------------------
# host entry point: sub-batch pipeline (H2D | kernels | D2H), pinned and pageable reads,
# sort-unique on the device
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('sub_bytes', [64, 1000, 30_000])
@pytest.mark.parametrize('pinned', [False, True, 'torch'])
def test_host_entry_pipeline_equals_one_batch(monkeypatch, sub_bytes, pinned):
    """A chunk cut into many sub-batches (ragged reads, reads with N, empty reads) gives the hits of
    the chunk answered in one piece; read ids stay global; the sorted result is the sorted chunk."""
    g, reads = _x_case()
    k, step = 21, 5
    ragged = []
    for i, r in enumerate(reads[:400]):
        ragged += [r, r[:k + (i % 7)], '', r[:40] + 'N' + r[41:]] if i % 5 == 0 else [r]
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    monkeypatch.setenv('PSIGPU_SUB_BYTES', str(1 << 30))
    want_raw = f.seeds_all(ragged, step=step, rec_offset=1000)
    want = psi_amd.sort_unique(want_raw)
    want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    bases, off = psi_amd.pack_reads(ragged)
    keep = []
    if pinned == 'torch':           # page-locked by somebody else's allocator
        import torch
        keep = [torch.from_numpy(bases).pin_memory(), torch.from_numpy(off.astype(np.int64)).pin_memory()]
        bases, off = keep[0].numpy(), keep[1].numpy().view(np.uint64)
    elif pinned:
        keep = [psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off)]
        bases, off = keep[0].array, keep[1].array
    monkeypatch.setenv('PSIGPU_SUB_BYTES', str(sub_bytes))
    raw = f.seeds_all((bases, off), step=step, rec_offset=1000)
    assert len(raw) == len(want_raw) and _eq(psi_amd.sort_unique(raw), psi_amd.sort_unique(want_raw))
    c = f.counters()
    assert c['n_reads'] == len(ragged) and c['n_hits'] == len(raw)
    su = f.seeds_all((bases, off), step=step, rec_offset=1000, sort_unique=True)
    assert _eq(su, want)
    assert f.counters()['n_hits'] == len(su)
    f.close()


def test_copy_ends_live_in_a_process_wide_pool():
    """The lifetime rule of the host entry's transfer buffers (round 6, include/psi_gpu.h psigpu_copy_pool_stats): contexts
    that come and go and calls that regrow hand both ends of their engine copies back to ONE pool -- a second finder's
    staging / landing buffers are the first one's, nothing goes to the driver while finders are closed, and a trim drains the
    engine queues first.  Records equal before and after (the reference's loop is deterministic, seed_finder.hpp:1724-1732)."""
    g, reads = _x_case()
    k, step = 21, 5
    bases, off = psi_amd.pack_reads(reads[:600])
    keep = [psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off)]
    got = []
    stats = []
    for life in range(4):
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(2, rng_seed=3)
        for sub in (1 << 12, 1 << 30):
            f.set_option('sub_bytes', sub)
            got.append(f.seeds_all((keep[0].array, keep[1].array), step=step, sort_unique=True))
            got.append(f.seeds_all(reads[:600], step=step, sort_unique=True))          # pageable reads: staged
        f.close()
        stats.append(psi_amd.copy_pool_stats())
    assert all(_eq(h, got[0]) for h in got[1:]) and len(got[0]) > 0
    # finders 2..4 are served what finder 1 handed back (a regrown buffer of finder 1 may leave finder 2 one size short, once)
    assert stats[0]['allocated'] > 0 and stats[1]['allocated'] - stats[0]['allocated'] <= 2, stats
    assert stats[-1]['allocated'] == stats[1]['allocated'], stats
    assert stats[-1]['reused'] > stats[0]['reused']
    assert stats[-1]['returned_to_driver'] == stats[0]['returned_to_driver']
    assert stats[-1]['idle'] > 0
    before = psi_amd.copy_pool_stats()
    after = psi_amd.copy_pool_stats(trim_all=True)
    assert after['idle'] == 0 and after['idle_device_bytes'] == 0 and after['idle_host_bytes'] == 0
    assert after['returned_to_driver'] == before['returned_to_driver'] + before['idle']
    assert after['queue_drains'] == before['queue_drains'] + 1          # the marker copies ran before the memory went
    assert after['in_use'] == before['in_use']                           # (keep[] is still ours)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    assert _eq(f.seeds_all((keep[0].array, keep[1].array), step=step, sort_unique=True), got[0])
    f.close()
    # hit arrays made ahead of time (psikt's side thread): they wait in the pool, the next taker of that size gets one of them
    s0 = psi_amd.copy_pool_stats()
    assert psi_amd.lib().psigpu_reserve_hit_arrays(200_000, 2) == 2
    s1 = psi_amd.copy_pool_stats()
    assert s1['allocated'] == s0['allocated'] + 2 and s1['idle'] == s0['idle'] + 2 and s1['in_use'] == s0['in_use']
    assert psi_amd.lib().psigpu_reserve_hit_arrays(200_000, 2) == 2
    assert psi_amd.copy_pool_stats()['allocated'] == s1['allocated']          # (the same two again)


@pytest.mark.parametrize('k,step', [(13, 4), (21, 21), (31, 7)])
def test_traverser_from_prefix_walks_equals_traverser_from_loci(k, step):
    """Traverse mode (TraverserBFS over every starting locus for every chunk, traverser_bfs.hpp:72-161) starts from the loci's
    tabulated 12-base prefix walks since round 4; started from the loci themselves (option no_pfx_roots) it must give the
    same records -- SNV graph with N blocks, a dense layered graph whose walks spill out of LDS, graph x without a path
    index (every locus a starting locus) -- and the same number of complete k-walks."""
    from oracle import brute
    cases = []
    sg = synth.snv_graph(200_000, 8_000, n_block=5_000, seed=k)
    cases.append((psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path]),
                  synth.sim_reads_snv(sg, 2000, 100, seed=3), 1))
    nid, lo, lab, eo, et, ref = synth.layered_graph(150, max_width=4, max_len=5, seed=k, p_n=0.01)
    lg = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    rng = np.random.default_rng(k)
    lreads = []
    for _ in range(300):
        v = int(rng.integers(0, len(nid)))
        seq = bytes(lab[int(lo[v]):int(lo[v + 1])]).decode()
        while len(seq) < 60 and eo[v + 1] > eo[v]:
            v = int(et[int(rng.integers(int(eo[v]), int(eo[v + 1])))])
            seq += bytes(lab[int(lo[v]):int(lo[v + 1])]).decode()
        lreads.append(seq[:60])
    cases.append((lg, lreads, 0))
    gx, xr = _x_case()
    cases.append((gx, xr[:200], 0))
    for g, reads, npaths in cases:
        res = []
        for no_roots in (0, 1):
            f = psi_amd.SeedFinder(g, k, mode='traverse')
            f.set_option('no_pfx_roots', no_roots)
            f.create_path_index(npaths, rng_seed=1)
            if no_roots == 0:
                f.prepare()
            hits = psi_amd.sort_unique(f.seeds_all(reads, step=step))
            res.append((hits, f.counters()['n_kpaths']))
            f.close()
        assert len(res[0][0]) and _eq(res[0][0], res[1][0])
        assert res[0][1] == res[1][1]


def _load_make_ref_paths():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_ref_paths', os.path.join(GOLDEN, 'make_ref_paths.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_reference_paths_file_decides_the_on_path_hits(tmp_path):
    """SURVEY 8(f) row 3, content-sensitive: an index made from a reference-written `<prefix>_paths`
    (PathIndex::save_paths_set pathindex.hpp:315-332 -> PathSet::serialize pathset.hpp:260-274 -> Path::serialize
    path_base.hpp:551-560) must answer seeds_on_paths -- the phase whose hit set DEPENDS on which paths and which trims
    were parsed -- like the oracle over the same three trimmed paths (the sequences the reference asserts,
    test_pathindex.cpp:166-168, 248-255), and leave the starting loci the brute-force definition leaves over them.
    (i) the committed fixture (graph x, context 10, Reversed); (ii) a file with long paths (> 300 nodes: several
    enc_vector samples), Forward direction; (iii) a graph whose ids DECREASE along the path (deltas wrap mod 2^64).
    What stays unpinnable here: that a given sdsl release writes exactly this layout (no sdsl-written file exists in
    the reference tree or this image)."""
    import oracle
    from oracle import brute
    m = _load_make_ref_paths()
    bg = brute.parse_gfa(os.path.join(REF, 'x.gfa'))
    g = _graph('x.gfa')
    reads = brute.read_seqs(os.path.join(REF, 'reads_n1000l100e0i0.seq'))[:400]
    bases, off = psi_amd.pack_reads(reads)
    k, step = 10, 3
    rank = {v: i for i, v in enumerate(bg.ids)}
    label_off = np.asarray(g.label_off, dtype=np.int64)
    og = oracle.OracleGraph.from_brute(bg)

    def on_path_oracle(px):
        paths = [p.tolist() for p in px.paths()]
        left = [int(label_off[p[0] + 1] - label_off[p[0]]) - h if h else 0 for p, (h, _) in zip(paths, px.trims())]
        right = [t for _, t in px.trims()]
        pidx = oracle.OraclePathIndex(og, paths, left=left, right=right)
        e = np.zeros(0, np.uint64)
        return oracle.sort_unique(oracle.seeds_all(og, pidx, bytes(bases), off, k, step, e, e, phases=1))

    def check(px, mode):
        f = psi_amd.SeedFinder(g, k, mode=mode)
        f.set_path_index(px)
        got = psi_amd.sort_unique(f.seeds_on_paths((bases, off), step=step))
        f.close()
        want = on_path_oracle(px)
        assert _eq(got, want)
        return got

    # (i) the reference test's three trimmed paths
    px = psi_amd.PathIndex.from_reference_paths(g, k, os.path.join(GOLDEN, 'ref_paths_x_trimmed.bin'))
    ids = [[bg.ids[v] for v in p.tolist()] for p in px.paths()]
    assert ids == [[205, 207, 209, 210], [187, 189, 191, 193, 194, 195, 197], [167, 168, 171, 172, 174]]
    hits_i = check(px, os.environ.get('PSI_AMD_MODE', 'kmer-table'))
    assert len(hits_i) > 0
    ln, lo = px.loci
    assert [(bg.ids[v], int(o)) for v, o in zip(ln.tolist(), lo.tolist())] == brute.uncovered_loci(bg, ids, k, px.trims())
    # the same three paths UNTRIMMED are another index: more on-path hits -- the trims of the file matter
    full = psi_amd.PathIndex.build_paths(g, k, [p.tolist() for p in px.paths()])
    hits_full = check(full, os.environ.get('PSI_AMD_MODE', 'kmer-table'))
    assert len(hits_full) > len(hits_i) and set(map(tuple, hits_i.tolist())) < set(map(tuple, hits_full.tolist()))
    # (ii) long walks, Forward: written by the independent statement of the format, read back, queried
    nl = m.x_node_lengths()
    walks = psi_amd.PathIndex.build(g, k, 3, rng_seed=5, patched=True, context=13)
    recs = []
    for p, (h, t) in zip(walks.paths(), walks.trims()):
        pid = [bg.ids[v] for v in p.tolist()]
        recs.append((pid, nl[pid[0]] - h if h else 0, t))
    assert max(len(r[0]) for r in recs) > 128
    fn = str(tmp_path / 'long_paths')
    open(fn, 'wb').write(m.paths_file(13, True, recs, nl))
    py = psi_amd.PathIndex.from_reference_paths(g, k, fn)
    assert py.ref_forward is True and py.trims() == walks.trims()
    a, b = check(py, os.environ.get('PSI_AMD_MODE', 'kmer-table')), check(walks, os.environ.get('PSI_AMD_MODE', 'kmer-table'))
    assert _eq(a, b) and len(a) > len(hits_i)
    pl, po = py.loci
    assert [(bg.ids[v], int(o)) for v, o in zip(pl.tolist(), po.tolist())] == \
        brute.uncovered_loci(bg, [r[0] for r in recs], k, py.trims())
    # (iii) ids that decrease along every path: graph x renumbered id -> 1000 - id
    rev = {v: 1000 - v for v in bg.ids}
    g2 = psi_amd.Graph.from_csr(np.array([rev[v] for v in bg.ids], dtype=np.uint64), g.label_off, g.labels, g.edge_off, g.edge_to,
                                paths=[p for p in g.paths()])
    recs2 = [([rev[v] for v in r[0]], r[1], r[2]) for r in recs]
    fn2 = str(tmp_path / 'rev_paths')
    open(fn2, 'wb').write(m.paths_file(13, False, recs2, {rev[v]: nl[v] for v in nl}))
    pz = psi_amd.PathIndex.from_reference_paths(g2, k, fn2)
    assert [p.tolist() for p in pz.paths()] == [p.tolist() for p in walks.paths()] and pz.trims() == walks.trims()
    f = psi_amd.SeedFinder(g2, k)
    f.set_path_index(pz)
    c = psi_amd.sort_unique(f.seeds_on_paths((bases, off), step=step))
    f.close()
    back = b.copy()
    back[:, 0] = np.uint64(1000) - back[:, 0]
    assert _eq(c, psi_amd.sort_unique(back))


def test_inversion_graph_followed_as_the_reference_follows_it(tmp_path):
    """PSIGPU_GRAPH_FOLLOW_REVERSING (psikt --follow-reversing-edges): a graph with reversing links is walked as the
    reference's traverser walks it -- every out-link's `to` node, forwards, `linktype` ignored
    (include/psi/traverser_bfs.hpp:146-160) -- i.e. the hit set is the brute-force definition over the edges AS WRITTEN."""
    from oracle import brute
    import random
    rng = random.Random(4)
    n = 40
    seqs = {i + 1: ''.join(rng.choice('ACGT') for _ in range(rng.choice([3, 5, 8, 13, 21]))) for i in range(n)}
    lines = ['H\tVN:Z:1.0'] + ['S\t%d\t%s' % (v, s) for v, s in seqs.items()]
    bg = brute.Graph()
    for v, sq in seqs.items():
        bg.add_node(v, sq)
    edges = set()
    for v in range(1, n):
        for t in {v + 1, min(n, v + rng.randint(1, 3))}:
            if (v, t) in edges:
                continue
            edges.add((v, t))
            a, b = rng.choice(['+', '+', '-']), rng.choice(['+', '+', '-'])        # a third of the links reverse a side
            lines.append('L\t%d\t%s\t%d\t%s\t0M' % (v, a, t, b))
            bg.add_edge(v, t)                                                      # as written: from -> to
    lines.append('P\tref\t' + ','.join('%d%s' % (v, rng.choice('+-')) for v in range(1, n + 1)) + '\t*')
    fn = str(tmp_path / 'inv.gfa')
    open(fn, 'w').write('\n'.join(lines) + '\n')
    g = psi_amd.Graph.load(fn, follow_reversing=True)
    reads = []
    for _ in range(120):
        v = rng.randint(1, n - 5)
        sq = bg.seq[v][rng.randrange(len(bg.seq[v])):]
        while len(sq) < 40 and bg.out[v]:
            v = rng.choice(bg.out[v])
            sq += bg.seq[v]
        reads.append(sq[:40])
    for k, npaths in ((8, 1), (13, 0), (21, 2)):
        want = np.array(brute.hit_set(bg, reads, k, 2), dtype=np.uint64).reshape(-1, 4)
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(npaths, rng_seed=1)
        assert len(want) and _eq(psi_amd.sort_unique(f.seeds_all(reads, step=2)), want)
        f.close()


def test_auto_query_mode_decides_by_expected_work():
    """PSIGPU_MODE_AUTO: a finder that expects one small chunk traverses (no tables made), one that expects many -- or
    does not know -- tabulates; same records either way (north_star: the traverser kernel and the tabulated default are
    the same function of the index)."""
    sg = synth.snv_graph(300_000, 9_000, n_block=20_000, seed=5)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 2000, 150, seed=6)
    k = 21
    px = psi_amd.PathIndex.build(g, k, 1, rng_seed=2)
    res = {}
    for label, calls, seeds, want_mode in (('one', 1, 14_000, 'traverse'), ('many', 100_000, 10 ** 9, 'kmer-table'),
                                           ('unknown', 0, 0, 'kmer-table')):
        f = psi_amd.SeedFinder(g, k, mode='auto')
        assert f.query_mode() == 'auto'
        f.set_option('expected_calls', calls)
        f.set_option('expected_seeds', seeds)
        f.set_path_index(px)
        f.prepare()
        assert f.query_mode() == want_mode, label
        res[label] = psi_amd.sort_unique(f.seeds_all((bases, off), step=k))
        c = f.counters()
        if want_mode == 'traverse':
            assert c['n_loci_traversed'] == c['n_loci'] and c['n_locus_kmers'] == 0       # nothing about the loci tabulated
        else:
            assert c['n_locus_kmers'] > 0 and c['n_loci_traversed'] < c['n_loci']        # (a walk cap leaves some loci over)
        f.close()
    assert len(res['one']) > 2000 and _eq(res['one'], res['many']) and _eq(res['one'], res['unknown'])


def test_finder_over_shared_views():
    """A context loaded from MAPPED views (psi_amd.shared: the arrangement of bench.py --gpus N, one host index for
    all ranks) answers like one loaded from the builder's own arrays -- one-part and multi-part indexes."""
    import shutil
    import tempfile
    from psi_amd import shared
    g, reads = _x_case()
    k, step = 21, 4
    for kw in (dict(), dict(patched=True, context=25, max_part_text=600)):
        px = psi_amd.PathIndex.build(g, k, 3, rng_seed=2, **kw)
        f = psi_amd.SeedFinder(g, k)
        f.set_path_index(px)
        want = f.seeds_all(reads[:300], step=step, sort_unique=True)
        f.close()
        d = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
        try:
            shared.export_views(d, g, px)
            sg, sx, _ = shared.import_views(d)
            f2 = psi_amd.SeedFinder(sg, k)
            f2.set_path_index(sx)
            assert _eq(f2.seeds_all(reads[:300], step=step, sort_unique=True), want) and len(want)
            f2.close()
        finally:
            shutil.rmtree(d)


def _ragged_reads(reads, k):
    out = []
    for i, r in enumerate(reads):
        out += [r, r[:k + (i % 7)], '', r[:40] + 'N' + r[41:], r[:30].lower() + r[30:], 'NNNN' + r[4:50]] if i % 5 == 0 else [r]
    return out


@pytest.mark.parametrize('k,step', [(21, 5), (8, 1), (31, 31), (40, 9), (63, 20)])
@pytest.mark.parametrize('pinned', [False, True])
def test_packed_reads_equal_ascii(k, step, pinned):
    """psigpu_find_seeds_packed (2-bit words + a "not ACGT" bit per base, read offsets in bases) returns the records of
    psigpu_find_seeds on the same reads -- ragged reads, N, lower case, reads shorter than k, empty reads -- in one
    piece and cut into many sub-batches (boundaries inside a word), raw and sorted; the reference keeps reads as one
    byte per base (sequence.hpp:1130-1294) and the packed form is the link format of round 4."""
    g, reads = _x_case()
    ragged = _ragged_reads(reads[:300], k)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    f.set_option('sub_bytes', 1 << 30)
    want_raw = f.seeds_all(ragged, step=step, rec_offset=1000)
    want = f.seeds_all(ragged, step=step, rec_offset=1000, sort_unique=True)
    assert len(want)
    bases, off = psi_amd.pack_reads(ragged)
    pr = psi_amd.PackedReads(bases, off, pinned=bool(pinned), threads=3)
    assert pr.n_not_acgt > 0 and pr.mask is not None
    for sub in (1 << 30, 31, 64, 1000, 30_000):
        f.set_option('sub_bytes', sub)
        for no_ahead in ((0, 1) if pinned else (0,)):
            f.set_option('no_ahead', no_ahead)
            raw = f.seeds_all_packed(pr, step=step, rec_offset=1000)
            assert len(raw) == len(want_raw) and _eq(psi_amd.sort_unique(raw), psi_amd.sort_unique(want_raw)), (sub, no_ahead)
            assert f.counters()['n_reads'] == len(ragged)
            assert _eq(f.seeds_all_packed(pr, step=step, rec_offset=1000, sort_unique=True), want), (sub, no_ahead)
    # a contiguous range of the chunk's reads (psikt --devices): the same word arrays, read_off[0] != 0
    nr = len(ragged)
    for (b, e), sub in (((nr // 3, 2 * nr // 3), 1 << 30), ((7, nr - 5), 97), ((nr - 1, nr), 64), ((5, 5), 64)):
        f.set_option('sub_bytes', sub)
        for no_ahead in ((0, 1) if pinned else (0,)):
            f.set_option('no_ahead', no_ahead)
            part = f.seeds_all_packed(pr, step=step, rec_offset=1000 + b, sort_unique=True, read_range=(b, e))
            assert _eq(part, want[(want[:, 2] >= 1000 + b) & (want[:, 2] < 1000 + e)]), (b, e, sub, no_ahead)
    # no mask at all when every base is ACGT
    clean = [r.upper().replace('N', 'A') for r in ragged]
    pc = psi_amd.PackedReads(*psi_amd.pack_reads(clean), pinned=bool(pinned))
    assert pc.mask is None
    f.set_option('sub_bytes', 500)
    assert _eq(f.seeds_all_packed(pc, step=step, sort_unique=True), f.seeds_all(clean, step=step, sort_unique=True))
    # phases
    assert _eq(psi_amd.sort_unique(f.seeds_all_packed(pr, step=step, flags=psi_amd.ON_PATHS)),
               psi_amd.sort_unique(f.seeds_on_paths(ragged, step=step)))
    # an empty chunk, a chunk of empty reads
    e = psi_amd.PackedReads(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(f.seeds_all_packed(e, step=step)) == 0
    e3 = psi_amd.PackedReads(np.zeros(0, np.uint8), np.zeros(4, np.uint64))
    assert len(f.seeds_all_packed(e3, step=step)) == 0
    f.close()


@pytest.mark.parametrize('k,step', [(21, 21), (12, 5), (31, 1)])
def test_uniform_reads_flag(k, step):
    """PSIGPU_UNIFORM_READS: with reads of one length a seed's read and offset follow from its number (no scan of the reads'
    seed counts); the claim is checked on the device, and a chunk for which it is false -- one read a base shorter, a
    ragged chunk, an empty read -- is answered again the general way: same records either way, through the device entry,
    the host entry (ASCII and packed, one piece and many sub-batches)."""
    import torch
    g, reads = _x_case()
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    equal = [r[:90] for r in reads[:500] if len(r) >= 90]
    equal[7] = equal[7][:40] + 'N' + equal[7][41:]
    almost = list(equal); almost[250] = almost[250][:-1]
    ragged = _ragged_reads(reads[:200], k)
    for name, rs, holds in (('equal', equal, True), ('almost', almost, False), ('ragged', ragged, False)):
        bases, off = psi_amd.pack_reads(rs)
        want = f.seeds_all((bases, off), step=step, rec_offset=3, sort_unique=True)
        c0 = f.counters()
        assert len(want)
        for sub in (1 << 30, 700):
            f.set_option('sub_bytes', sub)
            assert _eq(f.seeds_all((bases, off), step=step, rec_offset=3, sort_unique=True, uniform=True), want), (name, sub)
            c = f.counters()
            assert c['n_seeds'] == c0['n_seeds'] and c['n_seeds_valid'] == c0['n_seeds_valid']
            pr = psi_amd.PackedReads(bases, off, pinned=True)
            assert _eq(f.seeds_all_packed(pr, step=step, rec_offset=3, sort_unique=True, flags=psi_amd.ALL | psi_amd.UNIFORM_READS), want), (name, sub)
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        ptr, n = f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=3,
                                    flags=psi_amd.ALL | psi_amd.SORT_UNIQUE | psi_amd.UNIFORM_READS)
        assert _eq(f.copy_hits(ptr, n), want), name
        f.set_option('sub_bytes', 0)
    # traverse mode as well (the chunk's seed table is built from the same seeds)
    f2 = psi_amd.SeedFinder(g, k, mode='traverse')
    f2.set_path_index(f.pindex)
    bases, off = psi_amd.pack_reads(equal)
    assert _eq(f2.seeds_all((bases, off), step=step, sort_unique=True, uniform=True), f.seeds_all((bases, off), step=step, sort_unique=True))
    f2.close()
    f.close()


@pytest.mark.parametrize('query_mode', ['kmer-table', 'kmer-table-cap1'], indirect=True)
def test_probe_result_formats_agree(query_mode):
    """The k-mer table probe hands 8 bytes per seed to the emit kernel since round 4 (16 before: option res16): same raw
    stream -- order and multiplicities included -- for k-mers answered from their slot and for k-mers with a 32-byte
    record (several occurrences: four full copies of the paths; several loci), under gocc thresholds and for each phase."""
    g, reads = _x_case()
    for k, npaths in ((10, 4), (21, 1), (8, 0)):
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(npaths, rng_seed=2)
        for thr in (0, 1, 3):
            f.set_gocc_threshold(thr)
            out = {}
            for r16 in (0, 1):
                f.set_option('res16', r16)
                out[r16] = (f.seeds_all(reads[:300], step=3), f.seeds_on_paths(reads[:300], step=3), f.seeds_off_paths(reads[:300], step=3),
                            f.counters()['n_hits'])
            for a, b in zip(out[0][:3], out[1][:3]):
                # the same multiset of records (the table's hits come out seed by seed; a walk cap adds the traverser's in any order)
                assert _eq(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
            assert len(out[0][0]) > 0
        f.close()


@pytest.mark.parametrize('query_mode', ['kmer-table'], indirect=True)
def test_host_entry_with_two_sub_batches_in_flight(query_mode):
    """The host entry keeps two sub-batches in flight in the default mode when reads and offsets are pinned (the kernels of
    sub-batch i + 1 are queued before the host waits for sub-batch i: enqueue_default): the answer of the synchronous
    loop (option no_lookahead), from the second call of a context on; and what the five kernels alone do not settle
    hands the rest of the chunk to the synchronous loop -- a seed with more hits than the in-place ordering takes (four
    copies of the paths: duplicates), more hits than the sub-batch's buffer was sized for, a wire field too narrow."""
    sg = synth.snv_graph(400_000, 12_000, n_block=30_000, seed=3)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    bases, off = synth.sim_reads_snv(sg, 6000, 150, seed=4)
    k = 21
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=1)
    pin = (psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off))
    src = (pin[0].array, pin[1].array)
    pk = psi_amd.PackedReads(bases, off, pinned=True)
    f.set_option('sub_bytes', 60_000)                       # 15 sub-batches
    f.set_option('no_lookahead', 1)
    want = f.seeds_all(src, step=k, rec_offset=11, sort_unique=True)
    want_raw = f.seeds_all(src, step=k, rec_offset=11)
    assert f.counters()['lookahead_subbatches'] == 0 and len(want) > 6000
    f.set_option('no_lookahead', 0)
    for uniform in (False, True):
        got = f.seeds_all(src, step=k, rec_offset=11, sort_unique=True, uniform=uniform)
        c = f.counters()
        assert _eq(got, want) and c['lookahead_subbatches'] >= 10 and c['n_hits'] == len(want) and c['n_reads'] == 6000
        assert c['n_seeds'] == 6000 * 7
        raw = f.seeds_all(src, step=k, rec_offset=11, uniform=uniform)
        assert _eq(raw, want_raw) and f.counters()['lookahead_subbatches'] >= 10
        flags = psi_amd.ALL | (psi_amd.UNIFORM_READS if uniform else 0)
        assert _eq(f.seeds_all_packed(pk, step=k, rec_offset=11, sort_unique=True, flags=flags), want)
        assert f.counters()['lookahead_subbatches'] >= 10
    assert f.counters()['lookahead_fallbacks'] == 0
    # a wire field too narrow for some read offset: found when the sub-batch is finished, the chunk goes on synchronously
    f.set_option('wire8_roff_bits', 5)
    assert _eq(f.seeds_all(src, step=k, rec_offset=11, sort_unique=True), want)
    c = f.counters()
    assert c['lookahead_fallbacks'] == 1 and c['wire_bytes_per_hit'] == 16
    f.set_option('wire8_roff_bits', 0)
    assert _eq(f.seeds_all(src, step=k, rec_offset=11, sort_unique=True), want) and f.counters()['lookahead_subbatches'] >= 10
    # ragged reads claimed to be of one length: refuted on the device, answered the general way
    rag = np.concatenate([bases[:150 * 3000 - 7], bases[150 * 3000:]])
    roff = off.copy(); roff[3000:] -= np.uint64(7)
    rp = (psi_amd.pinned_copy(rag), psi_amd.pinned_copy(roff))
    w2 = f.seeds_all((rag, roff), step=k, sort_unique=True)
    assert _eq(f.seeds_all((rp[0].array, rp[1].array), step=k, sort_unique=True, uniform=True), w2)
    f_px = f.pindex
    f.close()
    # the host-side arrangement of round 5 switched off piece by piece (options read when the pipeline / its threads are made):
    # the records out on one copy engine, the library's threads wherever the scheduler puts them, three widening threads
    for opts in ((('one_out_engine', 1),), (('no_numa', 1), ('widen_threads', 3)), (('sub_bytes', 3_000_000),)):
        f = psi_amd.SeedFinder(g, k)
        f.set_path_index(f_px)
        f.set_option('sub_bytes', 60_000)
        for n_, v_ in opts:
            f.set_option(n_, v_)
        for _ in range(2):
            assert _eq(f.seeds_all(src, step=k, rec_offset=11, sort_unique=True), want)
            assert _eq(f.seeds_all_packed(pk, step=k, rec_offset=11, sort_unique=True), want)
        f.close()
    # duplicates (four full copies of each path): ordering a seed's hits in place is not enough -> the general sort
    f = psi_amd.SeedFinder(g, k, mode='locus-table')
    f.create_path_index(4, rng_seed=2)
    f.set_option('sub_bytes', 60_000)
    a = f.seeds_all(src, step=k, sort_unique=True)
    b = f.seeds_all(src, step=k, sort_unique=True)
    assert _eq(a, b) and f.counters()['lookahead_subbatches'] == 0          # (not the default mode: never the lookahead path)
    f.close()


def test_packed_device_entry_equals_ascii():
    """psigpu_find_seeds_device_packed: the device-resident chunk as 2-bit words."""
    import torch
    g, reads = _x_case()
    k, step = 21, 3
    ragged = _ragged_reads(reads[:400], k)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=3)
    bases, off = psi_amd.pack_reads(ragged)
    pr = psi_amd.PackedReads(bases, off)
    d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    d_w, d_m = torch.from_numpy(pr.words.view(np.int64)).cuda(), torch.from_numpy(pr._mask_store.view(np.int64)).cuda()
    for flags in (psi_amd.ALL, psi_amd.ALL | psi_amd.SORT_UNIQUE):
        ptr, n = f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(ragged), len(bases), step=step, rec_offset=5, flags=flags)
        a = f.copy_hits(ptr, n)
        ptr, n = f.seeds_all_device_packed(d_w.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), len(ragged), len(bases), step=step,
                                           rec_offset=5, flags=flags)
        b = f.copy_hits(ptr, n)
        assert len(a) and (_eq(a, b) if flags & psi_amd.SORT_UNIQUE else _eq(psi_amd.sort_unique(a), psi_amd.sort_unique(b)))
    f.close()


@pytest.mark.parametrize('pinned', [False, True])
def test_wire_record_formats_agree(pinned, query_mode):
    """The records cross the device-to-host link as packed records of 5 to 7 bytes (round 5: read id relative to the
    block of 256 records, read offset in seed distances), 8-byte keys (round 4), 16-byte records (round 3) or as they
    are (32 bytes): same records, same order.  An 8-byte key whose read-offset field is too narrow for a read of the
    sub-batch (forced here through the test hook) is detected on the device and the sub-batch goes out as 16-byte
    records instead; so is a packed record sized for shorter reads than the chunk holds."""
    g, reads = _x_case()
    k, step = 21, 5
    ragged = _ragged_reads(reads[:300], k)
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    bases, off = psi_amd.pack_reads(ragged)
    keep = [psi_amd.pinned_copy(bases), psi_amd.pinned_copy(off)] if pinned else []
    src = (keep[0].array, keep[1].array) if pinned else (bases, off)
    f.set_option('wire', 32)
    want_raw = psi_amd.sort_unique(f.seeds_all(src, step=step, rec_offset=9))
    want = f.seeds_all(src, step=step, rec_offset=9, sort_unique=True)
    assert f.counters()['wire_bytes_per_hit'] == 32
    traverser = f.counters()['n_loci_traversed'] > 0
    assert traverser == (query_mode in ('traverse', 'kmer-table-cap1'))
    for sub in (1 << 30, 700):
        f.set_option('sub_bytes', sub)
        for wire, expect in ((0, 5), (6, 6), (7, 7), (8, 8), (16, 16), (32, 32)):
            f.set_option('wire', wire)
            assert _eq(psi_amd.sort_unique(f.seeds_all(src, step=step, rec_offset=9)), want_raw), (sub, wire)
            # (raw records with the traverser's behind the others are not in read order: no packed records for them)
            assert f.counters()['wire_bytes_per_hit'] == (8 if expect < 8 and traverser else expect)
            assert _eq(f.seeds_all(src, step=step, rec_offset=9, sort_unique=True), want), (sub, wire)
            assert f.counters()['wire_bytes_per_hit'] == expect
        # a read-offset field of 4 bits: offsets beyond 15 do not fit -> 16-byte records, same answer
        f.set_option('wire', 0)
        f.set_option('wire8_roff_bits', 4)
        assert _eq(f.seeds_all(src, step=step, rec_offset=9, sort_unique=True), want)
        assert f.counters()['wire_bytes_per_hit'] == 16
        assert _eq(psi_amd.sort_unique(f.seeds_all(src, step=step, rec_offset=9)), want_raw)
        f.set_option('wire8_roff_bits', 0)
        assert _eq(f.seeds_all(src, step=step, rec_offset=9, sort_unique=True), want)
        assert f.counters()['wire_bytes_per_hit'] == 5
    with pytest.raises(psi_amd.PsiGpuError):
        f.set_option('no_such_option', 1)
    short = [r[:k + 2] for r in reads[:40]]
    long_ = [''.join(reads[i:i + 16]) for i in range(0, 320, 16)]
    f.set_option('wire', 32)
    want_long = f.seeds_all(long_, step=step, sort_unique=True)
    f_px = f.pindex
    f.close()
    # packed records are sized by the longest read the context has seen: a chunk of short reads, then one whose reads are
    # sixteen times as long -- its offsets do not fit, the kernel says so, the sub-batch goes out as 16-byte records and the
    # context goes on with one byte more per record
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(f_px)
    f.seeds_all(short, step=step, sort_unique=True)                  # (the first chunk of a context: nothing known yet, 8-byte keys)
    assert f.counters()['wire_bytes_per_hit'] == 8
    f.seeds_all(short, step=step, sort_unique=True)
    assert f.counters()['wire_bytes_per_hit'] == 5
    f.seeds_all(short, step=step, sort_unique=True)
    assert _eq(f.seeds_all(long_, step=step, sort_unique=True), want_long)
    assert f.counters()['wire_bytes_per_hit'] == 16
    assert _eq(f.seeds_all(long_, step=step, sort_unique=True), want_long)
    assert f.counters()['wire_bytes_per_hit'] == 6
    # a seed distance the offsets are no multiples of cannot happen through seeds_all; records out of read order cannot either
    f.close()


def test_wire_formats_through_the_radix_sort(monkeypatch):
    """... and when the general sort route runs (PSIGPU_NO_GROUPED_SORT), where the wire records are made after the sort."""
    monkeypatch.setenv('PSIGPU_NO_GROUPED_SORT', '1')
    g, reads = _x_case()
    k, step = 21, 5
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(3, rng_seed=1)
    f.set_option('wire', 32)
    want = f.seeds_all(reads[:300], step=step, sort_unique=True)
    for wire in (0, 16):
        f.set_option('wire', wire)
        f.set_option('sub_bytes', 2000)
        assert _eq(f.seeds_all(reads[:300], step=step, sort_unique=True), want)
    f.set_option('wire', 0)
    f.set_option('wire8_roff_bits', 3)
    assert _eq(f.seeds_all(reads[:300], step=step, sort_unique=True), want)
    assert f.counters()['wire_bytes_per_hit'] == 16
    f.close()


@pytest.mark.parametrize('query_mode', ['kmer-table', 'locus-table'], indirect=True)
def test_host_entry_with_the_host_oversubscribed(monkeypatch, query_mode):
    """The host entry's helper threads (widening of the 16-byte wire records, staging of pageable reads) next to
    three times as many spinning processes as the cores the test allows itself, one read per sub-batch: a thread descheduled in the middle
    of a sub-batch must not let its landing buffer be reused (found by three fuzz processes sharing a box)."""
    import subprocess
    import sys
    g, reads = _x_case()
    k, step = 21, 3
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    rr = reads[:120]
    monkeypatch.setenv('PSIGPU_SUB_BYTES', str(1 << 30))
    want = f.seeds_all(rr, step=step, rec_offset=7, sort_unique=True)
    monkeypatch.setenv('PSIGPU_SUB_BYTES', '16')
    # this process, the library's helper threads and the spinners all on four cores: oversubscribed without taking
    # the whole box
    cores = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(cores[:4]))
    burners = []
    try:
        burners = [subprocess.Popen([sys.executable, '-c', 'while True: pass']) for _ in range(12)]
        for _ in range(3):
            assert _eq(f.seeds_all(rr, step=step, rec_offset=7, sort_unique=True), want)
            assert _eq(psi_amd.sort_unique(f.seeds_all(rr, step=step, rec_offset=7)), psi_amd.sort_unique(want))
    finally:
        for b in burners:
            b.kill()
        for b in burners:
            b.wait()
        os.sched_setaffinity(0, set(cores))
    f.close()


def test_sort_unique_with_unordered_node_ids():
    """Node ids that are not rank + constant: the device sorter orders by id, not by rank."""
    g, reads = _random_graph(77)
    rng = np.random.default_rng(23)
    n = len(g.ids)
    new_ids = rng.permutation(np.arange(100, 100 + 3 * n, 3)).astype(np.uint64)
    rank = {v: i for i, v in enumerate(g.ids)}
    label_off = np.cumsum([0] + [len(g.seq[v]) for v in g.ids])
    labels = ''.join(g.seq[v] for v in g.ids).encode()
    edge_off = np.cumsum([0] + [len(g.out[v]) for v in g.ids])
    edge_to = [rank[t] for v in g.ids for t in g.out[v]]
    pg = psi_amd.Graph.from_csr(new_ids, label_off, labels, edge_off, edge_to,
                                paths=[[rank[v] for v in g.paths[0][1]]])
    k = 8
    f = psi_amd.SeedFinder(pg, k)
    f.create_path_index(1)
    raw = f.seeds_all(reads, step=2)
    assert len(raw)
    want = psi_amd.sort_unique(raw)
    want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    su = f.seeds_all(reads, step=2, sort_unique=True)
    assert _eq(su, want)
    assert set(np.unique(su[:, 0]).tolist()) <= set(new_ids.tolist())
    f.close()


def test_device_entry_sort_unique_and_prepare():
    """psigpu_find_seeds_device with PSIGPU_SORT_UNIQUE (device buffers in and out), after an explicit
    psigpu_prepare: the first query then builds nothing."""
    import torch
    g, reads = _x_case()
    k, step = 12, 4
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1)
    f.prepare()
    bases, off = psi_amd.pack_reads(reads[:500])
    d_bases = torch.from_numpy(bases).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    raw = f.seeds_all((bases, off), step=step, rec_offset=7)
    want = psi_amd.sort_unique(raw)
    want = want[np.lexsort((want[:, 1], want[:, 0], want[:, 3], want[:, 2]))]
    ptr, n = f.seeds_all_device(d_bases.data_ptr(), d_off.data_ptr(), 500, len(bases), step=step, rec_offset=7,
                                flags=psi_amd.ALL | psi_amd.SORT_UNIQUE,
                                stream=torch.cuda.current_stream().cuda_stream)
    assert n == len(want) == f.counters()['n_hits']
    assert _eq(f.copy_hits(ptr, n), want)
    f.close()


# ---------------------------------------------------------------------------------------
# MEM mode: find_mems, reference include/psi/index_iter.hpp:854-906
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,minlen,npaths,patched,gocc,max_mem', [
    ('x', 12, 1, False, 0, 0), ('x', 21, 3, False, 0, 0), ('x', 8, 2, True, 0, 0), ('x', 6, 2, False, 3, 0),
    ('tiny', 5, 2, False, 0, 4), ('multi', 15, 2, True, 0, 2), ('tiny', 3, 4, False, 2, 0),
])
def test_find_mems_vs_brute(name, minlen, npaths, patched, gocc, max_mem):
    """Greedy forward matching per read against the brute-force restatement (naive substring search over
    the indexed path texts): restarts behind hits, mismatches and N; gocc threshold; max_mem."""
    from oracle import brute
    import random
    b = brute.parse_gfa(os.path.join(REF, name + '.gfa'))
    g = _graph(name + '.gfa')
    f = psi_amd.SeedFinder(g, minlen, gocc_threshold=gocc)
    f.create_path_index(npaths, rng_seed=1, patched=patched, context=minlen + 4 if patched else 0)
    paths = [[b.ids[r] for r in p] for p in f.pindex.paths()]
    rng = random.Random(7)
    reads = []
    walk = ''.join(b.seq[v] for v in paths[0])
    for _ in range(60):
        a = rng.randrange(max(1, len(walk) - 80))
        r = list(walk[a:a + rng.randint(1, 80)])
        for _ in range(rng.choice([0, 0, 1, 2, 4])):                  # substitutions, some to N
            r[rng.randrange(len(r))] = rng.choice('ACGTN')
        reads.append(''.join(r))
    reads += ['', 'N', 'ACGT', walk[:minlen], walk[:minlen + 1], walk[5:5 + 2 * minlen + 3].lower()]
    got = f.find_mems(reads, max_mem=max_mem, rec_offset=9)
    want = np.array(brute.find_mems(b, paths, reads, minlen, f.pindex.trims(), gocc, max_mem, rec_offset=9),
                    dtype=np.uint64).reshape(-1, 6)
    assert len(want) > 10
    assert _eq(got, want)
    f.close()


# ---------------------------------------------------------------------------------------
# an index in several parts (texts beyond the 32-bit row limit; forced here with a small part size)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('fname', ['hits_x_reads_n1000l100e0i0_k21_d1.npz', 'hits_m_sim104_k12_d12.npz',
                                   'hits_multi_sim103_k31_d1.npz', 'hits_tiny_sim101_k10_d1.npz'])
@pytest.mark.parametrize('npaths,patched', [(3, False), (6, True)])
def test_index_in_several_parts(fname, npaths, patched, query_mode):
    """The same hit set from an index cut into parts of a few hundred symbols, in EVERY query mode: each part is
    a complete FM index (the FM modes search part after part, each part's hits behind the one before), the
    k-mer table tabulates the k-mers of all parts together (occurrences of one k-mer in several parts, k-mers
    whose first part is not the first); MEM mode extends a pattern in all parts at once (its occurrence count
    is the sum over the parts).  Also with the suffix array sampled, where the parts' hits are LF-walked."""
    z = np.load(os.path.join(GOLDEN, fname))
    reads = [str(r) for r in z['reads']]
    k, step = int(z['k']), int(z['step'])
    g = _graph(str(z['graph']))
    one = psi_amd.PathIndex.build(g, k, npaths, rng_seed=2, patched=patched)
    longest = max(sum(int(g.label_off[v + 1] - g.label_off[v]) for v in p) for p in one.paths())
    cut = max(longest + 40, one.text_len // 5)      # at most 8 parts
    px = psi_amd.PathIndex.build(g, k, npaths, rng_seed=2, patched=patched, max_part_text=cut, ftab_len=[0, 4, psi_amd.NO_FTAB][k % 3])
    assert px.view.n_more_parts >= 1
    assert [p.tolist() for p in px.paths()] == [p.tolist() for p in one.paths()] and px.trims() == one.trims()
    assert px.loci[0].tolist() == one.loci[0].tolist() and px.loci[1].tolist() == one.loci[1].tolist()
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(px)
    raw = f.seeds_all(reads, step=step)
    assert _eq(psi_amd.sort_unique(raw), z['hits'])
    f1 = psi_amd.SeedFinder(g, k)
    f1.set_path_index(one)
    raw1 = f1.seeds_all(reads, step=step)
    if query_mode != 'traverse':
        # raw emission: every on-path occurrence once, as from the one-part index
        assert len(raw) == len(raw1) and _eq(raw[np.lexsort(raw.T[::-1])], raw1[np.lexsort(raw1.T[::-1])])
    su = f.seeds_all(reads, step=step, sort_unique=True)
    want = z['hits'][np.lexsort((z['hits'][:, 1], z['hits'][:, 0], z['hits'][:, 3], z['hits'][:, 2]))]
    assert _eq(su, want)
    assert _eq(f.find_mems(reads[:40]), f1.find_mems(reads[:40]))
    f.close()
    # sampled suffix array in every part (FM modes: k_fm_walk per part; the k-mer table needs the whole array
    # and leaves such an index to the FM kernels)
    ps = psi_amd.PathIndex.build(g, k, npaths, rng_seed=2, patched=patched, max_part_text=cut, sa_rate=4)
    assert ps.view.n_more_parts == px.view.n_more_parts and ps.view.sa_rate == 4
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(ps)
    assert _eq(psi_amd.sort_unique(f.seeds_all(reads, step=step)), z['hits'])
    f.close(); f1.close()


def test_positions_deduplicated_at_table_build(query_mode):
    """Without a gocc threshold the k-mer table keeps one entry per graph position: a k-mer found at the
    same position on several full paths is one hit, not one per path.  A threshold counts occurrences in the
    path text (index_iter.hpp:843-847), so the table is then built (or rebuilt) with every occurrence."""
    if not query_mode.startswith('kmer-table'):
        pytest.skip('k-mer table build')
    g, reads = _x_case()
    k = 12
    px = psi_amd.PathIndex.build(g, k, 4, rng_seed=1)           # 4 full paths
    f = psi_amd.SeedFinder(g, k)
    f.set_path_index(px)
    on = f.seeds_on_paths(reads[:300], step=3)
    f2 = psi_amd.SeedFinder(g, k, gocc_threshold=1 << 30)
    f2.set_path_index(px)
    on2 = f2.seeds_on_paths(reads[:300], step=3)
    assert len(on) and len(on) < len(on2) and _eq(psi_amd.sort_unique(on), psi_amd.sort_unique(on2))
    assert len(np.unique(on, axis=0)) == len(on)
    # a threshold set after the table was built: same answer as a finder that had it from the start
    psi_amd.lib().psigpu_set_gocc_threshold(f.ctx, 3)
    f3 = psi_amd.SeedFinder(g, k, gocc_threshold=3)
    f3.set_path_index(px)
    a, b = f.seeds_all(reads[:300], step=3), f3.seeds_all(reads[:300], step=3)
    assert len(a) == len(b) and _eq(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    f.close(); f2.close(); f3.close()


def test_index_with_many_separators(monkeypatch, query_mode):
    """A rank block header counts the separators in front of it in 24 bits -- from the start of its SUPER-BLOCK of
    2^16 blocks; what lies in front of a super-block is a side array, so a text may hold any number (the 44.6 M
    patches of a whole genome x 3 walks).  Here the super-blocks are shrunk to 2 / 1 blocks through the test
    hook, so that every rank of a T and every exception lookup goes through the side array: same records as with
    the default layout, host and device builders byte-identical, every query mode and MEM mode."""
    g, reads = _x_case()
    k = 12
    one = psi_amd.PathIndex.build(g, k, 4, rng_seed=1, patched=True, context=k + 3, sa_rate=2)
    f1 = psi_amd.SeedFinder(g, k)
    f1.set_path_index(one)
    b = f1.seeds_all(reads[:400], step=3)
    for shift in ('1', '0'):
        monkeypatch.setenv('PSIGPU_TEST_EXC_SHIFT', shift)
        px = psi_amd.PathIndex.build(g, k, 4, rng_seed=1, patched=True, context=k + 3, device=0, sa_rate=2)
        py = psi_amd.PathIndex.build(g, k, 4, rng_seed=1, patched=True, context=k + 3, sa_rate=2)
        monkeypatch.delenv('PSIGPU_TEST_EXC_SHIFT')
        assert px.view.exc_shift == int(shift) and px.view.n_exc > 20
        nb = px.view.n_blocks
        assert nb == py.view.n_blocks == one.view.n_blocks
        assert (px._arr(px.view.bwt_blocks, nb * 16, np.uint32) == py._arr(py.view.bwt_blocks, nb * 16, np.uint32)).all()
        ns = ((nb - 1) >> int(shift)) + 1
        sup = px._arr(px.view.exc_super, ns, np.uint32)
        assert (sup == py._arr(py.view.exc_super, ns, np.uint32)).all() and sup[-1] > 0 and (np.diff(sup.astype(np.int64)) >= 0).all()
        for ix in (px, py):
            f = psi_amd.SeedFinder(g, k)
            f.set_path_index(ix)
            a = f.seeds_all(reads[:400], step=3)
            assert len(a) and _eq(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
            f.close()
    f1.close()
    # sa_rate 1: MEM mode and the k-mer table on the same layout
    one = psi_amd.PathIndex.build(g, k, 4, rng_seed=1, patched=True, context=k + 3)
    monkeypatch.setenv('PSIGPU_TEST_EXC_SHIFT', '0')
    px = psi_amd.PathIndex.build(g, k, 4, rng_seed=1, patched=True, context=k + 3, device=0)
    monkeypatch.delenv('PSIGPU_TEST_EXC_SHIFT')
    f, f1 = psi_amd.SeedFinder(g, k), psi_amd.SeedFinder(g, k)
    f.set_path_index(px); f1.set_path_index(one)
    a, b = f.seeds_all(reads[:400], step=3), f1.seeds_all(reads[:400], step=3)
    assert len(a) and _eq(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    assert _eq(f.find_mems(reads[:100]), f1.find_mems(reads[:100]))
    f.close(); f1.close()


def test_index_parts_roundtrip_and_gocc(tmp_path, query_mode):
    """Parts survive save / load; the gocc threshold counts a k-mer's occurrences over all parts (the FM modes:
    k_parts_combine adds the parts' counts up before the threshold is applied; MEM mode: the sum of the parts'
    intervals)."""
    g, reads = _x_case()
    k = 12
    one = psi_amd.PathIndex.build(g, k, 5, rng_seed=1)
    px = psi_amd.PathIndex.build(g, k, 5, rng_seed=1, max_part_text=2100)
    assert px.view.n_more_parts >= 2
    prefix = str(tmp_path / 'parts')
    px.save(prefix)
    py = psi_amd.PathIndex.load(prefix)
    assert py.view.n_more_parts == px.view.n_more_parts and py.text_len == px.text_len
    res = {}
    for name, ix in (('one', one), ('parts', py)):
        f = psi_amd.SeedFinder(g, k, gocc_threshold=3)
        f.set_path_index(ix)
        res[name] = psi_amd.sort_unique(f.seeds_all(reads[:300], step=3))
        f.close()
    assert len(res['one']) and _eq(res['one'], res['parts'])
    # ... equal to the oracle's seeds_all( gocc_thr ) over the same paths
    f = psi_amd.SeedFinder(g, k, gocc_threshold=3)
    f.set_path_index(py)
    bases, off = psi_amd.pack_reads(reads[:300])
    assert _eq(res['parts'], _oracle_hits(_graph_arrays(g), f, bases, off, k, 3, gocc=3))
    fm1 = psi_amd.SeedFinder(g, k, gocc_threshold=3)
    fm1.set_path_index(one)
    assert _eq(f.find_mems(reads[:60], max_mem=4), fm1.find_mems(reads[:60], max_mem=4))
    f.close(); fm1.close()


def test_device_entry_two_chunks_in_flight(query_mode):
    """psigpu_find_seeds_device_begin / _end (ABI 6): chunk i + 1 queued before chunk i is ended -- the same records as the
    synchronous entry for every chunk, whatever is in flight beside it: equal-length and ragged reads, the equal-length claim
    true and false, raw and sorted, packed words, a chunk that is larger than anything the context has seen (answered
    synchronously inside its end); in the other query modes, and with a walk-capped table, every chunk takes that route.  A third
    begin and any other entry point while chunks are begun are refused."""
    import torch
    g, reads = _x_case()
    k, step = 21, 3
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(1, rng_seed=3)
    equal = [r[:100] for r in reads[:900] if len(r) >= 100]
    ragged = _ragged_reads(reads[:500], k)
    long_ = [r for r in reads if len(r) >= 100]
    two_lengths = [r[:80] for r in long_[:120]] + [r[:100] for r in long_[120:240]]       # (their total is a multiple of their number)
    assert len(two_lengths) == 240
    chunks = []
    for rs, flags in ((equal[:300], psi_amd.ALL | psi_amd.UNIFORM_READS), (ragged[:200], psi_amd.ALL),
                      (two_lengths, psi_amd.ALL | psi_amd.UNIFORM_READS),              # the claim is false: answered again
                      (equal[300:600], psi_amd.ALL | psi_amd.SORT_UNIQUE | psi_amd.UNIFORM_READS),
                      (ragged[100:400], psi_amd.ALL | psi_amd.SORT_UNIQUE), (equal[100:800], psi_amd.ALL)):
        bases, off = psi_amd.pack_reads(rs)
        chunks.append((torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), len(rs), len(bases), flags))
    want = []
    for i, (d_b, d_o, nr, nb, flags) in enumerate(chunks):
        ptr, n = f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), nr, nb, step=step, rec_offset=1000 * i, flags=flags)
        want.append(f.copy_hits(ptr, n))
    assert all(len(w) for w in want)

    def begin(i):
        d_b, d_o, nr, nb, flags = chunks[i]
        f.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), nr, nb, step=step, rec_offset=1000 * i, flags=flags)

    def end(i):
        ptr, n = f.seeds_all_device_end()
        got = f.copy_hits(ptr, n)
        if chunks[i][4] & psi_amd.SORT_UNIQUE:
            assert _eq(got, want[i]), i
        else:
            assert _eq(psi_amd.sort_unique(got), psi_amd.sort_unique(want[i])), i
        assert f.counters()['n_hits'] == n

    for _ in range(2):
        begin(0)
        for i in range(len(chunks)):
            if i + 1 < len(chunks):
                begin(i + 1)
            end(i)
    # the hits of a chunk stay valid until the next end (psi_gpu.h), whatever is begun in between
    begin(0); begin(1)
    ptr0, n0 = f.seeds_all_device_end()
    begin(2)
    assert _eq(psi_amd.sort_unique(f.copy_hits(ptr0, n0)), psi_amd.sort_unique(want[0]))
    f.seeds_all_device_end()
    begin(3)
    with pytest.raises(psi_amd.PsiGpuError):
        begin(4)                                          # two are begun
    with pytest.raises(psi_amd.PsiGpuError):
        f.seeds_all_device(chunks[0][0].data_ptr(), chunks[0][1].data_ptr(), chunks[0][2], chunks[0][3], step=step)
    # ... and so is every other entry point that touches what the chunks in flight use (round-4 advisor)
    for refused in (lambda: f.find_mems(equal[:5]), lambda: f.set_gocc_threshold(3), lambda: f.set_tuning(0),
                    lambda: f.set_option('no_lookahead', 1), lambda: f.prepare(), lambda: f.verify_resident()):
        with pytest.raises(psi_amd.PsiGpuError):
            refused()
    # a chunk on another stream than the one in flight is refused: the workspace is shared, stream order is all that separates them
    other = torch.cuda.Stream()
    with pytest.raises(psi_amd.PsiGpuError):
        f.seeds_all_device_begin(chunks[0][0].data_ptr(), chunks[0][1].data_ptr(), chunks[0][2], chunks[0][3], step=step, stream=other.cuda_stream)
    f.seeds_all_device_end(); f.seeds_all_device_end()
    # (with nothing in flight any stream will do, and the next chunk must follow it there)
    torch.cuda.synchronize()
    d_b, d_o, nr, nb, flags = chunks[0]
    f.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), nr, nb, step=step, rec_offset=0, flags=flags, stream=other.cuda_stream)
    with pytest.raises(psi_amd.PsiGpuError):
        begin(1)
    f.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), nr, nb, step=step, rec_offset=0, flags=flags, stream=other.cuda_stream)
    for _ in range(2):
        ptr, n = f.seeds_all_device_end()
        assert _eq(psi_amd.sort_unique(f.copy_hits(ptr, n)), psi_amd.sort_unique(want[0]))
    # two chunks answered inside their end (the false claim, twice) back to back: each one's records are whole when handed out
    begin(2); begin(2)
    for _ in range(2):
        ptr, n = f.seeds_all_device_end()
        assert _eq(psi_amd.sort_unique(f.copy_hits(ptr, n)), psi_amd.sort_unique(want[2]))
    # a context that has to grow its hit buffers under a caller that still holds the previous end's records
    begin(0)
    ptr0, n0 = f.seeds_all_device_end()
    big = equal * 6
    bb, bo = psi_amd.pack_reads(big)
    d_bb, d_bo = torch.from_numpy(bb).cuda(), torch.from_numpy(bo.astype(np.int64)).cuda()
    f.seeds_all_device_begin(d_bb.data_ptr(), d_bo.data_ptr(), len(big), len(bb), step=1, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    assert _eq(psi_amd.sort_unique(f.copy_hits(ptr0, n0)), psi_amd.sort_unique(want[0]))      # (still chunk 0's, not freed)
    ptr, n = f.seeds_all_device_end()
    pb, nb_ = f.seeds_all_device(d_bb.data_ptr(), d_bo.data_ptr(), len(big), len(bb), step=1, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    assert n == nb_ and n > n0
    with pytest.raises(psi_amd.PsiGpuError):
        f.seeds_all_device_end()                          # nothing begun
    if query_mode == 'kmer-table':
        assert f.counters()['lookahead_fallbacks'] >= 1       # (the false claim at least)
        begin(0); begin(1)
        f.seeds_all_device_end()
        assert f.counters()['lookahead_subbatches'] == 1      # chunk 0 went through the queue
        f.seeds_all_device_end()
        assert f.counters()['lookahead_subbatches'] == 1      # and so did chunk 1, begun while chunk 0 was in flight
    # packed words
    rs = equal[:400]
    bases, off = psi_amd.pack_reads(rs)
    pr = psi_amd.PackedReads(bases, off)
    d_w, d_m = torch.from_numpy(pr.words.view(np.int64)).cuda(), torch.from_numpy(pr._mask_store.view(np.int64)).cuda()
    d_o = torch.from_numpy(off.astype(np.int64)).cuda()
    ptr, n = f.seeds_all_device_packed(d_w.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    wp = psi_amd.sort_unique(f.copy_hits(ptr, n))
    for _ in range(2):
        f.seeds_all_device_packed_begin(d_w.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    for _ in range(2):
        ptr, n = f.seeds_all_device_end()
        assert _eq(psi_amd.sort_unique(f.copy_hits(ptr, n)), wp)
    f.seeds_all_device_begin(chunks[0][0].data_ptr(), chunks[0][1].data_ptr(), chunks[0][2], chunks[0][3], step=step)
    f.close()                                             # (a chunk begun and never ended: the context drains it)


def test_resident_arrays_keep_their_checksums():
    """psigpu_verify_resident: every array the loaders put on the device is what it was when it was loaded -- after queries
    in every entry point -- and a word changed behind the library's back (test hook) is reported by the array's name; loading
    the index again makes the record whole.  PSIGPU_VERIFY_UPLOAD compares every upload with its host source."""
    g, reads = _x_case()
    k = 21
    f = psi_amd.SeedFinder(g, k)
    f.create_path_index(2, rng_seed=3)
    assert f.verify_resident() == ''
    want = psi_amd.sort_unique(f.seeds_all(reads[:300], step=5))
    f.find_mems(reads[:50])
    assert f.verify_resident() == ''
    f.set_option('corrupt_resident', 0x7FFFFFF0)
    rep = f.verify_resident()
    assert 'starting loci' in rep and 'rank blocks' not in rep, rep
    f.set_path_index(f.pindex)
    assert f.verify_resident() == ''
    assert _eq(psi_amd.sort_unique(f.seeds_all(reads[:300], step=5)), want)
    f.close()
    os.environ['PSIGPU_VERIFY_UPLOAD'] = '1'
    try:
        f = psi_amd.SeedFinder(g, k)
        f.create_path_index(2, rng_seed=3)
        assert f.verify_resident() == ''
        assert _eq(psi_amd.sort_unique(f.seeds_all(reads[:300], step=5)), want)
        f.close()
    finally:
        os.environ.pop('PSIGPU_VERIFY_UPLOAD', None)


@pytest.mark.parametrize('k,step', [(21, 1), (21, 21), (12, 3), (31, 2)])
def test_one_kernel_step_equals_the_three_kernel_step(query_mode, k, step):
    """k_kmer_step (round 5: seeding, table probe and emission of a tile of seeds in one kernel, output offsets by a decoupled
    look-back over the tiles) returns the RAW records of k_seed_pack -> k_kmer_probe -> k_kmer_emit in the same order:
    equal-length reads with and without the claim, ragged reads with N / lower case / short / empty reads, packed words,
    more tiles than one look-back window holds (> 64 tiles of 1024 seeds), a hit buffer that overflows (the call again with
    room), sort-unique on the device, the device-resident entry with two chunks in flight, the host entry in sub-batches."""
    import torch
    if query_mode != 'kmer-table':
        pytest.skip('the one-kernel step is the default mode without a traverser pass')
    g, reads = _x_case()
    f1 = psi_amd.SeedFinder(g, k)
    f1.create_path_index(2, rng_seed=5)
    f3 = psi_amd.SeedFinder(g, k)
    f3.set_option('no_fused', 1)
    f3.set_path_index(f1.pindex)
    equal = [r[:100] for r in reads if len(r) >= 100]
    ragged = _ragged_reads(reads[:600], k)
    many = equal * (1 if step > 1 else 1)
    cases = [(equal, psi_amd.ALL | psi_amd.UNIFORM_READS), (equal, psi_amd.ALL), (ragged, psi_amd.ALL), (ragged[:7], psi_amd.ALL),
             (equal, psi_amd.ON_PATHS), (ragged, psi_amd.OFF_PATHS | psi_amd.UNIFORM_READS), (many, psi_amd.ALL | psi_amd.SORT_UNIQUE)]
    for rs, flags in cases:
        bases, off = psi_amd.pack_reads(rs)
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        got, want = [], []
        for f, out in ((f1, got), (f3, want)):
            ptr, n = f.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags)
            out.append(f.copy_hits(ptr, n))
            out.append(f.counters())
        assert _eq(got[0], want[0]), (len(rs), flags)                 # raw records, same order
        assert got[1]['fused_step'] == 1 and want[1]['fused_step'] == 0
        for key in ('n_seeds', 'n_seeds_valid', 'n_seeds_on_path', 'n_hits_on_path', 'n_hits_off_path', 'n_hits'):
            assert got[1][key] == want[1][key], key
        # packed words through the same kernel
        pr = psi_amd.PackedReads(bases, off)
        d_w, d_m = torch.from_numpy(pr.words.view(np.int64)).cuda(), torch.from_numpy(pr._mask_store.view(np.int64)).cuda()
        ptr, n = f1.seeds_all_device_packed(d_w.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags)
        assert _eq(f1.copy_hits(ptr, n), want[0])
    if step == 1:
        assert got[1]['n_seeds'] > 64 * 1024                         # (the look-back moved its window)
    # PSIGPU_ANY_ORDER: the same raw records (as a multiset), tiles of 1024 seeds in the order they finished; with
    # PSIGPU_SORT_UNIQUE the flag is ignored
    rows = lambda h: h[np.lexsort(h.T[::-1])]                        # noqa: E731
    for rs, flags in cases[:3]:
        bases, off = psi_amd.pack_reads(rs)
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        ptr, n = f1.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags)
        ordered = f1.copy_hits(ptr, n)
        for _ in range(3):
            ptr, n = f1.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags | psi_amd.ANY_ORDER)
            anyo = f1.copy_hits(ptr, n)
            assert f1.counters()['n_hits'] == len(ordered)
            assert _eq(rows(anyo), rows(ordered))
        ptr, n = f1.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags | psi_amd.SORT_UNIQUE)
        su = f1.copy_hits(ptr, n)
        ptr, n = f1.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77,
                                     flags=flags | psi_amd.SORT_UNIQUE | psi_amd.ANY_ORDER)
        assert _eq(f1.copy_hits(ptr, n), su)
        # (the three-kernel step has one order only)
        ptr, n = f3.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(rs), len(bases), step=step, rec_offset=77, flags=flags | psi_amd.ANY_ORDER)
        assert _eq(f3.copy_hits(ptr, n), ordered)
    # the hit buffer overflows (a fresh context has no hint; many hits per seed at k = 12): the call is made again with room
    f1b = psi_amd.SeedFinder(g, k)
    f1b.set_path_index(f1.pindex)
    lots = equal * 3
    a = f1b.seeds_all(lots, step=step)
    b = f3.seeds_all(lots, step=step)
    assert _eq(a, b)
    # the host entry, cut into sub-batches, sorted on the device; and two chunks in flight
    os.environ['PSIGPU_SUB_BYTES'] = '20000'
    try:
        assert _eq(f1.seeds_all(ragged, step=step, sort_unique=True), f3.seeds_all(ragged, step=step, sort_unique=True))
        pb = psi_amd.pinned_copy(psi_amd.pack_reads(equal)[0]), psi_amd.pinned_copy(psi_amd.pack_reads(equal)[1])
        for _ in range(2):
            assert _eq(f1.seeds_all((pb[0].array, pb[1].array), step=step, sort_unique=True), f3.seeds_all(equal, step=step, sort_unique=True))
    finally:
        os.environ.pop('PSIGPU_SUB_BYTES')
    bases, off = psi_amd.pack_reads(equal)
    d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    ptr, n = f3.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(equal), len(bases), step=step, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    w = f3.copy_hits(ptr, n)
    f1.seeds_all_device(d_b.data_ptr(), d_o.data_ptr(), len(equal), len(bases), step=step, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    for _ in range(2):
        f1.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), len(equal), len(bases), step=step, flags=psi_amd.ALL | psi_amd.UNIFORM_READS)
    for _ in range(2):
        ptr, n = f1.seeds_all_device_end()
        assert _eq(f1.copy_hits(ptr, n), w)
        assert f1.counters()['fused_step'] == 1
    for _ in range(2):
        f1.seeds_all_device_begin(d_b.data_ptr(), d_o.data_ptr(), len(equal), len(bases), step=step,
                                  flags=psi_amd.ALL | psi_amd.UNIFORM_READS | psi_amd.ANY_ORDER)
    for _ in range(2):
        ptr, n = f1.seeds_all_device_end()
        assert _eq(rows(f1.copy_hits(ptr, n)), rows(w))
    f1.close(); f1b.close(); f3.close()


@pytest.mark.parametrize('k,step,npaths', [(21, 7, 2), (12, 1, 3), (31, 31, 1), (40, 13, 2)])
def test_header_api_callbacks_carry_gocc(query_mode, tmp_path, k, step, npaths):
    """psi::SeedFinder::seeds_all with one callback per phase (reference seed_finder.hpp:1734-1743) through the C++ shim: the
    records of both phases are the brute-force hit set, and every record carries Seed::gocc as the reference sets it -- on
    paths the number of occurrences of the seed's k-mer in the indexed path text (index_iter.hpp:743, counted here on the
    path sequences themselves, overlapping occurrences, not across paths), off paths the number of read positions of the
    chunk that hold the k-mer (traverser_bfs.hpp:107).  psigpu_count_occurrences against the same count, seed by seed."""
    import subprocess
    from oracle import brute
    if query_mode != 'kmer-table':
        pytest.skip('the shim drives the default mode')
    exe = str(tmp_path / 'seedfinder_api')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'psi_amd', 'include'),
                           os.path.join(ROOT, 'tests', 'cpp', 'seedfinder_api.cpp'), '-o', exe,
                           '-L' + os.path.join(ROOT, 'psi_amd'), '-lpsi_gpu', '-lz', '-Wl,-rpath,' + os.path.join(ROOT, 'psi_amd')])
    reads = brute.read_seqs(os.path.join(REF, 'reads_n1000l100e0i0.seq'))[:300]
    reads = [r.upper() for r in reads] + [reads[0].upper(), reads[1][:60].upper()]           # (repeated k-mers among the chunk's seeds)
    rf = tmp_path / 'reads.seq'
    rf.write_text('\n'.join(reads) + '\n')
    out = subprocess.check_output([exe, os.path.join(REF, 'x.gfa'), str(rf), str(k), str(npaths), str(step)]).decode().split('\n')
    paths = [l.split(' ', 1)[1] for l in out if l.startswith('pathseq ')]
    assert len(paths) == npaths
    recs = [l.split() for l in out if l.startswith('on ') or l.startswith('off ')]
    assert recs and all(int(r[5]) == k for r in recs)

    def occurrences(text, pat):
        n, at = 0, text.find(pat)
        while at >= 0:
            n, at = n + 1, text.find(pat, at + 1)
        return n
    in_reads = {}
    for r in reads:
        for j in range(0, len(r) - k + 1, step):
            in_reads[r[j:j + k]] = in_reads.get(r[j:j + k], 0) + 1
    on_cache = {}
    n_on = n_off = 0
    for phase, node, noff, rid, roff, _, gocc in recs:
        km = reads[int(rid)][int(roff):int(roff) + k]
        if phase == 'on':
            if km not in on_cache:
                on_cache[km] = sum(occurrences(p, km) for p in paths)
            assert int(gocc) == on_cache[km] and on_cache[km] >= 1, (phase, node, noff, rid, roff, gocc)
            n_on += 1
        else:
            assert int(gocc) == in_reads[km], (phase, node, noff, rid, roff, gocc)
            n_off += 1
    assert n_on and n_off
    g = brute.parse_gfa(os.path.join(REF, 'x.gfa'))
    want = brute.hit_set(g, reads, k, step)
    got = sorted({(int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in recs})
    assert got == want
    # the count itself, seed by seed, through the C ABI (a finder over the same graph picks the same paths: same seed)
    f = psi_amd.SeedFinder(_graph('x.gfa'), k)
    f.create_path_index(npaths, rng_seed=0)
    texts = []
    gb = g
    for p in f.pindex.paths():
        texts.append(''.join(gb.seq[gb.ids[r]] for r in p.tolist()))
    cnt = f.count_occurrences(reads, step=step)
    at = 0
    for r in reads:
        for j in range(0, len(r) - k + 1, step):
            km = r[j:j + k]
            assert cnt[at] == (0 if 'N' in km else sum(occurrences(t, km) for t in texts)), (at, km)
            at += 1
    assert at == len(cnt)
    f.close()
