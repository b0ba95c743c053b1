"""Device-side index construction (psi_amd/csrc/build_gpu.hip: suffix array by prefix doubling,
rank blocks, exceptions, samples, interval table, 4-bit text) against the host builder (SA-IS):
the two must produce the IDENTICAL index, byte for byte."""
import os
import time

import numpy as np
import pytest

import psi_amd
from psi_amd import synth

pytestmark = pytest.mark.gpu

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_data')


def _same_index(a, b):
    va, vb = a.view, b.view
    for f in ('seed_len', 'sa_rate', 'context', 'n_paths', 'text_len', 'n_blocks', 'n_samples', 'n_exc',
              'ftab_len', 'n_segs', 'n_dir', 'n_loci'):
        assert getattr(va, f) == getattr(vb, f), f
    assert list(va.C) == list(vb.C)
    arrays = [('bwt_blocks', va.n_blocks * 64, np.uint8), ('sa_samples', va.n_samples, np.uint32),
              ('exc_row', va.n_exc, np.uint32), ('exc_sa', va.n_exc, np.uint32),
              ('ftab', (2 << (2 * va.ftab_len)) if va.ftab_len else 0, np.uint32),
              ('text4', va.text_len // 16 + 2, np.uint64), ('seg_start', va.n_segs + 1, np.uint32),
              ('loci_node', va.n_loci, np.uint32), ('loci_off', va.n_loci, np.uint32)]
    for name, n, dt in arrays:
        x, y = a._arr(getattr(va, name), n, dt), b._arr(getattr(vb, name), n, dt)
        assert x.shape == y.shape and bool((x == y).all()), name


@pytest.mark.parametrize('name,k,npaths,sa_rate,ftab', [
    ('tiny.gfa', 10, 1, 1, 0), ('tiny.gfa', 12, 3, 4, 3), ('x.gfa', 21, 2, 1, 0), ('multi.gfa', 16, 2, 8, 5),
    ('m.gfa', 21, 4, 1, 0), ('m.gfa', 31, 1, 2, psi_amd.NO_FTAB), ('x.gfa', 21, 1, 1, 14),
])
def test_device_build_equals_host_build(name, k, npaths, sa_rate, ftab):
    g = psi_amd.Graph.load(os.path.join(REF, name))
    host = psi_amd.PathIndex.build(g, k, npaths, sa_rate=sa_rate, ftab_len=ftab, rng_seed=5, keep=True)
    dev = psi_amd.PathIndex.build(g, k, npaths, sa_rate=sa_rate, ftab_len=ftab, rng_seed=5, keep=True, device=0)
    assert (host.sa() == dev.sa()).all()
    _same_index(host, dev)


@pytest.mark.parametrize('seed', range(6))
def test_device_starting_loci_equal_host(seed):
    """Starting-loci detection on the device (k_loci_*) against the host routine -- which the CPU
    suite pins to the brute-force definition -- on layered DAGs with out-degree up to 4 and empty
    nodes, bubble graphs with indels, 0..5 indexed paths, locus steps 1..3; and on a graph with
    a cycle and a path that visits a node twice (no coverage bit for it)."""
    if seed < 3:
        nid, lo, lab, eo, et, ref = synth.layered_graph(300, max_width=4, max_len=7, seed=seed)
    else:
        nid, lo, lab, eo, et, ref = synth.bubble_graph(40_000, seed=seed)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    for k, npaths, step in ((11, 0, 1), (21, 1, 1), (21, 3, 2), (31, 5, 3), (8, 2, 1)):
        host = psi_amd.PathIndex.build(g, k, npaths, step=step, rng_seed=seed)
        dev = psi_amd.PathIndex.build(g, k, npaths, step=step, rng_seed=seed, device=0)
        hn, ho = host.loci
        dn, do = dev.loci
        assert len(hn) == len(dn) and (hn == dn).all() and (ho == do).all(), (k, npaths, step)
    if seed == 0:
        # a cycle 2 -> 3 -> 2 (ranks 1, 2) and a path that goes round it: the path is indexed but gets no bit
        lab = b'ACGTACGTAC' + b'GGTCA' + b'TTGAC' + b'CATGCATGCATG'
        cut = [0, 10, 15, 20, 32]
        g2 = psi_amd.Graph.from_csr([1, 2, 3, 4], cut, lab, [0, 1, 2, 4, 4], [1, 2, 3, 1], paths=[[0, 1, 2, 3]])
        for paths in ([[0, 1, 2, 3]], [[0, 1, 2, 1, 2, 3]], [[0, 1, 2, 3], [1, 2, 1, 2]]):
            host = psi_amd.PathIndex.build_paths(g2, 9, paths)
            dev = psi_amd.PathIndex.build_paths(g2, 9, paths, device=0)
            assert (host.loci[0] == dev.loci[0]).all() and (host.loci[1] == dev.loci[1]).all()


def test_device_starting_loci_of_patched_paths(monkeypatch, capfd):
    """Starting loci of TRIMMED paths (psikt's default: patches with a context), of more than 64 paths and of paths that
    come back to a node, found on the device by path steps (round 5: k_steps_loci_*) -- the host routine's own scheme --
    with the same result in the same order; the device routine is the one that ran (it says so under PSIGPU_TRACE)."""
    monkeypatch.setenv('PSIGPU_TRACE', '1')
    said = 'starting loci by path steps on the device: 0 node(s) left to the host'
    cases = []
    for name in ('x.gfa', 'm.gfa', 'multi.gfa', 'tiny.gfa'):
        g = psi_amd.Graph.load(os.path.join(REF, name))
        for k, npaths, context, step in ((21, 4, 0, 1), (21, 3, 26, 1), (12, 5, 12, 2), (31, 2, 36, 1), (10, 8, 11, 3)):
            cases.append((g, k, npaths, context, step))
    for seed in (3, 4):
        nid, lo, lab, eo, et, ref = synth.bubble_graph(40_000, seed=seed)
        g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
        cases += [(g, 21, 3, 0, 1), (g, 16, 6, 21, 2)]
    n_said = 0
    for g, k, npaths, context, step in cases:
        host = psi_amd.PathIndex.build(g, k, npaths, step=step, rng_seed=k, patched=True, context=context)
        capfd.readouterr()
        dev = psi_amd.PathIndex.build(g, k, npaths, step=step, rng_seed=k, patched=True, context=context, device=0)
        n_said += said in capfd.readouterr().err      # (patches that came out as whole simple paths take the coverage-bit routine)
        assert host.trims() == dev.trims() and [a.tolist() for a in host.paths()] == [a.tolist() for a in dev.paths()]
        hn, ho = host.loci
        dn, do = dev.loci
        assert len(hn) == len(dn) and (hn == dn).all() and (ho == do).all(), (k, npaths, context, step)
    assert n_said >= len(cases) // 2, (n_said, len(cases))
    # 70 full paths (more than the 64 coverage bits of the other device routine) and a path that goes round a cycle
    nid, lo, lab, eo, et, ref = synth.layered_graph(300, max_width=4, max_len=7, seed=1)
    g = psi_amd.Graph.from_csr(nid, lo, lab, eo, et, paths=[ref])
    # (seventy candidate steps per node and level: the lists may pass a thread's pool -- then the host routine answers, and says so)
    host = psi_amd.PathIndex.build(g, 15, 70, rng_seed=2)
    capfd.readouterr()
    dev = psi_amd.PathIndex.build(g, 15, 70, rng_seed=2, device=0)
    assert 'starting loci by path steps on the device' in capfd.readouterr().err
    assert (host.loci[0] == dev.loci[0]).all() and (host.loci[1] == dev.loci[1]).all()
    # the host routine on request
    monkeypatch.setenv('PSIGPU_HOST_LOCI', '1')
    capfd.readouterr()
    dev2 = psi_amd.PathIndex.build(g, 15, 70, rng_seed=2, device=0)
    assert 'by path steps' not in capfd.readouterr().err and (dev2.loci[0] == host.loci[0]).all()


def test_device_build_repeats_and_n_runs():
    """identical paths (LCP = whole path: many doubling rounds), N runs, separators"""
    lab = (b'ACGTTGCAACGTTGCA' * 40) + b'NNNN' + (b'GATTACA' * 30) + b'N' + b'ACGTTGCAACGTTGCA' * 10
    n_nodes = 12
    cut = np.linspace(0, len(lab), n_nodes + 1).astype(int)
    g = psi_amd.Graph.from_csr(np.arange(1, n_nodes + 1), cut, lab, list(range(n_nodes)) + [n_nodes - 1],
                               list(range(1, n_nodes)), paths=[list(range(n_nodes))])
    paths = [list(range(n_nodes))] * 3 + [list(range(3, n_nodes))]
    host = psi_amd.PathIndex.build_paths(g, 13, paths, sa_rate=1, keep=True, ftab_len=6)
    dev = psi_amd.PathIndex.build_paths(g, 13, paths, sa_rate=1, keep=True, ftab_len=6, device=0)
    assert (host.sa() == dev.sa()).all()
    _same_index(host, dev)


def test_device_build_at_scale_and_queries():
    """5 Mbp SNV graph: identical index, and the finder gives the same hits with either."""
    sg = synth.snv_graph(5_000_000, 110_000, n_block=300_000, seed=3)
    g = psi_amd.Graph.from_csr(sg.node_id, sg.label_off, sg.labels, sg.edge_off, sg.edge_to, paths=[sg.ref_path])
    t0 = time.time()
    host = psi_amd.PathIndex.build(g, 21, 2, rng_seed=1)
    t1 = time.time()
    dev = psi_amd.PathIndex.build(g, 21, 2, rng_seed=1, device=0)
    t2 = time.time()
    print('host build %.2f s, device build %.2f s (text %d)' % (t1 - t0, t2 - t1, host.text_len))
    _same_index(host, dev)
    bases, off = synth.sim_reads_snv(sg, 20_000, 150, seed=4)
    f = psi_amd.SeedFinder(g, 21)
    f.set_path_index(dev)
    a = psi_amd.sort_unique(f.seeds_all((bases, off)))
    f.set_path_index(host)
    b = psi_amd.sort_unique(f.seeds_all((bases, off)))
    assert a.shape == b.shape and (a == b).all() and len(a) > 100_000
    f.close()


def test_device_build_verified_against_the_host(monkeypatch):
    """PSIGPU_BUILD_VERIFY (the load campaigns' switch): the device build made again on the host inside the library and
    compared array by array; a build that passes is the one a plain device build returns."""
    g = psi_amd.Graph.load(os.path.join(REF, 'm.gfa'))
    plain = psi_amd.PathIndex.build(g, 21, 3, rng_seed=5, device=0)
    monkeypatch.setenv('PSIGPU_BUILD_VERIFY', '1')
    checked = psi_amd.PathIndex.build(g, 21, 3, rng_seed=5, device=0)
    _same_index(plain, checked)
    parts = psi_amd.PathIndex.build(g, 21, 3, rng_seed=5, device=0, max_part_text=int(plain.text_len) // 2 + 40)
    assert parts.view.n_more_parts >= 1
