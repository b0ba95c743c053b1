"""Seeded synthetic graphs and reads (no network, no real chr22 graph here).

The reference ships a simulator (tools/src/ggsim.cpp:231-264: uniform start on a simulated
haplotype, substitution errors); this is an independent, vectorised generator of the same
kind of input, shaped like a vg graph built from a linear reference plus bi-allelic SNVs:
nodes of at most `max_node` bases (vg's default chopping is 32), a two-node bubble per SNV.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

_ACGT = np.frombuffer(b'ACGT', dtype=np.uint8)


@dataclass
class SnvGraph:
    node_id: np.ndarray      # u64 [n]
    label_off: np.ndarray    # u64 [n+1]
    labels: np.ndarray       # u8  ASCII
    edge_off: np.ndarray     # u64 [n+1]
    edge_to: np.ndarray      # u32
    ref_path: np.ndarray     # u32 node ranks
    backbone: np.ndarray     # u8 ASCII, length L (reference allele everywhere)
    alt: np.ndarray          # u8 ASCII alt base at SNV positions, 0 elsewhere
    n_block: int             # leading N run

    @property
    def n_nodes(self):
        return len(self.node_id)


def clustered_positions(rng, lo: int, hi: int, n: int, frac: float = 0.2, window: int = 200, spacing: int = 8) -> np.ndarray:
    """`n` distinct sorted positions in [lo, hi): a share `frac` of them in windows of `window` bp where sites come every
    ~`spacing` bp (geometric gaps), the rest uniform -- real variation clusters (1000G: hotspots, HLA), uniform SNVs do not."""
    n_cl = int(n * frac)
    per = max(1, window // spacing)
    n_win = max(1, (n_cl + per - 1) // per) if n_cl else 0
    starts = rng.integers(lo, max(lo + 1, hi - window), size=n_win)
    gaps = rng.geometric(1.0 / spacing, size=(n_win, per * 2))
    offs = np.cumsum(gaps, axis=1)
    cl = (starts[:, None] + offs)[offs < window] if n_win else np.zeros(0, np.int64)
    if len(cl) > n_cl:
        cl = rng.choice(cl, size=n_cl, replace=False)
    uni = rng.integers(lo, hi, size=int((n - len(cl)) * 1.02) + 8)
    pos = np.unique(np.concatenate([cl.astype(np.int64), uni]))
    pos = pos[(pos >= lo) & (pos < hi)]
    if len(pos) > n:
        # thin the UNIFORM part only, so that the clusters keep their density
        is_cl = np.isin(pos, cl)
        drop = rng.choice(np.flatnonzero(~is_cl), size=min(len(pos) - n, int((~is_cl).sum())), replace=False)
        pos = np.delete(pos, drop)
    return np.sort(pos)


def snv_graph(length: int, n_snv: int, n_block: int = 0, max_node: int = 32,
              seed: int = 11, cluster_frac: float = 0.0, cluster_window: int = 200, cluster_spacing: int = 8) -> SnvGraph:
    """Linear backbone of `length` uniform ACGT bases (first `n_block` are N), `n_snv`
    bi-allelic SNV bubbles at uniform distinct positions outside the N block.
    `cluster_frac` > 0: that share of the sites sits in windows of `cluster_window` bp at one site per ~`cluster_spacing` bp
    (round 6: the uniform stand-in is kind to the tabulating modes -- 2.9 k-walks per locus; clustered sites are what
    1000G variation looks like)."""
    rng = np.random.default_rng(seed)
    code = rng.integers(0, 4, size=length, dtype=np.uint8)
    backbone = _ACGT[code]
    backbone[:n_block] = ord('N')
    lo = n_block + 1
    n_snv = min(n_snv, max(0, (length - 1 - lo)))
    rng2 = np.random.default_rng(seed + 1)
    if n_snv and cluster_frac > 0:
        pos = clustered_positions(rng2, lo, length - 1, n_snv, cluster_frac, cluster_window, cluster_spacing)
    elif n_snv:
        pos = np.unique(rng2.integers(lo, length - 1, size=int(n_snv * 1.02) + 8))
        if len(pos) > n_snv:
            pos = np.sort(rng2.choice(pos, size=n_snv, replace=False))
    else:
        pos = np.zeros(0, np.int64)
    alt = np.zeros(length, np.uint8)
    if len(pos):
        alt[pos] = _ACGT[(code[pos] + rng2.integers(1, 4, size=len(pos)).astype(np.uint8)) % 4]

    # intervals of plain backbone between SNVs: [a_i, b_i)
    a = np.concatenate([[0], pos + 1]).astype(np.int64)
    b = np.concatenate([pos, [length]]).astype(np.int64)
    ilen = b - a
    nchunk = (ilen + max_node - 1) // max_node                       # chunks per interval (0 if empty)
    # layers: for interval i: nchunk[i] single-node layers, then (if i < n_snv) one 2-node layer
    has_snv = np.zeros(len(a), np.int64)
    has_snv[:len(pos)] = 1
    # chunk nodes
    tot_chunks = int(nchunk.sum())
    chunk_iv = np.repeat(np.arange(len(a)), nchunk)
    first_chunk = np.cumsum(nchunk) - nchunk
    within = np.arange(tot_chunks) - first_chunk[chunk_iv]
    c_start = a[chunk_iv] + within * max_node
    c_end = np.minimum(c_start + max_node, b[chunk_iv])
    # node numbering: per interval, its chunks then ref allele, alt allele
    nodes_before_iv = np.cumsum(nchunk + 2 * has_snv) - (nchunk + 2 * has_snv)
    chunk_node = nodes_before_iv[chunk_iv] + within
    snv_ref_node = nodes_before_iv[:len(pos)] + nchunk[:len(pos)]
    snv_alt_node = snv_ref_node + 1
    n_nodes = int((nchunk + 2 * has_snv).sum())
    # node start / length in backbone coordinates
    nstart = np.zeros(n_nodes, np.int64)
    nlen = np.zeros(n_nodes, np.int64)
    nstart[chunk_node] = c_start
    nlen[chunk_node] = c_end - c_start
    nstart[snv_ref_node] = pos
    nstart[snv_alt_node] = pos
    nlen[snv_ref_node] = 1
    nlen[snv_alt_node] = 1
    is_alt = np.zeros(n_nodes, bool)
    is_alt[snv_alt_node] = True
    label_off = np.zeros(n_nodes + 1, np.uint64)
    label_off[1:] = np.cumsum(nlen)
    # labels: nodes follow the backbone in order, with every alt allele right behind its ref allele
    labels = np.insert(backbone, pos + 1, alt[pos]) if len(pos) else backbone.copy()
    # layers -> edges: every node of layer l points at every node of layer l+1
    layer_of = np.zeros(n_nodes, np.int64)
    # layer index: chunks and snv layers in order of node number, alt shares its ref's layer
    new_layer = np.ones(n_nodes, np.int64)
    new_layer[snv_alt_node] = 0
    layer_of = np.cumsum(new_layer) - 1
    n_layers = int(layer_of[-1]) + 1 if n_nodes else 0
    layer_first = np.zeros(n_layers + 1, np.int64)
    layer_first[:-1] = np.flatnonzero(new_layer)
    layer_first[-1] = n_nodes
    layer_size = np.diff(layer_first)
    next_size = np.concatenate([layer_size[1:], [0]])
    deg = next_size[layer_of]
    edge_off = np.zeros(n_nodes + 1, np.uint64)
    edge_off[1:] = np.cumsum(deg)
    n_edges = int(edge_off[-1])
    src = np.repeat(np.arange(n_nodes), deg)
    j = np.arange(n_edges) - edge_off[:-1].astype(np.int64)[src]
    edge_to = (layer_first[layer_of[src] + 1] + j).astype(np.uint32)
    ref_path = np.flatnonzero(~is_alt).astype(np.uint32)
    return SnvGraph(node_id=np.arange(1, n_nodes + 1, dtype=np.uint64), label_off=label_off,
                    labels=labels, edge_off=edge_off, edge_to=edge_to, ref_path=ref_path,
                    backbone=backbone, alt=alt, n_block=n_block)


def sim_reads_snv(sg: SnvGraph, n_reads: int, read_len: int = 150, seed: int = 13,
                  sub_rate: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """`n_reads` forward-strand reads of `read_len` bases; start uniform over the non-N
    backbone, every SNV inside a read takes the alt allele with probability 1/2 (a random
    haplotype walk per read); optional substitution errors.  Returns (bases u8, read_off u64)."""
    rng = np.random.default_rng(seed)
    L = len(sg.backbone)
    lo, hi = sg.n_block, L - read_len
    if hi <= lo:
        raise ValueError('backbone too short for the read length')
    out = np.empty(n_reads * read_len, np.uint8)
    step = max(1, (1 << 24) // read_len)
    for s in range(0, n_reads, step):
        m = min(step, n_reads - s)
        start = rng.integers(lo, hi, size=m)
        idx = start[:, None] + np.arange(read_len)[None, :]
        r = sg.backbone[idx]
        a = sg.alt[idx]
        take = (a != 0) & (rng.random(size=idx.shape) < 0.5)
        r = np.where(take, a, r)
        if sub_rate > 0:
            e = rng.random(size=idx.shape) < sub_rate
            r = np.where(e, _ACGT[rng.integers(0, 4, size=idx.shape)], r)
        out[s * read_len:(s + m) * read_len] = r.reshape(-1)
    off = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return out, off


def layered_graph(n_layers: int, max_width: int = 3, max_len: int = 9, seed: int = 0,
                  p_n: float = 0.0, p_edge: float = 0.8):
    """Random layered DAG for tests: layer widths 1..max_width, node lengths 1..max_len, each
    node keeps every edge to the next layer with probability p_edge (at least one).  Returns
    (node_id, label_off, labels, edge_off, edge_to, ref_path)."""
    rng = np.random.default_rng(seed)
    widths = rng.integers(1, max_width + 1, size=n_layers)
    first = np.concatenate([[0], np.cumsum(widths)])
    n = int(first[-1])
    lens = rng.integers(1, max_len + 1, size=n)
    label_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    labels = _ACGT[rng.integers(0, 4, size=int(label_off[-1]))]
    if p_n > 0:
        labels[rng.random(len(labels)) < p_n] = ord('N')
    edge_off = [0]
    edge_to: List[int] = []
    for l in range(n_layers):
        for v in range(first[l], first[l + 1]):
            if l + 1 < n_layers:
                nxt = np.arange(first[l + 1], first[l + 2])
                keep = nxt[rng.random(len(nxt)) < p_edge]
                if len(keep) == 0:
                    keep = nxt[:1]
                if v == first[l] and nxt[0] not in keep:      # keep the reference path connected
                    keep = np.concatenate([[nxt[0]], keep])
                edge_to.extend(int(t) for t in keep)
            edge_off.append(len(edge_to))
    ref_path = first[:-1].astype(np.uint32)
    return (np.arange(1, n + 1, dtype=np.uint64), label_off, labels,
            np.asarray(edge_off, np.uint64), np.asarray(edge_to, np.uint32), ref_path)


def bubble_graph(length: int, seed: int = 31, site_every: int = 20, max_node: int = 32,
                 hot_frac: float = 0.0, hot_len: int = 120, hot_every: int = 3):
    """HLA-like high-branching graph (BASELINE.json configs[4]): a backbone with a variant site
    every ~`site_every` bp; sites are SNVs with 2-4 alleles, insertions / deletions of 1-50 bp
    (a deletion is an edge that skips the deleted backbone, an insertion an extra node).
    `hot_frac` > 0: that share of the backbone lies in hot regions of `hot_len` bp with a site every ~`hot_every` bp (exon 2/3 of
    a class-I gene: ten sites inside one 31-mer window -> 2^10 and more k-walks from the loci in front of it).
    Returns (node_id, label_off, labels, edge_off, edge_to, ref_path).  Built with Python loops:
    meant for graphs up to a few Mbp."""
    rng = np.random.default_rng(seed)
    hot = np.zeros(length + 1, bool)
    if hot_frac > 0:
        for h0 in rng.integers(0, max(1, length - hot_len), size=max(1, int(length * hot_frac / hot_len))):
            hot[int(h0):int(h0) + hot_len] = True
    bb = _ACGT[rng.integers(0, 4, size=length)]
    labels: List[bytes] = []
    out: List[List[int]] = []
    ref_path: List[int] = []

    def new_node(seq: bytes) -> int:
        labels.append(seq)
        out.append([])
        return len(labels) - 1

    def chain(seq: bytes) -> Tuple[int, int]:
        first = prev = -1
        for i in range(0, len(seq), max_node):
            v = new_node(seq[i:i + max_node])
            if prev >= 0:
                out[prev].append(v)
            else:
                first = v
            prev = v
        return first, prev

    pos = 0
    tails: List[int] = []            # nodes whose next edge goes to the next backbone piece
    while pos < length:
        ev = hot_every if hot[pos] else site_every
        gap = int(rng.integers(max(2, ev // 2), max(3, ev * 3 // 2 + 1)))
        end = min(length, pos + gap)
        first, last = chain(bytes(bb[pos:end]))
        node = first
        while True:
            ref_path.append(node)
            if node == last:
                break
            node = out[node][0]
        for t in tails:
            out[t].append(first)
        tails = [last]
        pos = end
        if pos >= length - 60:
            continue
        kind = rng.random() * (0.6 if hot[pos] else 1.0)      # (hot regions: substitutions only, as in the exons)
        if kind < 0.6:                                   # SNV with 2..4 alleles
            n_all = int(rng.choice([2, 2, 2, 3, 4]))
            ref_base = int(np.where(_ACGT == bb[pos])[0][0])
            alleles = [ref_base] + [int(a) for a in rng.permutation([b for b in range(4) if b != ref_base])[:n_all - 1]]
            nodes = [new_node(bytes(_ACGT[a:a + 1])) for a in alleles]
            for v in nodes:
                out[last].append(v)
            ref_path.append(nodes[0])
            tails = nodes
            pos += 1
        elif kind < 0.8:                                 # insertion of 1..50 bp (optional node)
            ins = bytes(_ACGT[rng.integers(0, 4, size=int(rng.integers(1, 51)))])
            f, l = chain(ins)
            out[last].append(f)
            tails = [last, l]
        else:                                            # deletion of 1..50 bp (skip edge)
            dl = int(rng.integers(1, 51))
            f, l = chain(bytes(bb[pos:pos + dl]))
            out[last].append(f)
            node = f
            while True:
                ref_path.append(node)
                if node == l:
                    break
                node = out[node][0]
            tails = [last, l]
            pos += dl
    n = len(labels)
    label_off = np.zeros(n + 1, np.uint64)
    label_off[1:] = np.cumsum([len(x) for x in labels])
    lab = np.frombuffer(b''.join(labels), dtype=np.uint8).copy()
    edge_off = np.zeros(n + 1, np.uint64)
    edge_off[1:] = np.cumsum([len(x) for x in out])
    edge_to = np.array([t for x in out for t in x], dtype=np.uint32)
    return (np.arange(1, n + 1, dtype=np.uint64), label_off, lab, edge_off, edge_to,
            np.array(ref_path, dtype=np.uint32))


def sim_reads_walk(node_id, label_off, labels, edge_off, edge_to, n_reads: int, read_len: int,
                   seed: int = 0):
    """Reads as random walks from random loci (uniform over out-edges).  Python loop."""
    rng = np.random.default_rng(seed)
    n = len(node_id)
    lo = label_off.astype(np.int64)
    reads = []
    while len(reads) < n_reads:
        v = int(rng.integers(0, n))
        ln = int(lo[v + 1] - lo[v])
        if ln == 0:
            continue
        o = int(rng.integers(0, ln))
        s = bytes(labels[lo[v] + o:lo[v + 1]])
        while len(s) < read_len and edge_off[v + 1] > edge_off[v]:
            v = int(edge_to[int(rng.integers(int(edge_off[v]), int(edge_off[v + 1])))])
            s += bytes(labels[lo[v]:lo[v + 1]])
        if len(s) >= read_len:
            reads.append(s[:read_len])
    bases = np.frombuffer(b''.join(reads), dtype=np.uint8).copy()
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return bases, off
