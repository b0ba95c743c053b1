"""Brute-force DEFINITION of the seed-hit set.  TEST INFRASTRUCTURE ONLY.

This file is part of the oracle: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``psi_amd``) never does.

It states *what* the hit set is, with no index at all, straight from the
reference's semantics:

  H = { (node, offset, read_id, read_offset) :
        the k-mer of read `read_id` at `read_offset` (offsets 0, d, 2d, ... while
        i < len-k+1; reference include/psi/sequence.hpp:1711-1714) contains no N
        (include/psi/index_iter.hpp:831, include/psi/traverser_bfs.hpp:124) and is
        spelled by some walk in the graph that starts at base `offset` of `node`
        and follows out-edges (include/psi/traverser_bfs.hpp:114-161) }

The reference's own emission stream is a multiset superset of H whose order
depends on randomised path selection (include/psi/graph.hpp:172-187), so
parity is on sort-unique(H).  Pinned against the reference's own data:
``test/data/small/20-mers`` (test/src/test_graphiter.cpp:313-417) and the
traverser truth table (test/src/test_traverser.cpp:81-82) -- see
tests/test_oracle_golden.py.

Pure Python: small cases only.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple


@dataclass
class Graph:
    """Forward-only sequence graph.  `ids` keeps file order (= node rank order)."""
    ids: List[int] = field(default_factory=list)
    seq: Dict[int, str] = field(default_factory=dict)
    out: Dict[int, List[int]] = field(default_factory=dict)
    paths: List[Tuple[str, List[int]]] = field(default_factory=list)

    def add_node(self, nid: int, s: str) -> None:
        if nid not in self.seq:
            self.ids.append(nid)
            self.out[nid] = []
        self.seq[nid] = s

    def add_edge(self, a: int, b: int) -> None:
        if b not in self.out[a]:
            self.out[a].append(b)

    @property
    def total_len(self) -> int:
        return sum(len(self.seq[i]) for i in self.ids)


# --------------------------------------------------------------------------
# Loaders (the oracle has its own; the product's C++ loader is independent)
# --------------------------------------------------------------------------

def _strip_orient(tok: str) -> Tuple[int, bool]:
    return int(tok[:-1]), tok[-1] == '-'


def parse_gfa(path: str) -> Graph:
    """GFA 2.0 (S id len seq / E id a+ b+ ... / O name n+ ...) as in the reference's
    fixtures (test/data/tiny/tiny.gfa:1-37), and GFA 1 (S id seq / L a + b + 0M / P)."""
    g = Graph()
    edges: List[Tuple[int, int]] = []
    with open(path) as fh:
        for line in fh:
            f = line.rstrip('\n').split('\t')
            if not f or not f[0]:
                continue
            t = f[0]
            if t == 'S':
                # GFA2 has an integer length in column 3; GFA1 has the sequence there.
                if len(f) >= 4 and f[2].isdigit() and not f[3].startswith(('LN:', 'RC:')):
                    g.add_node(int(f[1]), f[3].upper())
                else:
                    g.add_node(int(f[1]), f[2].upper())
            elif t == 'E':
                (a, ar), (b, br) = _strip_orient(f[2]), _strip_orient(f[3])
                if ar and br:
                    a, b = b, a
                elif ar or br:
                    raise ValueError('reversing edge unsupported: ' + line)
                edges.append((a, b))
            elif t == 'L':
                a, ar, b, br = int(f[1]), f[2] == '-', int(f[3]), f[4] == '-'
                if ar and br:
                    a, b = b, a
                elif ar or br:
                    raise ValueError('reversing edge unsupported: ' + line)
                edges.append((a, b))
            elif t == 'O':
                g.paths.append((f[1], [_strip_orient(x)[0] for x in f[2].split()]))
            elif t == 'P':
                g.paths.append((f[1], [_strip_orient(x)[0] for x in f[2].split(',')]))
    for a, b in edges:
        g.add_edge(a, b)
    return g


def _varint(buf: bytes, p: int) -> Tuple[int, int]:
    x = s = 0
    while True:
        b = buf[p]
        p += 1
        x |= (b & 0x7F) << s
        if not b & 0x80:
            return x, p
        s += 7


def _pb_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    p = 0
    while p < len(buf):
        key, p = _varint(buf, p)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 2:
            ln, p = _varint(buf, p)
            v = buf[p:p + ln]
            p += ln
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        else:
            raise ValueError('bad wire type')
        yield fno, wt, v


def parse_vg(path: str) -> Graph:
    """vg protobuf stream (reference vg/vg.proto:13-103, vg/stream.hpp:81-130):
    gzip → repeated [varint count, count × (varint len, bytes)]; the first message of a
    group may be the type tag "VG" (skipped)."""
    raw = gzip.open(path, 'rb').read()
    g = Graph()
    edges: List[Tuple[int, int]] = []
    paths: Dict[str, List[Tuple[int, int]]] = {}
    p = 0
    while p < len(raw):
        cnt, p = _varint(raw, p)
        for _ in range(cnt):
            ln, p = _varint(raw, p)
            msg = raw[p:p + ln]
            p += ln
            if msg == b'VG':
                continue
            for fno, wt, v in _pb_fields(msg):
                if fno == 1 and wt == 2:      # Node {sequence=1, name=2, id=3}
                    nid, s = 0, ''
                    for a, _, b in _pb_fields(v):
                        if a == 1:
                            s = b.decode()
                        elif a == 3:
                            nid = b
                    g.add_node(nid, s.upper())
                elif fno == 2 and wt == 2:    # Edge {from=1,to=2,from_start=3,to_end=4}
                    d = {1: 0, 2: 0, 3: 0, 4: 0}
                    for a, _, b in _pb_fields(v):
                        if a in d:
                            d[a] = b
                    a_, b_ = d[1], d[2]
                    if d[3] and d[4]:
                        a_, b_ = b_, a_
                    elif d[3] or d[4]:
                        raise ValueError('reversing edge unsupported')
                    edges.append((a_, b_))
                elif fno == 3 and wt == 2:    # Path {name=1, mapping=2}
                    name, maps = '', []
                    for a, _, b in _pb_fields(v):
                        if a == 1:
                            name = b.decode()
                        elif a == 2:          # Mapping {position=1, edit=2, rank=5}
                            nid = rank = 0
                            for c, _, dd in _pb_fields(b):
                                if c == 1:    # Position {node_id=1, offset=2, is_reverse=4}
                                    for e, _, ff in _pb_fields(dd):
                                        if e == 1:
                                            nid = ff
                                elif c == 5:
                                    rank = dd
                            maps.append((rank, nid))
                    paths.setdefault(name, []).extend(maps)
    for a, b in edges:
        g.add_edge(a, b)
    for name, maps in paths.items():
        maps.sort()
        g.paths.append((name, [n for _, n in maps]))
    return g


def read_seqs(path: str) -> List[str]:
    """Plain one-read-per-line, or FASTQ (4-line records)."""
    with open(path) as fh:
        lines = [l.rstrip('\n') for l in fh]
    if lines and lines[0].startswith('@'):
        return [lines[i + 1].upper() for i in range(0, len(lines) - 1, 4)]
    return [l.upper() for l in lines if l]


# --------------------------------------------------------------------------
# The definition
# --------------------------------------------------------------------------

def seeding(reads: Sequence[str], k: int, step: int) -> List[Tuple[int, int, str]]:
    """(read index, offset, k-mer) for offsets 0, step, ... while i < len-k+1
    (reference include/psi/sequence.hpp:1711-1714).  A read shorter than k gives none
    (the reference's unsigned wrap at :1712 is a defect)."""
    if step == 0:
        step = k        # reference src/psikt.cpp:469
    out = []
    for r, s in enumerate(reads):
        for i in range(0, len(s) - k + 1, step):
            out.append((r, i, s[i:i + k]))
    return out


def kwalks_from(g: Graph, v: int, o: int, k: int) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """Every walk spelling exactly k bases from base o of node v, first-out-edge-first
    DFS (the Backtracker order of test/src/test_graphiter.cpp:313-417).  Yields
    (k-mer, node sequence of the walk).  Walks hitting a sink before k bases yield nothing
    (include/psi/traverser_bfs.hpp:141-144)."""
    stack = [(v, o, '', (v,))]
    while stack:
        n, off, acc, nodes = stack.pop()
        s = g.seq[n]
        take = s[off:off + k - len(acc)]
        acc2 = acc + take
        if len(acc2) == k:
            yield acc2, nodes
            continue
        for t in reversed(g.out[n]):
            stack.append((t, 0, acc2, nodes + (t,)))


def all_kwalks(g: Graph, k: int) -> Iterator[Tuple[str, int, int, Tuple[int, ...]]]:
    for v in g.ids:
        for o in range(len(g.seq[v])):
            for km, nodes in kwalks_from(g, v, o, k):
                yield km, v, o, nodes


def hit_set(g: Graph, reads: Sequence[str], k: int, step: int,
            rec_offset: int = 0) -> List[Tuple[int, int, int, int]]:
    """sort-unique(H) as a list of (node_id, node_offset, read_id, read_offset)."""
    table: Dict[str, List[Tuple[int, int]]] = {}
    for r, i, km in seeding(reads, k, step):
        if 'N' in km:
            continue
        table.setdefault(km, []).append((r + rec_offset, i))
    hits = set()
    for km, v, o, _ in all_kwalks(g, k):
        if 'N' in km:
            continue
        for r, i in table.get(km, ()):
            hits.add((v, o, r, i))
    return sorted(hits)


def uncovered_loci(g: Graph, paths: Iterable[Sequence[int]], k: int,
                   trims: Sequence[Tuple[int, int]] = ()) -> List[Tuple[int, int]]:
    """Loci with at least one k-walk that is not a contiguous run of some indexed path -- the intent
    of SeedFinder::add_uncovered_loci (reference include/psi/seed_finder.hpp:1481-1541) for step 1.
    With no paths every locus that has a k-walk qualifies (cf. add_all_loci, :1543-1585, which adds
    all).  `trims[i]` = (head offset, tail length) of path i, for patched paths
    (pathindex.hpp:496-560; Path::left / right, path_base.hpp:113-114): the path covers its first node
    from base `head` on and of its last node the first `tail` bases (0 = all); a walk is covered only
    if the path spells all of its bases."""
    paths = [tuple(p) for p in paths]
    trims = list(trims) + [(0, 0)] * (len(paths) - len(trims))
    where: List[Dict[int, List[int]]] = []          # per path: node -> its positions
    for p in paths:
        d: Dict[int, List[int]] = {}
        for i, u in enumerate(p):
            d.setdefault(u, []).append(i)
        where.append(d)
    out = set()
    for _, v, o, nodes in all_kwalks(g, k):
        m = len(nodes)
        # bases of the walk inside its last node
        used = k if m == 1 else k - (len(g.seq[v]) - o) - sum(len(g.seq[u]) for u in nodes[1:-1])
        covered = False
        for p, (head, tail), d in zip(paths, trims, where):
            for i in d.get(v, ()):
                if p[i:i + m] != nodes:
                    continue
                if i == 0 and o < head:
                    continue
                if i + m == len(p) and tail and (o + used if m == 1 else used) > tail:
                    continue
                covered = True
                break
            if covered:
                break
        if not covered:
            out.add((v, o))
    rank = {v: i for i, v in enumerate(g.ids)}
    return sorted(out, key=lambda t: (rank[t[0]], t[1]))


def path_texts(g: Graph, paths: Iterable[Sequence[int]], trims: Sequence[Tuple[int, int]] = ()):
    """The indexed (forward) text of every path with the graph position of each base:
    [(text, [(node, offset) per base])].  Trimmed as Path::left / right trim a patch
    (reference include/psi/path_base.hpp:113-114, :240-246)."""
    paths = [list(p) for p in paths]
    trims = list(trims) + [(0, 0)] * (len(paths) - len(trims))
    out = []
    for p, (head, tail) in zip(paths, trims):
        txt, pos = [], []
        for i, v in enumerate(p):
            s = g.seq[v]
            b = head if i == 0 else 0
            e = tail if (i + 1 == len(p) and tail) else len(s)
            for o in range(b, e):
                txt.append(s[o])
                pos.append((v, o))
        out.append((''.join(txt), pos))
    return out


def find_mems(g: Graph, paths: Iterable[Sequence[int]], reads: Sequence[str], minlen: int,
              trims: Sequence[Tuple[int, int]] = (), gocc_thr: int = 0, max_mem: int = 0,
              rec_offset: int = 0) -> List[Tuple[int, int, int, int, int, int]]:
    """find_mems of the reference (include/psi/index_iter.hpp:854-906) as driven by
    SeedFinder::seeds_on_paths( sequence, callback ) (seed_finder.hpp:1459-1479), with naive
    substring search standing in for the index iterator: go_down( c ) succeeds iff pattern + c
    occurs in a path text; count_occurrences = number of occurrences over all paths.
    Returns sorted (node_id, node_offset, read_id, read_offset, match_len, gocc)."""
    texts = path_texts(g, paths, trims)
    gocc_thr = gocc_thr or (1 << 62)
    max_mem = max_mem or (1 << 62)

    def occurrences(pat: str):
        occ = []
        for txt, pos in texts:
            i = txt.find(pat)
            while i >= 0:
                occ.append(pos[i])
                i = txt.find(pat, i + 1)
        return occ

    n_total = sum(len(t) for t, _ in texts)
    out = []
    for rid, pattern in enumerate(reads):
        pattern = pattern.upper()
        start = plen = 0
        has_hit = False
        nof = 0
        occ = None                      # occurrences of pattern[start:start+plen]; None = root (everything)
        while start + plen < len(pattern):
            cnt = n_total if occ is None else len(occ)
            if plen >= minlen and cnt <= gocc_thr:
                has_hit = True
                for v, o in occ:
                    out.append((v, o, rid + rec_offset, start, plen, len(occ)))
                nof += len(occ)
                if nof >= max_mem:
                    break
            c = pattern[start + plen]
            nxt = None
            if not has_hit and c in 'ACGT':
                nxt = occurrences(pattern[start:start + plen + 1])
            if not nxt:
                occ = None
                start, plen, has_hit = start + plen + 1, 0, False
                continue
            occ = nxt
            plen += 1
    return sorted(out, key=lambda t: (t[2], t[3], t[0], t[1]))
