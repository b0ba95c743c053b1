#!/usr/bin/env python3
"""Writes tests/golden/ref_paths_x_trimmed.bin: the bytes the reference's PathIndex::save_paths_set
(include/psi/pathindex.hpp:315-332) produces for the three trimmed paths of test/src/test_pathindex.cpp:248-255
on graph `x`, up to and including the last path -- the node-id index that follows in a real file
(pathset.hpp:266-272) is not needed to recover the paths and is not read.

The file format is sdsl-lite's (the library is not in the reference tree; restated from its published
serialisation, sdsl-lite 2.1.1 `int_vector.hpp` / `enc_vector.hpp` / `coder_elias_delta.hpp`):

  u64 context | u64 direction (1 Forward, 0 Reversed) | u64 #paths | per path:
    enc_vector< coder::elias_delta<>, 128 >:  u64 size
        int_vector<0> z:  u64 size in bits, u8 width (1), ceil(bits / 64) u64 words -- the Elias-delta codes of
                          v[i] - v[i-1] (mod 2^64) for every i that is not a multiple of 128, LSB first
        int_vector<0> samples_and_pointers:  u64 size in bits, u8 width w, words -- 2 * #samples + 2 entries of
                          w bits: (v[128 j], bit offset of the j-th run in z) ..., then 0, bits(z) + 1;
                          w = hi( max( max sample value, bits(z) + 1 ) ) + 1
    u64 left | u64 right   (bases of the first / last node that belong to the path, 0 = all; path_base.hpp:113-114)
    bit_vector node breaks:  u64 size in bits (= length of the path sequence), words; 1 at the last base of every node

    Elias-delta of x (x = 0 stands for 2^64):  len = bits of x (65 for x = 0), ll = hi(len):
        ll zero bits, a one bit, the low ll bits of len, the low len - 1 bits of x  -- all LSB first.

This writer is TEST infrastructure: an independent statement of the format against which the product's
reader (psi_amd/csrc/refio.cpp) is checked.  `python tests/golden/make_ref_paths.py` rewrites the fixture.
"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


class Bits:
    def __init__(self):
        self.v = 0
        self.n = 0

    def put(self, x, width):
        self.v |= (x & ((1 << width) - 1)) << self.n
        self.n += width

    def words(self):
        nw = (self.n + 63) // 64
        return b''.join(struct.pack('<Q', (self.v >> (64 * i)) & ((1 << 64) - 1)) for i in range(nw))


def hi(x):
    return x.bit_length() - 1


def elias_delta(bits, x):
    ln = x.bit_length() if x else 65
    ll = hi(ln)
    bits.put(1 << ll, ll + 1)
    if ll:
        bits.put(ln, ll)
        bits.put(x, ln - 1)


def int_vector0(bits, width):
    return struct.pack('<QB', bits.n, width) + bits.words()


def enc_vector(vals, dens=128):
    z, samples = Bits(), []
    for i, v in enumerate(vals):
        if i % dens == 0:
            samples.append((v, z.n))
        else:
            elias_delta(z, (v - vals[i - 1]) & ((1 << 64) - 1))
    out = struct.pack('<Q', len(vals))
    if not vals:                       # enc_vector of an empty container: two empty int_vectors (width 64 default)
        return out + struct.pack('<QB', 0, 64) + struct.pack('<QB', 0, 64)
    w = hi(max(max(s for s, _ in samples), z.n + 1)) + 1
    sp = Bits()
    for s, ptr in samples:
        sp.put(s, w)
        sp.put(ptr, w)
    sp.put(0, w)
    sp.put(z.n + 1, w)
    return out + int_vector0(z, 1) + int_vector0(sp, w)


def bit_vector(ones, n):
    b = Bits()
    b.n = n
    for i in ones:
        b.v |= 1 << i
    return struct.pack('<Q', n) + b.words()


def path_record(ids, node_len, left, right):
    lens = [node_len[v] for v in ids]
    if len(ids) == 1:                 # one node: bases [len - left, right)
        lens[0] = (right or lens[0]) - (lens[0] - left if left else 0)
    else:
        if left:
            lens[0] = left
        if right:
            lens[-1] = right
    ends, at = [], 0
    for ln in lens:
        at += ln
        ends.append(at - 1)
    return enc_vector(ids) + struct.pack('<QQ', left, right) + bit_vector(ends, at)


def paths_file(context, forward, paths, node_len):
    out = struct.pack('<QQQ', context, 1 if forward else 0, len(paths))
    for ids, left, right in paths:
        # Path::set_left_by_len / set_right_by_len (path_base.hpp:382-438) store 0 for "the whole node"
        left = 0 if left >= node_len[ids[0]] else left
        right = 0 if right >= node_len[ids[-1]] else right
        out += path_record(ids, node_len, left, right)
    return out


def x_node_lengths():
    lens = {}
    for line in open(os.path.join(HERE, 'ref_data', 'x.gfa')):
        f = line.rstrip('\n').split('\t')
        if f[0] == 'S':                # GFA 2: S id length sequence
            lens[int(f[1])] = int(f[2])
    return lens


def main():
    nl = x_node_lengths()
    paths = [([205, 207, 209, 210], 9, 9), ([187, 189, 191, 193, 194, 195, 197], 9, 9), ([167, 168, 171, 172, 174], 9, 9)]
    data = paths_file(10, False, paths, nl)
    open(os.path.join(HERE, 'ref_paths_x_trimmed.bin'), 'wb').write(data)
    print(len(data), 'bytes')


if __name__ == '__main__':
    sys.exit(main())
