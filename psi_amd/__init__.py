"""psi_amd -- host-side Python face of libpsi_gpu.so (the MI355X-native seed finder).

Thin ctypes binding over the C ABI in include/psi_gpu.h plus a mirror of the reference's
query surface (psi::SeedFinder, reference include/psi/seed_finder.hpp:761-1788) used by the
tests and by bench.py.  There is NO CPU fallback: every query goes to the HIP kernels, and
the package fails loudly when the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PSI_AMD_LIB') or os.path.join(_HERE, 'libpsi_gpu.so')   # env: experiment builds

ALL, ON_PATHS, OFF_PATHS, SORT_UNIQUE, UNIFORM_READS, ANY_ORDER = 3, 1, 2, 4, 8, 16
MAX_SEED_LEN = 63
TUNE_NO_DIRECT, TUNE_NO_VERIFY, TUNE_NO_ROWRECS, TUNE_NO_PATH_TABLE, TUNE_NO_SWEEP = 1, 2, 4, 8, 16


class PsiGpuError(RuntimeError):
    pass


class Hit(C.Structure):
    _fields_ = [('node_id', C.c_uint64), ('node_offset', C.c_uint64),
                ('read_id', C.c_uint64), ('read_offset', C.c_uint64)]


class Hits(C.Structure):
    _fields_ = [('n', C.c_uint64), ('data', C.POINTER(Hit))]


class MemHit(C.Structure):
    _fields_ = [('node_id', C.c_uint64), ('node_offset', C.c_uint64), ('read_id', C.c_uint64),
                ('read_offset', C.c_uint64), ('match_len', C.c_uint64), ('gocc', C.c_uint64)]


class Mems(C.Structure):
    _fields_ = [('n', C.c_uint64), ('data', C.POINTER(MemHit))]


class GraphView(C.Structure):
    _fields_ = [('n_nodes', C.c_uint64), ('node_id', C.c_void_p), ('label_off', C.c_void_p),
                ('labels', C.c_void_p), ('edge_off', C.c_void_p), ('edge_to', C.c_void_p)]


class IndexView(C.Structure):
    _fields_ = [('seed_len', C.c_uint32), ('sa_rate', C.c_uint32), ('context', C.c_uint32),
                ('n_paths', C.c_uint32), ('text_len', C.c_uint64), ('n_blocks', C.c_uint64),
                ('bwt_blocks', C.c_void_p), ('C', C.c_uint64 * 4), ('n_samples', C.c_uint64),
                ('sa_samples', C.c_void_p), ('n_exc', C.c_uint64), ('exc_row', C.c_void_p),
                ('exc_sa', C.c_void_p), ('ftab_len', C.c_uint32), ('exc_shift', C.c_uint32),
                ('ftab', C.c_void_p), ('text4', C.c_void_p), ('n_segs', C.c_uint64), ('seg_start', C.c_void_p),
                ('seg_node', C.c_void_p), ('seg_noff', C.c_void_p), ('n_dir', C.c_uint64),
                ('seg_dir', C.c_void_p), ('n_loci', C.c_uint64), ('loci_node', C.c_void_p),
                ('loci_off', C.c_void_p), ('n_more_parts', C.c_uint32), ('reserved2', C.c_uint32),
                ('more_parts', C.c_void_p), ('exc_super', C.c_void_p)]


class IndexOpts(C.Structure):
    _fields_ = [('seed_len', C.c_uint32), ('n_per_region', C.c_uint32), ('locus_step', C.c_uint32),
                ('sa_rate', C.c_uint32), ('ftab_len', C.c_uint32), ('keep_text_sa', C.c_uint32),
                ('rng_seed', C.c_uint64), ('build_on_device', C.c_uint32), ('context', C.c_uint32),
                ('patched', C.c_uint32), ('reserved1', C.c_uint32), ('max_part_text', C.c_uint64)]


MODE_KMER_TABLE, MODE_TRAVERSE, MODE_LOCUS_TABLE, MODE_AUTO = 0, 1, 2, 3
_MODES = {'kmer-table': MODE_KMER_TABLE, 'traverse': MODE_TRAVERSE, 'locus-table': MODE_LOCUS_TABLE, 'auto': MODE_AUTO}
NO_FTAB = 0xFFFFFFFF


class Counters(C.Structure):
    _fields_ = [('n_reads', C.c_uint64), ('n_seeds', C.c_uint64), ('n_seeds_valid', C.c_uint64),
                ('n_seeds_on_path', C.c_uint64), ('n_hits_on_path', C.c_uint64),
                ('n_hits_off_path', C.c_uint64), ('n_hits', C.c_uint64), ('n_kpaths', C.c_uint64),
                ('n_loci', C.c_uint64), ('n_spilled', C.c_uint64), ('n_lf_steps', C.c_uint64),
                ('n_rows_verified', C.c_uint64), ('n_locus_kmers', C.c_uint64), ('n_path_kmers', C.c_uint64), ('n_loci_traversed', C.c_uint64),
                ('ms_pack', C.c_float), ('ms_table', C.c_float), ('ms_search', C.c_float),
                ('ms_locate', C.c_float), ('ms_traverse', C.c_float), ('ms_sort', C.c_float),
                ('ms_total', C.c_float), ('ms_probe', C.c_float), ('ms_locus_table_build', C.c_float),
                ('search_launches', C.c_uint32),
                ('traverse_launches', C.c_uint32), ('sorted_in_place', C.c_uint32), ('wire_bytes_per_hit', C.c_uint32),
                ('n_locate_steps', C.c_uint64), ('lookahead_subbatches', C.c_uint32), ('fused_step', C.c_uint32),
                ('lookahead_fallbacks', C.c_uint64), ('stale_handbacks', C.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


# every symbol include/psi_gpu.h declares: (name, restype, argtypes)
_P = C.c_void_p
_U64P = C.POINTER(C.c_uint64)
_INTP = C.POINTER(C.c_int)
ABI = [
    ('psigpu_abi_version', C.c_uint32, []),
    ('psigpu_host_last_error', C.c_char_p, []),
    ('psigpu_graph_load', _P, [C.c_char_p, _INTP]),
    ('psigpu_graph_load_opts', _P, [C.c_char_p, C.c_uint32, _INTP]),
    ('psigpu_graph_from_csr', _P, [C.c_uint64, _P, _P, _P, _P, _P, C.c_uint64, _P, _P, _INTP]),
    ('psigpu_graph_free', None, [_P]),
    ('psigpu_graph_view_get', C.c_int, [_P, C.POINTER(GraphView)]),
    ('psigpu_graph_path_count', C.c_uint64, [_P]),
    ('psigpu_graph_edge_count', C.c_uint64, [_P]),
    ('psigpu_graph_path', C.c_uint64, [_P, C.c_uint64, _P, C.c_uint64]),
    ('psigpu_index_build', _P, [_P, C.POINTER(IndexOpts), _INTP]),
    ('psigpu_index_build_paths', _P, [_P, C.POINTER(IndexOpts), C.c_uint64, _P, _P, _INTP]),
    ('psigpu_index_build_patches', _P, [_P, C.POINTER(IndexOpts), C.c_uint64, _P, _P, _P, _P, _INTP]),
    ('psigpu_index_path_trim', C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ('psigpu_index_from_reference_paths', _P, [_P, C.POINTER(IndexOpts), C.c_char_p, _U64P, C.POINTER(C.c_uint32), _INTP]),
    ('psigpu_index_matches', C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    ('psigpu_index_locus_step', C.c_uint32, [_P]),
    ('psigpu_index_set_locus_step', C.c_int, [_P, _P, C.c_uint32]),
    ('psigpu_loci_save', C.c_int, [_P, _P, C.c_char_p]),
    ('psigpu_loci_load', C.c_int, [_P, _P, C.c_char_p, C.c_uint32]),
    ('psigpu_index_free', None, [_P]),
    ('psigpu_index_view_get', C.c_int, [_P, C.POINTER(IndexView)]),
    ('psigpu_index_save', C.c_int, [_P, C.c_char_p]),
    ('psigpu_index_load', _P, [C.c_char_p, _INTP]),
    ('psigpu_index_path_count', C.c_uint64, [_P]),
    ('psigpu_index_path', C.c_uint64, [_P, C.c_uint64, _P, C.c_uint64]),
    ('psigpu_index_text', _P, [_P]),
    ('psigpu_index_sa', _P, [_P]),
    ('psigpu_suffix_array', C.c_int, [_P, C.c_uint64, C.c_uint32, _P]),
    ('psigpu_create', _P, [C.c_int]),
    ('psigpu_destroy', None, [_P]),
    ('psigpu_last_error', C.c_char_p, [_P]),
    ('psigpu_load_graph', C.c_int, [_P, C.POINTER(GraphView)]),
    ('psigpu_load_index', C.c_int, [_P, C.POINTER(IndexView)]),
    ('psigpu_set_gocc_threshold', C.c_int, [_P, C.c_uint32]),
    ('psigpu_set_tuning', C.c_int, [_P, C.c_uint32]),
    ('psigpu_measure_random_loads', C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    ('psigpu_set_query_mode', C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ('psigpu_query_mode', C.c_uint32, [_P]),
    ('psigpu_find_seeds', C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.c_uint32, C.POINTER(Hits)]),
    ('psigpu_free_hits', None, [C.POINTER(Hits)]),
    ('psigpu_find_seeds_packed', C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                           C.c_uint32, C.POINTER(Hits)]),
    ('psigpu_pack_reads', C.c_uint64, [_P, C.c_uint64, C.c_uint64, _P, _P]),
    ('psigpu_count_occurrences', C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.c_uint64]),
    ('psigpu_find_seeds_device_packed', C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                                  C.c_uint64, C.c_uint32, _P, C.POINTER(_P), _U64P]),
    ('psigpu_set_option', C.c_int, [_P, C.c_char_p, C.c_uint64]),
    ('psigpu_find_seeds_device_begin', C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                                 C.c_uint64, C.c_uint32, _P]),
    ('psigpu_find_seeds_device_packed_begin', C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                                        C.c_uint64, C.c_uint32, _P]),
    ('psigpu_find_seeds_device_end', C.c_int, [_P, C.POINTER(_P), _U64P]),
    ('psigpu_verify_resident', C.c_int, [_P, C.POINTER(C.c_uint32), C.c_char_p, C.c_uint64]),
    ('psigpu_find_seeds_device', C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.c_uint64, C.c_uint32, _P, C.POINTER(_P), _U64P]),
    ('psigpu_get_counters', C.c_int, [_P, C.POINTER(Counters)]),
    ('psigpu_find_mems', C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Mems)]),
    ('psigpu_free_mems', None, [C.POINTER(Mems)]),
    ('psigpu_prepare', C.c_int, [_P, C.c_uint32]),
    ('psigpu_copy_hits', C.c_int, [_P, _P, _P, C.c_uint64]),
    ('psigpu_comm_available', C.c_int, []),
    ('psigpu_comm_unique_id', C.c_int, [_P]),
    ('psigpu_comm_create', _P, [C.c_int, _P, C.c_int, C.c_int]),
    ('psigpu_comm_destroy', None, [_P]),
    ('psigpu_comm_last_error', C.c_char_p, [_P]),
    ('psigpu_gather_hits', C.c_int, [_P, _P, C.c_uint64, C.c_int, C.POINTER(_P), _U64P, _P]),
    ('psigpu_host_alloc', _P, [C.c_uint64]),
    ('psigpu_host_free', None, [_P]),
    ('psigpu_copy_pool_stats', None, [C.POINTER(C.c_uint64), C.c_int]),
    ('psigpu_reserve_hit_arrays', C.c_uint32, [C.c_uint64, C.c_uint32]),
]

_lib = None


def lib():
    """Load libpsi_gpu.so (built in-tree by __graft_entry__.build()); raise if absent."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64
        # (same SONAME as /opt/rocm's).  Two HSA runtimes cannot share the GPU inside one
        # process, so when torch is installed it is imported FIRST and libpsi_gpu.so then binds
        # to the runtime torch already loaded.  Set PSI_AMD_NO_TORCH=1 for torch-free processes.
        if os.environ.get('PSI_AMD_NO_TORCH', '0') in ('', '0'):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise PsiGpuError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(there is no CPU fallback)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _host_err() -> str:
    return lib().psigpu_host_last_error().decode()


class PinnedArray:
    """numpy view of page-locked host memory from psigpu_host_alloc: a read chunk kept here is
    DMA'd in place by psigpu_find_seeds (no staging copy).  Keep the object alive while the
    array is in use; the memory is released with it."""

    def __init__(self, n: int, dtype):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.nbytes = max(1, self.n * self.dtype.itemsize)
        self.ptr = lib().psigpu_host_alloc(self.nbytes)
        if not self.ptr:
            raise PsiGpuError('psigpu_host_alloc(%d) failed (no GPU?)' % self.nbytes)
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=self.n)

    def __del__(self):
        if getattr(self, 'ptr', None) and _lib is not None:
            self.array = None
            _lib.psigpu_host_free(self.ptr)
            self.ptr = None


def copy_pool_stats(trim_all: bool = False) -> dict:
    """The process-wide pool of copy-engine transfer buffers (psigpu_copy_pool_stats, ABI 8)."""
    out = (C.c_uint64 * 8)()
    lib().psigpu_copy_pool_stats(out, 1 if trim_all else 0)
    keys = ('allocated', 'reused', 'returned_to_driver', 'queue_drains', 'idle_device_bytes', 'idle_host_bytes', 'idle', 'in_use')
    return dict(zip(keys, (int(v) for v in out)))


def pinned_copy(a: np.ndarray) -> PinnedArray:
    a = np.ascontiguousarray(a)
    p = PinnedArray(a.size, a.dtype)
    p.array[:] = a.ravel()
    return p


def pack_reads(reads: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if len(reads):
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(''.join(reads).encode(), dtype=np.uint8).copy()
    return bases, off


class PackedReads:
    """A chunk of reads as psigpu_find_seeds_packed takes it: 2-bit codes (32 bases per u64 word, first base most
    significant), a "not ACGT" bit per base (None when every base is ACGT) and the read offsets in BASES.  `pinned`:
    the arrays live in page-locked memory (DMA'd in place)."""

    def __init__(self, bases: np.ndarray, off: np.ndarray, pinned: bool = False, threads: int = 1):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = int(off[-1]) if len(off) else 0
        nw, nm = (n + 31) // 32 + 2, (n + 63) // 64 + 2          # (+ the words the seeding kernel loads behind the last seed)
        if pinned:
            self._pw, self._pm, self._po = PinnedArray(nw, np.uint64), PinnedArray(nm, np.uint64), pinned_copy(off)
            self.words, mask, self.off = self._pw.array, self._pm.array, self._po.array
            self.words[:] = 0
            mask[:] = 0
        else:
            self.words, mask, self.off = np.zeros(nw, np.uint64), np.zeros(nm, np.uint64), off
        self.n_bases = n
        self.n_not_acgt = 0
        if n:
            # (pieces of whole 64-base blocks may be packed by different threads at once)
            piece = max(64, ((n + threads - 1) // max(1, threads) + 63) // 64 * 64)
            cuts = list(range(0, n, piece)) + [n]
            if threads > 1 and len(cuts) > 2:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(threads) as ex:
                    bad = list(ex.map(lambda ab: lib().psigpu_pack_reads(bases[ab[0]:].ctypes.data_as(C.c_void_p), ab[0], ab[1] - ab[0],
                                                                          _ptr(self.words), _ptr(mask)), zip(cuts[:-1], cuts[1:])))
                self.n_not_acgt = int(sum(bad))
            else:
                self.n_not_acgt = int(lib().psigpu_pack_reads(_ptr(bases), 0, n, _ptr(self.words), _ptr(mask)))
        self.mask = mask if self.n_not_acgt else None
        self._mask_store = mask


class Graph:
    """Stand-in for the gum::SeqGraph the reference loads (src/psikt.cpp:249-251)."""

    def __init__(self, handle):
        self.h = handle
        v = GraphView()
        lib().psigpu_graph_view_get(self.h, C.byref(v))
        self.view = v
        self.n_nodes = v.n_nodes
        self.n_edges = lib().psigpu_graph_edge_count(self.h)

    @classmethod
    def load(cls, path: str, follow_reversing: bool = False) -> 'Graph':
        """`follow_reversing`: PSIGPU_GRAPH_FOLLOW_REVERSING -- reversing links and reverse path steps are walked as the
        reference walks them (the link's `to` node, read forwards) instead of being refused."""
        st = C.c_int(0)
        h = lib().psigpu_graph_load_opts(path.encode(), 1 if follow_reversing else 0, C.byref(st))
        if not h:
            raise PsiGpuError('cannot load graph %s: %s' % (path, _host_err()))
        return cls(h)

    @classmethod
    def from_csr(cls, node_id, label_off, labels, edge_off, edge_to,
                 paths: Sequence[Sequence[int]] = ()) -> 'Graph':
        node_id = np.ascontiguousarray(node_id, dtype=np.uint64)
        label_off = np.ascontiguousarray(label_off, dtype=np.uint64)
        edge_off = np.ascontiguousarray(edge_off, dtype=np.uint64)
        edge_to = np.ascontiguousarray(edge_to, dtype=np.uint32)
        if isinstance(labels, (bytes, bytearray)):
            labels = np.frombuffer(bytes(labels), dtype=np.uint8)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        poff = np.zeros(len(paths) + 1, dtype=np.uint64)
        if len(paths):
            poff[1:] = np.cumsum([len(p) for p in paths], dtype=np.uint64)
            pnodes = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32) for p in paths]))
        else:
            pnodes = np.zeros(0, np.uint32)
        st = C.c_int(0)
        h = lib().psigpu_graph_from_csr(len(node_id), _ptr(node_id), _ptr(label_off), _ptr(labels),
                                        _ptr(edge_off), _ptr(edge_to), len(paths), _ptr(poff),
                                        _ptr(pnodes), C.byref(st))
        if not h:
            raise PsiGpuError('bad graph: ' + _host_err())
        return cls(h)

    def _arr(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype)
        buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)

    @property
    def node_id(self):
        return self._arr(self.view.node_id, self.n_nodes, np.uint64)

    @property
    def label_off(self):
        return self._arr(self.view.label_off, self.n_nodes + 1, np.uint64)

    @property
    def labels(self):
        return self._arr(self.view.labels, int(self.label_off[-1]) if self.n_nodes else 0, np.uint8)

    @property
    def edge_off(self):
        return self._arr(self.view.edge_off, self.n_nodes + 1, np.uint64)

    @property
    def edge_to(self):
        return self._arr(self.view.edge_to, self.n_edges, np.uint32)

    def paths(self) -> List[np.ndarray]:
        out = []
        for i in range(lib().psigpu_graph_path_count(self.h)):
            n = lib().psigpu_graph_path(self.h, i, None, 0)
            a = np.zeros(n, np.uint32)
            lib().psigpu_graph_path(self.h, i, _ptr(a), n)
            out.append(a)
        return out

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.psigpu_graph_free(self.h)
            self.h = None


class PathIndex:
    """Stand-in for psi::PathIndex<..., Reversed> plus the finder's starting loci
    (reference include/psi/pathindex.hpp:40-333; seed_finder.hpp:1747-1752)."""

    def __init__(self, handle):
        self.h = handle
        v = IndexView()
        lib().psigpu_index_view_get(self.h, C.byref(v))
        self.view = v

    @classmethod
    def build(cls, g: Graph, k: int, n_paths: int, step: int = 1, sa_rate: int = 0,
              rng_seed: int = 0, ftab_len: int = 0, keep: bool = False, device: Optional[int] = None,
              patched: bool = False, context: int = 0, max_part_text: int = 0) -> 'PathIndex':
        """`device`: GPU ordinal to build the suffix array / FM arrays on (None = host SA-IS);
        `patched` / `context`: psikt's default indexing mode (no -P) and its -t; `max_part_text`: text
        symbols per index part (0 = the 32-bit row limit; tests set it small to get several parts)."""
        st = C.c_int(0)
        opts = IndexOpts(k, n_paths, step, sa_rate, ftab_len, int(keep), rng_seed,
                         0 if device is None else device + 1, context, int(patched), 0, max_part_text)
        h = lib().psigpu_index_build(g.h, C.byref(opts), C.byref(st))
        if not h:
            raise PsiGpuError('index build failed (%d): %s' % (st.value, _host_err()))
        return cls(h)

    @classmethod
    def build_paths(cls, g: Graph, k: int, paths: Sequence[Sequence[int]], step: int = 1,
                    sa_rate: int = 0, keep: bool = False, ftab_len: int = 0,
                    device: Optional[int] = None, head: Optional[Sequence[int]] = None,
                    tail: Optional[Sequence[int]] = None, context: int = 0, max_part_text: int = 0) -> 'PathIndex':
        """`head` / `tail`: per path, the offset of its first indexed base in its first node and the
        number of indexed bases of its last node (0 = all): a patch (Path::left / right)."""
        poff = np.zeros(len(paths) + 1, dtype=np.uint64)
        if len(paths):
            poff[1:] = np.cumsum([len(p) for p in paths], dtype=np.uint64)
            pnodes = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32) for p in paths]))
        else:
            pnodes = np.zeros(0, np.uint32)
        st = C.c_int(0)
        opts = IndexOpts(k, 0, step, sa_rate, ftab_len, int(keep), 0, 0 if device is None else device + 1, context, 0, 0,
                         max_part_text)
        hd = None if head is None else np.ascontiguousarray(head, dtype=np.uint32)
        tl = None if tail is None else np.ascontiguousarray(tail, dtype=np.uint32)
        h = lib().psigpu_index_build_patches(g.h, C.byref(opts), len(paths), _ptr(poff), _ptr(pnodes),
                                             _ptr(hd), _ptr(tl), C.byref(st))
        if not h:
            raise PsiGpuError('index build failed (%d): %s' % (st.value, _host_err()))
        return cls(h)

    @classmethod
    def from_reference_paths(cls, g: Graph, k: int, paths_file: str, step: int = 1, sa_rate: int = 0, ftab_len: int = 0,
                             device: Optional[int] = None) -> 'PathIndex':
        """An index over the paths of a `<prefix>_paths` file written by the reference's
        PathIndex::save_paths_set (enc_vector node lists, trims, node-break bit vectors)."""
        st = C.c_int(0)
        ctx, fwd = C.c_uint64(0), C.c_uint32(0)
        opts = IndexOpts(k, 0, step, sa_rate, ftab_len, 0, 0, 0 if device is None else device + 1, 0, 0, 0, 0)
        h = lib().psigpu_index_from_reference_paths(g.h, C.byref(opts), paths_file.encode(), C.byref(ctx), C.byref(fwd), C.byref(st))
        if not h:
            raise PsiGpuError('cannot read %s (%d): %s' % (paths_file, st.value, _host_err()))
        px = cls(h)
        px.ref_context, px.ref_forward = ctx.value, bool(fwd.value)
        return px

    @classmethod
    def load(cls, prefix: str) -> 'PathIndex':
        st = C.c_int(0)
        h = lib().psigpu_index_load(prefix.encode(), C.byref(st))
        if not h:
            raise PsiGpuError('cannot load index %s (%d)' % (prefix, st.value))
        return cls(h)

    def save(self, prefix: str) -> None:
        st = lib().psigpu_index_save(self.h, prefix.encode())
        if st:
            raise PsiGpuError('cannot save index to %s (%d)' % (prefix, st))

    def _arr(self, ptr, n, dtype):
        if n == 0 or not ptr:
            return np.zeros(0, dtype)
        buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)

    @property
    def loci(self) -> Tuple[np.ndarray, np.ndarray]:
        v = self.view
        return (self._arr(v.loci_node, v.n_loci, np.uint32).copy(),
                self._arr(v.loci_off, v.n_loci, np.uint32).copy())

    @property
    def text_len(self) -> int:
        """Indexed symbols over all parts."""
        return self.view.text_len + sum(v.text_len for v in self.more_parts())

    def more_parts(self) -> List['IndexView']:
        n = self.view.n_more_parts
        if not n:
            return []
        return list((IndexView * n).from_address(self.view.more_parts))

    def text(self) -> np.ndarray:
        """Indexed text (needs keep=True): 0 sentinel, 1 separator, 2..5 = ACGT."""
        p = lib().psigpu_index_text(self.h)
        return self._arr(p, self.view.text_len, np.uint8)

    def sa(self) -> np.ndarray:
        p = lib().psigpu_index_sa(self.h)
        return self._arr(p, self.view.text_len, np.int32)

    def paths(self) -> List[np.ndarray]:
        out = []
        for i in range(lib().psigpu_index_path_count(self.h)):
            n = lib().psigpu_index_path(self.h, i, None, 0)
            a = np.zeros(n, np.uint32)
            lib().psigpu_index_path(self.h, i, _ptr(a), n)
            out.append(a)
        return out

    def trims(self) -> List[Tuple[int, int]]:
        """(head offset, tail length) of every indexed path; (0, 0) = a full path."""
        out = []
        for i in range(lib().psigpu_index_path_count(self.h)):
            a, b = C.c_uint32(), C.c_uint32()
            lib().psigpu_index_path_trim(self.h, i, C.byref(a), C.byref(b))
            out.append((a.value, b.value))
        return out

    def save_loci(self, g: Graph, prefix: str) -> None:
        """`<prefix>_loci_e<E>l<K>` in the reference's format (SeedFinder::save_starts)."""
        if lib().psigpu_loci_save(self.h, g.h, prefix.encode()):
            raise PsiGpuError('cannot write the loci file')

    def load_loci(self, g: Graph, prefix: str, step: int = 1) -> None:
        """Replace the starting loci by those of `<prefix>_loci_e<step>l<K>` (SeedFinder::open_starts)."""
        st = lib().psigpu_loci_load(self.h, g.h, prefix.encode(), step)
        if st:
            raise PsiGpuError('cannot read the loci file (%d): %s' % (st, _host_err()))
        lib().psigpu_index_view_get(self.h, C.byref(self.view))

    def set_locus_step(self, g: Graph, step: int) -> None:
        """Starting loci recomputed for another locus step (psikt -e) from the index's own paths and trims."""
        st = lib().psigpu_index_set_locus_step(self.h, g.h, step)
        if st:
            raise PsiGpuError('cannot recompute the starting loci: ' + _host_err())
        lib().psigpu_index_view_get(self.h, C.byref(self.view))

    @property
    def locus_step(self) -> int:
        return lib().psigpu_index_locus_step(self.h)

    def matches(self, g: Graph, k: int, step: int = 1) -> bool:
        return bool(lib().psigpu_index_matches(self.h, g.h, k, step))

    def __del__(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.psigpu_index_free(self.h)
            self.h = None


def suffix_array(text: np.ndarray, sigma: int) -> np.ndarray:
    text = np.ascontiguousarray(text, dtype=np.uint8)
    sa = np.zeros(len(text), np.int32)
    st = lib().psigpu_suffix_array(_ptr(text), len(text), sigma, _ptr(sa))
    if st:
        raise PsiGpuError('suffix_array: bad input (%d)' % st)
    return sa


class SeedFinder:
    """Mirror of psi::SeedFinder's query surface (reference include/psi/seed_finder.hpp):
    ctor(graph, seed_len, gocc_threshold) :930; create_path_index :1330; load_path_index :1396;
    serialize_path_index :1372; seeds_on_paths :1426; seeds_off_paths :1703; seeds_all :1724;
    get_starting_loci.  Hits come back as an (n, 4) uint64 array
    (node_id, node_offset, read_id, read_offset) -- the 32-byte records psikt writes
    (src/psikt.cpp:172-181) -- instead of through a per-hit callback."""

    def __init__(self, graph: Graph, seed_len: int, gocc_threshold: int = 0, device: int = 0,
                 mode: Optional[str] = None, walk_cap: Optional[int] = None):
        """`mode`: 'kmer-table' (default: path k-mers and the starting loci's k-walks tabulated once
        in HBM, one probe per seed), 'locus-table' (FM index on the paths, table for the loci) or
        'traverse' (every starting locus traversed per chunk, as the reference does; paths by a table of their
        k-mers, or by the FM index: TUNE_NO_PATH_TABLE);
        `walk_cap`: loci with more k-walks than this stay with the traverser (0 = 256)."""
        if mode is None:
            mode = os.environ.get('PSI_AMD_MODE', 'kmer-table')
        if walk_cap is None:
            walk_cap = int(os.environ.get('PSI_AMD_WALK_CAP', '0'))
        if mode not in _MODES:
            raise PsiGpuError("mode must be one of " + ', '.join(sorted(_MODES)))
        if not 1 <= seed_len <= MAX_SEED_LEN:
            raise PsiGpuError('seed length out of range (1..%d)' % MAX_SEED_LEN)
        self.graph = graph
        self.seed_len = seed_len
        self.device = device
        self.pindex: Optional[PathIndex] = None
        self.ctx = lib().psigpu_create(device)
        if not self.ctx:
            raise PsiGpuError('psigpu_create: ' + lib().psigpu_last_error(None).decode())
        self._chk(lib().psigpu_load_graph(self.ctx, C.byref(graph.view)))
        if gocc_threshold:
            self._chk(lib().psigpu_set_gocc_threshold(self.ctx, gocc_threshold))
        self.set_query_mode(mode, walk_cap)
        if os.environ.get('PSI_AMD_TUNE'):                   # (tests: a measurement switch on every finder)
            self.set_tuning(int(os.environ['PSI_AMD_TUNE']))

    def set_query_mode(self, mode: str, walk_cap: int = 0) -> None:
        self._chk(lib().psigpu_set_query_mode(self.ctx, _MODES[mode], walk_cap))

    def query_mode(self) -> str:
        """The mode queries run in ('auto' until an AUTO finder has decided)."""
        m = lib().psigpu_query_mode(self.ctx)
        return next(n for n, v in _MODES.items() if v == m)

    def _chk(self, st: int) -> None:
        if st:
            raise PsiGpuError('psigpu error %d: %s' % (st, lib().psigpu_last_error(self.ctx).decode()))

    def set_tuning(self, flags: int) -> None:
        """Measurement switches (TUNE_*): which kernel answers the on-path phase of the FM modes."""
        self._chk(lib().psigpu_set_tuning(self.ctx, flags))

    def measure_random_loads(self, table_bytes: int, n_loads: int, quad_sectors: bool = False) -> float:
        """Independent random loads per second this device retires right now (16 bytes per lane, or one 64-byte
        sector per quad) on a scratch table of `table_bytes`: the bound of the table probe / of an LF step."""
        r = C.c_double()
        self._chk(lib().psigpu_measure_random_loads(self.ctx, table_bytes, n_loads, int(quad_sectors), C.byref(r)))
        return r.value

    def set_gocc_threshold(self, thr: int) -> None:
        """SeedFinder gocc_threshold (seed_finder.hpp:939): on-path k-mers with more than `thr` occurrences
        in the path text are skipped; 0 = unlimited.  May be changed at any time (a k-mer table built
        without a threshold is rebuilt by the next query)."""
        self._chk(lib().psigpu_set_gocc_threshold(self.ctx, thr))

    # -- index ------------------------------------------------------------------------
    def create_path_index(self, n: int, patched: bool = False, context: int = 0, step_size: int = 1,
                          sa_rate: int = 0, rng_seed: int = 0, ftab_len: int = 0,
                          build_on_device: bool = False) -> None:
        self.set_path_index(PathIndex.build(self.graph, self.seed_len, n, step_size, sa_rate, rng_seed,
                                            ftab_len, device=self.device if build_on_device else None,
                                            patched=patched, context=context))

    def set_path_index(self, pindex: PathIndex) -> None:
        self.pindex = pindex
        self._chk(lib().psigpu_load_index(self.ctx, C.byref(pindex.view)))

    def prepare(self) -> None:
        """Build the query mode's tables now (index load time) instead of inside the first query."""
        self._chk(lib().psigpu_prepare(self.ctx, self.seed_len))

    def load_path_index(self, prefix: str, step_size: int = 1) -> bool:
        """False (the caller then builds a new index) when the file is missing, unreadable, or was made
        for another graph / seed length / locus step."""
        try:
            px = PathIndex.load(prefix)
        except PsiGpuError:
            return False
        if not px.matches(self.graph, self.seed_len, step_size):
            # same graph and seed length, another locus step: the paths are good, the loci are recomputed
            # from them (never taken from a `_loci_e<E>l<K>` file found beside the index: it names no graph)
            if not px.matches(self.graph, self.seed_len, px.locus_step):
                return False
            px.set_locus_step(self.graph, step_size)
        self.set_path_index(px)
        return True

    def serialize_path_index(self, prefix: str) -> bool:
        if self.pindex is None:
            return False
        self.pindex.save(prefix)
        return True

    def get_starting_loci(self):
        return self.pindex.loci

    # -- queries ----------------------------------------------------------------------
    def _find(self, reads, step, rec_offset, flags) -> np.ndarray:
        if isinstance(reads, tuple):
            bases, off = reads
        else:
            bases, off = pack_reads(reads)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = Hits()
        self._chk(lib().psigpu_find_seeds(self.ctx, _ptr(bases), _ptr(off), len(off) - 1,
                                          self.seed_len, step, rec_offset, flags, C.byref(out)))
        if out.n:
            buf = (C.c_uint64 * (4 * out.n)).from_address(C.addressof(out.data.contents))
            arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
        else:
            arr = np.zeros((0, 4), np.uint64)
        lib().psigpu_free_hits(C.byref(out))
        return arr

    def seeds_all(self, reads, step: int = 0, rec_offset: int = 0, sort_unique: bool = False, uniform: bool = False):
        """`uniform`: PSIGPU_UNIFORM_READS -- the caller says all reads have one length (checked on the device)."""
        return self._find(reads, step, rec_offset, ALL | (SORT_UNIQUE if sort_unique else 0) | (UNIFORM_READS if uniform else 0))

    def seeds_all_packed(self, pr: 'PackedReads', step: int = 0, rec_offset: int = 0, sort_unique: bool = False,
                         flags: int = ALL, read_range: Optional[Tuple[int, int]] = None) -> np.ndarray:
        """psigpu_find_seeds_packed: the chunk's reads as 2-bit words (a quarter of the ASCII bytes on the host link).
        `read_range` (begin, end): only those reads of the chunk -- the word arrays as they are, the range's own offsets
        (read ids count from rec_offset at `begin`)."""
        out = Hits()
        off = pr.off if read_range is None else pr.off[read_range[0]:read_range[1] + 1]
        self._chk(lib().psigpu_find_seeds_packed(self.ctx, _ptr(pr.words), _ptr(pr.mask), _ptr(off), len(off) - 1,
                                                 self.seed_len, step, rec_offset, flags | (SORT_UNIQUE if sort_unique else 0),
                                                 C.byref(out)))
        if out.n:
            buf = (C.c_uint64 * (4 * out.n)).from_address(C.addressof(out.data.contents))
            arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
        else:
            arr = np.zeros((0, 4), np.uint64)
        lib().psigpu_free_hits(C.byref(out))
        return arr

    def set_option(self, name: str, value: int) -> None:
        """psigpu_set_option: per-context switches of the host entry ('sub_bytes', 'no_ahead', 'no_engine_copy', 'wire')."""
        self._chk(lib().psigpu_set_option(self.ctx, name.encode(), int(value)))

    def seeds_on_paths(self, reads, step: int = 0, rec_offset: int = 0):
        return self._find(reads, step, rec_offset, ON_PATHS)

    def seeds_off_paths(self, reads, step: int = 0, rec_offset: int = 0):
        return self._find(reads, step, rec_offset, OFF_PATHS)

    def count_occurrences(self, reads, step: int = 0) -> np.ndarray:
        """Seed::gocc of the on-path phase (psigpu_count_occurrences): per seed of the chunk -- reads in order, a read's seeds
        in offset order -- the number of occurrences of its k-mer in the indexed path text (index_iter.hpp:842-843)."""
        if isinstance(reads, tuple):
            bases, off = reads
        else:
            bases, off = pack_reads(reads)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        k, st = self.seed_len, step or self.seed_len
        lens = (off[1:] - off[:-1]).astype(np.int64)
        n = int(np.where(lens >= k, (lens - k) // st + 1, 0).sum())
        out = np.zeros(n, np.uint32)
        self._chk(lib().psigpu_count_occurrences(self.ctx, _ptr(bases), _ptr(off), len(off) - 1, k, st, _ptr(out), n))
        return out

    def find_mems(self, reads, max_mem: int = 0, rec_offset: int = 0) -> np.ndarray:
        """SeedFinder::seeds_on_paths( sequence, callback ) for every read (reference
        seed_finder.hpp:1459-1479 -> find_mems, index_iter.hpp:854-906), minimum length = seed length.
        Returns an (n, 6) uint64 array: node_id, node_offset, read_id, read_offset, match_len, gocc."""
        if isinstance(reads, tuple):
            bases, off = reads
        else:
            bases, off = pack_reads(reads)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = Mems()
        self._chk(lib().psigpu_find_mems(self.ctx, _ptr(bases), _ptr(off), len(off) - 1, self.seed_len, max_mem,
                                         rec_offset, C.byref(out)))
        if out.n:
            buf = (C.c_uint64 * (6 * out.n)).from_address(C.addressof(out.data.contents))
            arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).copy()
        else:
            arr = np.zeros((0, 6), np.uint64)
        lib().psigpu_free_mems(C.byref(out))
        return arr

    def seeds_all_device(self, d_bases_ptr: int, d_read_off_ptr: int, n_reads: int, n_bases: int,
                         step: int = 0, rec_offset: int = 0, flags: int = ALL, stream: int = 0):
        """Device-resident chunk in, device-resident hits out: returns (device pointer, n_hits).
        flags may include SORT_UNIQUE (sorted on the device)."""
        d_hits = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().psigpu_find_seeds_device(self.ctx, d_bases_ptr, d_read_off_ptr, n_reads, n_bases,
                                                 self.seed_len, step, rec_offset, flags, stream,
                                                 C.byref(d_hits), C.byref(n)))
        return d_hits.value, n.value

    def seeds_all_device_begin(self, d_bases_ptr: int, d_read_off_ptr: int, n_reads: int, n_bases: int,
                               step: int = 0, rec_offset: int = 0, flags: int = ALL, stream: int = 0) -> None:
        """Queue a device-resident chunk and return (psigpu_find_seeds_device_begin): at most two chunks begun and not
        ended; seeds_all_device_end() hands out the hits of the oldest."""
        self._chk(lib().psigpu_find_seeds_device_begin(self.ctx, d_bases_ptr, d_read_off_ptr, n_reads, n_bases,
                                                       self.seed_len, step, rec_offset, flags, stream))

    def seeds_all_device_packed_begin(self, d_words_ptr: int, d_mask_ptr: int, d_read_off_ptr: int, n_reads: int, n_bases: int,
                                      step: int = 0, rec_offset: int = 0, flags: int = ALL, stream: int = 0) -> None:
        self._chk(lib().psigpu_find_seeds_device_packed_begin(self.ctx, d_words_ptr, d_mask_ptr, d_read_off_ptr, n_reads, n_bases,
                                                              self.seed_len, step, rec_offset, flags, stream))

    def seeds_all_device_end(self):
        """(device pointer, n_hits) of the oldest chunk begun; valid until the next end() on this finder."""
        d_hits = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().psigpu_find_seeds_device_end(self.ctx, C.byref(d_hits), C.byref(n)))
        return d_hits.value, n.value

    def verify_resident(self) -> str:
        """'' when every array the loaders put on the device still has the content it was loaded with, else the names of
        those that changed (psigpu_verify_resident)."""
        n = C.c_uint32()
        buf = C.create_string_buffer(4096)
        self._chk(lib().psigpu_verify_resident(self.ctx, C.byref(n), buf, 4096))
        return buf.value.decode() if n.value else ''

    def seeds_all_device_packed(self, d_words_ptr: int, d_mask_ptr: int, d_read_off_ptr: int, n_reads: int, n_bases: int,
                                step: int = 0, rec_offset: int = 0, flags: int = ALL, stream: int = 0):
        """Device-resident PACKED chunk in (psigpu_find_seeds_device_packed), device-resident hits out."""
        d_hits = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().psigpu_find_seeds_device_packed(self.ctx, d_words_ptr, d_mask_ptr, d_read_off_ptr, n_reads, n_bases,
                                                        self.seed_len, step, rec_offset, flags, stream,
                                                        C.byref(d_hits), C.byref(n)))
        return d_hits.value, n.value

    def copy_hits(self, d_ptr: int, n: int) -> np.ndarray:
        """Device-resident hits (pointer from seeds_all_device) -> (n, 4) uint64 array."""
        out = np.zeros((n, 4), np.uint64)
        self._chk(lib().psigpu_copy_hits(self.ctx, _ptr(out), d_ptr, n))
        return out

    def counters(self) -> dict:
        c = Counters()
        lib().psigpu_get_counters(self.ctx, C.byref(c))
        return c.as_dict()

    def counters_into(self, c: 'Counters') -> 'Counters':
        """Same, into a caller-owned struct (no dict: a timing loop pays microseconds per call)."""
        lib().psigpu_get_counters(self.ctx, C.byref(c))
        return c

    def close(self):
        if getattr(self, 'ctx', None) and _lib is not None:
            _lib.psigpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()


class HitGather:
    """The gather of hit lists over RCCL / xGMI in C++ (psigpu_comm_*, psigpu_gather_hits): one per rank.  `id_bytes`:
    the 128 bytes of HitGather.unique_id() made on one rank and handed to all (e.g. torch.distributed.broadcast)."""

    @staticmethod
    def available() -> bool:
        return bool(lib().psigpu_comm_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        if lib().psigpu_comm_unique_id(buf):
            raise PsiGpuError('psigpu_comm_unique_id: ' + lib().psigpu_comm_last_error(None).decode())
        return bytes(buf)

    def __init__(self, device: int, id_bytes: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        self.rank, self.world = rank, world
        self.h = lib().psigpu_comm_create(device, buf, rank, world)
        if not self.h:
            raise PsiGpuError('psigpu_comm_create: ' + lib().psigpu_comm_last_error(None).decode())

    def gather(self, d_ptr: int, n: int, root: int = 0):
        """-> (device pointer of all records on the root | 0, total on the root | 0, every rank's count)"""
        d_all, n_all = C.c_void_p(), C.c_uint64()
        counts = np.zeros(self.world, np.uint64)
        if lib().psigpu_gather_hits(self.h, d_ptr, n, root, C.byref(d_all), C.byref(n_all), _ptr(counts)):
            raise PsiGpuError('psigpu_gather_hits: ' + lib().psigpu_comm_last_error(self.h).decode())
        return d_all.value or 0, n_all.value, counts

    def close(self):
        if getattr(self, 'h', None) and _lib is not None:
            _lib.psigpu_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class DeviceHits:
    """Zero-copy view of the records psigpu_find_seeds_device left in HBM -- (pointer, n) from
    SeedFinder.seeds_all_device -- for frameworks that speak the CUDA array interface:
    ``torch.as_tensor(DeviceHits(ptr, n), device='cuda')`` is an (n, 4) int64 tensor over the library's buffer
    (valid until the next call on the finder).  What a gather over RCCL sends, without a trip through the host."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {'shape': (int(n), 4), 'typestr': '<i8', 'data': (int(ptr), False), 'version': 2,
                                         'strides': None}


def sort_unique(hits: np.ndarray) -> np.ndarray:
    if len(hits) == 0:
        return hits.reshape(0, 4)
    return np.unique(hits, axis=0)
