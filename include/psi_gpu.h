/*
 * psi_gpu.h -- C ABI of the MI355X-native seed finder (libpsi_gpu.so).
 *
 * The reference (cartoonist/psi) has no FFI: it is header-only C++ templates called
 * in-process.  The seam this ABI cuts is the query surface of psi::SeedFinder as driven by
 * psikt's chunk loop (reference src/psikt.cpp:183-209):
 *
 *     finder.get_seeds(seeds, chunk, distance);          // include/psi/seed_finder.hpp:1099-1109
 *     seeds_index = finder.index_reads(seeds);           // :1089-1097
 *     finder.seeds_all(seeds, seeds_index, traverser, write_callback);   // :1724-1732
 *
 * Everything on the device side of that seam is in this library; INTEGRATION.md shows the
 * reference-side binding.  Plain pointers and sizes only; no C++ or torch types; no
 * exception crosses the boundary (int status, 0 = ok, psigpu_last_error() for the text).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference tree).
 */
#ifndef PSI_GPU_H
#define PSI_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSIGPU_ABI_VERSION 8
#define PSIGPU_MAX_SEED_LEN 63u   /* psikt takes any -l (src/psikt.cpp:327); seeds are 2-bit packed into one 64-bit word up
                                    to 31 bases and into two words from 32 to 63 */
#define PSIGPU_MAX_TABLE_SEED_LEN 31u   /* the tabulating query modes (k-mer table, locus table) hold one-word k-mers: longer
                                          seeds are answered by the FM index and the query-time traverser in every mode */
#define PSIGPU_MAX_PARTS 8u       /* parts of an index whose text passes the 32-bit row limit */

/* Status codes */
enum {
  PSIGPU_OK = 0,
  PSIGPU_ERR_ARG = 1,       /* invalid argument (also: seed length out of range) */
  PSIGPU_ERR_STATE = 2,     /* graph / index not loaded */
  PSIGPU_ERR_DEVICE = 3,    /* HIP runtime error, or no GPU */
  PSIGPU_ERR_IO = 4,
  PSIGPU_ERR_NOMEM = 5,
  PSIGPU_ERR_FORMAT = 6,    /* unsupported input (e.g. reversing edges) */
  PSIGPU_ERR_CONTEXT = 7    /* "seed length should not be larger than context size"
                               (include/psi/seed_finder.hpp:1434-1437) */
};

/* ------------------------------------------------------------------------------------
 * psi::Seed<> as psikt writes it: 4 x native-endian u64 = 32 bytes per hit
 * (include/psi/seed.hpp:32-46; src/psikt.cpp:172-181).  match_len == k and gocc are not
 * part of the on-wire record.
 * ---------------------------------------------------------------------------------- */
typedef struct psigpu_hit {
  uint64_t node_id;       /* external (file) node id of the first base of the occurrence */
  uint64_t node_offset;   /* offset of that base in the node label */
  uint64_t read_id;       /* rec_offset + index of the read in the batch (sequence.hpp:1277-1282) */
  uint64_t read_offset;   /* offset of the seed in the read (sequence.hpp:1207-1213) */
} psigpu_hit;

typedef struct psigpu_hits {
  uint64_t n;
  psigpu_hit* data;       /* library-owned pinned host memory; release with psigpu_free_hits */
} psigpu_hits;

/* ------------------------------------------------------------------------------------
 * Views: what the device needs, as plain arrays.  Host-owned, copied by psigpu_load_*.
 * They can be filled by any host code; psigpu_graph_* / psigpu_index_* below are this
 * library's own host-side builders.
 * ---------------------------------------------------------------------------------- */

/* Stand-in for the gum::SeqGraph accessors used on the path -- node_sequence,
 * node_length, has_edges_out, for_each_edges_out (include/psi/traverser_bfs.hpp:119,141,146).
 * Nodes are addressed by rank 0..n_nodes-1 (file order); forward strand only, as the
 * reference traverser ignores link orientation (traverser_bfs.hpp:146-160). */
typedef struct psigpu_graph_view {
  uint64_t n_nodes;
  const uint64_t* node_id;      /* [n_nodes] external ids */
  const uint64_t* label_off;    /* [n_nodes+1] offsets into labels */
  const char* labels;           /* ASCII bases, anything outside ACGT is N */
  const uint64_t* edge_off;     /* [n_nodes+1] */
  const uint32_t* edge_to;      /* [n_edges] target ranks, in for_each_edges_out order */
} psigpu_graph_view;

/* Stand-in for psi::PathIndex<Graph, DiskString, FMIndex<>, Reversed> + the finder's
 * starting loci (include/psi/pathindex.hpp:40-333; seed_finder.hpp:1747-1752).
 *
 * FM-index layout (own design): the text is the FORWARD concatenation of
 * the indexed path sequences, every maximal run of non-ACGT bases and every path end
 * collapsed to one separator, plus a final sentinel.  BWT rank blocks are 64 bytes:
 *   u32 cntA, cntC, cntG, u32 (exceptions_before << 8 | exceptions_in_block), then
 *   192 symbols as three 16-byte groups of 64 (u64 low-bit plane, u64 high-bit plane).
 *   cntT is derived.  Separators / the sentinel are stored as 'A' and
 *   listed in exc_row (sorted BWT rows) with their suffix-array values in exc_sa.
 *   exceptions_before counts from the start of the block's SUPER-BLOCK of 2^exc_shift blocks; the
 *   exceptions in front of every super-block are in exc_super -- any number of separators fits
 *   (the patches of a whole genome are tens of millions). */
typedef struct psigpu_index_view {
  uint32_t seed_len;            /* k the starting loci were computed for */
  uint32_t sa_rate;             /* SA-order sampling: SA[i] kept for i % sa_rate == 0 */
  uint32_t context;             /* 0 = full paths (psikt -P) */
  uint32_t n_paths;
  uint64_t text_len;            /* n, including separators and the sentinel; < 2^32 */
  uint64_t n_blocks;            /* ceil(n / 192) + 1 */
  const void* bwt_blocks;       /* [n_blocks] x 64 B */
  uint64_t C[4];                /* C[c] = #symbols smaller than c in the text */
  uint64_t n_samples;
  const uint32_t* sa_samples;   /* [n_samples] */
  uint64_t n_exc;
  const uint32_t* exc_row;      /* [n_exc] sorted */
  const uint32_t* exc_sa;       /* [n_exc] SA value of that row */
  /* interval table for the LAST ftab_len bases of a seed (the first ftab_len backward-search
   * steps collapsed into one lookup): entry c = SA interval [lo, hi) of the ftab_len-mer with
   * 2-bit code c (first base most significant); 4^ftab_len entries of 2 x u32; 0 = none */
  uint32_t ftab_len;
  uint32_t exc_shift;           /* log2 of the rank blocks per exception super-block (16) */
  const uint32_t* ftab;
  /* the indexed text itself, 4 bits per symbol (bits 0..1 base, bit 2 separator / sentinel),
   * 16 symbols per u64, first symbol in the top nibble: lets small SA intervals be finished by
   * comparing the remaining seed bases with the text instead of more LF steps */
  const uint64_t* text4;      /* [text_len / 16 + 2] */
  /* text position -> (node, offset): sorted segments + a directory every 64 positions */
  uint64_t n_segs;
  const uint32_t* seg_start;    /* [n_segs+1] text start of each segment (last = n) */
  const uint32_t* seg_node;     /* [n_segs] node rank, 0xFFFFFFFF for separator segments */
  const uint32_t* seg_noff;     /* [n_segs] node offset of the segment's first base */
  uint64_t n_dir;
  const uint32_t* seg_dir;      /* [n_dir] index of the segment containing position 64*i */
  /* starting loci, sorted by node rank then offset (seed_finder.hpp:1481-1585) */
  uint64_t n_loci;
  const uint32_t* loci_node;
  const uint32_t* loci_off;
  /* further PARTS of the index: rows, text positions and table indexes are 32 bits wide, so paths
   * whose concatenation would pass 2^32 symbols are indexed in groups, each a complete FM index of its
   * own -- text, rank blocks, exceptions, interval table, suffix array, segment table (the fields from
   * text_len to seg_dir and exc_super; seed_len, n_paths, the loci and this list are read from the first
   * part only).  The FM modes and MEM mode search every part (a pattern never spans two paths: the parts'
   * occurrences are disjoint, a k-mer's occurrence count is their sum); the k-mer table tabulates the
   * k-mers of all parts together at psigpu_prepare. */
  uint32_t n_more_parts;
  uint32_t reserved2;
  const struct psigpu_index_view* more_parts;     /* [n_more_parts] */
  const uint32_t* exc_super;    /* [((n_blocks - 1) >> exc_shift) + 1] exceptions in front of every super-block */
} psigpu_index_view;

/* ------------------------------------------------------------------------------------
 * Host side: graph loading and index construction (no GPU needed).
 * ---------------------------------------------------------------------------------- */
typedef struct psigpu_graph psigpu_graph;
typedef struct psigpu_index psigpu_index;

/* gum::util::load(graph, path, ...) as psikt calls it (src/psikt.cpp:249-251): .gfa
 * (GFA 1 / GFA 2) or .vg (gzip'd protobuf stream, vg/vg.proto:13-103, vg/stream.hpp:81-130).
 * Returns NULL on failure; *status gets the code. */
psigpu_graph* psigpu_graph_load(const char* path, int* status);
/* ... with options.  PSIGPU_GRAPH_FOLLOW_REVERSING: a link whose sides reverse (a+ -> b-, a- -> b+: an inversion) and reverse
 * steps of embedded paths are refused by default -- the device walks nodes forwards only -- while the reference follows every
 * out-link's `to` id and reads that node forwards, whatever side the link enters it by (include/psi/traverser_bfs.hpp:146-160
 * discards `linktype`).  With the flag the loader does exactly that: every link is the edge from -> to as written, a reverse
 * path step is the node; graphs with inversions load and the hit set is the reference's.  psikt: --follow-reversing-edges. */
#define PSIGPU_GRAPH_FOLLOW_REVERSING 1u
psigpu_graph* psigpu_graph_load_opts(const char* path, uint32_t flags, int* status);

/* In-memory construction (edge targets are node ranks). `path_nodes` holds node ranks of
 * the embedded (reference) paths; all path arguments may be NULL / 0. */
psigpu_graph* psigpu_graph_from_csr(uint64_t n_nodes, const uint64_t* node_id,
                                    const uint64_t* label_off, const char* labels,
                                    const uint64_t* edge_off, const uint32_t* edge_to,
                                    uint64_t n_paths, const uint64_t* path_off,
                                    const uint32_t* path_nodes, int* status);
void psigpu_graph_free(psigpu_graph* g);
int psigpu_graph_view_get(const psigpu_graph* g, psigpu_graph_view* out);
uint64_t psigpu_graph_path_count(const psigpu_graph* g);
uint64_t psigpu_graph_edge_count(const psigpu_graph* g);
/* Embedded path i as node ranks; returns its length, copies min(len, cap) entries. */
uint64_t psigpu_graph_path(const psigpu_graph* g, uint64_t i, uint32_t* out, uint64_t cap);

/* Index construction options. */
typedef struct psigpu_index_opts {
  uint32_t seed_len;       /* k (psikt -l): the starting loci are computed for this length */
  uint32_t n_per_region;   /* psikt -n: paths per embedded path (psigpu_index_build only) */
  uint32_t locus_step;     /* psikt -e: starting-locus sampling step, 0/1 = every locus */
  uint32_t sa_rate;        /* SA-order sampling rate, power of two; 0 = default (1: whole SA) */
  uint32_t ftab_len;       /* bases resolved by table lookup; 0 = auto (ceil(log4 n), <= 13; <= 15 when built on the device; 16 may be given explicitly),
                              0xFFFFFFFF = no table */
  uint32_t keep_text_sa;   /* keep the text and full suffix array for introspection (tests) */
  uint64_t rng_seed;       /* tie-breaking in path selection */
  uint32_t build_on_device; /* 0: suffix sorting and FM arrays on the host (SA-IS); d + 1: on GPU d
                              (prefix doubling); both give the identical index */
  uint32_t context;        /* psikt -t: patching context; 0 with `patched` = seed_len
                              (SeedFinder::set_context, seed_finder.hpp:1772-1787) */
  uint32_t patched;        /* psikt's default (no -P): index the first walk of a region whole and of every
                              further walk only the stretches no earlier walk covers (psigpu_index_build only) */
  uint32_t reserved1;
  uint64_t max_part_text;  /* text symbols per index part; 0 = the 32-bit row limit (2^32 - 256 on the device,
                              2^31 - 16 on the host).  An index whose paths do not fit one part is made in
                              several (whole-genome graphs with several indexed walks), each a complete FM
                              index; tests set it small */
} psigpu_index_opts;

/* SeedFinder::create_path_index(n, patched, context, step_size, ...) (seed_finder.hpp:1330-1355):
 * draw `n_per_region` walks per embedded path with the Haplotyper rules (graph_iter.hpp:537-731:
 * first out-edges for the first walk, then the out-edge that takes the walk where no earlier one
 * went; ties by an RNG seeded with `rng_seed`), index them whole or -- `patched` -- the first one
 * whole and of the others the patches no earlier walk covers (pathindex.hpp:496-560), then detect
 * the uncovered loci for seed length k and locus step (psikt -e; :1481-1541).
 * n_per_region == 0: no path index, every locus is a starting locus
 * (src/psikt.cpp:121-123; seed_finder.hpp:1543-1585). */
psigpu_index* psigpu_index_build(const psigpu_graph* g, const psigpu_index_opts* opts, int* status);
/* Same, over caller-chosen paths (node ranks); opts->n_per_region and opts->patched are ignored. */
psigpu_index* psigpu_index_build_paths(const psigpu_graph* g, const psigpu_index_opts* opts,
                                       uint64_t n_paths, const uint64_t* path_off,
                                       const uint32_t* path_nodes, int* status);
/* Same with trimmed paths -- what PathIndex::add_path takes (pathindex.hpp:153-161): path i covers
 * its first node from base head_off[i] on and of its last node the first tail_len[i] bases (0 = all
 * of it); Path::left / right (path_base.hpp:113-114, :240-246).  Either array may be NULL. */
psigpu_index* psigpu_index_build_patches(const psigpu_graph* g, const psigpu_index_opts* opts,
                                         uint64_t n_paths, const uint64_t* path_off,
                                         const uint32_t* path_nodes, const uint32_t* head_off,
                                         const uint32_t* tail_len, int* status);
/* PathIndex::load( prefix ) for a path index the REFERENCE wrote (pathindex.hpp:109-123): reads `paths_file` =
 * `<prefix>_paths` -- PathIndex::save_paths_set (:315-332): u64 context, u64 direction, PathSet::serialize
 * (pathset.hpp:260-274) = u64 #paths, then per path (path_base.hpp:551-560) an sdsl::enc_vector<elias_delta> of
 * external node ids, u64 left, u64 right, an sdsl::bit_vector of node breaks -- and builds this library's index
 * over those paths and trims (opts: seed_len, locus_step, sa_rate, ftab_len, build_on_device; the context comes
 * from the file).  The companion `<prefix>` file (sdsl::csa_wt over the reversed text) is not read: the FM index
 * is rebuilt.  sdsl's on-disk layouts are restated from its published sources (psi_amd/csrc/refio.cpp).
 * *context_out / *forward_out (either may be NULL): the file's context and sequence direction. */
psigpu_index* psigpu_index_from_reference_paths(const psigpu_graph* g, const psigpu_index_opts* opts, const char* paths_file,
                                                uint64_t* context_out, uint32_t* forward_out, int* status);
void psigpu_index_free(psigpu_index* x);
int psigpu_index_view_get(const psigpu_index* x, psigpu_index_view* out);
/* SeedFinder::serialize_path_index / load_path_index (seed_finder.hpp:1372-1413); own
 * container format, one file `<prefix>.psigpu`. */
int psigpu_index_save(const psigpu_index* x, const char* prefix);
psigpu_index* psigpu_index_load(const char* prefix, int* status);
/* Introspection used by tests: the indexed text (symbol 0 = sentinel, 1 = separator,
 * 2..5 = ACGT) and its full suffix array are kept only when opts->keep_text_sa was set. */
uint64_t psigpu_index_path_count(const psigpu_index* x);
uint64_t psigpu_index_path(const psigpu_index* x, uint64_t i, uint32_t* out, uint64_t cap);
int psigpu_index_path_trim(const psigpu_index* x, uint64_t i, uint32_t* head_off, uint32_t* tail_len);
/* Was this index made for this graph, seed length and locus step?  (load_path_index recomputes the
 * starting loci when its loci file does not match: seed_finder.hpp:1396-1413, utils.hpp:521-566;
 * here a mismatch means "no valid index": the caller builds a new one.)  1 = yes.  The graph is
 * recognised by a fingerprint over its ids, label bytes and edges, and the index's paths, trims and loci
 * are checked against it (nodes exist, consecutive nodes are joined by an edge, offsets inside the nodes). */
int psigpu_index_matches(const psigpu_index* x, const psigpu_graph* g, uint32_t seed_len, uint32_t locus_step);
uint32_t psigpu_index_locus_step(const psigpu_index* x);
/* The index's starting loci recomputed for another locus step (psikt -e) from its own paths and trims
 * (add_uncovered_loci( step ), seed_finder.hpp:1481-1541): what load_path_index does when the loci file of
 * the wanted step is missing (:1396-1413).  The index must have been made for `g`. */
int psigpu_index_set_locus_step(psigpu_index* x, const psigpu_graph* g, uint32_t locus_step);
/* The reference's own starting-loci file `<prefix>_loci_e<E>l<K>` (SeedFinder::save_starts /
 * open_starts, seed_finder.hpp:1640-1679; container serialisation utils.hpp:521-588): u64 count, then
 * raw psi::Position<> records { node id (external), offset } -- 8 + 8 bytes each, gum's id and offset
 * types (gum is not in the reference tree; a file that does not have this size is rejected).
 * save writes the index's loci for its (k, locus step); load replaces the index's loci by the file
 * made for (index k, locus_step) -- e.g. one written by the reference for the same paths.  The file carries
 * nothing that ties it to a graph or to the indexed paths: loading one is the caller's explicit decision
 * (SeedFinder::load_path_index never does it by itself; it recomputes: psigpu_index_set_locus_step). */
int psigpu_loci_save(const psigpu_index* x, const psigpu_graph* g, const char* prefix);
int psigpu_loci_load(psigpu_index* x, const psigpu_graph* g, const char* prefix, uint32_t locus_step);
const uint8_t* psigpu_index_text(const psigpu_index* x);      /* NULL unless kept */
const int32_t* psigpu_index_sa(const psigpu_index* x);        /* NULL unless kept */
/* Host helper: suffix array of a 0-terminated symbol string (own SA-IS). */
int psigpu_suffix_array(const uint8_t* text, uint64_t n, uint32_t sigma, int32_t* sa_out);

/* ------------------------------------------------------------------------------------
 * Device side.
 * ---------------------------------------------------------------------------------- */
typedef struct psigpu_ctx psigpu_ctx;

/* SeedFinder(graph, seed_len, gocc_threshold, ...) (seed_finder.hpp:930-942): one context
 * per GPU; calls on one context are serialised by the caller, distinct contexts are
 * independent.  Returns NULL when no usable GPU is present (no CPU fallback exists). */
psigpu_ctx* psigpu_create(int device);
void psigpu_destroy(psigpu_ctx* ctx);
const char* psigpu_last_error(const psigpu_ctx* ctx);   /* ctx may be NULL: last create error */

/* graph_ptr of the finder (seed_finder.hpp:1747): copied to HBM. */
int psigpu_load_graph(psigpu_ctx* ctx, const psigpu_graph_view* g);
/* load_path_index / create_path_index result (seed_finder.hpp:1330-1413): copied to HBM. */
int psigpu_load_index(psigpu_ctx* ctx, const psigpu_index_view* x);
/* SeedFinder gocc_threshold (seed_finder.hpp:939; index_iter.hpp:826-847): on-path k-mers
 * with more than `thr` path occurrences are skipped; 0 = unlimited. */
int psigpu_set_gocc_threshold(psigpu_ctx* ctx, uint32_t thr);
/* How a chunk is answered.  The seed length and the starting loci are fixed when the index is
 * made, so what the reference recomputes for EVERY chunk can be tabulated once, on the device,
 * when the first chunk arrives:
 *   PSIGPU_MODE_KMER_TABLE (default): a k-mer table in HBM maps every k-mer of the indexed paths
 *     to its suffix-array interval (the result of the backward search, fmindex.hpp:453-485) and
 *     every k-mer spelled by a k-walk from a starting locus to those loci (the result of the
 *     traverser, traverser_bfs.hpp:72-161); a seed is one probe.  Needs sa_rate 1 and the text on
 *     the device.  Falls back to LOCUS_TABLE, then TRAVERSE, when the tables do not fit.
 *   PSIGPU_MODE_LOCUS_TABLE: seeds_on_paths by FM-index backward search + locate (K1 / K2),
 *     seeds_off_paths from the table of the starting loci's k-walks.
 *   PSIGPU_MODE_TRAVERSE: nothing about the starting loci is tabulated (graphs with too many k-walks per
 *     locus): every starting locus is traversed for every chunk, pruned by the chunk's seeds, as in the
 *     reference.  The on-path phase is one probe of a table of the PATHS' k-mers (one entry per path position
 *     whatever the graph looks like) when that applies (sa_rate 1, text on the device, k <= 31) and fits, else
 *     FM-index backward search + locate -- the reference's scheme as written; PSIGPU_TUNE_NO_PATH_TABLE: always.
 * Loci with more than `walk_cap` k-walks (dense, high-degree regions) are left out of the tables
 * and traversed per chunk in every mode.  walk_cap = 0 is the default policy: 256 walks, then
 * further enumeration passes with larger caps (2^16, 2^22) while only few loci are over and a
 * budget of table entries holds.  Same hit set in all modes (tests/test_gpu_parity.py runs each
 * test in all of them).  Records of the k-mer table mode come out in seed order; raw emission
 * order is otherwise unspecified, as in the reference. */
#define PSIGPU_MODE_KMER_TABLE 0u
#define PSIGPU_MODE_TRAVERSE 1u
#define PSIGPU_MODE_LOCUS_TABLE 2u
/* PSIGPU_MODE_AUTO: KMER_TABLE or TRAVERSE, whichever costs less device time over the calls the caller expects
 * (psigpu_set_option "expected_calls": chunks the finder will be asked, 0 = unknown = many).  The tables are made by
 * the device in time proportional to the k-walks of the starting loci + the path positions (chr22-like: 48 ms; whole
 * genome: 6-14 s); they save the traverser's walk over ALL starting loci and the chunk's seed table on every chunk
 * (chr22-like: 1.2 ms per 1 M-read chunk; whole genome: ~90 ms per 10 M reads) -- so a finder that answers one chunk
 * (psikt on a small FASTQ) is better off traversing, one that answers hundreds is not.  Decided when the tables would be
 * made (psigpu_prepare / the first query) from the index's own sizes; psigpu_query_mode tells which it was. */
#define PSIGPU_MODE_AUTO 3u
int psigpu_set_query_mode(psigpu_ctx* ctx, uint32_t mode, uint32_t walk_cap);
uint32_t psigpu_query_mode(const psigpu_ctx* ctx);     /* the mode queries run in (AUTO: what it resolved to, AUTO until then) */

/* Measurement switches: A/B runs behind bench.py's roofline series (which kernel answers the on-path phase
 * of the FM modes).  The hit set never depends on them.  No counterpart in the reference. */
#define PSIGPU_TUNE_NO_DIRECT 1u     /* K1 by the quad LF kernel only (k_fm_search): no lane-per-seed kernel that finishes a
                                        seed from its interval-table entry and the rows' records */
#define PSIGPU_TUNE_NO_VERIFY 2u     /* every base of a seed by an LF step (fmindex.hpp:851-869 as written): small intervals
                                        are not finished by comparing the rows with the text */
#define PSIGPU_TUNE_NO_ROWRECS 4u    /* no per-row records (SaRec, located suffix array): locate through SA + segment table */
#define PSIGPU_TUNE_NO_PATH_TABLE 8u /* traverse mode: the FM index answers the on-path phase (the reference's scheme as written)
                                        instead of the table of the paths' k-mers */
#define PSIGPU_TUNE_NO_SWEEP 16u     /* the LF steps by one quad per seed (k_fm_search, rounds 1-5) instead of the level-synchronous
                                        sweeps (k_fm_sweep: seeds bucketed by interval, rank blocks staged in LDS, live intervals
                                        compacted by wave ballots) */
int psigpu_set_tuning(psigpu_ctx* ctx, uint32_t flags);

/* Per-context switches of the host entry (psigpu_find_seeds / _packed), by name -- what the PSIGPU_* environment
 * variables of the same meaning set for a whole process, settable per context and at run time (a test that flips an
 * environment variable while library threads are running races with their getenv).  The records returned never depend
 * on them.  PSIGPU_ERR_ARG for an unknown name.
 *   "sub_bytes"       bases per sub-batch of the pipeline (0 = default 16 Mi; tests cut small inputs into many)
 *   "no_ahead"        1: two-slot pipeline even when reads and offsets are pinned (no transfers queued ahead)
 *   "no_engine_copy"  1: every transfer through hipMemcpyAsync on the pipeline's streams instead of a named SDMA engine;
 *                     takes effect when set before the context's first host-entry call
 *   "wire"            bytes per record on the device-to-host link: 0 = the narrowest that fits (5-7 packed, 8, 16, 32), or
 *                     5 / 6 / 7 / 8 / 16 / 32 = nothing narrower than that
 *   "widen_threads"   host threads that widen the wire records (0 = min(12, cores / 4)); before the context's first host-entry call
 *   "expected_calls"  PSIGPU_MODE_AUTO: how many chunks this finder will be asked (0 = unknown: assume many)
 *   "expected_seeds"  PSIGPU_MODE_AUTO: ... and how many seeds over all of them (0 = unknown)
 *   "no_lookahead"    1: every sub-batch of the host entry is synchronised before the next one's kernels are queued (rounds 1-3)
 *   "res16"           1: the k-mer table probe leaves 16 bytes of results per seed for the emit kernel (rounds 1-3) instead of 8
 *   "no_pfx_roots"    1: the query-time traverser starts from the starting loci themselves (TraverserBFS as written,
 *                     traverser_bfs.hpp:72-161) instead of from their tabulated 12-base prefix walks */
int psigpu_set_option(psigpu_ctx* ctx, const char* name, uint64_t value);

/* Builds the tables of the current query mode for seed length k now (index load time) instead of
 * inside the first query: the k-walks of the starting loci enumerated by the traverser kernel, the
 * path k-mers read off the suffix array, sorted and hashed.  Blocks until the device is idle (it
 * allocates and frees gigabytes).  Without this call the first psigpu_find_seeds* call does the
 * same.  No counterpart in the reference (it re-traverses every chunk); closest in role:
 * SeedFinder::create_path_index / load_path_index (seed_finder.hpp:1330-1413). */
int psigpu_prepare(psigpu_ctx* ctx, uint32_t k);

/* flags for psigpu_find_seeds* */
#define PSIGPU_ON_PATHS 1u      /* SeedFinder::seeds_on_paths  (seed_finder.hpp:1426-1457) */
#define PSIGPU_OFF_PATHS 2u     /* SeedFinder::seeds_off_paths (seed_finder.hpp:1703-1722) */
#define PSIGPU_ALL 3u           /* SeedFinder::seeds_all       (seed_finder.hpp:1724-1732) */
#define PSIGPU_SORT_UNIQUE 4u   /* return sort-unique hits ordered by (read_id, read_offset,
                                   node_id, node_offset) instead of the raw emission stream (which
                                   may hold a hit more than once, like the reference's: a position
                                   on several indexed paths, found on a path and from a locus);
                                   done on the device: the hits of each seed ordered in place when
                                   they come out seed by seed, else 64-bit packed keys + radix sort */

#define PSIGPU_UNIFORM_READS 8u /* the caller knows that every read of the chunk has the same length (read_off[i] = i * L):
                                   a seed's read and offset then follow from its number -- no scan over the reads' seed
                                   counts, no search for a seed's read.  The claim is CHECKED on the device while the seeds
                                   are packed; a chunk for which it does not hold is answered again the general way (same
                                   records, a little later).  psi::Records sets it for chunks of equal-length reads. */

#define PSIGPU_ANY_ORDER 16u    /* raw records (no PSIGPU_SORT_UNIQUE: ignored with it) may come out in ANY order, as the
                                   reference's callback stream does (its order is unspecified: seed_finder.hpp:1724-1732 walks
                                   an index, not the reads).  Without the flag the default mode hands the records out seed by
                                   seed in read order, which costs its one kernel a look-back over the tiles in front of every
                                   tile (~0.06 ms per 7 M seeds); with it a tile takes its output range with one atomic add:
                                   same records, blocks of 1024 seeds in the order they finished.  Other query modes ignore it. */

/* One chunk of psikt's loop: get_seeds + index_reads + seeds_all (src/psikt.cpp:195-204).
 * `bases`/`read_off` are HOST buffers (read i = bases[read_off[i] .. read_off[i+1])),
 * `step` is psikt's -d (0 = k), `rec_offset` the number of reads consumed before this
 * chunk (sequence.hpp:1616).  Hits come back in library-owned pinned host memory.
 * This is SURVEY 8(d)'s timed region (H2D of the reads + kernels + D2H of the hits): the chunk
 * is cut into sub-batches that are pipelined over three streams, so the call costs about
 * max(bytes in, bytes out) / PCIe rate.  When the graph's ids are rank + constant the records cross the link narrower
 * than they are -- packed records of 5-7 bytes (blocks of 256 behind their first read id; read id relative to it, read
 * offset in seed distances, node rank, node offset) for records in read order, else one 64-bit key, else 4 x u32; every
 * field checked on the device, the next wider form when one does not fit -- and are
 * widened to psigpu_hit by host threads while the next sub-batch is in flight.  Reads held in pinned memory (psigpu_host_alloc) are
 * DMA'd in place; pageable reads are staged by a helper thread. */
int psigpu_find_seeds(psigpu_ctx* ctx, const char* bases, const uint64_t* read_off,
                      uint64_t n_reads, uint32_t k, uint32_t step, uint64_t rec_offset,
                      uint32_t flags, psigpu_hits* out);
void psigpu_free_hits(psigpu_hits* hits);

/* The same chunk with the reads PACKED to 2 bits per base -- what crosses the host link is then a quarter of the ASCII
 * bases (psikt's reader packs while it parses: psi::Records, psi_amd/include/psi/sequence.hpp; the reference holds reads
 * as seqan2::Dna5QString, one byte per base: sequence.hpp:1130-1294).  Base i of the chunk (reads back to back; read_off
 * counts BASES exactly as for psigpu_find_seeds) is the two bits 63 - 2 (i % 32), 62 - 2 (i % 32) of packed[i / 32]:
 * A 0, C 1, G 2, T 3, first base most significant.  n_mask (may be NULL: every base of the chunk is ACGT): bit i % 64 of
 * n_mask[i / 64] set = base i is not ACGT (its two code bits are ignored; a seed that covers it yields nothing, as an N
 * does in psigpu_find_seeds).  packed holds ceil(n / 32) words, n_mask ceil(n / 64), n = read_off[n_reads]; both may be
 * pinned (DMA'd in place) or pageable (staged).  read_off[0] need not be 0 here: the reads may be a contiguous RANGE of
 * a larger packed chunk (psikt --devices hands every GPU one), whose bases start at base read_off[0] of the arrays.  Same records, same order, same flags as psigpu_find_seeds on the
 * corresponding ASCII bases.  psigpu_pack_reads makes the two arrays from ASCII bases on the calling thread. */
int psigpu_find_seeds_packed(psigpu_ctx* ctx, const uint64_t* packed, const uint64_t* n_mask, const uint64_t* read_off,
                             uint64_t n_reads, uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags,
                             psigpu_hits* out);
/* ASCII bases [first, first + n) of a chunk -> their bits in packed / n_mask (arrays for the whole chunk, zeroed by the
 * caller or filled front to back: the words that [first, first + n) touches only partly are or-ed into).  Any byte that
 * is not one of ACGTacgt sets its mask bit.  Returns the number of non-ACGT bases in the range.  n_mask may be NULL only
 * for a caller that looks at that count: such a base is then packed as A, and a chunk packed that way and handed to
 * psigpu_find_seeds_packed with n_mask = NULL yields seeds ACROSS it -- not what psigpu_find_seeds does with an N -- so a
 * non-zero count without a mask array means "pack again with one".  Host only, no GPU needed; ranges that do not share a
 * 64-base block may be packed by different threads at once (32 bases per step where the host has AVX2 + BMI2). */
uint64_t psigpu_pack_reads(const char* bases, uint64_t first, uint64_t n, uint64_t* packed, uint64_t* n_mask);

/* Seed::gocc (seed.hpp:45, "genome occurrence count") of the on-path phase: how often each seed's k-mer occurs in the indexed
 * path text -- count_occurrences( fst_itr ) of kmer_exact_matches (index_iter.hpp:842-843), what _add_occurrences hands to
 * every hit of the seed (index_iter.hpp:743, :674).  counts[s] for seed s of the chunk, reads in order and a read's seeds in
 * offset order (offsets 0, step, 2 step ... as psigpu_find_seeds takes them; n_counts must be their number); 0 for a seed
 * with an N and for an index without paths; the gocc threshold is NOT applied (it is a filter on this very number).  The
 * counts come from the FM index (k_fm_search: the size of the seed's suffix-array interval, over all parts of the index),
 * whatever the query mode: needs an index view with rank blocks.  The 32-byte records of psigpu_find_seeds* do not carry
 * the field (psikt does not write it: src/psikt.cpp:176-179); the header API's callbacks get it from here
 * (psi_amd/include/psi/seed_finder.hpp).  The off-path phase's gocc is the number of READ positions with the k-mer
 * (traverser_bfs.hpp:103-108: length( saPositions ) of the reads index): a count over the chunk's seeds, made by the caller. */
int psigpu_count_occurrences(psigpu_ctx* ctx, const char* bases, const uint64_t* read_off, uint64_t n_reads, uint32_t k,
                             uint32_t step, uint32_t* counts, uint64_t n_counts);

/* MEM mode -- SeedFinder::seeds_on_paths( sequence, callback ) -> find_mems
 * (seed_finder.hpp:1459-1479, index_iter.hpp:854-906): per read, from `start` the pattern grows
 * base by base while it still occurs on the indexed paths; once it is at least `minlen` long and has
 * at most gocc_threshold occurrences (psigpu_set_gocc_threshold), every occurrence is reported
 * (read_offset = start, match_len = pattern length, gocc = number of occurrences) and the search
 * restarts one base behind the pattern's end; a base that cannot be appended, or an N, restarts it
 * one base behind that base; `max_mem` (psikt -E, 0 = unlimited) ends a read once that many records
 * are out.  Records are the reference's psi::Seed<> (seed.hpp:32-46), ordered by (read_id,
 * read_offset, node_id, node_offset).  Needs an index with sa_rate 1 (the default). */
typedef struct psigpu_mem_hit {
  uint64_t node_id, node_offset, read_id, read_offset, match_len, gocc;
} psigpu_mem_hit;
typedef struct psigpu_mems {
  uint64_t n;
  psigpu_mem_hit* data;   /* library-owned pinned host memory; release with psigpu_free_mems */
} psigpu_mems;
int psigpu_find_mems(psigpu_ctx* ctx, const char* bases, const uint64_t* read_off, uint64_t n_reads,
                     uint32_t minlen, uint32_t max_mem, uint64_t rec_offset, psigpu_mems* out);
void psigpu_free_mems(psigpu_mems* mems);

/* Pinned (page-locked) host memory for read chunks: what the reference keeps in
 * Records / seqan2::StringSet (sequence.hpp:1130-1294) the caller keeps here, and the copy
 * engine reads it in place.  NULL when no GPU is present. */
void* psigpu_host_alloc(uint64_t bytes);
void psigpu_host_free(void* p);

/* (ABI 8) The lifetime rule of the host entry's transfer buffers.  The reference's chunk loop (src/psikt.cpp:190-208) owns its
 * reads and its output for as long as it likes; here both ends of every raw copy-engine transfer (staging and landing buffers,
 * memory from psigpu_host_alloc, returned hit arrays) belong to a pool that lives as long as the process: psigpu_destroy,
 * psigpu_host_free, psigpu_free_hits and a regrowing call hand them back to the pool, never straight to the driver, and the
 * pool returns memory to the driver only when more than a budget sits idle and only after every engine queue it was used on
 * has consumed a marker submitted after the buffer came back.  out[0..7] = buffers allocated so far, takes served from the
 * pool, buffers returned to the driver, queue drains run, idle device bytes, idle host bytes, buffers idle, buffers in use.
 * `trim_all` != 0 first returns every idle buffer to the driver (by the rule above). */
void psigpu_copy_pool_stats(uint64_t out[8], int trim_all);
/* (ABI 8) `count` hit arrays of `n_records` records made ahead of time: taken from the pool above and handed straight back, so
 * that the first calls of a chunk loop find their output arrays waiting instead of pinning a few hundred megabytes each
 * (psikt does this on a side thread while the graph is parsed: src/psikt.cpp:190-208 has no such cost -- its records go to
 * the file one callback at a time).  Returns how many could be made.  Needs a GPU runtime, no context. */
uint32_t psigpu_reserve_hit_arrays(uint64_t n_records, uint32_t count);

/* Same with the chunk already resident in HBM and the hits left there (n_bases must be the
 * total length of the reads, d_read_off[n_reads]): `d_bases` and
 * `d_read_off` are DEVICE pointers; `stream` is a hipStream_t (NULL = default stream).
 * On return *d_hits points at library-owned device memory holding *n_hits records, valid
 * until the next call on this context.  The call is asynchronous up to the final count
 * read-back (one stream synchronise; one more with PSIGPU_SORT_UNIQUE when the radix sort is needed). */
int psigpu_find_seeds_device(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_read_off,
                             uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step,
                             uint64_t rec_offset, uint32_t flags, void* stream,
                             const psigpu_hit** d_hits, uint64_t* n_hits);
/* ... with the device-resident reads packed (layout of psigpu_find_seeds_packed; d_packed holds ceil(n_bases / 32) + 2
 * words, d_n_mask -- may be NULL -- ceil(n_bases / 64) + 1: the seeding kernel loads the word behind a seed's last). */
int psigpu_find_seeds_device_packed(psigpu_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_n_mask,
                                    const uint64_t* d_read_off, uint64_t n_reads, uint64_t n_bases, uint32_t k,
                                    uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream,
                                    const psigpu_hit** d_hits, uint64_t* n_hits);

/* Two chunks in flight (ABI 6).  A caller that has its next chunk resident before it needs the hits of the current one --
 * a loop that double-buffers its read batches, what the reference's chunked loop (src/psikt.cpp:190-206: load a chunk,
 * seeds_all, next chunk) becomes when loading and seeding overlap -- begins chunk i + 1 before it ends chunk i: the device
 * does not wait for the host between two chunks (the synchronisation, the counters' way back, the next call's launches).
 *   _begin   queues the chunk's kernels on `stream` and returns; at most two chunks may be begun and not ended
 *            (PSIGPU_ERR_STATE for a third); arguments as for psigpu_find_seeds_device / _device_packed; the reads and
 *            offsets must stay untouched until the chunk's _end.  The chunks in flight share one workspace and are kept
 *            apart by stream order alone: a _begin on another stream than the chunk still in flight is refused
 *            (PSIGPU_ERR_STATE); with nothing in flight any stream will do
 *   _end     waits for the OLDEST chunk begun and hands out its hits: library-owned device memory, valid until the NEXT
 *            _end of this context (or the next call of another entry point of it) -- a chunk answered inside its _end
 *            by the synchronous entry (below) leaves its records in the buffer the next such chunk reuses; work queued on
 *            the chunk's stream before that next _end is ordered before the reuse.  psigpu_get_counters describes that chunk
 * A chunk that needs more than the default mode's kernels (another query mode, a walk-capped table with a traverser pass,
 * tables not made yet, buffers that would have to grow under the chunk in flight, reads that are not of one length behind
 * PSIGPU_UNIFORM_READS) is answered by psigpu_find_seeds_device inside its _end: same records, no overlap.  While chunks
 * are begun and not ended every other entry point of the context returns PSIGPU_ERR_STATE. */
int psigpu_find_seeds_device_begin(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_read_off, uint64_t n_reads,
                                   uint64_t n_bases, uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream);
int psigpu_find_seeds_device_packed_begin(psigpu_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_n_mask,
                                          const uint64_t* d_read_off, uint64_t n_reads, uint64_t n_bases, uint32_t k,
                                          uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream);
int psigpu_find_seeds_device_end(psigpu_ctx* ctx, const psigpu_hit** d_hits, uint64_t* n_hits);

/* Has anything the finder READS on the device changed since it was loaded?  Every array psigpu_load_graph / psigpu_load_index
 * put on the device left a checksum behind; this recomputes them (a few milliseconds for 9 GB).  *n_changed = the arrays
 * whose content is no longer what was loaded, `report` (may be NULL) their names.  A diagnostic of the load campaigns
 * (tools/fuzz_modes.py asks it when a hit set is wrong); with PSIGPU_VERIFY_UPLOAD=1 in the environment every upload is
 * also compared with its host source when it is made (PSIGPU_ERR_DEVICE from the loader on a difference). */
int psigpu_verify_resident(psigpu_ctx* ctx, uint32_t* n_changed, char* report, uint64_t report_cap);

/* Copies `n` records that psigpu_find_seeds_device left in HBM into host memory (blocking;
 * through this library's HIP runtime, so that a binding never has to load one of its own). */
int psigpu_copy_hits(psigpu_ctx* ctx, psigpu_hit* host_dst, const psigpu_hit* d_src, uint64_t n);

/* SeedFinderStats / TraverserStats counters of the last call (seed_finder.hpp:111-494;
 * traverser_base.hpp:108-268) plus per-kernel device times from HIP events recorded on the
 * stream the kernels ran on. */
typedef struct psigpu_counters {
  uint64_t n_reads, n_seeds, n_seeds_valid;    /* valid = no N */
  uint64_t n_seeds_on_path;                    /* seeds with a non-empty SA interval */
  uint64_t n_hits_on_path, n_hits_off_path;    /* raw emissions */
  uint64_t n_hits;                             /* returned (after sort-unique if requested) */
  uint64_t n_kpaths;                           /* complete k-walks enumerated by the traverser */
  uint64_t n_loci;
  uint64_t n_spilled;                          /* traverser states spilled out of LDS */
  uint64_t n_lf_steps;                         /* LF (backward-search) steps executed by K1 */
  uint64_t n_rows_verified;                    /* SA rows K1 finished by comparing with the text */
  uint64_t n_locus_kmers;                      /* entries of the locus k-mer table (0: not in use) */
  uint64_t n_path_kmers;                       /* distinct path k-mers in the k-mer table (0: not in use) */
  uint64_t n_loci_traversed;                   /* starting loci the traverser walked for this chunk */
  float ms_pack, ms_table, ms_search, ms_locate, ms_traverse, ms_sort, ms_total;
  float ms_probe;                              /* k-mer table / locus table probe */
  float ms_locus_table_build;                  /* one-off: building the tables (first query) */
  uint32_t search_launches, traverse_launches;
  uint32_t sorted_in_place;                    /* PSIGPU_SORT_UNIQUE: sub-batches whose hits, emitted seed by seed, only needed
                                                * the hits of each seed put in order (no radix sort) */
  uint32_t wire_bytes_per_hit;                 /* psigpu_find_seeds: bytes per record on the device-to-host link (5-8 or 16: packed, widened
                                                * on the host; 32: as returned; the widest any sub-batch of the call used); 0 for the
                                                * device-resident entry */
  uint64_t n_locate_steps;                     /* LF steps K2 walked from occurrences to sampled suffix-array rows (sa_rate > 1) */
  uint32_t lookahead_subbatches;               /* psigpu_find_seeds*: sub-batches of the call whose kernels were queued while the one
                                                * before was still in flight (default mode, reads and offsets in pinned memory) */
  uint32_t fused_step;                         /* 1: the call's seeds were packed, probed and emitted by ONE kernel (k_kmer_step: default mode,
                                                * every seed answered by the k-mer table); ms_probe is then that kernel, ms_locate 0 */
  uint64_t lookahead_fallbacks;                /* since the context was made: chunks handed from that arrangement to the synchronous
                                                * loop (more hits than expected, a seed with many hits, a wire field too narrow ...) */
  uint64_t stale_handbacks;                    /* since the context was made: calls whose counter block came back from the device with
                                                * another call's serial number (detected, fetched again; expected 0) */
} psigpu_counters;
int psigpu_get_counters(const psigpu_ctx* ctx, psigpu_counters* out);

/* The secondary bound SURVEY 8(d) asks to be reported beside the HBM roofline -- "random-access efficiency
 * (achievable fraction of peak at 64-B granularity)" -- measured on this device, now: `n_loads` independent loads
 * at random addresses of a scratch table of `table_bytes`, launched as the query kernels are (8192 workgroups);
 * quad_sectors = 0: every lane loads 16 bytes from a sector of its own (the k-mer table probe's access),
 * 1: the four lanes of a quad load one 64-byte sector (a rank block, a segment record).  Best of five launches. */
int psigpu_measure_random_loads(psigpu_ctx* ctx, uint64_t table_bytes, uint64_t n_loads, uint32_t quad_sectors,
                                double* loads_per_s);

/* ------------------------------------------------------------------------------------
 * The gather of hit lists over RCCL / xGMI (BASELINE north_star: "RCCL over xGMI only to gather hit lists"): one
 * process per GPU, reads sharded in contiguous ranges, every rank's device-resident records (psigpu_find_seeds_device
 * with PSIGPU_SORT_UNIQUE) into the root's HBM in rank order -- which is the sorted chunk.  An all-gather of the counts,
 * then one point-to-point transfer per rank inside one group.  RCCL is loaded on first use (psigpu_comm_available).
 * No counterpart in the reference.  The unique id is made by one rank (psigpu_comm_unique_id) and handed to the others
 * by the launcher (MPI, torch.distributed, a file); psigpu_comm_create is collective over the `world` ranks.
 * ---------------------------------------------------------------------------------- */
#define PSIGPU_COMM_ID_BYTES 128u
typedef struct psigpu_comm psigpu_comm;
int psigpu_comm_available(void);                                  /* 1: RCCL could be loaded */
int psigpu_comm_unique_id(uint8_t id[PSIGPU_COMM_ID_BYTES]);
psigpu_comm* psigpu_comm_create(int device, const uint8_t id[PSIGPU_COMM_ID_BYTES], int rank, int world);
void psigpu_comm_destroy(psigpu_comm* comm);
const char* psigpu_comm_last_error(const psigpu_comm* comm);      /* comm may be NULL: last create / id error on this thread */
/* d_hits: `n` records in this rank's HBM.  On the root *d_all points at library-owned device memory (valid until the
 * next gather on this communicator) holding *n_all records, rank 0's first; elsewhere NULL / 0.  counts (may be NULL):
 * every rank's record count.
 * PRECONDITION: the records are COMPLETE when the call is made -- the gather runs on a stream of the communicator's
 * own, which is not ordered against the stream that produced d_hits.  psigpu_find_seeds_device synchronises its stream
 * before it returns, so records that came from it qualify; a caller that fills d_hits itself synchronises first.
 * Collective: every rank of the communicator calls it.  Errors: a rank that cannot take part (the root out of device
 * memory) makes the call return PSIGPU_ERR_NOMEM on EVERY rank before any transfer is posted, and the communicator
 * stays usable; a failure of the collective calls themselves aborts the communicator (ncclCommAbort: the peers'
 * pending operations fail instead of hanging) and later gathers on it return PSIGPU_ERR_STATE. */
int psigpu_gather_hits(psigpu_comm* comm, const psigpu_hit* d_hits, uint64_t n, int root, const psigpu_hit** d_all,
                       uint64_t* n_all, uint64_t* counts);

uint32_t psigpu_abi_version(void);
/* Text of the last host-side (graph / index) failure on this thread. */
const char* psigpu_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PSI_GPU_H */
