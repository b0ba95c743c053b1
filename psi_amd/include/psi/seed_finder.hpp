// psi::SeedFinder -- drop-in for the query surface of the reference's
// psi::SeedFinder (include/psi/seed_finder.hpp:761-1788) as psikt drives it
// (src/psikt.cpp:83-212).  Same member names and argument meaning; exceptions are
// std::runtime_error as in the reference.  All compute goes through the C ABI
// (include/psi_gpu.h) into the HIP kernels; there is no CPU path here.
//
// Differences, all documented in DESIGN.md:
//  * one concrete type instead of template<TStats, TTraits>; both parameters are accepted and
//    ignored so that `SeedFinder< NoStats, Traits >` in caller code still compiles;
//  * get_seeds()/index_reads() only describe the chunk (k, distance): seeding and the seeds
//    index are built on the device inside seeds_all();
//  * path selection, indexing and the starting loci are made together by the library
//    (psigpu_index_build): pick_paths() does all three, index_paths() / add_uncovered_loci() are kept
//    for callers that spell the steps out (reference test/src/test_seedfinder.cpp:98-163).
#ifndef PSI_AMD_SEED_FINDER_HPP__
#define PSI_AMD_SEED_FINDER_HPP__

#include <cctype>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "graph.hpp"
#include "pathindex.hpp"
#include "seed.hpp"
#include "sequence.hpp"
#include "psi_gpu.h"

namespace psi {
  struct NoStats {};
  struct WithStats {};
  struct PerComponent {};
  struct Whole {};
  struct DefaultTraits {};

  /** The seeds of one chunk: what seeding() would produce, described rather than copied. */
  struct SeedsRecord {
    Records const* chunk = nullptr;
    unsigned int seed_len = 0;
    unsigned int distance = 0;
  };
  /** Placeholder for the reads (seeds) index; the real one lives in HBM. */
  struct ReadsIndex { SeedsRecord const* seeds = nullptr; };
  /** Placeholder for the Traverser object psikt creates once and passes back in. */
  struct Traverser {
    typedef Seed<> output_type;
    unsigned int seed_len = 0;
    SeedsRecord const* reads = nullptr;   /**< set by SeedFinder::setup_traverser */
  };

  template< typename TStatsSpec = NoStats, typename TTraits = DefaultTraits >
  class SeedFinder {
  public:
    typedef Graph graph_type;
    typedef Records readsrecord_type;
    typedef ReadsIndex readsindex_type;
    typedef Traverser traverser_type;
    typedef PathIndex pathindex_type;
    typedef Seed<> output_type;

    /** SeedFinder( graph, seed_len, gocc_threshold, max_mem ) (reference :930-942). */
    SeedFinder( graph_type const& g, unsigned int len, unsigned int gocc_thr = 0,
                unsigned int mxmem = 0, int device = 0 )
      : graph_ptr( &g ), seed_len( len ), gocc_threshold( gocc_thr ), max_mem( mxmem ), device_( device ), pindex( g )
    {
      if ( len == 0 || len > PSIGPU_MAX_SEED_LEN )
        throw std::runtime_error( "seed length out of range (1.." + std::to_string( PSIGPU_MAX_SEED_LEN ) + ")" );
      ctx = psigpu_create( device );
      if ( ctx == nullptr ) throw std::runtime_error( psigpu_last_error( nullptr ) );
      check( psigpu_load_graph( ctx, &g.view() ) );
      if ( gocc_thr ) check( psigpu_set_gocc_threshold( ctx, gocc_thr ) );
    }
    /** How chunks are answered (no counterpart in the reference; include/psi_gpu.h):
     *  PSIGPU_MODE_KMER_TABLE (default), PSIGPU_MODE_LOCUS_TABLE, PSIGPU_MODE_TRAVERSE (the
     *  reference's own scheme).  Same hits in every mode. */
    void set_query_mode( unsigned int mode, unsigned int walk_cap = 0 )
    {
      check( psigpu_set_query_mode( ctx, mode, walk_cap ) );
    }
    /** PSIGPU_MODE_AUTO: how much work the caller expects (chunks, seeds over all of them; 0 = unknown = a lot). */
    void set_expected_work( std::uint64_t calls, std::uint64_t seeds )
    {
      check( psigpu_set_option( ctx, "expected_calls", calls ) );
      check( psigpu_set_option( ctx, "expected_seeds", seeds ) );
    }
    unsigned int query_mode() const { return psigpu_query_mode( ctx ); }
    SeedFinder( SeedFinder const& ) = delete;
    SeedFinder& operator=( SeedFinder const& ) = delete;
    ~SeedFinder() { psigpu_destroy( ctx ); }

    /* ---- index ------------------------------------------------------------------------ */
    /** pick_paths( n, patched, context, callback, info, warn ) (reference :1138-1167): `n` walks per
     *  embedded path drawn with the Haplotyper rules, whole or patched.  The library indexes them and
     *  detects the uncovered loci in the same call. */
    void pick_paths( unsigned int n, bool patched = true, unsigned int context = 0,
                     std::function< void( std::string const&, int ) > callback = nullptr,
                     std::function< void( std::string const& ) > info = nullptr,
                     std::function< void( std::string const& ) > warn = nullptr )
    {
      (void)callback; (void)info;
      if ( n != 0 && graph_ptr->get_path_count() == 0 )
        throw std::runtime_error( "no reference path found in the input graph" );      /* reference :1145-1147 */
      if ( patched && context == 0 && warn )                                             /* set_context, :1772-1787 */
        warn( "The context size cannot be zero for patching. Assuming the seed length as the context size..." );
      pick_n = n; pick_patched = patched; pick_context = context;
      build( 1 );
    }
    /** index_paths() (reference :1169-1176): done by pick_paths. */
    void index_paths() {}
    /** add_uncovered_loci( step ) (reference :1481-1541): the loci for step 1 exist after pick_paths;
     *  another step rebuilds. */
    void add_uncovered_loci( unsigned int step = 1 ) { if ( step != built_step ) build( step ); }

    /** create_path_index( n, patched, context, step_size, dmin, dmax, mode, info, warn )
     *  (reference :1330-1355). */
    template< typename TMode = PerComponent >
    void create_path_index( unsigned int n, bool patched = true, unsigned int context = 0,
                            unsigned int step_size = 1, unsigned int dmin = 0, unsigned int dmax = 0,
                            TMode = {}, std::function< void( std::string const& ) > info = nullptr,
                            std::function< void( std::string const& ) > warn = nullptr )
    {
      if ( ( dmin || dmax ) && warn ) warn( "the distance index is not part of the seed-finding path; ignored" );
      if ( n != 0 && graph_ptr->get_path_count() == 0 )
        throw std::runtime_error( "no reference path found in the input graph" );
      if ( patched && context == 0 && n != 0 && warn )
        warn( "The context size cannot be zero for patching. Assuming the seed length as the context size..." );
      if ( info ) info( "Selecting and indexing " + std::to_string( n ) + ( patched ? " patched" : "" ) +
                        " path(s) per region..." );
      pick_n = n; pick_patched = patched; pick_context = context;
      if ( info ) info( "Detecting uncovered loci..." );
      build( step_size );
    }

    /** load_path_index( prefix, context, step, dmin, dmax ) -> bool (reference :1396-1413).  A file made
     *  for another graph, seed length or locus step is not a valid index for this finder (the reference
     *  keys its loci file on seed length and step and recomputes on a mismatch, utils.hpp:521-566). */
    bool load_path_index( std::string const& prefix, unsigned int /*context*/ = 0,
                          unsigned int step_size = 1, unsigned int = 0, unsigned int = 0 )
    {
      if ( !pindex.load( prefix ) ) {
        /* no container of this library: a path index the reference wrote?  `<prefix>_paths` gives the paths
         * and their trims; the FM index and the starting loci are made from them */
        psigpu_index_opts o{};
        o.seed_len = seed_len; o.locus_step = step_size;
        o.build_on_device = device_ < 0 ? 0u : static_cast< unsigned int >( device_ ) + 1u;
        if ( !pindex.load_reference( prefix, o ) ) return false;
      }
      if ( !psigpu_index_matches( pindex.handle(), graph_ptr->handle(), seed_len, step_size ) ) {
        /* same graph and seed length, other locus step: the paths are still good, and the loci for this
         * step are recomputed from them (the reference recomputes too when open_starts finds no file for
         * the step, :1396-1413).  A `_loci_e<E>l<K>` file that happens to lie beside the index is NOT
         * used here: it names no graph and no paths, a stale one would silently lose off-path hits. */
        bool ok = psigpu_index_matches( pindex.handle(), graph_ptr->handle(), seed_len,
                                        psigpu_index_locus_step( pindex.handle() ) ) &&
                  pindex.set_locus_step( step_size );
        if ( !ok ) { pindex.clear(); return false; }
      }
      check( psigpu_load_index( ctx, &pindex.view() ) );
      check( psigpu_prepare( ctx, seed_len ) );
      built_step = step_size;
      return true;
    }

    /** serialize_path_index( prefix, step ) -> bool (reference :1372-1394). */
    bool serialize_path_index( std::string const& prefix, unsigned int /*step_size*/ = 1 )
    { return pindex.serialize( prefix ) && pindex.save_loci( prefix ); }      /* + save_starts (:1659-1679) */

    /* ---- per chunk ---------------------------------------------------------------------- */
    readsrecord_type create_readrecord() const { return readsrecord_type(); }
    traverser_type create_traverser() const { return traverser_type{ seed_len }; }

    /** get_seeds( seeds, chunk, distance ) (reference :1099-1109): distance 0 = seed length. */
    void get_seeds( SeedsRecord& seeds, readsrecord_type const& chunk, unsigned int distance ) const
    { seeds.chunk = &chunk; seeds.seed_len = seed_len; seeds.distance = distance ? distance : seed_len; }

    /** index_reads( seeds ) (reference :1089-1097). */
    readsindex_type index_reads( SeedsRecord const& seeds ) const { return readsindex_type{ &seeds }; }

    typedef std::function< void( output_type const& ) > callback_type;

    /** seeds_on_paths( reads, reads_index, callback ) (reference :1426-1457). */
    void seeds_on_paths( SeedsRecord const& seeds, readsindex_type&, callback_type callback ) const
    { run( seeds, PSIGPU_ON_PATHS, callback ); }

    /** seeds_on_paths( sequence, callback ) (reference :1459-1479): MEM mode -- find_mems
     *  (index_iter.hpp:854-906) with minimum length = seed length, the finder's gocc threshold and
     *  max_mem.  The callback gets psi::Seed<> with match_len and gocc filled; read_id is 0. */
    void seeds_on_paths( std::string const& sequence, callback_type callback ) const
    {
      std::uint64_t off[ 2 ] = { 0, sequence.size() };
      psigpu_mems mems{};
      check( psigpu_find_mems( ctx, sequence.data(), off, 1, seed_len, max_mem, 0, &mems ) );
      output_type h{};
      for ( std::uint64_t i = 0; i < mems.n; ++i ) {
        h.node_id = mems.data[ i ].node_id; h.node_offset = mems.data[ i ].node_offset;
        h.read_id = mems.data[ i ].read_id; h.read_offset = mems.data[ i ].read_offset;
        h.match_len = mems.data[ i ].match_len; h.gocc = mems.data[ i ].gocc;
        callback( h );
      }
      psigpu_free_mems( &mems );
    }

    /** setup_traverser( traverser, reads, reads_index ) (reference :1695-1701). */
    void setup_traverser( traverser_type& traverser, SeedsRecord const& seeds, readsindex_type& ) const
    { traverser.reads = &seeds; }

    /** seeds_off_paths( traverser, callback ) (reference :1703-1722). */
    void seeds_off_paths( traverser_type& traverser, callback_type callback ) const
    {
      if ( traverser.reads == nullptr ) throw std::runtime_error( "setup_traverser() has not been called" );
      run( *traverser.reads, PSIGPU_OFF_PATHS, callback );
    }

    /** seeds_all( reads, reads_index, traverser, callback ) (reference :1724-1732): both phases in
     *  one device call. */
    void seeds_all( SeedsRecord const& seeds, readsindex_type& index, traverser_type& traverser,
                    callback_type callback ) const
    {
      setup_traverser( traverser, seeds, index );
      run( seeds, PSIGPU_ALL, callback );
    }

    /** seeds_all without the per-hit callback: the chunk's sort-unique hits as one array of 32-byte
     *  records in library-owned pinned memory (release with psigpu_free_hits).  No counterpart in
     *  the reference; psikt uses it to write a chunk with one fwrite. */
    psigpu_hits seeds_all_hits( SeedsRecord const& seeds, readsindex_type& index, traverser_type& traverser ) const
    {
      setup_traverser( traverser, seeds, index );
      return find( seeds, PSIGPU_ALL );
    }

    /** The same for the reads [begin, end) of the chunk only: psikt --devices gives every GPU one
     *  contiguous range of each chunk (reads are independent given the index; read ids stay global). */
    psigpu_hits seeds_all_hits( SeedsRecord const& seeds, std::uint64_t begin, std::uint64_t end ) const
    {
      if ( seeds.chunk == nullptr ) throw std::runtime_error( "get_seeds() has not been called" );
      Records const& c = *seeds.chunk;
      if ( begin > end || end > c.size() ) throw std::runtime_error( "read range out of bounds" );
      psigpu_hits hits{};
      if ( c.is_packed ) {
        /* a range of the packed chunk: the word arrays as they are, the range's own offsets (read_off[0] != 0) */
        check( psigpu_find_seeds_packed( ctx, c.packed.data(), c.n_not_acgt ? c.not_acgt.data() : nullptr, c.offsets.data() + begin,
                                         end - begin, seeds.seed_len, seeds.distance, c.get_record_offset() + begin,
                                         PSIGPU_ALL | PSIGPU_SORT_UNIQUE | ( c.uniform() ? PSIGPU_UNIFORM_READS : 0u ), &hits ) );
        return hits;
      }
      std::vector< std::uint64_t > off( end - begin + 1 );
      for ( std::uint64_t i = begin; i <= end; ++i ) off[ i - begin ] = c.offsets[ i ] - c.offsets[ begin ];
      check( psigpu_find_seeds( ctx, c.bases.data() + c.offsets[ begin ], off.data(), end - begin, seeds.seed_len,
                                seeds.distance, c.get_record_offset() + begin, PSIGPU_ALL | PSIGPU_SORT_UNIQUE, &hits ) );
      return hits;
    }

    /** Another finder (another GPU) takes this finder's path index and starting loci: the index is made
     *  or loaded once and copied to every device. */
    void share_path_index( SeedFinder const& other )
    {
      if ( other.pindex.empty() ) throw std::runtime_error( "the other finder has no path index" );
      check( psigpu_load_index( ctx, &other.pindex.view() ) );
      check( psigpu_prepare( ctx, seed_len ) );
      shared_from = &other;
    }

    /** seeds_all with one callback per phase (reference :1734-1743). */
    void seeds_all( SeedsRecord const& seeds, readsindex_type& index, traverser_type& traverser,
                    callback_type callback1, callback_type callback2 ) const
    {
      seeds_on_paths( seeds, index, callback1 );
      setup_traverser( traverser, seeds, index );
      seeds_off_paths( traverser, callback2 );
    }

    /* ---- accessors ------------------------------------------------------------------------ */
    graph_type const* get_graph_ptr() const { return graph_ptr; }
    pathindex_type const& get_pindex() const { return pindex; }
    unsigned int get_seed_len() const { return seed_len; }
    std::vector< Position > get_starting_loci() const
    {
      std::vector< Position > out;
      auto const& v = pindex.view();
      out.reserve( v.n_loci );
      for ( std::uint64_t i = 0; i < v.n_loci; ++i ) {
        Position p;
        p.set_node_id( graph_ptr->view().node_id[ v.loci_node[ i ] ] );
        p.set_offset( v.loci_off[ i ] );
        out.push_back( p );
      }
      return out;
    }
    std::uint64_t get_nof_starting_loci() const { return ( shared_from ? shared_from->pindex : pindex ).view().n_loci; }
    std::uint64_t get_nof_uniq_nodes() const
    {
      auto const& v = pindex.view();
      std::uint64_t n = 0;
      for ( std::uint64_t i = 0; i < v.n_loci; ++i )
        if ( i == 0 || v.loci_node[ i ] != v.loci_node[ i - 1 ] ) ++n;
      return n;
    }
    psigpu_counters get_stats() const { psigpu_counters c{}; psigpu_get_counters( ctx, &c ); return c; }

  private:
    void check( int st ) const
    { if ( st != PSIGPU_OK ) throw std::runtime_error( psigpu_last_error( ctx ) ); }

    void build( unsigned int step_size )
    {
      psigpu_index_opts o{};
      o.seed_len = seed_len; o.n_per_region = pick_n; o.locus_step = step_size;
      o.patched = pick_patched ? 1u : 0u; o.context = pick_context;
      o.build_on_device = (unsigned int)device_ + 1;      /* suffix sorting on the GPU the finder runs on */
      pindex.create( *graph_ptr, o );
      check( psigpu_load_index( ctx, &pindex.view() ) );
      check( psigpu_prepare( ctx, seed_len ) );           /* the query mode's tables: index time, not query time */
      built_step = step_size;
    }

    psigpu_hits find( SeedsRecord const& seeds, unsigned int flags ) const
    {
      if ( seeds.chunk == nullptr ) throw std::runtime_error( "get_seeds() has not been called" );
      Records const& c = *seeds.chunk;
      psigpu_hits hits{};
      if ( c.is_packed )      /* readRecords packed the chunk: a quarter of the bytes on the host link */
        check( psigpu_find_seeds_packed( ctx, c.packed.data(), c.n_not_acgt ? c.not_acgt.data() : nullptr, c.offsets.data(), c.size(),
                                         seeds.seed_len, seeds.distance, c.get_record_offset(),
                                         flags | PSIGPU_SORT_UNIQUE | ( c.uniform() ? PSIGPU_UNIFORM_READS : 0u ), &hits ) );
      else
      check( psigpu_find_seeds( ctx, c.bases.data(), c.offsets.data(), c.size(), seeds.seed_len,
                                seeds.distance, c.get_record_offset(),
                                flags | PSIGPU_SORT_UNIQUE | ( c.uniform() ? PSIGPU_UNIFORM_READS : 0u ), &hits ) );
      return hits;
    }

    /** The hits of one phase through the callback, with Seed::gocc as the reference sets it: on paths the number of
     *  occurrences of the seed's k-mer in the path text (index_iter.hpp:743 -- length( occurrences1 ), from
     *  psigpu_count_occurrences), off paths the number of read positions of the chunk that hold the k-mer
     *  (traverser_bfs.hpp:107 -- length( saPositions ) of the reads index, counted here over the chunk's seeds).  Both
     *  phases asked for at once: on paths first, then off paths, as seeds_all does (reference :1724-1732). */
    void run( SeedsRecord const& seeds, unsigned int flags, callback_type const& callback ) const
    {
      if ( ( flags & PSIGPU_ALL ) == PSIGPU_ALL ) {
        run( seeds, ( flags & ~PSIGPU_ALL ) | PSIGPU_ON_PATHS, callback );
        run( seeds, ( flags & ~PSIGPU_ALL ) | PSIGPU_OFF_PATHS, callback );
        return;
      }
      psigpu_hits hits = find( seeds, flags );
      if ( hits.n == 0 ) { psigpu_free_hits( &hits ); return; }
      Records const& c = *seeds.chunk;
      std::uint64_t const k = seeds.seed_len, d = seeds.distance;
      /* first seed of every read (offsets 0, d, 2d ... while i < len - k + 1: sequence.hpp:1711-1714) */
      std::vector< std::uint64_t > first( c.size() + 1, 0 );
      for ( std::size_t r = 0; r < c.size(); ++r ) {
        std::uint64_t const len = c.offsets[ r + 1 ] - c.offsets[ r ];
        first[ r + 1 ] = first[ r ] + ( len >= k ? ( len - k ) / d + 1 : 0 );
      }
      std::vector< std::uint32_t > gocc( first.back(), 0 );
      try {
        if ( flags & PSIGPU_ON_PATHS ) {
          if ( !gocc.empty() )
            check( psigpu_count_occurrences( ctx, c.bases.data(), c.offsets.data(), c.size(), seeds.seed_len, seeds.distance,
                                             gocc.data(), gocc.size() ) );
        } else {
          std::unordered_map< std::string, std::uint32_t > occ;
          occ.reserve( gocc.size() * 2 );
          auto kmer = [ & ]( std::size_t r, std::uint64_t j ) {
            std::string s( c.bases.data() + c.offsets[ r ] + j * d, k );
            for ( auto& ch : s ) ch = static_cast< char >( std::toupper( static_cast< unsigned char >( ch ) ) );
            return s;
          };
          for ( std::size_t r = 0; r < c.size(); ++r )
            for ( std::uint64_t j = 0; j < first[ r + 1 ] - first[ r ]; ++j ) ++occ[ kmer( r, j ) ];
          for ( std::size_t r = 0; r < c.size(); ++r )
            for ( std::uint64_t j = 0; j < first[ r + 1 ] - first[ r ]; ++j ) gocc[ first[ r ] + j ] = occ[ kmer( r, j ) ];
        }
      } catch ( ... ) { psigpu_free_hits( &hits ); throw; }
      output_type h{};
      h.match_len = seeds.seed_len;
      for ( std::uint64_t i = 0; i < hits.n; ++i ) {
        h.node_id = hits.data[ i ].node_id;
        h.node_offset = hits.data[ i ].node_offset;
        h.read_id = hits.data[ i ].read_id;
        h.read_offset = hits.data[ i ].read_offset;
        std::uint64_t const r = h.read_id - c.get_record_offset();
        h.gocc = gocc[ first[ r ] + h.read_offset / d ];
        callback( h );
      }
      psigpu_free_hits( &hits );
    }

    graph_type const* graph_ptr;
    unsigned int seed_len;
    unsigned int gocc_threshold;
    unsigned int max_mem = 0;
    unsigned int pick_n = 0, pick_context = 0, built_step = 0;
    bool pick_patched = true;
    int device_ = 0;
    SeedFinder const* shared_from = nullptr;
    pathindex_type pindex;
    psigpu_ctx* ctx = nullptr;
  };
}  /* --- end of namespace psi --- */

#endif
