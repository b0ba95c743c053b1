// psi::Path / psi::PathIndex -- the indexed path set + FM-index + starting loci.
// Mirrors the public surface of reference include/psi/pathindex.hpp:40-333 (ctor( graph[, context][,
// lazy] ), get_paths_set, get_context / set_context, add_path / push_back, size, reserve,
// create_index, load, serialize, clear; free functions length, position_to_id / position_to_offset
// :360-416, covered_by :430-435) and of the part of Path (path_base.hpp) those functions need.  The data
// itself is the device layout built by libpsi_gpu.so (include/psi_gpu.h).  Template parameters of the
// reference (graph, text, index spec) do not apply; the sequence direction is a constructor argument
// of the position functions' callers (the device index is over the forward text; `Reversed` positions
// are mapped as the reference maps them, pathindex.hpp:378-387).
#ifndef PSI_AMD_PATHINDEX_HPP__
#define PSI_AMD_PATHINDEX_HPP__

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "graph.hpp"
#include "psi_gpu.h"

namespace psi {
  struct Forward {};
  struct Reversed {};

  /** A path by external node ids, optionally trimmed at both ends the way the reference's Path is
   *  (path_base.hpp:113-114): `left` = number of bases of the FIRST node that belong to the path
   *  (its suffix), `right` = number of bases of the LAST node that belong to it (its prefix);
   *  0 = the whole node. */
  class Path {
  public:
    typedef Graph graph_type;
    typedef std::uint64_t id_type;
    typedef std::uint64_t offset_type;

    explicit Path( graph_type const* g = nullptr ) : graph_ptr( g ) {}
    Path( graph_type const* g, std::vector< id_type > p, offset_type l = 0, offset_type r = 0 )
      : graph_ptr( g ), nodes( std::move( p ) )
    {
      if ( !nodes.empty() ) {
        left = ( l == 0 || l >= node_len( nodes.front() ) ) ? 0 : l;
        right = ( r == 0 || r >= node_len( nodes.back() ) ) ? 0 : r;
      }
    }
    graph_type const* get_graph_ptr() const { return graph_ptr; }
    std::vector< id_type > const& get_nodes() const { return nodes; }
    std::size_t size() const { return nodes.size(); }
    bool empty() const { return nodes.empty(); }
    void push_back( id_type id ) { nodes.push_back( id ); right = 0; }
    /** Path::get_head_offset (path_base.hpp:240-246). */
    offset_type get_head_offset() const { return left == 0 ? 0 : node_len( nodes.front() ) - left; }
    offset_type get_seqlen_head() const { return left ? left : node_len( nodes.front() ); }
    offset_type get_seqlen_tail() const { return right ? right : node_len( nodes.back() ); }
    offset_type get_left() const { return left; }
    offset_type get_right() const { return right; }
    offset_type get_sequence_len() const
    {
      if ( nodes.empty() ) return 0;
      if ( nodes.size() == 1 ) {
        /* one node: bases [head offset, right or end) */
        offset_type end = right ? right : node_len( nodes.front() );
        return end - get_head_offset();
      }
      offset_type n = get_seqlen_head() + get_seqlen_tail();
      for ( std::size_t i = 1; i + 1 < nodes.size(); ++i ) n += node_len( nodes[ i ] );
      return n;
    }
    std::string sequence() const
    {
      std::string s;
      for ( std::size_t i = 0; i < nodes.size(); ++i ) {
        std::string l = graph_ptr->node_sequence( graph_ptr->id_to_rank( nodes[ i ] ) );
        std::size_t b = ( i == 0 ) ? get_head_offset() : 0;
        std::size_t e = ( i + 1 == nodes.size() && right ) ? right : l.size();
        s += l.substr( b, e - b );
      }
      return s;
    }
    /** position in the path's (forward) sequence -> index of the node holding it and the offset
     *  inside that node (position_to_id / position_to_offset, path_interface.hpp:172-197: the head
     *  offset is added when the position falls into the first node). */
    std::pair< std::size_t, offset_type > locate( offset_type pos ) const
    {
      offset_type acc = 0;
      for ( std::size_t i = 0; i < nodes.size(); ++i ) {
        offset_type len = ( i == 0 ) ? get_seqlen_head()
                        : ( i + 1 == nodes.size() ) ? get_seqlen_tail() : node_len( nodes[ i ] );
        if ( nodes.size() == 1 ) len = get_sequence_len();
        if ( pos < acc + len ) return { i, pos - acc + ( i == 0 ? get_head_offset() : 0 ) };
        acc += len;
      }
      throw std::runtime_error( "position out of range" );
    }
  private:
    offset_type node_len( id_type id ) const { return graph_ptr->node_length( graph_ptr->id_to_rank( id ) ); }
    graph_type const* graph_ptr;
    std::vector< id_type > nodes;
    offset_type left = 0, right = 0;
  };

  inline Path::offset_type position_to_offset( Path const& p, Path::offset_type pos ) { return p.locate( pos ).second; }
  inline Path::id_type position_to_id( Path const& p, Path::offset_type pos ) { return p.get_nodes()[ p.locate( pos ).first ]; }

  /** ( string id, offset ) in the string set of the index: seqan2::SAValue of the reference. */
  struct PathPosition { std::uint64_t i1; std::uint64_t i2; };

  class PathIndex {
  public:
    typedef Graph graph_type;
    typedef Path value_type;
    typedef std::vector< Path > container_type;
    typedef std::uint64_t size_type;
    typedef std::uint64_t context_type;

    PathIndex() = default;
    explicit PathIndex( graph_type const& graph, context_type ct = 0, bool l = false )
      : graph_ptr( &graph ), context( ct ), lazy_mode( l ) {}
    PathIndex( graph_type const& graph, bool lazy ) : graph_ptr( &graph ), context( 0 ), lazy_mode( lazy ) {}
    PathIndex( PathIndex const& ) = delete;
    PathIndex& operator=( PathIndex const& ) = delete;
    ~PathIndex() { psigpu_index_free( h_ ); }

    void set_graph( graph_type const& graph ) { graph_ptr = &graph; }
    container_type& get_paths_set() { return paths_set; }
    container_type const& get_paths_set() const { return paths_set; }
    context_type get_context() const { return h_ ? view_.context : context; }
    void set_context( context_type value ) { context = value; }

    /** add_path / push_back (reference :153-190): the path joins the set; the device index is made by
     *  create_index() (the reference re-creates its SeqAn index per added path unless in lazy mode). */
    void add_path( value_type new_path ) { paths_set.push_back( std::move( new_path ) ); }
    void push_back( value_type new_path ) { add_path( std::move( new_path ) ); }
    size_type size() const { return paths_set.size(); }
    void reserve( size_type n ) { paths_set.reserve( n ); }
    bool empty() const { return h_ == nullptr; }

    /** create_index (reference :235-243) + the starting loci for `seed_len` (the device layout holds both:
     *  SeedFinder::index_paths and add_uncovered_loci in one step).  `device` < 0: suffix sorting on the host. */
    void create_index( unsigned int seed_len, unsigned int step_size = 1, int device = -1, bool keep_text = false )
    {
      if ( graph_ptr == nullptr ) throw std::runtime_error( "PathIndex has no graph" );
      std::vector< std::uint64_t > off{ 0 };
      std::vector< std::uint32_t > nodes, head, tail;
      for ( auto const& p : paths_set ) {
        for ( auto id : p.get_nodes() ) nodes.push_back( static_cast< std::uint32_t >( graph_ptr->id_to_rank( id ) - 1 ) );
        off.push_back( nodes.size() );
        head.push_back( static_cast< std::uint32_t >( p.empty() ? 0 : p.get_head_offset() ) );
        /* a one-node path's `right` counts from the node start, as the tail length does */
        tail.push_back( static_cast< std::uint32_t >( p.get_right() ) );
      }
      psigpu_index_opts o{};
      o.seed_len = seed_len; o.locus_step = step_size; o.context = static_cast< unsigned int >( context );
      o.build_on_device = device < 0 ? 0u : static_cast< unsigned int >( device ) + 1u;
      o.keep_text_sa = keep_text;
      int st = 0;
      psigpu_index* x = psigpu_index_build_patches( graph_ptr->handle(), &o, paths_set.size(), off.data(), nodes.data(),
                                                    head.data(), tail.data(), &st );
      if ( x == nullptr ) throw std::runtime_error( psigpu_host_last_error() );
      reset( x );
    }

    /** SeedFinder::pick_paths + index_paths + add_uncovered_loci in one call
     *  (seed_finder.hpp:1138-1176, :1481-1541): the paths are drawn by the library. */
    void create( graph_type const& graph, psigpu_index_opts const& opts )
    {
      graph_ptr = &graph;
      int st = 0;
      psigpu_index* x = psigpu_index_build( graph.handle(), &opts, &st );
      if ( x == nullptr ) throw std::runtime_error( psigpu_host_last_error() );
      reset( x );
      sync_paths();
    }

    /** PathIndex::load( prefix ) (reference pathindex.hpp:109-123). */
    bool load( std::string const& prefix )
    {
      if ( prefix.empty() ) return false;
      int st = 0;
      psigpu_index* x = psigpu_index_load( prefix.c_str(), &st );
      if ( x == nullptr ) return false;
      reset( x );
      if ( graph_ptr ) sync_paths();
      return true;
    }

    /** PathIndex::load( prefix ) for an index the REFERENCE wrote: `<prefix>_paths` (save_paths_set,
     *  reference pathindex.hpp:315-332) holds the paths and their trims in sdsl's enc_vector / bit_vector
     *  layouts; the FM index over them is rebuilt (the `<prefix>` file, an sdsl::csa_wt, is not read). */
    bool load_reference( std::string const& prefix, psigpu_index_opts const& opts )
    {
      if ( prefix.empty() || graph_ptr == nullptr ) return false;
      int st = 0;
      std::uint64_t ctx_in_file = 0;
      psigpu_index* x = psigpu_index_from_reference_paths( graph_ptr->handle(), &opts, ( prefix + "_paths" ).c_str(),
                                                           &ctx_in_file, nullptr, &st );
      if ( x == nullptr ) return false;
      if ( context != 0 && context != ctx_in_file ) { psigpu_index_free( x ); return false; }      /* load_paths_set :288-289 */
      context = ctx_in_file;
      reset( x );
      sync_paths();
      return true;
    }

    /** PathIndex::serialize( prefix ) (reference pathindex.hpp:135-143). */
    bool serialize( std::string const& prefix ) const
    {
      if ( prefix.empty() || h_ == nullptr ) return false;
      return psigpu_index_save( h_, prefix.c_str() ) == PSIGPU_OK;
    }

    /** `<prefix>_loci_e<E>l<K>` in the reference's format (SeedFinder::save_starts / open_starts,
     *  seed_finder.hpp:1640-1679). */
    bool save_loci( std::string const& prefix ) const
    { return h_ && graph_ptr && psigpu_loci_save( h_, graph_ptr->handle(), prefix.c_str() ) == PSIGPU_OK; }
    bool load_loci( std::string const& prefix, unsigned int step_size )
    {
      if ( !h_ || !graph_ptr || psigpu_loci_load( h_, graph_ptr->handle(), prefix.c_str(), step_size ) != PSIGPU_OK ) return false;
      psigpu_index_view_get( h_, &view_ );
      return true;
    }

    /** The starting loci recomputed for another locus step from the index's own paths and trims
     *  (SeedFinder::add_uncovered_loci( step ), seed_finder.hpp:1481-1541). */
    bool set_locus_step( unsigned int step_size )
    {
      if ( !h_ || !graph_ptr || psigpu_index_set_locus_step( h_, graph_ptr->handle(), step_size ) != PSIGPU_OK ) return false;
      psigpu_index_view_get( h_, &view_ );
      return true;
    }

    psigpu_index_view const& view() const { return view_; }
    psigpu_index const* handle() const { return h_; }
    void clear() { reset( nullptr ); paths_set.clear(); }
  private:
    void reset( psigpu_index* x )
    {
      psigpu_index_free( h_ );
      h_ = x;
      view_ = psigpu_index_view{};
      if ( h_ ) psigpu_index_view_get( h_, &view_ );
    }
    /* paths of a built / loaded index -> paths_set */
    void sync_paths()
    {
      paths_set.clear();
      std::uint64_t n = psigpu_index_path_count( h_ );
      for ( std::uint64_t i = 0; i < n; ++i ) {
        std::uint64_t len = psigpu_index_path( h_, i, nullptr, 0 );
        std::vector< std::uint32_t > ranks( len );
        psigpu_index_path( h_, i, ranks.data(), len );
        std::vector< Path::id_type > ids( len );
        for ( std::uint64_t j = 0; j < len; ++j ) ids[ j ] = graph_ptr->rank_to_id( ranks[ j ] + 1 );
        std::uint32_t head = 0, tail = 0;
        psigpu_index_path_trim( h_, i, &head, &tail );
        Path::offset_type left = 0;
        if ( head && len ) left = graph_ptr->node_length( ranks[ 0 ] + 1 ) - head;
        paths_set.emplace_back( graph_ptr, std::move( ids ), left, tail );
      }
    }
    graph_type const* graph_ptr = nullptr;
    container_type paths_set;
    context_type context = 0;
    bool lazy_mode = false;
    psigpu_index* h_ = nullptr;
    psigpu_index_view view_{};
  };

  /* PathIndex interface functions (reference pathindex.hpp:352-435) ------------------------------ */
  inline PathIndex::size_type length( PathIndex const& pindex ) { return pindex.size(); }

  inline Path::offset_type position_to_offset( PathIndex const& pindex, PathPosition pos, Forward = {} )
  { return position_to_offset( pindex.get_paths_set().at( pos.i1 ), pos.i2 ); }
  inline Path::id_type position_to_id( PathIndex const& pindex, PathPosition pos, Forward = {} )
  { return position_to_id( pindex.get_paths_set().at( pos.i1 ), pos.i2 ); }
  /** Reversed text: `pos` is the END position of the occurrence in the reversed string
   *  (reference :366-387). */
  inline Path::offset_type position_to_offset( PathIndex const& pindex, PathPosition pos, Reversed )
  {
    auto const& p = pindex.get_paths_set().at( pos.i1 );
    return position_to_offset( p, p.get_sequence_len() - pos.i2 - 1 );
  }
  inline Path::id_type position_to_id( PathIndex const& pindex, PathPosition pos, Reversed )
  {
    auto const& p = pindex.get_paths_set().at( pos.i1 );
    return position_to_id( p, p.get_sequence_len() - pos.i2 - 1 );
  }

  /** covered_by( path, pindex ) (reference :428-435): is the node sequence a contiguous run of an
   *  indexed path? */
  inline bool covered_by( std::vector< Path::id_type > const& nodes, PathIndex const& pindex )
  {
    if ( nodes.empty() ) return false;
    for ( auto const& p : pindex.get_paths_set() ) {
      auto const& pn = p.get_nodes();
      if ( std::search( pn.begin(), pn.end(), nodes.begin(), nodes.end() ) != pn.end() ) return true;
    }
    return false;
  }
  inline bool covered_by( Path const& path, PathIndex const& pindex ) { return covered_by( path.get_nodes(), pindex ); }
}  /* --- end of namespace psi --- */

#endif
