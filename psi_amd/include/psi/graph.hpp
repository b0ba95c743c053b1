// psi::Graph / psi::Position -- host-side stand-ins for the gum::SeqGraph the reference
// loads (src/psikt.cpp:249-251) and for psi::Position<> (include/psi/graph.hpp:33-82).
// Thin RAII wrappers over the C ABI (include/psi_gpu.h); no template machinery.
#ifndef PSI_AMD_GRAPH_HPP__
#define PSI_AMD_GRAPH_HPP__

#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "psi_gpu.h"

namespace psi {
  struct Position {
    std::uint64_t id = 0;
    std::uint64_t off = 0;
    std::uint64_t node_id() const { return id; }
    std::uint64_t offset() const { return off; }
    void set_node_id( std::uint64_t v ) { id = v; }
    void set_offset( std::uint64_t v ) { off = v; }
  };

  class Graph {
  public:
    typedef std::uint64_t id_type;
    typedef std::uint64_t offset_type;
    typedef std::uint64_t rank_type;

    Graph() = default;
    explicit Graph( std::string const& path, bool follow_reversing = false ) { load( path, follow_reversing ); }
    Graph( Graph const& ) = delete;
    Graph& operator=( Graph const& ) = delete;
    ~Graph() { psigpu_graph_free( h_ ); }

    /** gum::util::load(graph, path, ...): .gfa or .vg; throws std::runtime_error. */
    void load( std::string const& path, bool follow_reversing = false )
    {
      int st = 0;
      psigpu_graph* g = psigpu_graph_load_opts( path.c_str(), follow_reversing ? PSIGPU_GRAPH_FOLLOW_REVERSING : 0u, &st );
      if ( g == nullptr )
        throw std::runtime_error( "cannot load graph '" + path + "': " + psigpu_host_last_error() );
      psigpu_graph_free( h_ );
      h_ = g;
      psigpu_graph_view_get( h_, &view_ );
    }

    std::uint64_t get_node_count() const { return view_.n_nodes; }
    std::uint64_t get_edge_count() const { return psigpu_graph_edge_count( h_ ); }
    std::uint64_t get_path_count() const { return psigpu_graph_path_count( h_ ); }
    id_type rank_to_id( rank_type rank ) const { return view_.node_id[ rank - 1 ]; }   /* ranks are 1-based as in gum */
    /** gum's id_to_rank: 0 when there is no such node. */
    rank_type id_to_rank( id_type id ) const
    {
      std::uint64_t n = view_.n_nodes;
      if ( n == 0 ) return 0;
      /* ids of vg / GFA files are almost always first id + rank */
      std::uint64_t guess = id - view_.node_id[ 0 ];
      if ( id >= view_.node_id[ 0 ] && guess < n && view_.node_id[ guess ] == id ) return guess + 1;
      if ( id_rank_.empty() ) {
        id_rank_.reserve( n * 2 );
        for ( std::uint64_t r = 0; r < n; ++r ) id_rank_.emplace( view_.node_id[ r ], r + 1 );
      }
      auto it = id_rank_.find( id );
      return it == id_rank_.end() ? 0 : it->second;
    }
    bool has_node( id_type id ) const { return id_to_rank( id ) != 0; }
    offset_type node_length( rank_type rank ) const
    { return view_.label_off[ rank ] - view_.label_off[ rank - 1 ]; }
    std::string node_sequence( rank_type rank ) const
    { return std::string( view_.labels + view_.label_off[ rank - 1 ], node_length( rank ) ); }

    psigpu_graph const* handle() const { return h_; }
    psigpu_graph_view const& view() const { return view_; }
  private:
    psigpu_graph* h_ = nullptr;
    psigpu_graph_view view_{};
    mutable std::unordered_map< id_type, rank_type > id_rank_;
  };
}  /* --- end of namespace psi --- */

#endif
