// The header API with per-hit callbacks (psi::SeedFinder::seeds_all, two callbacks: reference seed_finder.hpp:1734-1743)
// on a graph and a reads file: one line per hit -- phase, node id, node offset, read id, read offset, match_len, gocc.
// Driven by tests/test_gpu_parity.py::test_header_api_callbacks_carry_gocc.   usage: seedfinder_api graph reads k n_paths step
#include <cstdio>
#include <cstdlib>
#include <string>

#include <psi/seed_finder.hpp>

int main( int argc, char** argv )
{
  if ( argc < 6 ) return 2;
  try {
    psi::Graph graph( argv[ 1 ], false );
    unsigned const k = std::strtoul( argv[ 3 ], nullptr, 10 ), n = std::strtoul( argv[ 4 ], nullptr, 10 );
    unsigned const step = std::strtoul( argv[ 5 ], nullptr, 10 );
    psi::SeedFinder< psi::NoStats > finder( graph, k );
    finder.create_path_index( n, false );
    for ( auto const& p : finder.get_pindex().get_paths_set() ) std::printf( "pathseq %s\n", p.sequence().c_str() );
    psi::SeqStreamIn in( argv[ 2 ] );
    auto chunk = finder.create_readrecord();
    if ( !psi::readRecords( chunk, in, 0 ) ) return 3;
    psi::SeedsRecord seeds;
    finder.get_seeds( seeds, chunk, step );
    auto index = finder.index_reads( seeds );
    auto traverser = finder.create_traverser();
    auto print = []( char const* phase ) {
      return [ phase ]( psi::Seed<> const& h ) {
        std::printf( "%s %llu %llu %llu %llu %llu %llu\n", phase, (unsigned long long)h.node_id, (unsigned long long)h.node_offset,
                     (unsigned long long)h.read_id, (unsigned long long)h.read_offset, (unsigned long long)h.match_len,
                     (unsigned long long)h.gocc );
      };
    };
    finder.seeds_all( seeds, index, traverser, print( "on" ), print( "off" ) );
  } catch ( std::exception const& e ) { std::fprintf( stderr, "error: %s\n", e.what() ); return 1; }
  return 0;
}
