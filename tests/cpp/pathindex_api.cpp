// Test driver for the psi::Path / psi::PathIndex shim (psi_amd/include/psi/pathindex.hpp): prints what
// reference test/src/test_pathindex.cpp:94-288 asserts, one "key value" line each, for the test to compare
// with the reference's numbers.  usage: pathindex_api small/x.gfa
#include <cstdio>
#include <string>
#include <psi/pathindex.hpp>

using namespace psi;

static std::string text_of( PathIndex const& px )
{
  static const char sym[] = { '#', '$', 'A', 'C', 'G', 'T' };
  const uint8_t* t = psigpu_index_text( px.handle() );
  std::string s;
  for ( uint64_t i = 0; i < px.view().text_len; ++i ) s += sym[ t[ i ] ];
  return s;
}

int main( int argc, char** argv )
{
  if ( argc < 2 ) return 2;
  Graph graph( argv[ 1 ] );
  {
    PathIndex pindex( graph );
    Path path( &graph, { 205, 207, 209, 210 } );
    printf( "seqlen %llu\nlength %zu\n", (unsigned long long)path.get_sequence_len(), path.size() );
    pindex.add_path( std::move( path ) );
    for ( unsigned pos : { 0u, 14u, 26u, 27u, 30u, 51u, 52u, 53u } )
      printf( "fwd %u %llu %llu\n", pos, (unsigned long long)position_to_id( pindex, { 0, pos } ),
              (unsigned long long)position_to_offset( pindex, { 0, pos } ) );
    printf( "plen %llu\n", (unsigned long long)length( pindex ) );
  }
  {
    uint64_t context = 10;
    PathIndex pindex( graph, context, true );
    pindex.add_path( Path( &graph, { 205, 207, 209, 210 }, context - 1, context - 1 ) );
    pindex.add_path( Path( &graph, { 187, 189, 191, 193, 194, 195, 197 }, context - 1, context - 1 ) );
    pindex.add_path( Path( &graph, { 167, 168, 171, 172, 174 }, context - 1, context - 1 ) );
    pindex.create_index( 10, 1, -1, true );
    printf( "text %s\n", text_of( pindex ).c_str() );
    for ( auto const& p : pindex.get_paths_set() ) printf( "pathseq %s\n", p.sequence().c_str() );
    printf( "context %llu\n", (unsigned long long)pindex.get_context() );
    for ( unsigned pos : { 10u, 11u, 12u, 20u } )
      printf( "fwd2 %u %llu %llu\n", pos, (unsigned long long)position_to_id( pindex, { 2, pos } ),
              (unsigned long long)position_to_offset( pindex, { 2, pos } ) );
    for ( unsigned pos : { 0u, 1u, 2u, 20u, 26u, 27u, 29u, 35u } )
      printf( "rev0 %u %llu %llu\n", pos, (unsigned long long)position_to_id( pindex, { 0, pos }, Reversed() ),
              (unsigned long long)position_to_offset( pindex, { 0, pos }, Reversed() ) );
    printf( "covered %d %d %d\n", (int)covered_by( std::vector< Path::id_type >{ 207, 209 }, pindex ),
            (int)covered_by( std::vector< Path::id_type >{ 207, 210 }, pindex ),
            (int)covered_by( Path( &graph, { 168, 171, 172 } ), pindex ) );
    /* round trip through a file keeps paths and trimming */
    if ( argc > 2 ) {
      if ( !pindex.serialize( argv[ 2 ] ) ) return 3;
      PathIndex loaded( graph );
      if ( !loaded.load( argv[ 2 ] ) ) return 4;
      printf( "loaded %llu", (unsigned long long)loaded.size() );
      for ( auto const& p : loaded.get_paths_set() ) printf( " %s", p.sequence().c_str() );
      printf( "\n" );
    }
  }
  return 0;
}
