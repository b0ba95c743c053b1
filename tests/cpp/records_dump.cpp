// Test driver: reads a FASTQ / FASTA / plain-text (optionally gzip'd) file through
// psi::SeqStreamIn + psi::readRecords in chunks and prints "offset<TAB>name<TAB>sequence" per record.
#include <cstdio>
#include <cstdlib>
#include <psi/sequence.hpp>

int main( int argc, char** argv )
{
  if ( argc < 3 ) return 2;
  psi::SeqStreamIn in( argv[ 1 ] );
  unsigned long chunk = strtoul( argv[ 2 ], nullptr, 10 );
  psi::Records rec;
  while ( psi::readRecords( rec, in, chunk ) ) {
    printf( "#chunk %llu %zu %llu\n", (unsigned long long)rec.get_record_offset(), rec.size(),
            (unsigned long long)rec.length_sum() );
    for ( std::size_t i = 0; i < rec.size(); ++i )
      printf( "%llu\t%s\t%s\n", (unsigned long long)( rec.get_record_offset() + i ), rec.name[ i ].c_str(),
              rec[ i ].c_str() );
  }
  return 0;
}
