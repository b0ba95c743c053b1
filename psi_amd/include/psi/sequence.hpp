// psi::Records + readRecords -- a chunk of reads and its reader.
// Mirrors the parts of reference include/psi/sequence.hpp the seed-finding loop touches:
// Records (name/str, record offset :1130-1294), readRecords (:1590-1624).  FASTQ (optionally
// gzip'd) or one-sequence-per-line text; kseq++ / SeqAn are not used.
//
// The bases of a chunk live in page-locked memory (psigpu_host_alloc): the copy engine of the GPU
// reads them in place, psigpu_find_seeds does not stage them.
#ifndef PSI_AMD_SEQUENCE_HPP__
#define PSI_AMD_SEQUENCE_HPP__

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "psi_gpu.h"

namespace psi {
  /** Grow-only byte buffer in pinned host memory (plain malloc when there is no GPU runtime:
   *  the finder itself still refuses to run without one). */
  class PinnedChars {
  public:
    PinnedChars() = default;
    PinnedChars( PinnedChars const& ) = delete;
    PinnedChars& operator=( PinnedChars const& ) = delete;
    ~PinnedChars() { release(); }
    char const* data() const { return p_; }
    char* data() { return p_; }
    std::size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    void append( char const* s, std::size_t len )
    {
      if ( n_ + len > cap_ ) grow( n_ + len );
      std::memcpy( p_ + n_, s, len );
      n_ += len;
    }
    PinnedChars& operator+=( std::string const& s ) { append( s.data(), s.size() ); return *this; }
    /** n bytes, contents unspecified (the parallel reader fills them): one allocation, no doubling */
    void resize_uninitialized( std::size_t n ) { if ( n > cap_ ) grow_exact( n ); n_ = n; }
    std::string substr( std::size_t pos, std::size_t len ) const { return std::string( p_ + pos, len ); }
  private:
    void grow_exact( std::size_t want )
    {
      std::size_t const keep = n_;
      n_ = 0;                                   /* (nothing to carry over) */
      grow( want + want / 16 + 4096, true );
      n_ = keep;
    }
    void grow( std::size_t want, bool exact = false )
    {
      std::size_t cap = cap_ ? cap_ : ( 1u << 20 );
      while ( cap < want ) cap *= 2;
      if ( exact ) cap = want;
      bool pinned = true;
      char* q = static_cast< char* >( psigpu_host_alloc( cap ) );
      if ( q == nullptr ) { q = static_cast< char* >( std::malloc( cap ) ); pinned = false; }
      if ( q == nullptr ) throw std::bad_alloc();
      if ( n_ ) std::memcpy( q, p_, n_ );
      release();
      p_ = q; cap_ = cap; pinned_ = pinned;
    }
    void release()
    {
      if ( p_ == nullptr ) return;
      if ( pinned_ ) psigpu_host_free( p_ ); else std::free( p_ );
      p_ = nullptr; cap_ = 0;
    }
    char* p_ = nullptr;
    std::size_t n_ = 0, cap_ = 0;
    bool pinned_ = false;
  };

  /** Grow-only array of 64-bit words in pinned host memory: the 2-bit form of a chunk's reads. */
  class PinnedWords {
  public:
    PinnedWords() = default;
    PinnedWords( PinnedWords const& ) = delete;
    PinnedWords& operator=( PinnedWords const& ) = delete;
    ~PinnedWords() { release(); }
    std::uint64_t const* data() const { return p_; }
    std::uint64_t* data() { return p_; }
    std::size_t size() const { return n_; }
    /** n words, all zero */
    void assign_zero( std::size_t n )
    {
      if ( n > cap_ ) {
        release();
        std::size_t cap = n + n / 8 + 64;
        p_ = static_cast< std::uint64_t* >( psigpu_host_alloc( cap * 8 ) );
        pinned_ = p_ != nullptr;
        if ( p_ == nullptr ) p_ = static_cast< std::uint64_t* >( std::malloc( cap * 8 ) );
        if ( p_ == nullptr ) throw std::bad_alloc();
        cap_ = cap;
      }
      n_ = n;
      if ( n ) std::memset( p_, 0, n * 8 );
    }
  private:
    void release()
    {
      if ( p_ == nullptr ) return;
      if ( pinned_ ) psigpu_host_free( p_ ); else std::free( p_ );
      p_ = nullptr; cap_ = 0; n_ = 0;
    }
    std::uint64_t* p_ = nullptr;
    std::size_t n_ = 0, cap_ = 0;
    bool pinned_ = false;
  };

  /** A set of reads stored back to back (what the device consumes) plus their names. */
  class Records {
  public:
    std::vector< std::string > name;
    PinnedChars bases;                         /**< concatenated sequences */
    std::vector< std::uint64_t > offsets{ 0 }; /**< size()+1 offsets into bases */
    /** The same bases at 2 bits each + one "not ACGT" bit per base (layout: psigpu_find_seeds_packed): what crosses
     *  the host link.  Made by pack() once the chunk is complete (readRecords does); the reference keeps a byte per
     *  base (seqan2::Dna5QString, sequence.hpp:1130-1294). */
    PinnedWords packed, not_acgt;
    std::uint64_t n_not_acgt = 0;
    bool is_packed = false;

    void pack( unsigned threads = 0 )
    {
      std::uint64_t const n = bases.size();
      packed.assign_zero( ( n + 31 ) / 32 + 2 );
      not_acgt.assign_zero( ( n + 63 ) / 64 + 2 );
      n_not_acgt = 0;
      if ( threads == 0 ) threads = std::max( 1u, std::min( 8u, std::thread::hardware_concurrency() / 2 ) );
      std::uint64_t piece = ( ( n + threads - 1 ) / threads + 63 ) / 64 * 64;     /* whole 64-base blocks per thread */
      if ( piece < ( 1u << 20 ) ) piece = 1u << 20;
      std::vector< std::thread > th;
      std::vector< std::uint64_t > bad( ( n + piece - 1 ) / piece + 1, 0 );
      std::size_t j = 0;
      for ( std::uint64_t a = 0; a < n; a += piece, ++j ) {
        std::uint64_t const len = std::min( piece, n - a );
        auto job = [ this, a, len, j, &bad ] { bad[ j ] = psigpu_pack_reads( bases.data() + a, a, len, packed.data(), not_acgt.data() ); };
        if ( a + piece < n ) th.emplace_back( job ); else job();
      }
      for ( auto& t : th ) t.join();
      for ( auto b : bad ) n_not_acgt += b;
      is_packed = true;
    }

    std::size_t size() const { return offsets.size() - 1; }
    std::uint64_t length_sum() const { return bases.size(); }
    std::uint64_t get_record_offset() const { return rec_offset; }
    void set_record_offset( std::uint64_t v ) { rec_offset = v; }
    void clear() { name.clear(); bases.clear(); offsets.assign( 1, 0 ); rec_offset = 0; is_packed = false; n_not_acgt = 0; }
    /** false: readRecords leaves `name` empty (psikt never looks at a read's name; a million std::strings per chunk are
     *  a fifth of its parsing time) */
    bool keep_names = true;
    /** (the parallel reader: `offsets` and `bases` were filled in bulk) */
    void set_bulk_lengths( std::size_t first, bool same ) { first_len = first; same_len = same; is_packed = false; }
    void push_back( std::string const& n, std::string const& s )
    {
      push_back( n.data(), n.size(), s.data(), s.size() );
    }
    void push_back( char const* n, std::size_t nlen, char const* s, std::size_t slen )
    {
      if ( offsets.size() == 1 ) { first_len = slen; same_len = true; }
      else if ( slen != first_len ) same_len = false;
      name.emplace_back( n, nlen );
      bases.append( s, slen );
      offsets.push_back( bases.size() );
      is_packed = false;
    }
    /** every read of the chunk has the same length (PSIGPU_UNIFORM_READS: the device then skips the scan over the reads) */
    bool uniform() const { return size() != 0 && same_len; }
    std::string operator[]( std::size_t i ) const
    { return bases.substr( offsets[ i ], offsets[ i + 1 ] - offsets[ i ] ); }
  private:
    std::uint64_t rec_offset = 0;
    std::size_t first_len = 0;
    bool same_len = false;
  };

  /** Sequence input stream: counts the records handed out so far (kseq++'s `counts()`).
   *  The file is inflated in 4-MiB blocks and lines are cut in place. */
  class SeqStreamIn {
  public:
    explicit SeqStreamIn( std::string const& path ) : buf_( BLOCK + 1 )
    {
      gz_ = gzopen( path.c_str(), "rb" );
      if ( gz_ == nullptr ) throw std::runtime_error( "cannot open file '" + path + "'" );
      gzbuffer( gz_, 1 << 20 );
      /* A plain (not gzip'd) FASTQ file of four-line records is mapped and parsed by several threads (round 6: the serial
       * parser below reads 2-4 M reads/s, a tenth of what the device answers); anything else -- gzip, FASTA, one sequence
       * per line, a record the fast path does not recognise -- goes through the serial parser, from the same byte on. */
      int fd = ::open( path.c_str(), O_RDONLY );
      if ( fd >= 0 ) {
        struct stat sb;
        unsigned char head[ 2 ] = { 0, 0 };
        if ( fstat( fd, &sb ) == 0 && S_ISREG( sb.st_mode ) && sb.st_size >= 2 && pread( fd, head, 2, 0 ) == 2 &&
             !( head[ 0 ] == 0x1f && head[ 1 ] == 0x8b ) && head[ 0 ] == '@' ) {
          void* m = mmap( nullptr, (std::size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0 );
          if ( m != MAP_FAILED ) { map_ = static_cast< char const* >( m ); map_len_ = (std::size_t)sb.st_size; fast_ = true; }
        }
        ::close( fd );
      }
      unsigned const hw = std::thread::hardware_concurrency();
      threads_ = std::max( 1u, std::min( 16u, hw / 4 ? hw / 4 : 1u ) );
      if ( char const* e = std::getenv( "PSI_READER_THREADS" ) ) threads_ = std::max( 1, std::atoi( e ) );
      if ( std::getenv( "PSI_READER_SERIAL" ) ) fast_ = false;
    }
    SeqStreamIn( SeqStreamIn const& ) = delete;
    ~SeqStreamIn()
    {
      if ( gz_ ) gzclose( gz_ );
      if ( map_ ) munmap( const_cast< char* >( map_ ), map_len_ );
    }
    std::uint64_t counts() const { return count_; }
    bool fast() const { return fast_; }

    /** Up to `num` records (0 = all that are left) by the parallel parser.  Returns 1 when records were read, 0 at the end
     *  of the input, -1 when the input is not what the fast path takes (the stream then continues with the serial parser
     *  from the same byte; `records` is left cleared). */
    int read_chunk_fast( Records& records, std::uint64_t num )
    {
      if ( !fast_ ) return -1;
      if ( fpos_ >= map_len_ ) return 0;
      unsigned const T = threads_;
      /* ---- 1. the positions of the newlines in a window that holds the chunk ---- */
      struct Slice { std::size_t begin, end; std::vector< std::uint32_t > nl; std::uint64_t first_line; };
      std::vector< Slice > slices;
      std::uint64_t lines = 0;
      std::uint64_t const want_lines = num ? 4 * num : ~0ull;
      std::size_t wpos = fpos_;
      std::size_t guess = map_len_ - fpos_;
      if ( num ) {
        /* the first record's length x num, and a little more */
        char const* p = map_ + fpos_;
        char const* e = map_ + map_len_;
        int seen = 0;
        while ( p < e && seen < 4 ) { char const* q = static_cast< char const* >( std::memchr( p, '\n', e - p ) ); if ( !q ) { p = e; break; } p = q + 1; ++seen; }
        std::size_t const rec_len = std::max< std::size_t >( 8, p - ( map_ + fpos_ ) );
        guess = std::min< std::size_t >( guess, (std::size_t)( (double)rec_len * (double)num * 1.01 ) + ( 1u << 16 ) );
      }
      while ( lines < want_lines && wpos < map_len_ ) {
        std::size_t const wend = std::min( map_len_, wpos + guess );
        std::size_t const per = ( wend - wpos + T - 1 ) / T;
        std::size_t const s0 = slices.size();
        for ( unsigned t = 0; t < T; ++t ) {
          std::size_t const b = std::min( wend, wpos + t * per ), e2 = std::min( wend, b + per );
          if ( e2 > b ) slices.push_back( Slice{ b, e2, {}, 0 } );
        }
        if ( per >= ( 1ull << 32 ) ) return leave_fast();          /* (offsets inside a slice are 32 bits) */
        auto scan = [ & ]( std::size_t i ) {
          Slice& sl = slices[ i ];
          sl.nl.reserve( ( sl.end - sl.begin ) / 64 + 16 );
          char const* base = map_ + sl.begin;
          char const* p = base;
          char const* e = map_ + sl.end;
          while ( p < e ) {
            char const* q = static_cast< char const* >( std::memchr( p, '\n', e - p ) );
            if ( !q ) break;
            sl.nl.push_back( (std::uint32_t)( q - base ) );
            p = q + 1;
          }
        };
        run_parallel( slices.size() - s0, [ & ]( std::size_t i ) { scan( s0 + i ); } );
        for ( std::size_t i = s0; i < slices.size(); ++i ) { slices[ i ].first_line = lines; lines += slices[ i ].nl.size(); }
        wpos = wend;
        guess = std::max< std::size_t >( guess / 8, 1u << 20 );
      }
      bool const at_eof = wpos >= map_len_;
      bool const virtual_nl = at_eof && map_[ map_len_ - 1 ] != '\n';      /* the last line ends with the file */
      std::uint64_t const lines_all = lines + ( virtual_nl ? 1 : 0 );
      std::uint64_t n_rec = std::min< std::uint64_t >( num ? num : ~0ull, lines_all / 4 );
      if ( n_rec == 0 || ( lines_all < want_lines && lines_all % 4 != 0 ) ) return leave_fast();      /* a truncated record, blank lines at the end ...: the serial parser says what it is */
      /* position of newline `i` (the virtual one at the end of the file included) */
      auto nl_pos = [ & ]( std::uint64_t i, std::size_t& hint ) -> std::size_t {
        if ( i >= lines ) return map_len_;
        while ( i >= slices[ hint ].first_line + slices[ hint ].nl.size() ) ++hint;
        return slices[ hint ].begin + slices[ hint ].nl[ i - slices[ hint ].first_line ];
      };
      auto slice_of = [ & ]( std::uint64_t i ) -> std::size_t {
        std::size_t lo = 0, hi = slices.size();
        while ( hi - lo > 1 ) { std::size_t const mid = ( lo + hi ) / 2; if ( slices[ mid ].first_line <= i ) lo = mid; else hi = mid; }
        return lo;
      };
      /* ---- 2. every record's sequence line (start, length), checked; lengths summed per thread ---- */
      std::vector< std::uint64_t > seq_at( n_rec );
      std::vector< std::uint32_t > seq_len( n_rec );
      std::vector< std::uint64_t > name_at;
      std::vector< std::uint32_t > name_len;
      bool const names = records.keep_names;
      if ( names ) { name_at.resize( n_rec ); name_len.resize( n_rec ); }
      std::vector< std::uint64_t > part_sum( T + 1, 0 );
      std::vector< int > part_bad( T, 0 ), part_same( T, 1 );
      std::size_t const fpos = fpos_;
      run_parallel( T, [ & ]( std::size_t t ) {
        std::uint64_t const r0 = n_rec * t / T, r1 = n_rec * ( t + 1 ) / T;
        if ( r0 == r1 ) return;
        std::size_t hint = slice_of( r0 ? 4 * r0 - 1 : 0 );
        std::size_t start = r0 ? nl_pos( 4 * r0 - 1, hint ) + 1 : fpos;
        std::uint64_t sum = 0;
        std::uint32_t first = 0;
        for ( std::uint64_t r = r0; r < r1; ++r ) {
          std::size_t const e0 = nl_pos( 4 * r, hint ), e1 = nl_pos( 4 * r + 1, hint ), e2 = nl_pos( 4 * r + 2, hint ), e3 = nl_pos( 4 * r + 3, hint );
          if ( e0 == start || map_[ start ] != '@' || e2 == e1 + 1 || map_[ e1 + 1 ] != '+' ) { part_bad[ t ] = 1; return; }
          std::size_t sl = e1 - ( e0 + 1 );
          if ( sl && map_[ e1 - 1 ] == '\r' ) --sl;
          if ( sl == 0 || sl >= ( 1ull << 32 ) ) { part_bad[ t ] = 1; return; }      /* (an empty sequence: the serial parser's business) */
          seq_at[ r ] = e0 + 1; seq_len[ r ] = (std::uint32_t)sl;
          if ( r == r0 ) first = (std::uint32_t)sl; else if ( sl != first ) part_same[ t ] = 0;
          if ( names ) {
            std::size_t nl = 0, ll = e0 - start;
            if ( ll && map_[ e0 - 1 ] == '\r' ) --ll;
            while ( 1 + nl < ll && map_[ start + 1 + nl ] != ' ' && map_[ start + 1 + nl ] != '\t' ) ++nl;
            name_at[ r ] = start + 1; name_len[ r ] = (std::uint32_t)nl;
          }
          sum += sl;
          start = e3 + 1;
        }
        part_sum[ t + 1 ] = sum;
      } );
      for ( unsigned t = 0; t < T; ++t ) if ( part_bad[ t ] ) return leave_fast();
      for ( unsigned t = 0; t < T; ++t ) part_sum[ t + 1 ] += part_sum[ t ];
      /* ---- 3. bases back to back, offsets, names ---- */
      records.clear();
      records.set_record_offset( count_ );
      records.bases.resize_uninitialized( part_sum[ T ] );
      records.offsets.resize( n_rec + 1 );
      if ( names ) records.name.resize( n_rec );
      char* const dst = records.bases.data();
      run_parallel( T, [ & ]( std::size_t t ) {
        std::uint64_t const r0 = n_rec * t / T, r1 = n_rec * ( t + 1 ) / T;
        std::uint64_t at = part_sum[ t ];
        for ( std::uint64_t r = r0; r < r1; ++r ) {
          records.offsets[ r ] = at;
          std::memcpy( dst + at, map_ + seq_at[ r ], seq_len[ r ] );
          at += seq_len[ r ];
          if ( names ) records.name[ r ].assign( map_ + name_at[ r ], name_len[ r ] );
        }
      } );
      records.offsets[ n_rec ] = part_sum[ T ];
      bool same = true;
      for ( unsigned t = 0; t < T; ++t ) same = same && part_same[ t ];
      for ( std::uint64_t t = 1; same && t < T; ++t ) {
        std::uint64_t const r0 = n_rec * t / T;
        if ( r0 < n_rec && r0 != n_rec * ( t + 1 ) / T && seq_len[ r0 ] != seq_len[ 0 ] ) same = false;
      }
      records.set_bulk_lengths( seq_len[ 0 ], same );
      std::size_t h2 = slice_of( 4 * n_rec - 1 );
      fpos_ = std::min( map_len_, nl_pos( 4 * n_rec - 1, h2 ) + 1 );
      count_ += n_rec;
      return 1;
    }

    /** Next record; false at end of input. */
    bool next( std::string& name, std::string& seq )
    {
      char const *n, *s; std::size_t nl, sl;
      if ( !next_view( n, nl, s, sl ) ) return false;
      name.assign( n, nl ); seq.assign( s, sl );
      return true;
    }

    /** Next record appended to `records` without intermediate strings. */
    bool next_into( Records& records )
    {
      char const *n, *s; std::size_t nl, sl;
      if ( !next_view( n, nl, s, sl ) ) return false;
      records.push_back( n, nl, s, sl );
      return true;
    }
  private:
    static constexpr std::size_t BLOCK = 4u << 20;

    template < typename F >
    static void run_parallel( std::size_t n, F fn )
    {
      if ( n <= 1 ) { if ( n ) fn( 0 ); return; }
      std::vector< std::thread > th;
      for ( std::size_t i = 1; i < n; ++i ) th.emplace_back( [ i, &fn ] { fn( i ); } );
      fn( 0 );
      for ( auto& t : th ) t.join();
    }
    /* the input is not what the fast path takes: the serial parser goes on from the byte the fast path stands at */
    int leave_fast()
    {
      fast_ = false;
      if ( gzseek( gz_, (z_off_t)fpos_, SEEK_SET ) < 0 ) throw std::runtime_error( "read error" );
      pos_ = end_ = 0; eof_ = false;
      return -1;
    }

    /* Views stay valid until the next call.  `keep` is where the current record starts in the
     * block buffer: line() preserves everything from there on when it refills, and positions
     * inside the record are held relative to it. */
    bool next_view( char const*& name, std::size_t& nlen, char const*& seq, std::size_t& slen )
    {
      while ( true ) {
        std::size_t keep = pos_, lo, ll;
        if ( !line( lo, ll, keep ) ) return false;
        if ( ll == 0 ) continue;
        char const first = buf_[ lo ];
        if ( first == '@' || first == '>' ) {
          bool const fq = first == '@';
          std::size_t name_len = 0;
          while ( 1 + name_len < ll && buf_[ lo + 1 + name_len ] != ' ' && buf_[ lo + 1 + name_len ] != '\t' ) ++name_len;
          std::size_t const name_rel = lo + 1 - keep;
          std::size_t so, sl, to, tl;
          if ( !line( so, sl, keep ) ) throw std::runtime_error( fq ? "truncated FASTQ record" : "truncated FASTA record" );
          std::size_t const seq_rel = so - keep;
          if ( fq && ( !line( to, tl, keep ) || !line( to, tl, keep ) ) ) throw std::runtime_error( "truncated FASTQ record" );
          name = buf_.data() + keep + name_rel; nlen = name_len;
          seq = buf_.data() + keep + seq_rel; slen = sl;
        } else {                                    /* plain text: one sequence per line */
          tmp_name_ = std::to_string( count_ );
          name = tmp_name_.data(); nlen = tmp_name_.size();
          seq = buf_.data() + lo; slen = ll;
        }
        ++count_;
        return true;
      }
    }

    /* Offset and length of the next line (terminator stripped).  Refilling moves [keep, end) to
     * the front of the buffer and sets keep to 0. */
    bool line( std::size_t& off, std::size_t& len, std::size_t& keep )
    {
      while ( true ) {
        char* base = buf_.data();
        char* nl = end_ > pos_ ? static_cast< char* >( std::memchr( base + pos_, '\n', end_ - pos_ ) ) : nullptr;
        if ( nl != nullptr || ( eof_ && pos_ < end_ ) ) {
          std::size_t const stop = nl ? static_cast< std::size_t >( nl - base ) : end_;
          off = pos_;
          len = stop - pos_;
          pos_ = nl ? stop + 1 : end_;
          if ( len && base[ off + len - 1 ] == '\r' ) --len;
          return true;
        }
        if ( eof_ ) return false;
        if ( keep ) {
          std::memmove( base, base + keep, end_ - keep );
          pos_ -= keep; end_ -= keep; keep = 0;
        }
        if ( buf_.size() - end_ < BLOCK / 2 ) { buf_.resize( buf_.size() * 2 ); base = buf_.data(); }
        int got = gzread( gz_, base + end_, static_cast< unsigned >( std::min< std::size_t >( buf_.size() - end_, 1u << 30 ) ) );
        if ( got < 0 ) throw std::runtime_error( "read error" );
        if ( got == 0 ) eof_ = true;
        end_ += static_cast< std::size_t >( got );
      }
    }

    gzFile gz_ = nullptr;
    std::vector< char > buf_;
    std::size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    std::string tmp_name_;
    std::uint64_t count_ = 0;
    char const* map_ = nullptr;               /* the fast path: the file, mapped */
    std::size_t map_len_ = 0, fpos_ = 0;
    bool fast_ = false;
    unsigned threads_ = 1;
  };

  /**
   *  Load up to `num` records (0 = all) into `records`; its record offset becomes the number
   *  of records consumed before this chunk, so read ids stay global across chunks
   *  (reference sequence.hpp:1616).  Returns false when nothing was read.
   */
  inline bool
  readRecords( Records& records, SeqStreamIn& iss, std::uint64_t num = 0 )
  {
    records.clear();
    records.set_record_offset( iss.counts() );
    int const fast = iss.read_chunk_fast( records, num );
    if ( fast == 0 ) return false;
    if ( fast < 0 ) {
      records.clear();
      records.set_record_offset( iss.counts() );
      bool const names = records.keep_names;
      while ( ( num == 0 || records.size() < num ) && iss.next_into( records ) ) { }
      if ( !names ) records.name.clear();
    }
    if ( records.size() != 0 ) records.pack();    /* 2 bits per base for the host link (a few threads, ~10 ms per 150 Mbp) */
    return records.size() != 0;
  }
}  /* --- end of namespace psi --- */

#endif
