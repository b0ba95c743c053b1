// psi::Records + readRecords -- a chunk of reads and its reader.
// Mirrors the parts of reference include/psi/sequence.hpp the seed-finding loop touches:
// Records (name/str, record offset :1130-1294), readRecords (:1590-1624).  FASTQ (optionally
// gzip'd) or one-sequence-per-line text; kseq++ / SeqAn are not used.
//
// The bases of a chunk live in page-locked memory (psigpu_host_alloc): the copy engine of the GPU
// reads them in place, psigpu_find_seeds does not stage them.
#ifndef PSI_AMD_SEQUENCE_HPP__
#define PSI_AMD_SEQUENCE_HPP__

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
    std::string substr( std::size_t pos, std::size_t len ) const { return std::string( p_ + pos, len ); }
  private:
    void grow( std::size_t want )
    {
      std::size_t cap = cap_ ? cap_ : ( 1u << 20 );
      while ( cap < want ) cap *= 2;
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
    }
    SeqStreamIn( SeqStreamIn const& ) = delete;
    ~SeqStreamIn() { if ( gz_ ) gzclose( gz_ ); }
    std::uint64_t counts() const { return count_; }

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
    while ( ( num == 0 || records.size() < num ) && iss.next_into( records ) ) { }
    if ( records.size() != 0 ) records.pack();    /* 2 bits per base for the host link (a few threads, ~10 ms per 150 Mbp) */
    return records.size() != 0;
  }
}  /* --- end of namespace psi --- */

#endif
