// psikt -- seed finder command line, MI355X build.
//
// Same command line and output bytes as the reference CLI (reference src/psikt.cpp:292-471 for
// the options, :172-181 for the 32-byte hit records, :190-208 for the chunk loop), driving the
// psi::SeedFinder shim (psi_amd/include/psi/seed_finder.hpp) and, through it, the HIP kernels.
// Argument parsing, logging and I/O are plain C++; SeqAn's ArgumentParser / spdlog are not used.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include <psi/seed_finder.hpp>

using namespace psi;

namespace {

struct Options {                       // reference src/options.hpp:67-93
  std::string graph_path, fq_path, output_path = "out.gam", pindex_path, log_path = "psi.log";
  std::string dindex_mode = "per-component", index = "WOTD";
  unsigned int seed_len = 0, step_size = 1, distance = 0, path_num = 0, context = 0;
  unsigned int gocc_threshold = 0, max_mem = 0, dindex_min_ris = 0, dindex_max_ris = 0;
  unsigned long chunk_size = 0;
  bool patched = true, indexonly = false, nologfile = false, quiet = false, nocolor = false;
  bool nolog = false, verbose = false;
  std::vector< int > devices{ 0 };
  unsigned int query_mode = PSIGPU_MODE_KMER_TABLE;
  bool follow_reversing = false;
};

struct Logger {
  FILE* file = nullptr;
  bool console_info = false, quiet = false, off = false;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void line( const char* level, std::string const& msg, bool to_console )
  {
    if ( off ) return;
    double t = std::chrono::duration< double >( std::chrono::steady_clock::now() - t0 ).count();   /* seconds since start */
    if ( file ) { fprintf( file, "[%9.3f] [%s] %s\n", t, level, msg.c_str() ); fflush( file ); }
    if ( to_console && !quiet ) fprintf( stderr, "[psikt] [%s] %s\n", level, msg.c_str() );
  }
  void info( std::string const& m ) { line( "info", m, console_info ); }
  void warn( std::string const& m ) { line( "warning", m, true ); }
  void error( std::string const& m ) { line( "error", m, true ); }
};

const char* USAGE =
  "psikt [OPTIONS] GRAPH_FILE\n"
  "  GRAPH_FILE                 vg or gfa graph\n"
  "  -f, --fastq FILE           reads (fq, fastq, optionally .gz) [required]\n"
  "  -o, --output FILE          output file (default: out.gam)\n"
  "  -I, --path-index PREFIX    path index to load / save\n"
  "  -l, --seed-length INT      seed length [required]\n"
  "  -c, --chunk-size INT       reads per chunk, 0 = all (default: 0)\n"
  "  -e, --step-size INT        starting-locus sampling step (default: 1)\n"
  "  -d, --distance INT         distance between seeds, 0 = seed length (default: 0)\n"
  "  -n, --path-num INT         paths per region to index (default: 0)\n"
  "  -P, --no-patched           index full genome-wide paths\n"
  "  -t, --context INT          context length in patching (default: 0)\n"
  "  -r, --gocc-threshold INT   skip seeds with more path occurrences, 0 = off (default: 0)\n"
  "  -E, --max-mem INT          maximum number of MEMs, 0 = off (default: 0)\n"
  "  -m, --min-insert-size INT  distance index minimum insert size (default: 0)\n"
  "  -M, --max-insert-size INT  distance index maximum insert size (default: 0)\n"
  "      --dindex-mode MODE     per-component | whole (default: per-component)\n"
  "  -i, --index NAME           reads index: SA ESA WOTD DFI QGRAM FM (default: WOTD; accepted,\n"
  "                             the device builds its own seed table)\n"
  "  -x, --index-only           only build the path index\n"
  "  -L, --log-file FILE        log file (default: psi.log)\n"
  "  -Q, --no-log-file          no log file\n"
  "  -q, --quiet                quiet console\n"
  "  -C, --no-color             no colour (accepted)\n"
  "  -D, --disable-log          disable logging\n"
  "  -v, --verbose              info messages on the console\n"
  "      --device INT           GPU ordinal (default: 0)\n"
  "      --devices LIST         several GPUs, e.g. 0-7 or 0,2,3: the index is copied to each, every chunk's\n"
  "                             reads are split into one contiguous range per GPU (same output)\n"
  "      --query-mode MODE      kmer-table | locus-table | traverse | auto (default: kmer-table; same hits;\n"
  "                             auto: traverse for a small FASTQ, tables for a large one)\n"
  "      --follow-reversing-edges  links whose sides reverse (inversions) and reverse path steps are walked as the\n"
  "                             reference walks them -- the link's `to` node, read forwards -- instead of being refused\n"
  "  -h, --help\n";

bool ends_with( std::string const& s, const char* suf )
{
  size_t n = strlen( suf );
  return s.size() >= n && s.compare( s.size() - n, n, suf ) == 0;
}

unsigned long to_uint( std::string const& opt, std::string const& v )
{
  char* end = nullptr;
  unsigned long x = strtoul( v.c_str(), &end, 10 );
  if ( v.empty() || *end != '\0' ) throw std::runtime_error( "invalid value for " + opt + ": '" + v + "'" );
  return x;
}

Options parse_args( int argc, char** argv )
{
  static const std::map< std::string, std::string > long2short = {
    { "--fastq", "-f" }, { "--output", "-o" }, { "--path-index", "-I" }, { "--seed-length", "-l" },
    { "--chunk-size", "-c" }, { "--step-size", "-e" }, { "--distance", "-d" }, { "--path-num", "-n" },
    { "--no-patched", "-P" }, { "--context", "-t" }, { "--gocc-threshold", "-r" }, { "--max-mem", "-E" },
    { "--min-insert-size", "-m" }, { "--max-insert-size", "-M" }, { "--index", "-i" },
    { "--index-only", "-x" }, { "--log-file", "-L" }, { "--no-log-file", "-Q" }, { "--quiet", "-q" },
    { "--no-color", "-C" }, { "--disable-log", "-D" }, { "--verbose", "-v" }, { "--help", "-h" } };
  Options o;
  bool have_f = false, have_l = false;
  std::vector< std::string > pos;
  for ( int i = 1; i < argc; ++i ) {
    std::string a = argv[ i ], val;
    bool has_val = false;
    if ( a.rfind( "--", 0 ) == 0 ) {
      size_t eq = a.find( '=' );
      if ( eq != std::string::npos ) { val = a.substr( eq + 1 ); a = a.substr( 0, eq ); has_val = true; }
      auto it = long2short.find( a );
      if ( it != long2short.end() ) a = it->second;
    }
    auto need = [&]() -> std::string {
      if ( has_val ) return val;
      if ( i + 1 >= argc ) throw std::runtime_error( "option " + a + " needs a value" );
      return argv[ ++i ];
    };
    if ( a == "-h" ) { fputs( USAGE, stdout ); exit( 0 ); }
    else if ( a == "-f" ) { o.fq_path = need(); have_f = true; }
    else if ( a == "-o" ) o.output_path = need();
    else if ( a == "-I" ) o.pindex_path = need();
    else if ( a == "-l" ) { o.seed_len = (unsigned)to_uint( a, need() ); have_l = true; }
    else if ( a == "-c" ) o.chunk_size = to_uint( a, need() );
    else if ( a == "-e" ) o.step_size = (unsigned)to_uint( a, need() );
    else if ( a == "-d" ) o.distance = (unsigned)to_uint( a, need() );
    else if ( a == "-n" ) o.path_num = (unsigned)to_uint( a, need() );
    else if ( a == "-P" ) o.patched = false;
    else if ( a == "-t" ) o.context = (unsigned)to_uint( a, need() );
    else if ( a == "-r" ) o.gocc_threshold = (unsigned)to_uint( a, need() );
    else if ( a == "-E" ) o.max_mem = (unsigned)to_uint( a, need() );
    else if ( a == "-m" ) o.dindex_min_ris = (unsigned)to_uint( a, need() );
    else if ( a == "-M" ) o.dindex_max_ris = (unsigned)to_uint( a, need() );
    else if ( a == "--dindex-mode" ) {
      o.dindex_mode = need();
      if ( o.dindex_mode != "per-component" && o.dindex_mode != "whole" )
        throw std::runtime_error( "Unknown distance index construction mode: " + o.dindex_mode );
    }
    else if ( a == "-i" ) {
      o.index = need();
      static const char* valid[] = { "SA", "ESA", "WOTD", "DFI", "QGRAM", "FM" };
      bool ok = false;
      for ( auto v : valid ) ok = ok || o.index == v;
      if ( !ok ) throw std::runtime_error( "invalid reads index: " + o.index );
    }
    else if ( a == "-x" ) o.indexonly = true;
    else if ( a == "-L" ) o.log_path = need();
    else if ( a == "-Q" ) o.nologfile = true;
    else if ( a == "-q" ) o.quiet = true;
    else if ( a == "-C" ) o.nocolor = true;
    else if ( a == "-D" ) o.nolog = true;
    else if ( a == "-v" ) o.verbose = true;
    else if ( a == "--device" ) o.devices.assign( 1, (int)to_uint( a, need() ) );
    else if ( a == "--devices" ) {
      o.devices.clear();
      std::string list = need();
      size_t p = 0;
      while ( p <= list.size() ) {
        size_t q = list.find( ',', p );
        if ( q == std::string::npos ) q = list.size();
        std::string tok = list.substr( p, q - p );
        size_t dash = tok.find( '-' );
        if ( dash == std::string::npos ) o.devices.push_back( (int)to_uint( a, tok ) );
        else {
          unsigned long lo = to_uint( a, tok.substr( 0, dash ) ), hi = to_uint( a, tok.substr( dash + 1 ) );
          if ( hi < lo || hi - lo > 1024 ) throw std::runtime_error( "invalid device range '" + tok + "'" );
          for ( unsigned long d = lo; d <= hi; ++d ) o.devices.push_back( (int)d );
        }
        p = q + 1;
      }
      if ( o.devices.empty() ) throw std::runtime_error( "--devices needs at least one GPU" );
    }
    else if ( a == "--follow-reversing-edges" ) o.follow_reversing = true;
    else if ( a == "--query-mode" ) {
      std::string m = need();
      if ( m == "kmer-table" ) o.query_mode = PSIGPU_MODE_KMER_TABLE;
      else if ( m == "locus-table" ) o.query_mode = PSIGPU_MODE_LOCUS_TABLE;
      else if ( m == "traverse" ) o.query_mode = PSIGPU_MODE_TRAVERSE;
      else if ( m == "auto" ) o.query_mode = PSIGPU_MODE_AUTO;
      else throw std::runtime_error( "unknown query mode " + m );
    }
    else if ( !a.empty() && a[ 0 ] == '-' ) throw std::runtime_error( "unknown option " + a );
    else pos.push_back( a );
  }
  if ( pos.size() != 1 ) throw std::runtime_error( "exactly one GRAPH_FILE is required" );
  o.graph_path = pos[ 0 ];
  if ( !ends_with( o.graph_path, ".vg" ) && !ends_with( o.graph_path, ".gfa" ) )
    throw std::runtime_error( "GRAPH_FILE must be a vg or gfa file" );
  if ( !have_f ) throw std::runtime_error( "option -f/--fastq is required" );
  if ( !have_l ) throw std::runtime_error( "option -l/--seed-length is required" );
  if ( o.distance == 0 ) o.distance = o.seed_len;       // reference src/psikt.cpp:469
  return o;
}

/* Writes hit arrays in the order they are pushed, on its own thread; owns them until written.  The thread also counts the
 * reads a chunk covers (the arrays are sorted by read id: the changes of read id along them) -- round 6: that count was an
 * OpenMP loop on the main thread, and the 128 spinning OpenMP threads starved the FASTQ reader's threads of their cores
 * (0.07 s per 7 M records, and the parse of the next chunk took 0.14 s instead of 0.02).  A regular file is written by a few
 * threads at once (pwrite at the chunk's offsets), anything else by one write after the other. */
class HitWriter {
public:
  explicit HitWriter( FILE* f ) : out_( f ), th_( [ this ] { loop(); } )
  {
    fflush( f );
    fd_ = fileno( f );
    struct stat sb;
    seekable_ = fd_ >= 0 && fstat( fd_, &sb ) == 0 && S_ISREG( sb.st_mode );
    if ( seekable_ ) { off_t const at = lseek( fd_, 0, SEEK_CUR ); if ( at < 0 ) seekable_ = false; else pos_ = (std::uint64_t)at; }
  }
  ~HitWriter() { finish(); }
  void push( psigpu_hits h )
  {
    std::unique_lock< std::mutex > lk( mu_ );
    cv_.wait( lk, [ this ] { return q_.size() < 2; } );      /* at most two chunks of hits in flight */
    q_.push_back( h );
    cv_.notify_all();
  }
  bool finish()
  {
    {
      std::lock_guard< std::mutex > lk( mu_ );
      if ( done_ ) return ok_;
      done_ = true;
      cv_.notify_all();
    }
    th_.join();
    if ( seekable_ && ok_ ) ok_ = lseek( fd_, (off_t)pos_, SEEK_SET ) >= 0;
    return ok_;
  }
  /** reads covered by the chunks written so far (complete after finish()) */
  unsigned long long covered() const { return covered_; }
private:
  static constexpr unsigned MAX_THREADS = 32;
  unsigned const THREADS = [] { char const* e = getenv( "PSIKT_WRITE_THREADS" ); unsigned const t = e ? (unsigned)atoi( e ) : 8u; return t < 1 ? 1u : t > MAX_THREADS ? MAX_THREADS : t; }();
  static unsigned long long count_changes( psigpu_hit const* d, std::uint64_t a, std::uint64_t b )
  {
    unsigned long long n = 0;
    for ( std::uint64_t i = a ? a : 1; i < b; ++i ) n += d[ i ].read_id != d[ i - 1 ].read_id;
    return n;
  }
  void loop()
  {
    while ( true ) {
      psigpu_hits h;
      {
        std::unique_lock< std::mutex > lk( mu_ );
        cv_.wait( lk, [ this ] { return !q_.empty() || done_; } );
        if ( q_.empty() ) return;
        h = q_.front();
      }
      if ( h.n ) {
        std::uint64_t const n = h.n;
        unsigned long long part[ MAX_THREADS ] = { 0 };
        bool wrote[ MAX_THREADS ];
        std::vector< std::thread > th;
        for ( unsigned t = 0; t < THREADS; ++t ) {
          wrote[ t ] = true;
          auto job = [ &, t ] {
            std::uint64_t const a = n * t / THREADS, b = n * ( t + 1 ) / THREADS;
            if ( seekable_ ) {
              char const* p = reinterpret_cast< char const* >( h.data + a );
              std::uint64_t left = ( b - a ) * sizeof( psigpu_hit ), at = pos_ + a * sizeof( psigpu_hit );
              while ( left ) {
                ssize_t const w = pwrite( fd_, p, left, (off_t)at );
                if ( w <= 0 ) { wrote[ t ] = false; break; }
                p += w; at += (std::uint64_t)w; left -= (std::uint64_t)w;
              }
            }
            part[ t ] = count_changes( h.data, a, b );
          };
          if ( t + 1 < THREADS ) th.emplace_back( job ); else job();
        }
        for ( auto& t : th ) t.join();
        if ( !seekable_ && fwrite( h.data, sizeof( psigpu_hit ), n, out_ ) != n ) ok_ = false;
        for ( unsigned t = 0; t < THREADS; ++t ) { covered_ += part[ t ]; ok_ = ok_ && wrote[ t ]; }
        covered_ += 1;                                      /* (the chunk's first read) */
        pos_ += n * sizeof( psigpu_hit );
      }
      psigpu_free_hits( &h );
      std::lock_guard< std::mutex > lk( mu_ );
      q_.pop_front();
      cv_.notify_all();
    }
  }
  FILE* out_;
  int fd_ = -1;
  bool seekable_ = false;
  std::uint64_t pos_ = 0;
  unsigned long long covered_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque< psigpu_hits > q_;
  bool done_ = false, ok_ = true;
  std::thread th_;
};

double seconds_since( std::chrono::steady_clock::time_point t0 )
{
  return std::chrono::duration< double >( std::chrono::steady_clock::now() - t0 ).count();
}

int run( Options const& o, Logger& log )
{
  /* the HIP runtime takes a few hundred milliseconds to come up: let it do so while the graph file is parsed */
  /* ... and the output arrays of the first chunks are pinned there too (three go round: one being filled, two with the writer):
   * the library's first call assumes ten records per read, later ones what the chunks before had */
  std::uint64_t const warm_records = o.chunk_size ? (std::uint64_t)( (double)o.chunk_size * 10.0 * 1.25 ) + 8192 : 0;
  std::thread warm( [ warm_records ] {
    void* p = psigpu_host_alloc( 4096 );
    if ( p != nullptr ) psigpu_host_free( p );
    if ( p != nullptr && warm_records && warm_records < ( 1ull << 26 ) ) (void)psigpu_reserve_hit_arrays( warm_records, 3 );
  } );
  struct Joiner { std::thread& t; ~Joiner() { if ( t.joinable() ) t.join(); } } warm_guard{ warm };
  log.info( "Loading input graph from file '" + o.graph_path + "'..." );
  Graph graph( o.graph_path, o.follow_reversing );
  log.info( "Number of nodes: " + std::to_string( graph.get_node_count() ) + ", edges: " +
            std::to_string( graph.get_edge_count() ) + ", paths: " + std::to_string( graph.get_path_count() ) );
  SeqStreamIn reads_iss( o.fq_path );
  /* The reference opens the output OPEN_CREATE | OPEN_WRONLY without truncation
   * (src/psikt.cpp:265-267), leaving stale bytes behind a shorter run; truncate instead. */
  FILE* out = fopen( o.output_path.c_str(), "wb" );
  if ( out == nullptr ) throw std::runtime_error( "cannot open file '" + o.output_path + "'" );

  typedef SeedFinder< NoStats > finder_type;
  warm.join();
  finder_type finder( graph, o.seed_len, o.gocc_threshold, o.max_mem, o.devices[ 0 ] );
  finder.set_query_mode( o.query_mode );
  if ( o.query_mode == PSIGPU_MODE_AUTO ) {
    /* how much is there to answer?  A plain FASTQ is ~2 bytes per base (bases + qualities): seeds ~ size / (2 d); the
     * chunks follow from -c (0: the whole file is one chunk).  A gzip'd file says nothing: assume a lot. */
    std::uint64_t calls = 0, seeds = 0;
    struct stat sb;
    bool const gz = o.fq_path.size() > 3 && o.fq_path.compare( o.fq_path.size() - 3, 3, ".gz" ) == 0;
    if ( !gz && stat( o.fq_path.c_str(), &sb ) == 0 && sb.st_size > 0 ) {
      unsigned const d = o.distance ? o.distance : o.seed_len;
      seeds = (std::uint64_t)sb.st_size / ( 2ull * std::max( 1u, d ) ) + 1;
      std::uint64_t const reads = (std::uint64_t)sb.st_size / 320 + 1;
      calls = o.chunk_size ? ( reads + o.chunk_size - 1 ) / o.chunk_size : 1;
    }
    finder.set_expected_work( calls, seeds );
  }
  /* the first chunk of reads is parsed (into page-locked memory) while the index is loaded or made */
  /* two chunk buffers: while the device answers one chunk the next is parsed and packed into the other (round 5; the
   * reference's loop loads, seeds, loads: src/psikt.cpp:190-208 -- parsing 1 M reads costs several device calls) */
  auto chunk_a = finder.create_readrecord();
  auto chunk_b = finder.create_readrecord();
  chunk_a.keep_names = chunk_b.keep_names = false;      /* (nothing here looks at a read's name: a million std::strings per chunk) */
  /* how long parsing + packing a chunk took on the thread that did it (the loop itself only waits for it) */
  double t_parse_a = 0, t_parse_b = 0, t_parse_all = 0;
  auto timed_read = [ & ]( decltype( chunk_a )& into, double& took ) {
    auto const t = std::chrono::steady_clock::now();
    bool const have = readRecords( into, reads_iss, o.chunk_size );
    took = seconds_since( t );
    return have;
  };
  auto* chunk_p = &chunk_a;
  auto* next_p = &chunk_b;
  std::future< bool > first_chunk;
  if ( !o.indexonly )
    first_chunk = std::async( std::launch::async, [ & ] { return timed_read( chunk_a, t_parse_a ); } );
  log.info( "Looking for an existing path index..." );
  auto t0 = std::chrono::steady_clock::now();
  if ( finder.load_path_index( o.pindex_path, o.context, o.step_size, o.dindex_min_ris, o.dindex_max_ris ) ) {
    log.info( "The path index has been found and loaded." );
  } else {
    if ( o.path_num == 0 ) log.info( "No path has been specified. Skipping path indexing..." );
    else log.info( "No valid path index found. Creating the path index..." );
    if ( o.step_size > 1 )
      log.warn( "-e " + std::to_string( o.step_size ) + ": every e-th uncovered locus of a node is kept; the reference steps "
                "along its backtracked k-paths (seed_finder.hpp:1520-1524). Neither is fully sensitive, and the two hit "
                "sets differ; only -e 1 is covered by the parity claim." );
    auto info_cb = [ &log ]( std::string const& m ) { log.info( m ); };
    auto warn_cb = [ &log ]( std::string const& m ) { log.warn( m ); };
    finder.create_path_index( o.path_num, o.patched, o.context, o.step_size, o.dindex_min_ris, o.dindex_max_ris,
                              PerComponent{}, info_cb, warn_cb );
    log.info( "Created path index in " + std::to_string( seconds_since( t0 ) ) + " s." );
    if ( o.path_num != 0 ) {
      if ( o.pindex_path.empty() ) log.warn( "No path index file is specified. Skipping..." );
      else if ( !finder.serialize_path_index( o.pindex_path, o.step_size ) )
        log.warn( "Specified path index file is not writable. Skipping..." );
      else log.info( "Saved path index." );
    }
  }
  log.info( "Number of starting loci (in " + std::to_string( finder.get_nof_uniq_nodes() ) + " nodes of total " +
            std::to_string( graph.get_node_count() ) + "): " + std::to_string( finder.get_nof_starting_loci() ) );
  if ( o.indexonly ) {
    log.info( "Skipping seed finding as requested..." );
    fclose( out );
    return 0;
  }

  /* The reference's write_callback (src/psikt.cpp:172-181) appends one 32-byte record per hit:
   * 4 x native-endian u64 = the layout of psigpu_hit, so a chunk's hits are ONE fwrite.  It runs on
   * a writer thread while the device answers the next chunk.  Hits arrive sorted by read id, so
   * the reads covered are the changes of read id along the array. */
  unsigned long long found = 0, covered = 0;
  HitWriter writer( out );
  /* --devices: one finder (one psigpu_ctx) per further GPU, each with a copy of the graph and the index */
  std::vector< std::unique_ptr< finder_type > > more;
  for ( size_t d = 1; d < o.devices.size(); ++d ) {
    more.emplace_back( new finder_type( graph, o.seed_len, o.gocc_threshold, o.max_mem, o.devices[ d ] ) );
    more.back()->set_query_mode( o.query_mode );
    more.back()->share_path_index( finder );
  }
  if ( !more.empty() ) log.info( "Index copied to " + std::to_string( o.devices.size() ) + " devices." );

  SeedsRecord seeds;
  auto traverser = finder.create_traverser();
  log.info( "Finding seeds..." );
  auto t_all = std::chrono::steady_clock::now();
  double t_device = 0, t_wait_reads = 0, t_call = 0, t_count = 0, t_push = 0, t_last_call = 0;
  std::future< bool > next_chunk;      /* the chunk being read into *next_p while the device is busy */
  double* parse_p = &t_parse_a;
  double* parse_next_p = &t_parse_b;
  while ( true ) {
    log.info( "Loading a read chunk..." );
    auto t_load = std::chrono::steady_clock::now();
    bool have;
    if ( first_chunk.valid() ) have = first_chunk.get();
    else if ( next_chunk.valid() ) { have = next_chunk.get(); std::swap( chunk_p, next_p ); std::swap( parse_p, parse_next_p ); }
    else have = timed_read( *chunk_p, *parse_p );
    t_wait_reads += seconds_since( t_load );
    if ( !have ) break;
    auto& chunk = *chunk_p;
    t_parse_all += *parse_p;
    /* (parsed and packed in ...: on the thread that read the chunk, mostly while the device answered the chunk before;
     * waited ...: what this loop stood still for it) */
    log.info( "Fetched " + std::to_string( chunk.size() ) + " reads with total length of " +
              std::to_string( chunk.length_sum() ) + "bp in " + std::to_string( *parse_p ) + " s (waited " +
              std::to_string( seconds_since( t_load ) ) + " s)." );
    /* (the reader is this thread's again only after the get() above: one thread at a time on the stream) */
    if ( o.chunk_size != 0 )
      next_chunk = std::async( std::launch::async, [ &, np = next_p, tp = parse_next_p ] { return timed_read( *np, *tp ); } );
    finder.get_seeds( seeds, chunk, o.distance );
    auto seeds_index = finder.index_reads( seeds );
    log.info( "Finding all seeds..." );
    auto st = finder.get_stats();
    if ( more.empty() ) {
      auto const t_a = std::chrono::steady_clock::now();
      psigpu_hits hits = finder.seeds_all_hits( seeds, seeds_index, traverser );
      auto const t_b = std::chrono::steady_clock::now();
      found += hits.n;
      auto const t_c = std::chrono::steady_clock::now();
      writer.push( hits );                             // takes ownership, frees after writing
      t_call += std::chrono::duration< double >( t_b - t_a ).count();
      t_last_call = std::chrono::duration< double >( t_b - t_a ).count();
      t_count += std::chrono::duration< double >( t_c - t_b ).count();
      t_push += seconds_since( t_c );
      st = finder.get_stats();
    } else {
      /* reads are independent given the index: GPU r answers the r-th contiguous range of the chunk
       * (read ids stay global through the record offset); the ranges' sorted hit arrays, in range
       * order, are the sorted chunk.  Each GPU returns its hits over its own host link: there is
       * nothing to exchange between the devices. */
      const size_t nd = o.devices.size();
      std::vector< psigpu_hits > part( nd );
      std::vector< std::string > err( nd );
      std::vector< std::thread > th;
      const std::uint64_t n = chunk.size();
      for ( size_t r = 0; r < nd; ++r )
        th.emplace_back( [ &, r ] {
          const std::uint64_t b = n * r / nd, e = n * ( r + 1 ) / nd;
          try { part[ r ] = ( r == 0 ? finder : *more[ r - 1 ] ).seeds_all_hits( seeds, b, e ); }
          catch ( std::exception const& ex ) { err[ r ] = ex.what(); }
        } );
      for ( auto& t : th ) t.join();
      for ( size_t r = 0; r < nd; ++r )
        if ( !err[ r ].empty() ) {
          for ( auto& h : part ) psigpu_free_hits( &h );      /* what the other devices returned (pinned memory) */
          throw std::runtime_error( "device " + std::to_string( o.devices[ r ] ) + ": " + err[ r ] );
        }
      st = finder.get_stats();
      for ( size_t r = 0; r < nd; ++r ) {
        found += part[ r ].n;
        writer.push( part[ r ] );
        if ( r ) {
          auto s2 = more[ r - 1 ]->get_stats();
          st.n_hits_on_path += s2.n_hits_on_path; st.n_hits_off_path += s2.n_hits_off_path;
          st.ms_total = std::max( st.ms_total, s2.ms_total );
        }
      }
    }
    t_device += st.ms_total * 1e-3;
    log.info( "Found seeds on paths: " + std::to_string( st.n_hits_on_path ) + ", off paths: " +
              std::to_string( st.n_hits_off_path ) + " (raw), device time " + std::to_string( st.ms_total ) + " ms, call " +
              std::to_string( t_last_call * 1e3 ) + " ms." );
  }
  auto const t_fin = std::chrono::steady_clock::now();
  if ( !writer.finish() ) throw std::runtime_error( "cannot write to '" + o.output_path + "'" );
  covered = writer.covered();
  fclose( out );
  double const t_finish = seconds_since( t_fin );
  log.info( "Found seed in " + std::to_string( seconds_since( t_all ) ) + " s (" + std::to_string( t_device ) +
            " s on the device)." );
  /* where the loop's time went (round 6): waiting for the reader, inside the library call (transfers, kernels, widening),
   * counting covered reads, waiting for the writer's queue, and the writer's last chunk; beside it what the reader's thread
   * spent parsing + packing (overlapped with the calls) */
  log.info( "Seed loop breakdown: wait_reads " + std::to_string( t_wait_reads ) + " s, call " + std::to_string( t_call ) +
            " s, count " + std::to_string( t_count ) + " s, push " + std::to_string( t_push ) + " s, finish_write " +
            std::to_string( t_finish ) + " s; parse+pack (reader thread) " + std::to_string( t_parse_all ) + " s." );
  log.info( "Total number of seeds found: " + std::to_string( found ) );            // src/psikt.cpp:59-80
  log.info( "Number of reads covered: " + std::to_string( covered ) );
  if ( !getenv( "PSIKT_CLEAN_EXIT" ) ) {
    /* everything is written: end the process here.  Handing 10+ GB of device and page-locked memory back
     * piece by piece (the destructors of the finder, the index, the record buffers) takes a few hundred
     * milliseconds that the operating system does not need. */
    fflush( nullptr );
    _exit( 0 );
  }
  return 0;
}

}  // namespace

int main( int argc, char** argv )
{
  Options o;
  try { o = parse_args( argc, argv ); }
  catch ( std::exception const& e ) {
    fprintf( stderr, "psikt: %s\n%s", e.what(), USAGE );
    return 1;
  }
  Logger log;
  log.quiet = o.quiet; log.off = o.nolog; log.console_info = o.verbose;
  if ( !o.nologfile && !o.nolog ) log.file = fopen( o.log_path.c_str(), "a" );
  int rc = 0;
  try { rc = run( o, log ); }
  catch ( std::exception const& e ) {
    log.error( e.what() );
    if ( log.quiet || log.off ) fprintf( stderr, "psikt: %s\n", e.what() );
    rc = 1;
  }
  if ( log.file ) fclose( log.file );
  return rc;
}
