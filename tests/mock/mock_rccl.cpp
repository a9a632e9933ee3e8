// mock_rccl.cpp -- test infrastructure: the eleven entries of librccl.so that libtmc2gof.so's sharded mode calls
// (mpeg-pcc-tmc2_amd/host/gof_runner.cpp, loaded through TMC2_RCCL_LIBRARY), carried out between PROCESSES through files in
// $MOCK_RCCL_DIR -- "device" buffers are host memory here (tests/mock/mock_tmc2hip.cpp) -- and logged per rank in
// $MOCK_RCCL_DIR/log_<rank>: one line per call.  tests/test_native_gof_schedule.py checks what crosses and how often.  Never shipped.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Id {
  char internal[128];
};
struct Comm {
  int rank, world;
  long seq = 0;
  bool grouped = false;
  std::atomic<bool> aborted{false};  // (ncclCommAbort from the runner's watchdog thread: every pending wait gives up)
  struct P2p {
    bool        send;
    void*       buf;
    size_t      bytes;
    int         peer;
  };
  std::vector<P2p> pending;
};
std::string dir() { return getenv( "MOCK_RCCL_DIR" ) ? getenv( "MOCK_RCCL_DIR" ) : "/tmp"; }
void        log( const Comm* c, const std::string& line ) {
  std::ofstream f( dir() + "/log_" + std::to_string( c->rank ), std::ios::app );
  f << line << "\n";
}
size_t width( int type ) { return type == 8 ? 8 : ( type == 2 || type == 3 ) ? 4 : 1; }
void   put( const std::string& name, const void* p, size_t n ) {
  const std::string tmp = dir() + "/" + name + ".tmp", path = dir() + "/" + name;
  {
    std::ofstream f( tmp, std::ios::binary );
    f.write( static_cast<const char*>( p ), std::streamsize( n ) );
  }
  rename( tmp.c_str(), path.c_str() );
}
bool get( const Comm* c, const std::string& name, void* p, size_t n ) {
  const auto limit = std::chrono::steady_clock::now() + std::chrono::seconds( 60 );
  for ( ;; ) {
    if ( c->aborted.load() ) return false;
    std::ifstream f( dir() + "/" + name, std::ios::binary );
    if ( f && f.read( static_cast<char*>( p ), std::streamsize( n ) ) ) return true;
    if ( std::chrono::steady_clock::now() > limit ) return false;
    std::this_thread::sleep_for( std::chrono::milliseconds( 2 ) );
  }
}
int flush( Comm* c ) {  // sends first (they only write), then the receives
  const long s = c->seq++;
  for ( const auto& x : c->pending )
    if ( x.send ) put( "p2p_" + std::to_string( s ) + "_" + std::to_string( c->rank ) + "_" + std::to_string( x.peer ), x.buf, x.bytes );
  for ( const auto& x : c->pending )
    if ( !x.send && !get( c, "p2p_" + std::to_string( s ) + "_" + std::to_string( x.peer ) + "_" + std::to_string( c->rank ), x.buf, x.bytes ) ) return 1;
  c->pending.clear();
  return 0;
}
Comm* g_last = nullptr;  // (ncclGroupStart / End carry no communicator)
}  // namespace

extern "C" {
int ncclGetUniqueId( Id* id ) {
  memset( id->internal, 0, sizeof( id->internal ) );
  snprintf( id->internal, sizeof( id->internal ), "mock-rccl-id" );
  return 0;
}
int ncclCommInitRank( void** comm, int world, Id id, int rank ) {
  if ( strcmp( id.internal, "mock-rccl-id" ) != 0 ) return 5;
  Comm* c = new Comm();
  c->rank = rank, c->world = world;
  *comm   = c;
  g_last  = c;
  log( c, "init " + std::to_string( rank ) + " " + std::to_string( world ) );
  return 0;
}
int ncclCommDestroy( void* comm ) {
  log( static_cast<Comm*>( comm ), "destroy" );
  if ( g_last == comm ) g_last = nullptr;
  delete static_cast<Comm*>( comm );
  return 0;
}
int ncclBroadcast( const void* send, void* recv, size_t count, int type, int root, void* comm, void* ) {
  Comm*        c = static_cast<Comm*>( comm );
  const long   s = c->seq++;
  const size_t n = count * width( type );
  log( c, "broadcast " + std::to_string( n ) + " root " + std::to_string( root ) );
  if ( c->rank == root ) {
    put( "bcast_" + std::to_string( s ), send, n );
    if ( recv != send ) memcpy( recv, send, n );
    return 0;
  }
  return get( c, "bcast_" + std::to_string( s ), recv, n ) ? 0 : 1;
}
int ncclAllReduce( const void* send, void* recv, size_t count, int type, int op, void* comm, void* ) {
  Comm*      c = static_cast<Comm*>( comm );
  const long s = c->seq++;
  log( c, "allreduce " + std::to_string( count * width( type ) ) + " op " + std::to_string( op ) );
  if ( type != 2 || op != 2 || count < 1 || count > 16 ) return 4;  // (the runner reduces a few int32 with max)
  int32_t mine[16], best[16];
  memcpy( mine, send, 4 * count ), memcpy( best, send, 4 * count );
  put( "ar_" + std::to_string( s ) + "_" + std::to_string( c->rank ), mine, 4 * count );
  for ( int r = 0; r < c->world; ++r ) {
    int32_t v[16];
    if ( !get( c, "ar_" + std::to_string( s ) + "_" + std::to_string( r ), v, 4 * count ) ) return 1;
    for ( size_t k = 0; k < count; ++k ) best[k] = v[k] > best[k] ? v[k] : best[k];
  }
  memcpy( recv, best, 4 * count );
  return 0;
}
int ncclCommAbort( void* comm ) {  // (from another thread than the one that waits: the object stays, a mock may leak)
  Comm* c = static_cast<Comm*>( comm );
  log( c, "abort" );
  c->aborted.store( true );
  if ( g_last == c ) g_last = nullptr;
  return 0;
}
int ncclGroupStart() {
  if ( g_last ) g_last->grouped = true;
  return 0;
}
int ncclGroupEnd() {
  if ( !g_last ) return 0;
  g_last->grouped = false;
  log( g_last, "group " + std::to_string( g_last->pending.size() ) );
  return flush( g_last );
}
int ncclSend( const void* buf, size_t count, int type, int peer, void* comm, void* ) {
  Comm* c = static_cast<Comm*>( comm );
  log( c, "send " + std::to_string( count * width( type ) ) + " to " + std::to_string( peer ) );
  c->pending.push_back( {true, const_cast<void*>( buf ), count * width( type ), peer} );
  return c->grouped ? 0 : flush( c );
}
int ncclRecv( void* buf, size_t count, int type, int peer, void* comm, void* ) {
  Comm* c = static_cast<Comm*>( comm );
  log( c, "recv " + std::to_string( count * width( type ) ) + " from " + std::to_string( peer ) );
  c->pending.push_back( {false, buf, count * width( type ), peer} );
  return c->grouped ? 0 : flush( c );
}
const char* ncclGetErrorString( int rc ) { return rc == 1 ? "mock: peer never arrived" : rc == 4 ? "mock: unexpected arguments" : "mock error"; }
}
