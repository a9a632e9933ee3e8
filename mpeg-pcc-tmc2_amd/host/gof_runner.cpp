// gof_runner.cpp -- libtmc2gof.so: the frame loop of a GOF pass as native host code over the C-ABI of libtmc2hip.so
// (include/tmc2gof.h).  No device code and no HIP call here: threads, the call sequence of INTEGRATION.md section 4, one rendezvous.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <functional>

#include <chrono>
#include <exception>

#include "tmc2gof.h"

namespace {
// the message of the LAST call of the calling thread (two GOFs encoded from two threads do not overwrite each other's; a pass
// collects the message of whichever of its slot threads failed first and hands it to its caller's thread on return)
thread_local std::string t_err;

// One core per slot, round-robin over the last-level caches (SMT siblings dropped); empty if the topology cannot be read.
// The mask is the PROCESS' (its main thread's), not the calling thread's: a caller that has pinned itself to one core (a worker of
// another pool) must not pin every slot thread of every later pass to that core.  Reading the topology costs milliseconds, so the
// list is kept per mask.
std::vector<int> coresOf( const cpu_set_t& allowed );
std::vector<int> coresByCacheDomain() {
  cpu_set_t allowed;
  CPU_ZERO( &allowed );
  if ( sched_getaffinity( getpid(), sizeof( allowed ), &allowed ) != 0 ) return {};
  static std::mutex       lock;
  static cpu_set_t        cachedMask;
  static std::vector<int> cached;
  static bool             have = false;
  std::lock_guard<std::mutex> g( lock );
  if ( !have || !CPU_EQUAL( &allowed, &cachedMask ) ) cached = coresOf( allowed ), cachedMask = allowed, have = true;
  return cached;
}
std::vector<int> coresOf( const cpu_set_t& allowed ) {
  std::map<std::string, std::vector<int>> domains;
  for ( int cpu = 0; cpu < CPU_SETSIZE; ++cpu ) {
    if ( !CPU_ISSET( cpu, &allowed ) ) continue;
    const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string( cpu ) + "/";
    std::ifstream     sib( base + "topology/thread_siblings_list" ), l3( base + "cache/index3/shared_cpu_list" );
    std::string       s, d;
    if ( !std::getline( sib, s ) || !std::getline( l3, d ) ) return {};
    if ( std::atoi( s.c_str() ) != cpu ) continue;
    domains[d].push_back( cpu );
  }
  std::vector<int> order;
  for ( size_t k = 0;; ++k ) {
    bool any = false;
    for ( auto& kv : domains )
      if ( k < kv.second.size() ) order.push_back( kv.second[k] ), any = true;
    if ( !any ) break;
  }
  return order;
}

// the CTC lossy settings (cfg/common/ctc-common.cfg + the sequence's), as tmc2_amd.lib.ctc_params
tmc2_segmenter_params ctcParams( const tmc2_gof_config& c, const double w[3] ) {
  tmc2_segmenter_params p{};
  p.nnNormalEstimation = 16, p.normalOrientation = 1, p.gridBasedRefineSegmentation = 1, p.maxNNCountRefineSegmentation = 1024;
  p.iterationCountRefineSegmentation = c.iterationCountRefineSegmentation;
  p.voxelDimensionRefineSegmentation = c.voxelDimensionRefineSegmentation, p.searchRadiusRefineSegmentation = 192;
  p.occupancyResolution = 16, p.enablePatchSplitting = 1, p.maxPatchSize = 1024, p.quantizerSizeX = 16, p.quantizerSizeY = 16;
  p.minPointCountPerCCPatchSegmentation = 16, p.maxNNCountPatchSegmentation = 16, p.surfaceThickness = 4, p.mapCountMinus1 = 1;
  p.minLevel = 64, p.maxAllowedDepth = 255, p.geometryBitDepth2D = 8, p.geometryBitDepth3D = c.geometryBitDepth3D;
  p.maxAllowedDist2RawPointsDetection = 9, p.maxAllowedDist2RawPointsSelection = 1, p.lambdaRefineSegmentation = 3;
  for ( int k = 0; k < 3; ++k ) p.weightNormal[k] = w[k];
  return p;
}

// The slot threads of the passes: parked between passes instead of being born and joined twice per pass (32 births per 180 ms GOF
// with sixteen slots).  A pass LEASES as many as it has slots -- two passes of two caller threads never share a thread -- and hands
// them back when its phase is over; the set only grows.  Never destroyed (a thread parked in a condition variable whose owner's
// destructor runs at exit is the classic way to hang a process on its way out).
class SlotThreads {
  struct Worker {
    std::mutex              lock;
    std::condition_variable wake;
    std::function<void()>   job;
    bool                    busy = false;
    int                     pinned = -1;
    std::thread             thread;
  };
  std::mutex                           lock_;
  std::vector<std::unique_ptr<Worker>> all_;
  std::vector<Worker*>                 idle_;
  static void loop( Worker* w ) {
    for ( ;; ) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> g( w->lock );
        w->wake.wait( g, [&] { return bool( w->job ); } );
        job.swap( w->job );
      }
      job();
    }
  }

 public:
  static SlotThreads& instance() {
    static SlotThreads* pool = new SlotThreads();  // (leaked on purpose: see above)
    return *pool;
  }
  // fn( slot ) on `slots` threads side by side, slot s pinned to cores[s % cores.size()]; returns when all are done
  void run( int slots, const std::vector<int>& cores, const std::function<void( int )>& fn ) {
    std::vector<Worker*> mine;
    {
      std::lock_guard<std::mutex> g( lock_ );
      while ( int( mine.size() ) < slots ) {
        if ( idle_.empty() ) {
          all_.emplace_back( new Worker() );
          Worker* w = all_.back().get();
          w->thread = std::thread( loop, w );
          w->thread.detach();
          idle_.push_back( w );
        }
        mine.push_back( idle_.back() );
        idle_.pop_back();
      }
    }
    std::mutex              doneLock;
    std::condition_variable doneCv;
    int                     left = slots;
    for ( int sl = 0; sl < slots; ++sl ) {
      Worker*   w    = mine[size_t( sl )];
      const int core = cores.empty() ? -1 : cores[size_t( sl ) % cores.size()];
      std::lock_guard<std::mutex> g( w->lock );
      w->job = [&, w, sl, core] {
        if ( core >= 0 && w->pinned != core ) {
          cpu_set_t one;
          CPU_ZERO( &one );
          CPU_SET( core, &one );
          (void)pthread_setaffinity_np( pthread_self(), sizeof( one ), &one );
          w->pinned = core;
        }
        try {
          fn( sl );
        } catch ( ... ) {  // (fn reports through the pass; nothing may unwind a parked thread)
        }
        std::lock_guard<std::mutex> d( doneLock );
        if ( --left == 0 ) doneCv.notify_one();
      };
      w->wake.notify_one();
    }
    {
      std::unique_lock<std::mutex> d( doneLock );
      doneCv.wait( d, [&] { return left == 0; } );
    }
    std::lock_guard<std::mutex> g( lock_ );
    for ( Worker* w : mine ) idle_.push_back( w );
  }
};

struct Pass {
  std::atomic<int> status{TMC2_OK};
  std::mutex       lock;
  std::string      message;
  void             fail( int rc, const char* what, const char* detail = nullptr ) {
    int expected = TMC2_OK;
    if ( status.compare_exchange_strong( expected, rc ) ) {
      std::lock_guard<std::mutex> g( lock );
      message = std::string( what ) + ": " + ( detail ? detail : tmc2_last_error() );  // (the failing thread's own message)
    }
  }
  int done() {  // on the calling thread: the pass' status, its message for tmc2_gof_last_error()
    std::lock_guard<std::mutex> g( lock );
    t_err = message;
    return status.load();
  }
};
#define GOF_TRY( call )                               \
  do {                                                \
    const int rc_ = ( call );                         \
    if ( rc_ != TMC2_OK ) {                           \
      pass.fail( rc_, #call );                        \
      return;                                         \
    }                                                 \
  } while ( 0 )
}  // namespace

extern "C" const char* tmc2_gof_last_error( void ) { return t_err.c_str(); }


// ---- the ranks of a sharded GOF (one process per GPU): RCCL, loaded at run time --------------------------------------------
// (librccl.so is looked up when a communicator is made, not when this library is loaded: a single-GPU host needs no RCCL.  The
//  test tier loads a recorder in its place through TMC2_RCCL_LIBRARY, tests/mock/mock_rccl.cpp.)
namespace {
struct NcclId {
  char internal[128];
};
enum { kNcclUint8 = 1, kNcclInt32 = 2, kNcclFloat64 = 8, kNcclMax = 2 };
struct Rccl {
  void* lib = nullptr;
  int ( *GetUniqueId )( NcclId* )                                                         = nullptr;
  int ( *CommInitRank )( void**, int, NcclId, int )                                       = nullptr;
  int ( *CommDestroy )( void* )                                                           = nullptr;
  int ( *CommAbort )( void* )                                                             = nullptr;  // (optional)
  int ( *Broadcast )( const void*, void*, size_t, int, int, void*, void* )                = nullptr;
  int ( *AllReduce )( const void*, void*, size_t, int, int, void*, void* )                = nullptr;
  int ( *Send )( const void*, size_t, int, int, void*, void* )                            = nullptr;
  int ( *Recv )( void*, size_t, int, int, void*, void* )                                  = nullptr;
  int ( *GroupStart )()                                                                   = nullptr;
  int ( *GroupEnd )()                                                                     = nullptr;
  const char* ( *GetErrorString )( int )                                                  = nullptr;
  bool load( std::string& why ) {
    const char* named = getenv( "TMC2_RCCL_LIBRARY" );
    for ( const char* name : {named, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"} ) {
      if ( !name ) continue;
      lib = dlopen( name, RTLD_NOW | RTLD_LOCAL );
      if ( lib ) break;
      why = dlerror();
    }
    if ( !lib ) return false;
    const auto sym = [&]( const char* n, bool required = true ) {
      void* p = dlsym( lib, n );
      if ( !p && required ) why = std::string( "librccl: no symbol " ) + n;
      return p;
    };
    GetUniqueId    = reinterpret_cast<decltype( GetUniqueId )>( sym( "ncclGetUniqueId" ) );
    CommInitRank   = reinterpret_cast<decltype( CommInitRank )>( sym( "ncclCommInitRank" ) );
    CommDestroy    = reinterpret_cast<decltype( CommDestroy )>( sym( "ncclCommDestroy" ) );
    CommAbort      = reinterpret_cast<decltype( CommAbort )>( sym( "ncclCommAbort", false ) );
    Broadcast      = reinterpret_cast<decltype( Broadcast )>( sym( "ncclBroadcast" ) );
    AllReduce      = reinterpret_cast<decltype( AllReduce )>( sym( "ncclAllReduce" ) );
    Send           = reinterpret_cast<decltype( Send )>( sym( "ncclSend" ) );
    Recv           = reinterpret_cast<decltype( Recv )>( sym( "ncclRecv" ) );
    GroupStart     = reinterpret_cast<decltype( GroupStart )>( sym( "ncclGroupStart" ) );
    GroupEnd       = reinterpret_cast<decltype( GroupEnd )>( sym( "ncclGroupEnd" ) );
    GetErrorString = reinterpret_cast<decltype( GetErrorString )>( sym( "ncclGetErrorString" ) );
    return GetUniqueId && CommInitRank && CommDestroy && Broadcast && AllReduce && Send && Recv && GroupStart && GroupEnd && GetErrorString;
  }
};
}  // namespace

struct tmc2_gof_comm {
  int       rank = 0, world = 1;
  tmc2_ctx* ctx  = nullptr;  // a context on this rank's device: its stream carries the collectives
  Rccl      rccl;
  void*     comm    = nullptr;
  void*     dSmall  = nullptr;  // 64 bytes on the device: weights, heights, headers
  void*     dBlocks = nullptr;  // the records of this rank's frames, then (rank 0) those of every rank
  size_t    blocksBytes = 0;
  int64_t   passes  = 0;        // passes this communicator has carried (the ranks call them in lock-step: every block carries it)
  bool      broken  = false;    // a collective failed or timed out: the communicator was aborted, every later call is refused
  std::mutex abortLock;         // (the watchdog's thread and the calling thread both may abort: once)
  double    timeoutSeconds = 600.0;
  std::vector<uint8_t> hostBlocks;
};

namespace {
void breakComm( tmc2_gof_comm* comm ) {  // a collective failed: nobody may be left waiting for this rank's part of it, no later call may use it
  std::lock_guard<std::mutex> g( comm->abortLock );
  if ( comm->comm && comm->rccl.CommAbort ) (void)comm->rccl.CommAbort( comm->comm ), comm->comm = nullptr;
  comm->broken = true;
}
// A rank that waits in a collective for a rank that will never arrive (it died, or left the pass through a path that skipped the
// collective) would wait for ever -- in the call itself (a group's connection set-up) or in the stream synchronisation behind it.
// Every collective of this file therefore runs under a watchdog: when it is not over after comm->timeoutSeconds
// (TMC2_GOF_COLLECTIVE_TIMEOUT, default 600 s: far beyond any pass) the communicator is ABORTED from the watchdog's thread
// (ncclCommAbort: pending calls and kernels give up, the synchronisation returns) and the call fails with TMC2_E_STATE.
class Watchdog {
  tmc2_gof_comm*          comm_;
  std::mutex              lock_;
  std::condition_variable cv_;
  bool                    done_ = false, fired_ = false;
  std::thread             thread_;

 public:
  explicit Watchdog( tmc2_gof_comm* comm ) : comm_( comm ) {
    if ( !comm_->rccl.CommAbort || comm_->timeoutSeconds <= 0.0 ) return;
    thread_ = std::thread( [this] {
      std::unique_lock<std::mutex> g( lock_ );
      if ( cv_.wait_for( g, std::chrono::duration<double>( comm_->timeoutSeconds ), [this] { return done_; } ) ) return;
      fired_ = true;
      g.unlock();
      breakComm( comm_ );
    } );
  }
  bool finish() {  // -> true if the watchdog had to abort the communicator
    if ( thread_.joinable() ) {
      {
        std::lock_guard<std::mutex> g( lock_ );
        done_ = true;
      }
      cv_.notify_one();
      thread_.join();
    }
    return fired_;
  }
  ~Watchdog() { (void)finish(); }
};
#define COMM_TRY( call, what )                                                                              \
  do {                                                                                                      \
    const int rc_ = ( call );                                                                               \
    if ( rc_ != 0 ) {                                                                                       \
      t_err = std::string( what ) + ": " + ( comm->rccl.GetErrorString ? comm->rccl.GetErrorString( rc_ ) : "?" ); \
      breakComm( comm );                                                                                    \
      return TMC2_E_HIP;                                                                                    \
    }                                                                                                       \
  } while ( 0 )
#define HIP_TRY( call, what )                              \
  do {                                                     \
    const int rc_ = ( call );                              \
    if ( rc_ != TMC2_OK ) {                                \
      t_err = std::string( what ) + ": " + tmc2_last_error(); \
      return rc_;                                          \
    }                                                      \
  } while ( 0 )
int refuseBroken( tmc2_gof_comm* comm ) {
  if ( !comm->broken ) return TMC2_OK;
  t_err = "tmc2_gof_comm: the communicator was aborted after a collective failed or timed out; make a new one";
  return TMC2_E_STATE;
}
template <typename F>
int underWatchdog( tmc2_gof_comm* comm, const char* what, F&& body ) {
  if ( const int rc = refuseBroken( comm ) ) return rc;
  Watchdog  dog( comm );
  const int rc = body();
  if ( dog.finish() ) {
    t_err = std::string( what ) + ": no answer from the other ranks within " + std::to_string( int( comm->timeoutSeconds ) ) +
            " s (TMC2_GOF_COLLECTIVE_TIMEOUT): the communicator was aborted";
    return TMC2_E_STATE;
  }
  return rc;
}
// the wait behind a collective.  host == nullptr: only the synchronisation
int waitFor( tmc2_gof_comm* comm, void* host, const void* device, size_t bytes, const char* what ) {
  const int rc = host ? tmc2_ctx_download( comm->ctx, host, device, bytes ) : tmc2_ctx_synchronize( comm->ctx );
  if ( rc != TMC2_OK ) t_err = std::string( what ) + ": " + tmc2_last_error();
  return rc;
}

// One small collective on device words: every rank ALWAYS issues it -- a rank whose upload failed still takes part (with whatever
// the buffer holds) and reports its own failure afterwards, so that nobody waits for it; the stage behind it carries the failure.
// 24 bytes from rank 0 to everybody (the axis weights of frame 0: PCCEncoder::calculateWeightNormal runs on the first frame only)
int commBroadcastBody( tmc2_gof_comm* comm, void* host, size_t count, int type, size_t width, const char* what ) {
  void*             st = tmc2_ctx_stream( comm->ctx );
  const int         up = tmc2_ctx_upload( comm->ctx, comm->dSmall, host, count * width );
  const std::string upError = up != TMC2_OK ? std::string( what ) + ": upload: " + tmc2_last_error() : std::string();
  COMM_TRY( comm->rccl.Broadcast( comm->dSmall, comm->dSmall, count, type, 0, comm->comm, st ), what );
  const int rc = waitFor( comm, host, comm->dSmall, count * width, what );
  if ( up != TMC2_OK ) {
    t_err = upError;
    return up;
  }
  return rc;
}
int commBroadcast( tmc2_gof_comm* comm, void* host, size_t count, int type, size_t width, const char* what ) {
  return underWatchdog( comm, what, [&] { return commBroadcastBody( comm, host, count, type, width, what ); } );
}
int commBroadcastWeights( tmc2_gof_comm* comm, double w[3] ) { return commBroadcast( comm, w, 3, kNcclFloat64, 8, "ncclBroadcast( weights )" ); }
// max over the ranks of `count` int32 words (count <= 16)
int commMaxBody( tmc2_gof_comm* comm, int32_t* words, size_t count, const char* what ) {
  void*             st = tmc2_ctx_stream( comm->ctx );
  const int         up = tmc2_ctx_upload( comm->ctx, comm->dSmall, words, 4 * count );
  const std::string upError = up != TMC2_OK ? std::string( what ) + ": upload: " + tmc2_last_error() : std::string();
  COMM_TRY( comm->rccl.AllReduce( comm->dSmall, comm->dSmall, count, kNcclInt32, kNcclMax, comm->comm, st ), what );
  const int rc = waitFor( comm, words, comm->dSmall, 4 * count, what );
  if ( up != TMC2_OK ) {
    t_err = upError;
    return up;
  }
  return rc;
}
int commMax( tmc2_gof_comm* comm, int32_t* words, size_t count, const char* what ) {
  return underWatchdog( comm, what, [&] { return commMaxBody( comm, words, count, what ); } );
}
// the canvas height of the GOF: max over the ranks of what their frames packed into (resizeGeometryVideo, PCCEncoder.cpp:5546-5591)
int commMaxHeight( tmc2_gof_comm* comm, int32_t* h ) { return commMax( comm, h, 1, "ncclAllReduce( height, max )" ); }

int ensureBlocks( tmc2_gof_comm* comm, size_t need ) {
  if ( need <= comm->blocksBytes ) return TMC2_OK;
  if ( comm->dBlocks ) HIP_TRY( tmc2_ctx_device_free( comm->ctx, comm->dBlocks ), "records: free" );
  comm->dBlocks = nullptr, comm->blocksBytes = 0;
  HIP_TRY( tmc2_ctx_device_alloc( comm->ctx, need, &comm->dBlocks ), "records: device buffer" );
  comm->blocksBytes = need;
  return TMC2_OK;
}
// every rank's block of `mine` bytes (hostBlocks[0 .. mine)) to rank 0 (hostBlocks[mine * (r + 1) ..)): one grouped send / receive
int commBlocksToRootBody( tmc2_gof_comm* comm, size_t mine, const char* what ) {
  void*     st  = tmc2_ctx_stream( comm->ctx );
  uint8_t*  dev = static_cast<uint8_t*>( comm->dBlocks );
  const int up  = tmc2_ctx_upload( comm->ctx, dev, comm->hostBlocks.data(), mine );  // (failed: a stale block travels; its pass number gives it away)
  const std::string upError = up != TMC2_OK ? std::string( what ) + ": upload: " + tmc2_last_error() : std::string();
  COMM_TRY( comm->rccl.GroupStart(), "ncclGroupStart" );
  if ( comm->rank == 0 )
    for ( int r = 0; r < comm->world; ++r ) COMM_TRY( comm->rccl.Recv( dev + mine * size_t( r + 1 ), mine, kNcclUint8, r, comm->comm, st ), what );
  COMM_TRY( comm->rccl.Send( dev, mine, kNcclUint8, 0, comm->comm, st ), what );
  COMM_TRY( comm->rccl.GroupEnd(), "ncclGroupEnd" );
  const int rc = comm->rank == 0 ? waitFor( comm, comm->hostBlocks.data() + mine, dev + mine, mine * size_t( comm->world ), what )
                                 : waitFor( comm, nullptr, nullptr, 0, what );
  if ( up != TMC2_OK ) {
    t_err = upError;
    return up;
  }
  return rc;
}
int commBlocksToRoot( tmc2_gof_comm* comm, size_t mine, const char* what ) {
  return underWatchdog( comm, what, [&] { return commBlocksToRootBody( comm, mine, what ); } );
}
// rank 0's block for rank r (hostBlocks[mine * r ..), r = 0 .. world-1) to rank r (hostBlocks[0 .. mine)): the way back
int commBlocksFromRootBody( tmc2_gof_comm* comm, size_t mine, const char* what ) {
  void*     st  = tmc2_ctx_stream( comm->ctx );
  uint8_t*  dev = static_cast<uint8_t*>( comm->dBlocks );
  int       up  = TMC2_OK;
  std::string upError;
  if ( comm->rank == 0 ) {
    up = tmc2_ctx_upload( comm->ctx, dev + mine, comm->hostBlocks.data(), mine * size_t( comm->world ) );
    if ( up != TMC2_OK ) upError = std::string( what ) + ": upload: " + tmc2_last_error();
  }
  COMM_TRY( comm->rccl.GroupStart(), "ncclGroupStart" );
  if ( comm->rank == 0 )
    for ( int r = 0; r < comm->world; ++r ) COMM_TRY( comm->rccl.Send( dev + mine * size_t( r + 1 ), mine, kNcclUint8, r, comm->comm, st ), what );
  COMM_TRY( comm->rccl.Recv( dev, mine, kNcclUint8, 0, comm->comm, st ), what );
  COMM_TRY( comm->rccl.GroupEnd(), "ncclGroupEnd" );
  const int rc = waitFor( comm, comm->hostBlocks.data(), dev, mine, what );
  if ( up != TMC2_OK ) {
    t_err = upError;
    return up;
  }
  return rc;
}

int commBlocksFromRoot( tmc2_gof_comm* comm, size_t mine, const char* what ) {
  return underWatchdog( comm, what, [&] { return commBlocksFromRootBody( comm, mine, what ); } );
}

constexpr size_t roundUp( size_t n, size_t to ) { return ( n + to - 1 ) / to * to; }

// The final gather: the packed patch records of every frame (the side information the bitstream carries: ~ 100 bytes a patch) to
// rank 0 -- one grouped send / receive per pass.  Block of a frame: int64 count, int64 pass number, then recordSlots records in
// list order.
int commGatherRecords( tmc2_gof_comm* comm, tmc2_frame** frames, int32_t count, int32_t recordSlots, tmc2_patch* gathered,
                       int64_t* gatheredCounts, int failed ) {
  const size_t frameBytes = 16 + size_t( recordSlots ) * sizeof( tmc2_patch ), mine = frameBytes * size_t( count );
  const size_t need       = mine * ( comm->rank == 0 ? size_t( comm->world ) + 1 : 1 );
  int         localStatus = failed;
  std::string localError  = failed != TMC2_OK ? t_err : std::string();
  if ( const int rc = ensureBlocks( comm, need ); rc != TMC2_OK && localStatus == TMC2_OK ) localStatus = rc, localError = t_err;
  comm->hostBlocks.assign( need, 0 );
  std::vector<tmc2_patch> list;
  std::vector<int32_t>    order;
  // (A rank that cannot fill its block -- its pass failed after the rendezvous, or a frame has more patches than the block holds --
  //  STILL takes part in the exchange, with a negative count in the block: the other ranks must not be left waiting in a receive.)
  for ( int i = 0; i < count; ++i ) {
    uint8_t* at  = comm->hostBlocks.data() + frameBytes * size_t( i );
    int64_t  n64 = -1;
    if ( localStatus == TMC2_OK ) {
      const int n = tmc2_frame_patch_count( frames[i] );
      if ( n < 0 || n > recordSlots ) {
        localStatus = TMC2_E_INVALID;
        localError  = "tmc2_gof_encode_sharded: a frame with " + std::to_string( n ) + " patches, the gather holds " + std::to_string( recordSlots );
      } else {
        list.resize( size_t( n ) ), order.resize( size_t( n ) );
        int rc = tmc2_frame_get_patches( frames[i], list.data(), nullptr, nullptr, nullptr );
        if ( rc == TMC2_OK ) rc = tmc2_frame_get_patch_order( frames[i], order.data() );
        if ( rc != TMC2_OK ) {
          localStatus = rc, localError = std::string( "tmc2_frame_get_patches: " ) + tmc2_last_error();
        } else {
          n64 = n;
          for ( int k = 0; k < n; ++k ) memcpy( at + 16 + size_t( k ) * sizeof( tmc2_patch ), &list[size_t( order[size_t( k )] )], sizeof( tmc2_patch ) );
        }
      }
    }
    memcpy( at, &n64, 8 );
    memcpy( at + 8, &comm->passes, 8 );
  }
  if ( comm->blocksBytes < need ) {  // (no device buffer: this rank cannot take part at all -- the others must not wait for it)
    breakComm( comm );
    t_err = localError;
    return localStatus;
  }
  const int rc = commBlocksToRoot( comm, mine, "ncclSend / ncclRecv( records )" );
  if ( localStatus != TMC2_OK ) {
    t_err = localError;
    return localStatus;
  }
  if ( rc != TMC2_OK || comm->rank != 0 ) return rc;
  for ( int r = 0; r < comm->world; ++r )
    for ( int i = 0; i < count; ++i ) {
      const uint8_t* at = comm->hostBlocks.data() + mine * size_t( r + 1 ) + frameBytes * size_t( i );
      int64_t        n = 0, pass = 0;
      memcpy( &n, at, 8 ), memcpy( &pass, at + 8, 8 );
      if ( n < 0 || n > recordSlots || pass != comm->passes ) {
        t_err = "tmc2_gof_encode_sharded: rank " + std::to_string( r ) + " could not deliver the records of its frame " + std::to_string( i ) +
                " (its own call says why)";
        return TMC2_E_STATE;
      }
      if ( gatheredCounts ) gatheredCounts[size_t( r ) * size_t( count ) + size_t( i )] = n;
      if ( gathered )
        memcpy( gathered + ( size_t( r ) * size_t( count ) + size_t( i ) ) * size_t( recordSlots ), at + 16, size_t( n ) * sizeof( tmc2_patch ) );
    }
  return TMC2_OK;
}

// ---- the packing chains of a sharded GOF (low-delay: spatialConsistencyPackFlexible, PCCEncoder.cpp:1183-1412; random access: +
// performDataAdaptiveGPAMethod, :6821-6971).  They run over ALL frames of the GOF in frame order, microseconds per frame, on patch
// RECORDS ("gather to rank 0 of the per-frame patch table before packing"): every rank sends the records and
// block-occupancy pools of its frames to rank 0 (one grouped send / receive; the block size from an all-reduce of the largest
// frame), rank 0 runs PCCEncoder::placeSegments over them (tmc2_host_place_segments: no device), and every rank gets the packed
// lists of ITS frames back (a 32-byte header broadcast with the canvas and the block size, one grouped send / receive) and
// installs them (tmc2_frame_set_packing).  Rank 0 is left with every frame's records in list order: no gather at the end.
constexpr int32_t kFailedWord = 0x7FFFFFF0;
struct ChainResult {
  int32_t W = 0, H = 0;
  // per frame of this rank: the packed list, matches, pool, tile size
  struct Frame {
    std::vector<tmc2_patch> list;
    std::vector<int32_t>    matches;
    std::vector<uint8_t>    occupancy;
    int32_t                 packedW = 0, packedH = 0;
  };
  std::vector<Frame> frames;
};
int chainOverRanks( tmc2_gof_comm* comm, tmc2_frame** frames, int32_t count, const tmc2_gof_config& c, int32_t recordSlots,
                    tmc2_patch* gathered, int64_t* gatheredCounts, int failed, ChainResult& out ) {
  const int world = comm->world;
  int         localStatus = failed;
  std::string localError  = failed != TMC2_OK ? t_err : std::string();
  auto        localFail   = [&]( int rc, const std::string& why ) {
    if ( localStatus == TMC2_OK ) localStatus = rc, localError = why;
  };
  // (1) what this rank's frames hold; the largest frame of the GOF sizes everybody's blocks
  std::vector<std::vector<tmc2_patch>> recs( static_cast<size_t>( count ) );
  std::vector<std::vector<uint8_t>>    pools( static_cast<size_t>( count ) );
  int32_t sizes[3] = {0, 0, 0};  // failed?, most patches of a frame, largest pool of a frame
  for ( int i = 0; i < count && localStatus == TMC2_OK; ++i ) {
    const int n = tmc2_frame_patch_count( frames[i] );
    int64_t   depth = 0, occ = 0;
    if ( n < 0 || tmc2_frame_patch_pool_sizes( frames[i], &depth, &occ ) != TMC2_OK || occ < 0 || occ > 0x3FFFFFFF ) {
      localFail( TMC2_E_STATE, std::string( "tmc2_frame_patch_pool_sizes: " ) + tmc2_last_error() );
      break;
    }
    recs[size_t( i )].resize( size_t( n ) ), pools[size_t( i )].resize( size_t( occ ) );
    if ( const int rc = tmc2_frame_get_patches( frames[i], recs[size_t( i )].data(), nullptr, nullptr, pools[size_t( i )].data() ) ) {
      localFail( rc, std::string( "tmc2_frame_get_patches: " ) + tmc2_last_error() );
      break;
    }
    sizes[1] = std::max( sizes[1], int32_t( n ) ), sizes[2] = std::max( sizes[2], int32_t( occ ) );
  }
  if ( localStatus != TMC2_OK ) sizes[0] = kFailedWord;
  {
    const int rc = commMax( comm, sizes, 3, "ncclAllReduce( records of the largest frame, max )" );
    if ( rc != TMC2_OK ) return localStatus != TMC2_OK ? ( t_err = localError, localStatus ) : rc;
  }
  if ( sizes[0] != 0 ) {  // (every rank leaves here, together)
    if ( localStatus != TMC2_OK ) return t_err = localError, localStatus;
    t_err = "tmc2_gof_encode_sharded: another rank's frames failed before the packing chain (its own call says why)";
    return TMC2_E_STATE;
  }
  if ( sizes[1] > recordSlots ) {
    t_err = "tmc2_gof_encode_sharded: a frame with " + std::to_string( sizes[1] ) + " patches, the records hold " + std::to_string( recordSlots );
    return TMC2_E_INVALID;
  }
  const size_t slotsN = size_t( std::max( sizes[1], 1 ) ), poolN = roundUp( size_t( sizes[2] ), 64 );
  const size_t inFrame = 32 + slotsN * sizeof( tmc2_patch ) + poolN, inMine = inFrame * size_t( count );
  // (2) records + pools -> rank 0
  int rc = ensureBlocks( comm, inMine * ( comm->rank == 0 ? size_t( world ) + 1 : 1 ) );
  if ( rc != TMC2_OK ) {  // (cannot take part: the others must not wait)
    breakComm( comm );
    return rc;
  }
  comm->hostBlocks.assign( inMine * ( comm->rank == 0 ? size_t( world ) + 1 : 1 ), 0 );
  for ( int i = 0; i < count; ++i ) {
    uint8_t*      at = comm->hostBlocks.data() + inFrame * size_t( i );
    const int64_t hd[4] = {int64_t( recs[size_t( i )].size() ), int64_t( pools[size_t( i )].size() ), comm->passes, 0};
    memcpy( at, hd, 32 );
    memcpy( at + 32, recs[size_t( i )].data(), recs[size_t( i )].size() * sizeof( tmc2_patch ) );
    memcpy( at + 32 + slotsN * sizeof( tmc2_patch ), pools[size_t( i )].data(), pools[size_t( i )].size() );
  }
  rc = commBlocksToRoot( comm, inMine, "ncclSend / ncclRecv( records for the packing chain )" );
  if ( rc != TMC2_OK ) return rc;
  // (3) rank 0: PCCEncoder::placeSegments over the GOF in frame order (frame f = slot f / world of rank f mod world)
  const int            gofFrames = count * world;
  int32_t              header[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // status, W, H, bytes of the largest packed pool
  std::vector<int32_t> counts, matches, widths, heights;
  std::vector<tmc2_patch> all;
  std::vector<uint8_t>    occOut;
  std::vector<int64_t>    outBase;
  if ( comm->rank == 0 ) {
    std::vector<uint8_t> occIn;
    std::vector<int64_t> inBase;
    counts.resize( size_t( gofFrames ) ), inBase.resize( size_t( gofFrames ) ), outBase.resize( size_t( gofFrames ) + 1 );
    widths.resize( size_t( gofFrames ) ), heights.resize( size_t( gofFrames ) );
    int64_t maxU0 = 1, maxV0 = 1;
    bool    ok    = true;
    for ( int f = 0; f < gofFrames && ok; ++f ) {
      const uint8_t* at = comm->hostBlocks.data() + inMine * size_t( f % world + 1 ) + inFrame * size_t( f / world );
      int64_t        hd[4];
      memcpy( hd, at, 32 );
      if ( hd[0] < 0 || hd[0] > int64_t( slotsN ) || hd[1] < 0 || hd[1] > int64_t( poolN ) || hd[2] != comm->passes ) {
        ok = false;
        break;
      }
      counts[size_t( f )] = int32_t( hd[0] ), inBase[size_t( f )] = int64_t( occIn.size() );
      const tmc2_patch* p = reinterpret_cast<const tmc2_patch*>( at + 32 );
      all.insert( all.end(), p, p + hd[0] );
      occIn.insert( occIn.end(), at + 32 + slotsN * sizeof( tmc2_patch ), at + 32 + slotsN * sizeof( tmc2_patch ) + hd[1] );
      for ( int64_t k = 0; k < hd[0]; ++k ) maxU0 = std::max<int64_t>( maxU0, p[k].sizeU0 ), maxV0 = std::max<int64_t>( maxV0, p[k].sizeV0 );
    }
    if ( !ok ) {
      header[0] = kFailedWord;
      localFail( TMC2_E_STATE, "tmc2_gof_encode_sharded: a rank delivered a block of another pass or of impossible sizes" );
    } else {
      // (random access: a tracked patch grows to the box of its track's union -- never beyond the largest box of the GOF in either direction)
      const int64_t cap = c.packing == 2 ? int64_t( all.size() ) * maxU0 * maxV0 : int64_t( occIn.size() );
      occOut.assign( size_t( std::max<int64_t>( cap, 1 ) ), 0 ), matches.assign( std::max<size_t>( all.size(), 1 ), -1 );
      const int prc = tmc2_host_place_segments( gofFrames, counts.data(), all.data(), occIn.data(), inBase.data(), c.packing, c.minimumImageWidth,
                                                c.minimumImageHeight, 2, 1.0, matches.data(), occOut.data(), cap, outBase.data(), widths.data(),
                                                heights.data() );
      int32_t tileW = c.minimumImageWidth, tileH = 0;
      for ( int f = 0; f < gofFrames; ++f ) tileW = std::max( tileW, widths[size_t( f )] ), tileH = std::max( tileH, heights[size_t( f )] );
      if ( c.packing == 2 ) tileH = std::max( tileH, c.minimumImageHeight );
      int32_t   W = 0, H = 0;
      const int crc = prc != TMC2_OK ? prc : tmc2_encoder_canvas_size( &tileH, 1, tileW, c.minimumImageWidth, c.minimumImageHeight, &W, &H );
      if ( crc != TMC2_OK ) {
        header[0] = kFailedWord;
        localFail( crc, std::string( prc != TMC2_OK ? "tmc2_host_place_segments: " : "tmc2_encoder_canvas_size: " ) + tmc2_last_error() );
      } else {
        int64_t largest = 0;
        for ( int f = 0; f < gofFrames; ++f ) largest = std::max( largest, outBase[size_t( f ) + 1] - outBase[size_t( f )] );
        header[1] = W, header[2] = H, header[3] = int32_t( largest );
      }
    }
  }
  rc = commBroadcast( comm, header, 8, kNcclInt32, 4, "ncclBroadcast( canvas of the GOF )" );
  if ( rc != TMC2_OK ) return rc;
  if ( header[0] != 0 ) {
    if ( localStatus != TMC2_OK ) return t_err = localError, localStatus;
    t_err = "tmc2_gof_encode_sharded: the packing chain failed on rank 0 (its own call says why)";
    return TMC2_E_STATE;
  }
  // (4) the packed lists back to the ranks that hold the frames
  const size_t outPool = roundUp( size_t( header[3] ), 64 ), outFrame = 32 + slotsN * ( sizeof( tmc2_patch ) + 4 ) + outPool;
  const size_t outMine = outFrame * size_t( count );
  rc = ensureBlocks( comm, outMine * ( comm->rank == 0 ? size_t( world ) + 1 : 1 ) );
  if ( rc != TMC2_OK ) {
    breakComm( comm );
    return rc;
  }
  comm->hostBlocks.assign( outMine * ( comm->rank == 0 ? size_t( world ) : 1 ), 0 );
  if ( comm->rank == 0 ) {
    size_t first = 0;
    for ( int f = 0; f < gofFrames; ++f ) {
      uint8_t*      at = comm->hostBlocks.data() + outMine * size_t( f % world ) + outFrame * size_t( f / world );
      const size_t  n  = size_t( counts[size_t( f )] );
      const int64_t hd[4] = {int64_t( n ), outBase[size_t( f ) + 1] - outBase[size_t( f )],
                             ( int64_t( widths[size_t( f )] ) << 32 ) | uint32_t( heights[size_t( f )] ), comm->passes};
      memcpy( at, hd, 32 );
      memcpy( at + 32, all.data() + first, n * sizeof( tmc2_patch ) );
      memcpy( at + 32 + slotsN * sizeof( tmc2_patch ), matches.data() + first, n * 4 );
      memcpy( at + 32 + slotsN * ( sizeof( tmc2_patch ) + 4 ), occOut.data() + outBase[size_t( f )], size_t( hd[1] ) );
      const size_t slot = size_t( f % world ) * size_t( count ) + size_t( f / world );
      if ( gatheredCounts ) gatheredCounts[slot] = int64_t( n );
      if ( gathered ) memcpy( gathered + slot * size_t( recordSlots ), all.data() + first, n * sizeof( tmc2_patch ) );
      first += n;
    }
  }
  rc = commBlocksFromRoot( comm, outMine, "ncclSend / ncclRecv( packed lists )" );
  if ( rc != TMC2_OK ) return rc;
  out.W = header[1], out.H = header[2];
  out.frames.resize( size_t( count ) );
  for ( int i = 0; i < count; ++i ) {
    const uint8_t* at = comm->hostBlocks.data() + outFrame * size_t( i );
    int64_t        hd[4];
    memcpy( hd, at, 32 );
    if ( hd[0] < 0 || hd[0] > int64_t( slotsN ) || hd[1] < 0 || hd[1] > int64_t( outPool ) || hd[3] != comm->passes ) {
      t_err = "tmc2_gof_encode_sharded: the packed list of frame " + std::to_string( i ) + " did not arrive";
      return TMC2_E_STATE;
    }
    ChainResult::Frame& fr = out.frames[size_t( i )];
    fr.list.resize( size_t( hd[0] ) ), fr.matches.resize( size_t( hd[0] ) ), fr.occupancy.resize( size_t( hd[1] ) );
    memcpy( fr.list.data(), at + 32, size_t( hd[0] ) * sizeof( tmc2_patch ) );
    memcpy( fr.matches.data(), at + 32 + slotsN * sizeof( tmc2_patch ), size_t( hd[0] ) * 4 );
    memcpy( fr.occupancy.data(), at + 32 + slotsN * ( sizeof( tmc2_patch ) + 4 ), size_t( hd[1] ) );
    fr.packedW = int32_t( hd[2] >> 32 ), fr.packedH = int32_t( hd[2] & 0xFFFFFFFF );
  }
  return TMC2_OK;
}

// Where rank 0 publishes the communicator's id when the caller names no file: /dev/shm/tmc2_gof_id_<uid>_$MASTER_PORT.  The port
// is what tells two jobs of one node apart, so several ranks without it are refused.
int defaultRendezvous( int worldSize, std::string& path ) {
  const char* port = getenv( "MASTER_PORT" );
  if ( !port && worldSize > 1 ) {
    t_err = "tmc2_gof_comm_create: no rendezvous file was named and MASTER_PORT is not set: two jobs on this node would read each other's id";
    return TMC2_E_INVALID;
  }
  path = "/dev/shm/tmc2_gof_id_" + std::to_string( long( getuid() ) ) + "_" + ( port ? port : "0" );
  return TMC2_OK;
}
// seconds since the epoch at which this process started (/proc/self/stat field 22 in clock ticks since boot + /proc/stat btime);
// 0 if it cannot be read
double processStart() {
  std::ifstream st( "/proc/self/stat" ), boot( "/proc/stat" );
  std::string   line;
  if ( !std::getline( st, line ) ) return 0.0;
  const size_t close = line.rfind( ')' );
  if ( close == std::string::npos ) return 0.0;
  unsigned long long ticks = 0;
  {
    const char* p = line.c_str() + close + 1;
    for ( int field = 3; field <= 22; ++field ) {
      while ( *p == ' ' ) ++p;
      if ( field == 22 ) ticks = strtoull( p, nullptr, 10 );
      while ( *p && *p != ' ' ) ++p;
    }
  }
  double btime = 0.0;
  while ( std::getline( boot, line ) )
    if ( line.compare( 0, 6, "btime " ) == 0 ) btime = atof( line.c_str() + 6 );
  const long hz = sysconf( _SC_CLK_TCK );
  return btime > 0.0 && hz > 0 ? btime + double( ticks ) / double( hz ) : 0.0;
}
constexpr char kIdMagic[8] = {'t', 'm', 'c', '2', 'i', 'd', '0', '1'};
}  // namespace

extern "C" int tmc2_gof_comm_create( int rank, int worldSize, tmc2_ctx* ctx, const char* rendezvous, tmc2_gof_comm** out ) {
  if ( !out ) return TMC2_E_INVALID;
  *out = nullptr;
  if ( rank < 0 || worldSize < 1 || rank >= worldSize || !ctx ) {
    t_err = "tmc2_gof_comm_create: invalid argument";
    return TMC2_E_INVALID;
  }
  // (every failure below goes through tmc2_gof_comm_destroy: the RCCL communicator, the device words and the library handle with it)
  std::unique_ptr<tmc2_gof_comm, void ( * )( tmc2_gof_comm* )> owner( new tmc2_gof_comm(), tmc2_gof_comm_destroy );
  tmc2_gof_comm*                                                comm = owner.get();
  comm->rank = rank, comm->world = worldSize, comm->ctx = ctx;
  if ( const char* t = getenv( "TMC2_GOF_COLLECTIVE_TIMEOUT" ) ) comm->timeoutSeconds = atof( t );
  std::string why;
  if ( !comm->rccl.load( why ) ) {
    t_err = "tmc2_gof_comm_create: RCCL not available (" + why + ")";
    return TMC2_E_UNSUPPORTED;
  }
  HIP_TRY( tmc2_ctx_make_current( ctx ), "tmc2_ctx_make_current" );
  // The 128-byte id of the communicator: rank 0 makes it and publishes it in a file (same node: /dev/shm), the others wait for it.
  // The file is created exclusively (O_EXCL | O_NOFOLLOW, mode 0600: no symbolic link is followed, nobody else's file is reused)
  // after whatever an earlier run left under the name has been removed, and renamed into place complete.  A reader accepts only a
  // complete file of ITS user that is not older than the reader's own process: what a crashed run of the same port left behind
  // was written before this job's processes were started.
  std::string path = rendezvous ? rendezvous : "";
  if ( path.empty() )
    if ( const int rc = defaultRendezvous( worldSize, path ) ) return rc;
  NcclId id{};
  if ( rank == 0 ) {
    COMM_TRY( comm->rccl.GetUniqueId( &id ), "ncclGetUniqueId" );
    if ( worldSize > 1 ) {
      const std::string tmp = path + ".tmp." + std::to_string( long( getpid() ) );
      (void)unlink( path.c_str() ), (void)unlink( tmp.c_str() );
      const int fd = open( tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600 );
      bool      ok = fd >= 0;
      ok = ok && write( fd, kIdMagic, sizeof( kIdMagic ) ) == ssize_t( sizeof( kIdMagic ) );
      ok = ok && write( fd, id.internal, sizeof( id.internal ) ) == ssize_t( sizeof( id.internal ) );
      if ( fd >= 0 ) ok = ( close( fd ) == 0 ) && ok;
      if ( !ok || rename( tmp.c_str(), path.c_str() ) != 0 ) {
        (void)unlink( tmp.c_str() );
        t_err = "tmc2_gof_comm_create: cannot publish the communicator id in " + path;
        return TMC2_E_INVALID;
      }
    }
  } else {
    const double started = processStart();
    const auto   limit   = std::chrono::steady_clock::now() + std::chrono::seconds( 120 );
    for ( ;; ) {
      const int fd = open( path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC );
      if ( fd >= 0 ) {
        struct stat info;
        char        magic[sizeof( kIdMagic )];
        const bool  mine  = fstat( fd, &info ) == 0 && S_ISREG( info.st_mode ) && info.st_uid == getuid();
        const bool  fresh = mine && ( started <= 0.0 || double( info.st_mtim.tv_sec ) + 1e-9 * double( info.st_mtim.tv_nsec ) + 1.0 >= started );
        const bool  whole = fresh && read( fd, magic, sizeof( magic ) ) == ssize_t( sizeof( magic ) ) && memcmp( magic, kIdMagic, sizeof( magic ) ) == 0 &&
                           read( fd, id.internal, sizeof( id.internal ) ) == ssize_t( sizeof( id.internal ) );
        close( fd );
        if ( whole ) break;
      }
      if ( std::chrono::steady_clock::now() > limit ) {
        t_err = "tmc2_gof_comm_create: rank 0 never published the communicator id in " + path + " (a file that is older than this process, "
                "another user's, or incomplete does not count)";
        return TMC2_E_STATE;
      }
      std::this_thread::sleep_for( std::chrono::milliseconds( 5 ) );
    }
  }
  COMM_TRY( comm->rccl.CommInitRank( &comm->comm, worldSize, id, rank ), "ncclCommInitRank" );
  HIP_TRY( tmc2_ctx_device_alloc( ctx, 64, &comm->dSmall ), "tmc2_ctx_device_alloc" );
  int32_t probe = 100 + rank;  // pre-flight: one collective through the new communicator, checked
  {
    const int rc = commMaxHeight( comm, &probe );
    if ( rank == 0 && worldSize > 1 ) unlink( path.c_str() );  // (every rank has read it -- or will never: the all-reduce is over)
    if ( rc != TMC2_OK ) return rc;
  }
  if ( probe != 100 + worldSize - 1 ) {
    t_err = "tmc2_gof_comm_create: the pre-flight all-reduce gave " + std::to_string( probe );
    return TMC2_E_STATE;
  }
  *out = owner.release();
  return TMC2_OK;
}

extern "C" void tmc2_gof_comm_destroy( tmc2_gof_comm* comm ) {
  if ( !comm ) return;
  if ( comm->dSmall ) (void)tmc2_ctx_device_free( comm->ctx, comm->dSmall );
  if ( comm->dBlocks ) (void)tmc2_ctx_device_free( comm->ctx, comm->dBlocks );
  if ( comm->comm ) (void)comm->rccl.CommDestroy( comm->comm );
  if ( comm->rccl.lib ) dlclose( comm->rccl.lib );
  delete comm;
}

namespace {
// One pass over the frames this process holds.  comm == nullptr: they are the whole GOF.  Otherwise they are this rank's share
// (frame f of the GOF on rank f mod world): the weights come from rank 0, the canvas height is the maximum over the ranks, the
// packed records of every frame end on rank 0.  THE RULE of the sharded pass: a rank that fails locally goes on through every
// collective of the pass, with a value that says so -- nobody is ever left waiting for a rank that has returned.
// resume: the frames are segmented and packed already (a pass that ended with "the GOF needs a larger canvas than the buffers hold"):
// the pass starts at the canvas size, from what the packers left in the frames.
int encodeGof( tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots, const tmc2_gof_config* config,
               uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch, uint16_t** geometryD0, uint16_t** geometryD1,
               uint8_t** attribute, int32_t capacityWidth, int32_t capacityHeight, int32_t* width, int32_t* height,
               int32_t recordSlots, tmc2_patch* gathered, int64_t* gatheredCounts, bool resume ) {
  t_err.clear();
  if ( !frames || !slotOf || !config || count <= 0 || slots <= 0 || !width || !height ) {
    t_err = "tmc2_gof_encode: invalid argument";
    return TMC2_E_INVALID;
  }
  for ( int i = 0; i < count; ++i )
    if ( !frames[i] || slotOf[i] < 0 || slotOf[i] >= slots ) {
      t_err = "tmc2_gof_encode: frame " + std::to_string( i ) + " is null or on a slot outside [0, " + std::to_string( slots ) + ")";
      return TMC2_E_INVALID;
    }
  const tmc2_gof_config& c = *config;
  if ( c.packing < 0 || c.packing > 2 ) {
    t_err = "tmc2_gof_encode: packing " + std::to_string( c.packing ) + " (0 all-intra, 1 low-delay, 2 random-access)";
    return TMC2_E_INVALID;
  }
  if ( comm ) {
    if ( const int rc = refuseBroken( comm ) ) return rc;
    ++comm->passes;
  }
  // several ranks under a chained condition: the chain runs on rank 0 over the records (TMC2_GOF_RECORDS_CHAIN=1 forces that route
  // for a communicator of one rank too: the GPU tier runs it that way on a one-GPU box)
  const bool chained = c.packing != 0, guess = !chained && c.guessCanvas != 0;
  const bool recordsChain = comm && chained && ( comm->world > 1 || ( getenv( "TMC2_GOF_RECORDS_CHAIN" ) && atoi( getenv( "TMC2_GOF_RECORDS_CHAIN" ) ) != 0 ) );
  const std::vector<int> cores = coresByCacheDomain();
  Pass                   pass;
  auto perSlot = [&]( auto fn ) {  // the frames of one slot in order, slots side by side
    SlotThreads::instance().run( slots, cores, [&]( int sl ) {
      for ( int i = 0; i < count; ++i )
        if ( slotOf[i] == sl && pass.status.load() == TMC2_OK ) fn( i );
    } );
  };
  for ( int i = 0; i < count && pass.status.load() == TMC2_OK && !resume; ++i ) {
    const int rc = tmc2_frame_reset( frames[i] );
    if ( rc != TMC2_OK ) pass.fail( rc, "tmc2_frame_reset" );
  }
  if ( !comm && pass.status.load() != TMC2_OK ) return pass.done();
  double w[3] = {0, 0, 0};
  if ( ( !comm || comm->rank == 0 ) && pass.status.load() == TMC2_OK && !resume ) {
    const int rc = tmc2_weight_normal( frames[0], c.geometryBitDepth3D, 0.6, w );  // S0: frame 0 of the GOF only (rank 0's first)
    if ( rc != TMC2_OK ) {
      pass.fail( rc, "tmc2_weight_normal" );
      if ( !comm ) return pass.done();
    }
  }
  if ( comm && !resume ) {
    // (no axis weight is negative: from a negative one the other ranks learn that rank 0 has no pass to offer)
    if ( comm->rank == 0 && pass.status.load() != TMC2_OK ) w[0] = w[1] = w[2] = -1.0;
    const int rc = commBroadcastWeights( comm, w );
    if ( rc != TMC2_OK ) pass.fail( rc, "tmc2_gof_encode_sharded", t_err.c_str() );
    if ( w[0] < 0.0 ) {  // (the same on every rank: all leave here)
      if ( pass.status.load() != TMC2_OK ) return pass.done();
      t_err = "tmc2_gof_encode_sharded: rank 0 could not start the pass (its own call says why)";
      return TMC2_E_STATE;
    }
    if ( comm->broken ) return pass.done();
  }
  const tmc2_segmenter_params params = ctcParams( c, w );
  std::vector<int32_t>        heights( static_cast<size_t>( count ), 0 ), guessW( static_cast<size_t>( count ), 0 ),
      guessH( static_cast<size_t>( count ), 0 );
  const bool anyOut = occupancy || occVideo || blockToPatch || geometryD0 || geometryD1 || attribute;  // (none: nothing leaves the device)
  if ( !anyOut ) capacityWidth = capacityHeight = INT32_MAX;
  // what follows the packing of one frame, on a canvas of W x H: S12-S22 and the copies of its finished canvases
  auto images = [&]( int i, int32_t W, int32_t H ) {
    if ( W > capacityWidth || H > capacityHeight ) return;  // (refused after the rendezvous, with the size the GOF needs)
    GOF_TRY( tmc2_encoder_generate_geometry_images( frames[i], W, H, c.occupancyPrecision ) );
    GOF_TRY( tmc2_encoder_generate_attribute_images( frames[i] ) );
    auto at = []( auto** arr, int k ) { return arr ? arr[k] : nullptr; };
    if ( occupancy || occVideo || blockToPatch || geometryD0 || geometryD1 )
      GOF_TRY( tmc2_frame_get_geometry_images( frames[i], at( occupancy, i ), at( occVideo, i ), at( blockToPatch, i ), at( geometryD0, i ),
                                               at( geometryD1, i ) ) );
    if ( attribute && attribute[i] ) GOF_TRY( tmc2_frame_get_attribute_images( frames[i], attribute[i] ) );
  };
  // All-intra with guessCanvas: a frame does not wait for the others.  It goes through its whole chain on the canvas ITS OWN packed height gives
  // (with the CTC sequences: the minimum canvas, for every frame); the rendezvous then only compares, and a frame whose guess
  // was short rasterises again on the common canvas (same bytes as the two-phase order: the images depend on the final size only).
  if ( pass.status.load() == TMC2_OK && !resume )
    perSlot( [&]( int i ) {
      GOF_TRY( tmc2_segmenter_compute( frames[i], &params ) );
      if ( chained ) return;
      GOF_TRY( tmc2_encoder_pack_flexible( frames[i], c.minimumImageWidth, 2, 1.0, &heights[size_t( i )] ) );
      if ( !guess ) return;
      GOF_TRY( tmc2_encoder_canvas_size( &heights[size_t( i )], 1, c.minimumImageWidth, c.minimumImageWidth, c.minimumImageHeight,
                                         &guessW[size_t( i )], &guessH[size_t( i )] ) );
      images( i, guessW[size_t( i )], guessH[size_t( i )] );
    } );
  int32_t W = 0, H = 0;
  if ( resume ) {
    // ---- the canvas of a GOF whose frames are packed: the tiles the packers left (tmc2_frame_get_packed_size) ---------------
    int32_t words[3] = {0, c.minimumImageWidth, 0};  // failed?, widest tile, tallest tile
    for ( int i = 0; i < count; ++i ) {
      int32_t   pw = 0, ph = 0;
      const int rc = tmc2_frame_get_packed_size( frames[i], &pw, &ph );
      if ( rc != TMC2_OK ) pass.fail( rc, "tmc2_frame_get_packed_size" );
      else if ( pw <= 0 ) pass.fail( TMC2_E_STATE, "tmc2_gof_encode_resume", "a frame is not packed (the pass to resume must have ended with 'the GOF needs a larger canvas')" );
      words[1] = std::max( words[1], pw ), words[2] = std::max( words[2], ph );
    }
    if ( c.packing == 2 ) words[2] = std::max( words[2], c.minimumImageHeight );
    if ( pass.status.load() != TMC2_OK ) words[0] = kFailedWord;
    if ( comm ) {
      const int rc = commMax( comm, words, 3, "ncclAllReduce( tile of the GOF, max )" );
      if ( pass.status.load() != TMC2_OK ) return pass.done();
      if ( rc != TMC2_OK ) return rc;
      if ( words[0] != 0 ) {
        t_err = "tmc2_gof_encode_sharded_resume: another rank's frames are not packed (its own call says why)";
        return TMC2_E_STATE;
      }
    } else if ( pass.status.load() != TMC2_OK ) {
      return pass.done();
    }
    const int rc = tmc2_encoder_canvas_size( &words[2], 1, words[1], c.minimumImageWidth, c.minimumImageHeight, &W, &H );
    if ( rc != TMC2_OK ) {
      pass.fail( rc, "tmc2_encoder_canvas_size" );
      return pass.done();
    }
  } else if ( recordsChain ) {
    // ---- the rendezvous of a sharded chained GOF: records to rank 0, the chain there, the packed lists back ------------------
    ChainResult chain;
    const int   rc = chainOverRanks( comm, frames, count, c, recordSlots, gathered, gatheredCounts, pass.done(), chain );
    if ( rc != TMC2_OK ) return rc;  // (every rank leaves at the same place: chainOverRanks)
    W = chain.W, H = chain.H;
    // (installed before the buffers are looked at: a pass that ends with "the GOF needs a larger canvas" leaves its frames packed)
    perSlot( [&]( int i ) {
      const ChainResult::Frame& fr = chain.frames[size_t( i )];
      GOF_TRY( tmc2_frame_set_packing( frames[i], fr.list.data(), int( fr.list.size() ), fr.matches.data(), fr.occupancy.data(),
                                       int64_t( fr.occupancy.size() ), fr.packedW, fr.packedH ) );
    } );
  } else {
    // (several ranks: a rank whose frames failed still goes to the rendezvous -- with a height no canvas has -- so that every rank
    //  leaves the pass at the same place instead of waiting in a collective for one that has returned)
    if ( pass.status.load() != TMC2_OK ) {
      if ( comm ) {
        const int   rc  = pass.done();
        const auto  why = t_err;
        int32_t     h   = kFailedWord;
        (void)commMaxHeight( comm, &h );
        t_err = why;
        return rc;
      }
      return pass.done();
    }
    // ---- the rendezvous: the packing chain (if any) and the common canvas size -----------------------------------------------
    int32_t tileW = c.minimumImageWidth, gofH = 0;
    if ( chained ) {
      auto once = [&]( int rc, const char* what ) {
        if ( rc != TMC2_OK ) pass.fail( rc, what );
        return rc == TMC2_OK;
      };
      bool ok = once( tmc2_encoder_pack_flexible( frames[0], c.minimumImageWidth, 2, 1.0, &heights[0] ), "tmc2_encoder_pack_flexible" );
      for ( int i = 1; i < count && ok; ++i )
        ok = once( tmc2_encoder_pack_spatial_consistency( frames[i], frames[i - 1], c.minimumImageWidth, 2, 1.0, &heights[size_t( i )] ),
                   "tmc2_encoder_pack_spatial_consistency" );
      if ( ok && c.packing == 2 ) {
        std::vector<int32_t> widths( static_cast<size_t>( count ), 0 );
        ok = once( tmc2_encoder_global_patch_allocation( frames, count, c.minimumImageWidth, c.minimumImageHeight, widths.data(), heights.data() ),
                   "tmc2_encoder_global_patch_allocation" );
        for ( int i = 0; i < count && ok; ++i ) {
          tileW                = std::max( tileW, widths[size_t( i )] );
          heights[size_t( i )] = std::max( heights[size_t( i )], c.minimumImageHeight );
        }
      } else {
        for ( int i = 0; i < count && ok; ++i ) {
          int32_t pw = 0;
          ok         = once( tmc2_frame_get_packed_size( frames[i], &pw, nullptr ), "tmc2_frame_get_packed_size" );
          tileW      = std::max( tileW, pw );
        }
      }
    }
    for ( int i = 0; i < count; ++i ) gofH = std::max( gofH, heights[size_t( i )] );
    if ( comm ) {  // the one number the ranks of an all-intra GOF share
      if ( pass.status.load() != TMC2_OK ) gofH = kFailedWord;
      const int rc = commMaxHeight( comm, &gofH );
      if ( pass.status.load() != TMC2_OK ) return pass.done();
      if ( rc != TMC2_OK ) return rc;
      if ( gofH == kFailedWord ) {
        t_err = "tmc2_gof_encode_sharded: another rank's frames failed before the rendezvous (its own call says why)";
        return TMC2_E_STATE;
      }
    } else if ( pass.status.load() != TMC2_OK ) {
      return pass.done();
    }
    const int rc = tmc2_encoder_canvas_size( &gofH, 1, tileW, c.minimumImageWidth, c.minimumImageHeight, &W, &H );
    if ( rc != TMC2_OK ) {  // (a function of the GOF's numbers: the same on every rank)
      pass.fail( rc, "tmc2_encoder_canvas_size" );
      return pass.done();
    }
  }
  *width = W, *height = H;
  if ( pass.status.load() != TMC2_OK && !comm ) return pass.done();
  if ( W > capacityWidth || H > capacityHeight ) {  // (the same on every rank: W and H are the GOF's)
    char msg[160];
    std::snprintf( msg, sizeof( msg ), "tmc2_gof_encode: the GOF needs a %d x %d canvas, the buffers hold %d x %d", W, H, capacityWidth, capacityHeight );
    t_err = msg;
    return TMC2_E_INVALID;
  }
  // ---- from here the frames are independent: images, attribute images, copies, each on its slot -----------------------------
  perSlot( [&]( int i ) {
    if ( guess && !resume && guessW[size_t( i )] == W && guessH[size_t( i )] == H ) return;  // (already there)
    images( i, W, H );
  } );
  if ( !comm ) return pass.done();
  const int failed = pass.done();  // (a pass that failed after the rendezvous still takes part in the exchange)
  if ( !recordsChain || resume ) return commGatherRecords( comm, frames, count, recordSlots, gathered, gatheredCounts, failed );
  // the chained GOF left every record on rank 0 already; what the ranks still owe each other is whether the pass held
  const std::string why = t_err;
  int32_t           bad = failed != TMC2_OK ? 1 : 0;
  const int         rc  = commMax( comm, &bad, 1, "ncclAllReduce( status of the pass, max )" );
  if ( failed != TMC2_OK ) return t_err = why, failed;
  if ( rc != TMC2_OK ) return rc;
  if ( bad != 0 ) {
    t_err = "tmc2_gof_encode_sharded: another rank's frames failed after the rendezvous (its own call says why)";
    return TMC2_E_STATE;
  }
  return TMC2_OK;
}
}  // namespace

// (nothing may leave through the C boundary but a status: a std::bad_alloc / std::system_error of the thread pool included)
template <typename F>
static int guarded( F&& f ) {
  try {
    return f();
  } catch ( const std::exception& e ) {
    t_err = std::string( "tmc2_gof_encode: " ) + e.what();
    return TMC2_E_STATE;
  } catch ( ... ) {
    t_err = "tmc2_gof_encode: unknown exception";
    return TMC2_E_STATE;
  }
}

extern "C" int tmc2_gof_encode( tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots, const tmc2_gof_config* config,
                                uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch, uint16_t** geometryD0,
                                uint16_t** geometryD1, uint8_t** attribute, int32_t capacityWidth, int32_t capacityHeight,
                                int32_t* width, int32_t* height ) {
  return guarded( [&] {
    return encodeGof( nullptr, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                      capacityWidth, capacityHeight, width, height, 0, nullptr, nullptr, false );
  } );
}

extern "C" int tmc2_gof_encode_resume( tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots, const tmc2_gof_config* config,
                                       uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch, uint16_t** geometryD0,
                                       uint16_t** geometryD1, uint8_t** attribute, int32_t capacityWidth, int32_t capacityHeight,
                                       int32_t* width, int32_t* height ) {
  return guarded( [&] {
    return encodeGof( nullptr, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                      capacityWidth, capacityHeight, width, height, 0, nullptr, nullptr, true );
  } );
}

static int sharded( bool resume, tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots,
                    const tmc2_gof_config* config, uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch, uint16_t** geometryD0,
                    uint16_t** geometryD1, uint8_t** attribute, int32_t capacityWidth, int32_t capacityHeight, int32_t* width, int32_t* height,
                    int32_t recordSlots, tmc2_patch* gathered, int64_t* gatheredCounts ) {
  if ( !comm || recordSlots <= 0 ) {
    t_err = "tmc2_gof_encode_sharded: invalid argument";
    return TMC2_E_INVALID;
  }
  return guarded( [&] {
    return encodeGof( comm, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                      capacityWidth, capacityHeight, width, height, recordSlots, gathered, gatheredCounts, resume );
  } );
}

extern "C" int tmc2_gof_encode_sharded( tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots,
                                        const tmc2_gof_config* config, uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch,
                                        uint16_t** geometryD0, uint16_t** geometryD1, uint8_t** attribute, int32_t capacityWidth,
                                        int32_t capacityHeight, int32_t* width, int32_t* height, int32_t recordSlots, tmc2_patch* gathered,
                                        int64_t* gatheredCounts ) {
  return sharded( false, comm, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                  capacityWidth, capacityHeight, width, height, recordSlots, gathered, gatheredCounts );
}

extern "C" int tmc2_gof_encode_sharded_resume( tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots,
                                               const tmc2_gof_config* config, uint8_t** occupancy, uint8_t** occVideo,
                                               uint32_t** blockToPatch, uint16_t** geometryD0, uint16_t** geometryD1, uint8_t** attribute,
                                               int32_t capacityWidth, int32_t capacityHeight, int32_t* width, int32_t* height,
                                               int32_t recordSlots, tmc2_patch* gathered, int64_t* gatheredCounts ) {
  return sharded( true, comm, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                  capacityWidth, capacityHeight, width, height, recordSlots, gathered, gatheredCounts );
}
