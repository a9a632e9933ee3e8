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
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

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
    const auto sym = [&]( const char* n ) {
      void* p = dlsym( lib, n );
      if ( !p ) why = std::string( "librccl: no symbol " ) + n;
      return p;
    };
    GetUniqueId    = reinterpret_cast<decltype( GetUniqueId )>( sym( "ncclGetUniqueId" ) );
    CommInitRank   = reinterpret_cast<decltype( CommInitRank )>( sym( "ncclCommInitRank" ) );
    CommDestroy    = reinterpret_cast<decltype( CommDestroy )>( sym( "ncclCommDestroy" ) );
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
  void*     dSmall  = nullptr;  // 64 bytes on the device: weights, heights
  void*     dBlocks = nullptr;  // the records of this rank's frames, then (rank 0) those of every rank
  size_t    blocksBytes = 0;
  std::vector<uint8_t> hostBlocks;
};

namespace {
#define COMM_TRY( call, what )                                                                              \
  do {                                                                                                      \
    const int rc_ = ( call );                                                                               \
    if ( rc_ != 0 ) {                                                                                       \
      t_err = std::string( what ) + ": " + ( comm->rccl.GetErrorString ? comm->rccl.GetErrorString( rc_ ) : "?" ); \
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

// 24 bytes from rank 0 to everybody (the axis weights of frame 0: PCCEncoder::calculateWeightNormal runs on the first frame only)
int commBroadcastWeights( tmc2_gof_comm* comm, double w[3] ) {
  void* st = tmc2_ctx_stream( comm->ctx );
  HIP_TRY( tmc2_ctx_upload( comm->ctx, comm->dSmall, w, 24 ), "weights: upload" );
  COMM_TRY( comm->rccl.Broadcast( comm->dSmall, comm->dSmall, 3, kNcclFloat64, 0, comm->comm, st ), "ncclBroadcast( weights )" );
  HIP_TRY( tmc2_ctx_download( comm->ctx, w, comm->dSmall, 24 ), "weights: download" );
  return TMC2_OK;
}
// the canvas height of the GOF: max over the ranks of what their frames packed into (resizeGeometryVideo, PCCEncoder.cpp:5546-5591)
int commMaxHeight( tmc2_gof_comm* comm, int32_t* h ) {
  void* st = tmc2_ctx_stream( comm->ctx );
  HIP_TRY( tmc2_ctx_upload( comm->ctx, comm->dSmall, h, 4 ), "height: upload" );
  COMM_TRY( comm->rccl.AllReduce( comm->dSmall, comm->dSmall, 1, kNcclInt32, kNcclMax, comm->comm, st ), "ncclAllReduce( height, max )" );
  HIP_TRY( tmc2_ctx_download( comm->ctx, h, comm->dSmall, 4 ), "height: download" );
  return TMC2_OK;
}
// The final gather: the packed patch records of every frame (the side information the bitstream carries: ~ 100 bytes a patch) to
// rank 0 -- one grouped send / receive per pass.  Block of a frame: int64 count, then recordSlots records in list order.
int commGatherRecords( tmc2_gof_comm* comm, tmc2_frame** frames, int32_t count, int32_t recordSlots, tmc2_patch* gathered,
                       int64_t* gatheredCounts, int failed ) {
  const size_t frameBytes = 8 + size_t( recordSlots ) * sizeof( tmc2_patch ), mine = frameBytes * size_t( count );
  const size_t need       = mine * ( comm->rank == 0 ? size_t( comm->world ) + 1 : 1 );
  if ( need > comm->blocksBytes ) {
    if ( comm->dBlocks ) HIP_TRY( tmc2_ctx_device_free( comm->ctx, comm->dBlocks ), "records: free" );
    comm->dBlocks = nullptr, comm->blocksBytes = 0;
    HIP_TRY( tmc2_ctx_device_alloc( comm->ctx, need, &comm->dBlocks ), "records: device buffer" );
    comm->blocksBytes = need;
  }
  comm->hostBlocks.assign( need, 0 );
  std::vector<tmc2_patch> list;
  std::vector<int32_t>    order;
  // (A rank that cannot fill its block -- its pass failed after the rendezvous, or a frame has more patches than the block holds --
  //  STILL takes part in the exchange, with a negative count in the block: the other ranks must not be left waiting in a receive.)
  int         localStatus = failed;
  std::string localError  = failed != TMC2_OK ? t_err : std::string();
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
          for ( int k = 0; k < n; ++k ) memcpy( at + 8 + size_t( k ) * sizeof( tmc2_patch ), &list[size_t( order[size_t( k )] )], sizeof( tmc2_patch ) );
        }
      }
    }
    memcpy( at, &n64, 8 );
  }
  void*    st   = tmc2_ctx_stream( comm->ctx );
  uint8_t* dev  = static_cast<uint8_t*>( comm->dBlocks );
  HIP_TRY( tmc2_ctx_upload( comm->ctx, dev, comm->hostBlocks.data(), mine ), "records: upload" );
  COMM_TRY( comm->rccl.GroupStart(), "ncclGroupStart" );
  if ( comm->rank == 0 )
    for ( int r = 0; r < comm->world; ++r )
      COMM_TRY( comm->rccl.Recv( dev + mine * size_t( r + 1 ), mine, kNcclUint8, r, comm->comm, st ), "ncclRecv( records )" );
  COMM_TRY( comm->rccl.Send( dev, mine, kNcclUint8, 0, comm->comm, st ), "ncclSend( records )" );
  COMM_TRY( comm->rccl.GroupEnd(), "ncclGroupEnd" );
  if ( comm->rank != 0 ) {
    const int rc = tmc2_ctx_synchronize( comm->ctx );
    if ( localStatus != TMC2_OK ) t_err = localError;
    return localStatus != TMC2_OK ? localStatus : rc;
  }
  HIP_TRY( tmc2_ctx_download( comm->ctx, comm->hostBlocks.data() + mine, dev + mine, mine * size_t( comm->world ) ), "records: download" );
  if ( localStatus != TMC2_OK ) {
    t_err = localError;
    return localStatus;
  }
  for ( int r = 0; r < comm->world; ++r )
    for ( int i = 0; i < count; ++i ) {
      const uint8_t* at = comm->hostBlocks.data() + mine * size_t( r + 1 ) + frameBytes * size_t( i );
      int64_t        n  = 0;
      memcpy( &n, at, 8 );
      if ( n < 0 || n > recordSlots ) {
        t_err = "tmc2_gof_encode_sharded: rank " + std::to_string( r ) + " could not deliver the records of its frame " + std::to_string( i ) +
                " (its own call says why)";
        return TMC2_E_STATE;
      }
      if ( gatheredCounts ) gatheredCounts[size_t( r ) * size_t( count ) + size_t( i )] = n;
      if ( gathered )
        memcpy( gathered + ( size_t( r ) * size_t( count ) + size_t( i ) ) * size_t( recordSlots ), at + 8, size_t( n ) * sizeof( tmc2_patch ) );
    }
  return TMC2_OK;
}
}  // namespace

extern "C" int tmc2_gof_comm_create( int rank, int worldSize, tmc2_ctx* ctx, const char* rendezvous, tmc2_gof_comm** out ) {
  if ( !out ) return TMC2_E_INVALID;
  *out = nullptr;
  if ( rank < 0 || worldSize < 1 || rank >= worldSize || !ctx ) {
    t_err = "tmc2_gof_comm_create: invalid argument";
    return TMC2_E_INVALID;
  }
  std::unique_ptr<tmc2_gof_comm> owner( new tmc2_gof_comm() );
  tmc2_gof_comm*                 comm = owner.get();
  comm->rank = rank, comm->world = worldSize, comm->ctx = ctx;
  std::string why;
  if ( !comm->rccl.load( why ) ) {
    t_err = "tmc2_gof_comm_create: RCCL not available (" + why + ")";
    return TMC2_E_UNSUPPORTED;
  }
  HIP_TRY( tmc2_ctx_make_current( ctx ), "tmc2_ctx_make_current" );
  // The 128-byte id of the communicator: rank 0 makes it and publishes it in a file (same node: /dev/shm), the others wait for it.
  std::string path = rendezvous ? rendezvous : "";
  if ( path.empty() ) {
    const char* port = getenv( "MASTER_PORT" );
    path             = std::string( "/dev/shm/tmc2_gof_id_" ) + ( port ? port : "0" );
  }
  NcclId id{};
  if ( rank == 0 ) {
    COMM_TRY( comm->rccl.GetUniqueId( &id ), "ncclGetUniqueId" );
    if ( worldSize > 1 ) {
      const std::string tmp = path + ".tmp";
      std::ofstream     f( tmp, std::ios::binary );
      f.write( id.internal, sizeof( id.internal ) );
      f.close();
      if ( !f || rename( tmp.c_str(), path.c_str() ) != 0 ) {
        t_err = "tmc2_gof_comm_create: cannot publish the communicator id in " + path;
        return TMC2_E_INVALID;
      }
    }
  } else {
    const auto limit = std::chrono::steady_clock::now() + std::chrono::seconds( 120 );
    for ( ;; ) {
      std::ifstream f( path, std::ios::binary );
      if ( f && f.read( id.internal, sizeof( id.internal ) ) ) break;
      if ( std::chrono::steady_clock::now() > limit ) {
        t_err = "tmc2_gof_comm_create: rank 0 never published the communicator id in " + path;
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
    if ( rc != TMC2_OK ) return rc;
  }
  if ( rank == 0 && worldSize > 1 ) unlink( path.c_str() );  // (every rank has read it: the all-reduce above has completed)
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
// packed records of every frame end on rank 0.
int encodeGof( tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots, const tmc2_gof_config* config,
               uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch, uint16_t** geometryD0, uint16_t** geometryD1,
               uint8_t** attribute, int32_t capacityWidth, int32_t capacityHeight, int32_t* width, int32_t* height,
               int32_t recordSlots, tmc2_patch* gathered, int64_t* gatheredCounts ) {
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
  const bool             sharded = comm && comm->world > 1;
  if ( sharded && c.packing != 0 ) {
    t_err = "tmc2_gof_encode_sharded: the low-delay / random-access packing chains run over ALL frames of the GOF in order; with the "
            "frames on several ranks that is the caller's (records to rank 0, tmc2_host_place_segments, tmc2_frame_set_packing)";
    return TMC2_E_UNSUPPORTED;
  }
  const std::vector<int> cores = coresByCacheDomain();
  Pass                   pass;
  auto perSlot = [&]( auto fn ) {  // the frames of one slot in order, slots side by side
    std::vector<std::thread> pool;
    for ( int sl = 0; sl < slots; ++sl )
      pool.emplace_back( [&, sl] {
        if ( !cores.empty() ) {
          cpu_set_t one;
          CPU_ZERO( &one );
          CPU_SET( cores[size_t( sl ) % cores.size()], &one );
          (void)pthread_setaffinity_np( pthread_self(), sizeof( one ), &one );
        }
        for ( int i = 0; i < count; ++i )
          if ( slotOf[i] == sl && pass.status.load() == TMC2_OK ) fn( i );
      } );
    for ( auto& t : pool ) t.join();
  };
  for ( int i = 0; i < count; ++i ) {
    const int rc = tmc2_frame_reset( frames[i] );
    if ( rc != TMC2_OK ) {
      pass.fail( rc, "tmc2_frame_reset" );
      return pass.done();
    }
  }
  double w[3] = {0, 0, 0};
  if ( !comm || comm->rank == 0 ) {
    const int rc = tmc2_weight_normal( frames[0], c.geometryBitDepth3D, 0.6, w );  // S0: frame 0 of the GOF only (rank 0's first)
    if ( rc != TMC2_OK ) {
      pass.fail( rc, "tmc2_weight_normal" );
      if ( !comm ) return pass.done();
      w[0] = w[1] = w[2] = -1.0;  // (no axis weight is negative: the other ranks learn from the broadcast that there is no pass)
    }
  }
  if ( comm ) {
    const int rc = commBroadcastWeights( comm, w );
    if ( rc != TMC2_OK ) return rc;
    if ( pass.status.load() != TMC2_OK ) return pass.done();
    if ( w[0] < 0.0 ) {
      t_err = "tmc2_gof_encode_sharded: rank 0 could not compute the axis weights of frame 0 (its own call says why)";
      return TMC2_E_STATE;
    }
  }
  const tmc2_segmenter_params params = ctcParams( c, w );
  std::vector<int32_t>        heights( static_cast<size_t>( count ), 0 ), guessW( static_cast<size_t>( count ), 0 ),
      guessH( static_cast<size_t>( count ), 0 );
  const bool                  chained = c.packing != 0, guess = !chained && c.guessCanvas != 0;
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
  perSlot( [&]( int i ) {
    GOF_TRY( tmc2_segmenter_compute( frames[i], &params ) );
    if ( chained ) return;
    GOF_TRY( tmc2_encoder_pack_flexible( frames[i], c.minimumImageWidth, 2, 1.0, &heights[size_t( i )] ) );
    if ( !guess ) return;
    GOF_TRY( tmc2_encoder_canvas_size( &heights[size_t( i )], 1, c.minimumImageWidth, c.minimumImageWidth, c.minimumImageHeight,
                                       &guessW[size_t( i )], &guessH[size_t( i )] ) );
    images( i, guessW[size_t( i )], guessH[size_t( i )] );
  } );
  // (several ranks: a rank whose frames failed still goes to the rendezvous -- with a height no canvas has -- so that every rank
  //  leaves the pass at the same place instead of waiting in a collective for one that has returned)
  constexpr int32_t kFailedHeight = 0x7FFFFFF0;
  if ( pass.status.load() != TMC2_OK ) {
    if ( comm ) {
      int32_t h = kFailedHeight;
      (void)commMaxHeight( comm, &h );
    }
    return pass.done();
  }
  // ---- the rendezvous: the packing chain (if any) and the common canvas size -------------------------------------------------
  int32_t tileW = c.minimumImageWidth, gofH = 0;
  if ( chained ) {
    auto once = [&]( int rc, const char* what ) {
      if ( rc != TMC2_OK ) pass.fail( rc, what );
      return rc == TMC2_OK;
    };
    if ( !once( tmc2_encoder_pack_flexible( frames[0], c.minimumImageWidth, 2, 1.0, &heights[0] ), "tmc2_encoder_pack_flexible" ) )
      return pass.done();
    for ( int i = 1; i < count; ++i )
      if ( !once( tmc2_encoder_pack_spatial_consistency( frames[i], frames[i - 1], c.minimumImageWidth, 2, 1.0, &heights[size_t( i )] ),
                  "tmc2_encoder_pack_spatial_consistency" ) )
        return pass.done();
    if ( c.packing == 2 ) {
      std::vector<int32_t> widths( static_cast<size_t>( count ), 0 );
      if ( !once( tmc2_encoder_global_patch_allocation( frames, count, c.minimumImageWidth, c.minimumImageHeight, widths.data(), heights.data() ),
                  "tmc2_encoder_global_patch_allocation" ) )
        return pass.done();
      for ( int i = 0; i < count; ++i ) {
        tileW                = std::max( tileW, widths[size_t( i )] );
        heights[size_t( i )] = std::max( heights[size_t( i )], c.minimumImageHeight );
      }
    } else {
      for ( int i = 0; i < count; ++i ) {
        int32_t pw = 0;
        if ( !once( tmc2_frame_get_packed_size( frames[i], &pw, nullptr ), "tmc2_frame_get_packed_size" ) ) return pass.done();
        tileW = std::max( tileW, pw );
      }
    }
  }
  for ( int i = 0; i < count; ++i ) gofH = std::max( gofH, heights[size_t( i )] );
  if ( comm ) {  // the one number the ranks of an all-intra GOF share
    const int rc = commMaxHeight( comm, &gofH );
    if ( rc != TMC2_OK ) return rc;
    if ( gofH == kFailedHeight ) {
      t_err = "tmc2_gof_encode_sharded: another rank's frames failed before the rendezvous (its own call says why)";
      return TMC2_E_STATE;
    }
  }
  int32_t W = 0, H = 0;
  {
    const int rc = tmc2_encoder_canvas_size( &gofH, 1, tileW, c.minimumImageWidth, c.minimumImageHeight, &W, &H );
    if ( rc != TMC2_OK ) {
      pass.fail( rc, "tmc2_encoder_canvas_size" );
      return pass.done();
    }
  }
  *width = W, *height = H;
  if ( W > capacityWidth || H > capacityHeight ) {  // (the same on every rank: W and H are the GOF's)
    char msg[160];
    std::snprintf( msg, sizeof( msg ), "tmc2_gof_encode: the GOF needs a %d x %d canvas, the buffers hold %d x %d", W, H, capacityWidth, capacityHeight );
    t_err = msg;
    return TMC2_E_INVALID;
  }
  // ---- from here the frames are independent: images, attribute images, copies, each on its slot -----------------------------
  perSlot( [&]( int i ) {
    if ( guess && guessW[size_t( i )] == W && guessH[size_t( i )] == H ) return;  // (already there)
    images( i, W, H );
  } );
  if ( !comm ) return pass.done();
  const int failed = pass.done();  // (a pass that failed after the rendezvous still takes part in the exchange: see commGatherRecords)
  return commGatherRecords( comm, frames, count, recordSlots, gathered, gatheredCounts, failed );
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
                      capacityWidth, capacityHeight, width, height, 0, nullptr, nullptr );
  } );
}

extern "C" int tmc2_gof_encode_sharded( tmc2_gof_comm* comm, tmc2_frame** frames, const int32_t* slotOf, int32_t count, int32_t slots,
                                        const tmc2_gof_config* config, uint8_t** occupancy, uint8_t** occVideo, uint32_t** blockToPatch,
                                        uint16_t** geometryD0, uint16_t** geometryD1, uint8_t** attribute, int32_t capacityWidth,
                                        int32_t capacityHeight, int32_t* width, int32_t* height, int32_t recordSlots, tmc2_patch* gathered,
                                        int64_t* gatheredCounts ) {
  if ( !comm || recordSlots <= 0 ) {
    t_err = "tmc2_gof_encode_sharded: invalid argument";
    return TMC2_E_INVALID;
  }
  return guarded( [&] {
    return encodeGof( comm, frames, slotOf, count, slots, config, occupancy, occVideo, blockToPatch, geometryD0, geometryD1, attribute,
                      capacityWidth, capacityHeight, width, height, recordSlots, gathered, gatheredCounts );
  } );
}
