// api.cpp -- extern "C" surface of libtmc2hip.so (see include/tmc2hip.h for the reference seams).
#include <algorithm>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <map>
#include <memory>
#include <mutex>

#include "internal.h"

extern char** environ;

namespace tmc2 {
// Opt-in to more than 48 KB of dynamic LDS, once per device and kernel: hipFuncSetAttribute mutates runtime-wide kernel
// state, and every in-flight frame's host thread comes through here at the same time.
int allowLargeLds( const void* kernel, size_t bytes, int device, size_t staticBytes ) {
  static std::mutex                                 lock;
  static std::map<std::pair<const void*, int>, int> granted;
  std::lock_guard<std::mutex>                       g( lock );
  int& have = granted[{kernel, device}];
  if ( int( bytes ) <= have ) return TMC2_OK;
  // the whole LDS of a gfx950 CU less the kernel's static part: asked for once, whatever this frame needs
  const int want = 160 * 1024 - int( ( staticBytes + 63 ) & ~size_t( 63 ) );
  if ( int( bytes ) > want ) {
    setError( "%zu bytes of dynamic LDS requested (+ %zu static): more than a gfx950 workgroup has", bytes, staticBytes );
    return TMC2_E_INVALID;
  }
  TMC2_HIP( hipFuncSetAttribute( kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want ) );
  have = want;
  return TMC2_OK;
}
}  // namespace tmc2

namespace tmc2 {

static thread_local std::string g_lastError;

void setError( const char* fmt, ... ) {
  char    buf[1024];
  va_list ap;
  va_start( ap, fmt );
  vsnprintf( buf, sizeof( buf ), fmt, ap );
  va_end( ap );
  g_lastError = buf;
}

// The gate around the host-resident, cache-hungry steps.  Round 6: a gate is an OBJECT (tmc2_host_gate_create) that an encoder
// shares among its own contexts (tmc2_ctx_set_host_gate) -- two encoders of one process no longer draw on one process-wide count;
// a context without one uses the process' default gate, whose limit tmc2_set_host_parallelism presets (0: none).
}  // namespace tmc2
struct tmc2_host_gate {
  std::mutex              lock;
  std::condition_variable freed;
  int                     limit = 0, busy = 0;
};
namespace tmc2 {
static std::shared_ptr<tmc2_host_gate> defaultGate() {
  static std::shared_ptr<tmc2_host_gate> g = std::make_shared<tmc2_host_gate>();
  return g;
}
void setHostParallelism( int n ) {
  const auto                  g = defaultGate();
  std::lock_guard<std::mutex> lk( g->lock );
  g->limit = n < 0 ? 0 : n;
  g->freed.notify_all();
}
HostGate::HostGate( const tmc2_ctx* ctx, bool wait ) {
  {
    std::shared_ptr<tmc2_host_gate> own;
    if ( ctx ) {
      std::lock_guard<std::mutex> lk( ctx->optionsLock );
      own = ctx->hostGate;
    }
    gate = own ? own : defaultGate();
  }
  std::unique_lock<std::mutex> lk( gate->lock );
  if ( !wait && gate->limit != 0 && gate->busy >= gate->limit ) return;  // no free slot: not held
  gate->freed.wait( lk, [this] { return gate->limit == 0 || gate->busy < gate->limit; } );
  ++gate->busy;
  held = true;
}
HostGate::~HostGate() { release(); }
void HostGate::release() {
  if ( !held ) return;
  held = false;
  std::lock_guard<std::mutex> g( gate->lock );
  --gate->busy;
  gate->freed.notify_one();
}

static thread_local tmc2_ctx* g_tlsCtx = nullptr;
DevicePool* currentPool() { return g_tlsCtx ? &g_tlsCtx->pool : nullptr; }
ApiScope::ApiScope( tmc2_ctx* ctx ) : prev( g_tlsCtx ) {
  g_tlsCtx = ctx;
  if ( ctx ) (void)hipSetDevice( ctx->device );
}
ApiScope::~ApiScope() { g_tlsCtx = prev; }

int DevicePool::acquire( size_t bytes, void** out, size_t* got ) {
  if ( bytes > ( size_t( 1 ) << 40 ) ) {  // 1 TiB: beyond any device; also what a wrapped n * sizeof( T ) looks like
    setError( "device allocation of %zu bytes refused", bytes );
    return TMC2_E_INVALID;
  }
  const size_t cls = sizeClass( bytes );
  {
    std::lock_guard<std::mutex> g( lock );
    auto                        it = freeBlocks.find( cls );
    if ( it != freeBlocks.end() && !it->second.empty() ) {
      *out = it->second.back();
      it->second.pop_back();
      *got = cls;
      return TMC2_OK;
    }
  }
  {
    std::lock_guard<std::mutex> g( lock );
    for ( Slab& sl : slabs ) {  // reserved memory first (256-byte aligned pieces; a piece goes back to the free lists like any block)
      if ( sl.size - sl.used >= cls ) {
        *out = sl.base + sl.used;
        *got = cls;
        sl.used += cls;
        ++carved;
        return TMC2_OK;
      }
    }
  }
  void*      p  = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  TMC2_HIP( hipMalloc( &p, cls ) );
  const auto t1 = std::chrono::steady_clock::now();
  {
    std::lock_guard<std::mutex> g( lock );
    bytesHeld += cls;
    ++mallocCalls;
    mallocMs += std::chrono::duration<double, std::milli>( t1 - t0 ).count();
  }
  *out = p;
  *got = cls;
  return TMC2_OK;
}
int DevicePool::reserve( size_t bytes ) {
  bytes = ( bytes + 255 ) & ~size_t( 255 );
  void* p = nullptr;
  TMC2_HIP( hipMalloc( &p, bytes ) );
  std::lock_guard<std::mutex> g( lock );
  slabs.push_back( Slab{static_cast<char*>( p ), bytes, 0} );
  bytesHeld += bytes;
  return TMC2_OK;
}
void DevicePool::recycle( void* p, size_t cls ) {
  std::lock_guard<std::mutex> g( lock );
  freeBlocks[cls].push_back( p );
}
void DevicePool::drain() {
  std::lock_guard<std::mutex> g( lock );
  const auto inSlab = [&]( const void* p ) {
    for ( const Slab& sl : slabs )
      if ( static_cast<const char*>( p ) >= sl.base && static_cast<const char*>( p ) < sl.base + sl.size ) return true;
    return false;
  };
  for ( auto& kv : freeBlocks )
    for ( void* p : kv.second )
      if ( !inSlab( p ) ) (void)hipFree( p );
  freeBlocks.clear();
  for ( const Slab& sl : slabs ) (void)hipFree( sl.base );
  slabs.clear();
}

void orientNormalsSpanningTree( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals, void* scratch );

}  // namespace tmc2

using namespace tmc2;

int tmc2_ctx::stageBegin( const char* name ) {
  int id = -1;
  for ( size_t i = 0; i < stages.size(); ++i )
    if ( stages[i].name == name ) id = int( i );
  if ( id < 0 ) {
    StageTimer t;
    t.name = name;
    stages.push_back( t );
    id = int( stages.size() ) - 1;
  }
  stages[id].calls++;
  if ( !timing ) return id;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if ( freeEvents.size() >= 2 ) {
    e0 = freeEvents.back();
    freeEvents.pop_back();
    e1 = freeEvents.back();
    freeEvents.pop_back();
  } else {
    (void)hipEventCreate( &e0 );
    (void)hipEventCreate( &e1 );
  }
  (void)hipEventRecord( e0, stream );
  stages[id].pending.emplace_back( e0, e1 );
  return id;
}
void tmc2_ctx::stageEnd( int id ) {
  if ( id < 0 || !timing || stages[id].pending.empty() ) return;
  (void)hipEventRecord( stages[id].pending.back().second, stream );
}
void tmc2_ctx::foldStage( StageTimer& t ) {
  for ( auto& pr : t.pending ) {
    float ms = 0.f;
    if ( hipEventSynchronize( pr.second ) == hipSuccess && hipEventElapsedTime( &ms, pr.first, pr.second ) == hipSuccess )
      t.ms += ms;
    freeEvents.push_back( pr.first );
    freeEvents.push_back( pr.second );
  }
  t.pending.clear();
}
void tmc2_ctx::stageAddHostMs( const char* name, double ms ) {
  for ( auto& s : stages )
    if ( s.name == name ) {
      s.ms += ms;
      s.calls++;
      return;
    }
  StageTimer t;
  t.name  = name;
  t.ms    = ms;
  t.calls = 1;
  stages.push_back( t );
}

extern "C" {

const char* tmc2_last_error( void ) { return g_lastError.c_str(); }

int tmc2_ctx_create( int device, tmc2_ctx** out ) {
  if ( !out ) return TMC2_E_INVALID;
  *out      = nullptr;
  int count = 0;
  if ( hipGetDeviceCount( &count ) != hipSuccess || count <= 0 ) {
    setError( "no HIP device visible: this library has no CPU path" );
    return TMC2_E_NO_DEVICE;
  }
  if ( device < 0 || device >= count ) {
    setError( "device %d out of range (%d visible)", device, count );
    return TMC2_E_INVALID;
  }
  TMC2_HIP( hipSetDevice( device ) );
  tmc2_ctx* c = new tmc2_ctx();
  c->device   = device;
  hipDeviceProp_t prop;
  if ( hipGetDeviceProperties( &prop, device ) == hipSuccess ) c->cuCount = prop.multiProcessorCount;
  // the defaults of this context's options: every TMC2_* variable of the process environment, read here and never again
  for ( char** e = environ; e && *e; ++e ) {
    if ( strncmp( *e, "TMC2_", 5 ) != 0 ) continue;
    const char* eq = strchr( *e, '=' );
    if ( !eq ) continue;
    c->options[std::string( *e + 5, size_t( eq - ( *e + 5 ) ) )] = std::string( eq + 1 );
  }
  // (Round 6 tried to confine a context's stream to some of the chip's eight XCDs with hipExtStreamCreateWithCUMask, so that a
  //  frame's next kernel finds in "its" L2 what the previous one left: in this partition mode the workgroups of a masked stream
  //  still land on all eight XCCs, whichever bits are set -- the mask thins the CUs inside every XCC.  profiles/r06_stream_cu_mask.txt.)
  if ( hipStreamCreateWithFlags( &c->stream, hipStreamNonBlocking ) != hipSuccess ) {
    setError( "hipStreamCreate failed" );
    delete c;
    return TMC2_E_HIP;
  }
  if ( hipHostMalloc( reinterpret_cast<void**>( &c->mailbox ), tmc2_ctx::kMailboxWords * 4, hipHostMallocDefault ) != hipSuccess ) {
    setError( "hipHostMalloc failed (the context's mailbox)" );
    (void)hipStreamDestroy( c->stream );
    delete c;
    return TMC2_E_HIP;
  }
  memset( c->mailbox, 0, tmc2_ctx::kMailboxWords * 4 );
  *out = c;
  return TMC2_OK;
}

/* Per-context options: no process-global state, no environment look-ups at run time.  key: the knob's name without the
   TMC2_ prefix ("REFINE_OVERLAP", "KDTREE_HOST", "UF_CHECK", "KD_FORM", ...: the names DESIGN.md lists); value NULL unsets it. */
int tmc2_ctx_set_option( tmc2_ctx* ctx, const char* key, const char* value ) {
  if ( !ctx || !key || !*key ) {
    tmc2::setError( "tmc2_ctx_set_option: invalid argument" );
    return TMC2_E_INVALID;
  }
  const std::string k = strncmp( key, "TMC2_", 5 ) == 0 ? key + 5 : key;
  std::lock_guard<std::mutex> g( ctx->optionsLock );  // (the context's worker thread may be reading: stage code looks options up)
  if ( value )
    ctx->options[k] = value;
  else
    ctx->options.erase( k );
  return TMC2_OK;
}
const char* tmc2_ctx_get_option( tmc2_ctx* ctx, const char* key ) {
  if ( !ctx || !key ) return nullptr;
  return tmc2::ctxOption( ctx, strncmp( key, "TMC2_", 5 ) == 0 ? key + 5 : key );
}

/* Device memory for the frames this context will see, allocated NOW: every buffer a stage asks the context's pool for is carved
   from it, so that the first GOFs of a sequence make no hipMalloc (each one synchronises the device under all frames in flight).
   The figure is an upper estimate of what ONE frame in flight holds at its peak (tmc2_ctx_pool_stats of the BASELINE
   configurations, with the pool's power-of-two size classes): 2.1 KB per point with voxels of 4, 4.6 KB with voxels of 2 (the
   85 M-entry forward and reverse rows of the refinement land in 1 GiB classes), 64 bytes
   per canvas pixel, the dense occupancy words of the refinement grid.  A frame that needs more falls back to hipMalloc. */
int tmc2_ctx_reserve( tmc2_ctx* ctx, uint64_t maxPoints, int voxelDimRefine, int bits3d, int maxCanvasWidth, int maxCanvasHeight ) {
  if ( !ctx || maxPoints == 0 || maxPoints > ( uint64_t( 1 ) << 32 ) || bits3d < 1 || bits3d > 16 || maxCanvasWidth < 0 || maxCanvasHeight < 0 ) {
    tmc2::setError( "tmc2_ctx_reserve: invalid argument" );
    return TMC2_E_INVALID;
  }
  tmc2::ApiScope scope( ctx );
  const uint64_t perPoint = voxelDimRefine > 0 && voxelDimRefine <= 2 ? 5500 : 2400;  // (round 6: + the balls' hits kept between S5's two passes, 640 keys per voxel)
  const int      gridShift = std::max( 1, bits3d - 1 - ( voxelDimRefine >= 4 ? 2 : ( voxelDimRefine >= 2 ? 1 : 0 ) ) );
  const uint64_t dense     = ( uint64_t( 1 ) << std::min( 33, 3 * gridShift + 1 ) ) / 32 * 8;  // uint2 per 32 keys
  const uint64_t bytes     = maxPoints * perPoint + uint64_t( maxCanvasWidth ) * uint64_t( maxCanvasHeight ) * 64 + dense + ( uint64_t( 64 ) << 20 );
  // ... and the page-locked staging of the host-resident step of the default path (S3: cluster records, the reduced cross-edge
  // list and the cluster signs): grown on first use otherwise -- hipHostMalloc / hipHostFree pairs in the middle of the first GOFs
  if ( !ctx->hostC.get<uint32_t>( 4 + ( size_t( maxPoints ) + 4 ) / 4 + 4 ) || !ctx->hostA.get<tmc2::OrientClusterRec>( 64 * 1024 ) ||
       !ctx->hostE.get<tmc2::OrientCompactEdge>( 384 * 1024 ) ) {
    tmc2::setError( "tmc2_ctx_reserve: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  return ctx->pool.reserve( size_t( bytes ) );
}
/* what the context's pool holds (bytes of device memory, reserved slabs included), how many hipMalloc calls its misses have cost
   so far and the host time spent in them, and how many blocks were carved from reserved memory instead */
int tmc2_ctx_pool_stats( tmc2_ctx* ctx, uint64_t* bytesHeld, uint64_t* mallocCalls, double* mallocMs, uint64_t* carvedBlocks ) {
  if ( !ctx ) return TMC2_E_INVALID;
  std::lock_guard<std::mutex> g( ctx->pool.lock );
  if ( bytesHeld ) *bytesHeld = ctx->pool.bytesHeld;
  if ( mallocCalls ) *mallocCalls = ctx->pool.mallocCalls;
  if ( mallocMs ) *mallocMs = ctx->pool.mallocMs;
  if ( carvedBlocks ) *carvedBlocks = ctx->pool.carved;
  return TMC2_OK;
}

/* Device staging for a host that moves small records between ranks itself (libtmc2gof.so: RCCL needs device buffers and the
   stream they are ready on): plain device memory of the context's device, copies ordered on the context's stream. */
int tmc2_ctx_device_alloc( tmc2_ctx* ctx, size_t bytes, void** out ) {
  if ( !ctx || !out ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  *out = nullptr;
  TMC2_HIP( hipMalloc( out, std::max<size_t>( bytes, 1 ) ) );
  return TMC2_OK;
}
int tmc2_ctx_device_free( tmc2_ctx* ctx, void* p ) {
  if ( !ctx ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  if ( p ) TMC2_HIP( hipFree( p ) );
  return TMC2_OK;
}
int tmc2_ctx_upload( tmc2_ctx* ctx, void* deviceDst, const void* hostSrc, size_t bytes ) {
  if ( !ctx || ( bytes && ( !deviceDst || !hostSrc ) ) ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  if ( bytes ) TMC2_HIP( hipMemcpyAsync( deviceDst, hostSrc, bytes, hipMemcpyHostToDevice, ctx->stream ) );
  return TMC2_OK;
}
int tmc2_ctx_download( tmc2_ctx* ctx, void* hostDst, const void* deviceSrc, size_t bytes ) {
  if ( !ctx || ( bytes && ( !hostDst || !deviceSrc ) ) ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  if ( bytes ) TMC2_HIP( hipMemcpyAsync( hostDst, deviceSrc, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  return TMC2_OK;
}
void* tmc2_ctx_stream( tmc2_ctx* ctx ) { return ctx ? static_cast<void*>( ctx->stream ) : nullptr; }
int   tmc2_ctx_device( tmc2_ctx* ctx ) { return ctx ? ctx->device : -1; }
int   tmc2_ctx_make_current( tmc2_ctx* ctx ) {  /* hipSetDevice( the context's device ) on the calling thread */
  if ( !ctx ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( ctx->device ) );
  return TMC2_OK;
}

void tmc2_ctx_destroy( tmc2_ctx* ctx ) {
  if ( !ctx ) return;
  ctx->destroyRequested.store( true );
  if ( ctx->liveFrames.load() > 0 ) return;  // frames of this context are alive: the last of them destroys it (FrameTicket)
  if ( ctx->destroyClaimed.exchange( true ) ) return;
  tmc2::destroyContextNow( ctx );
}
}  // extern "C"

void tmc2::destroyContextNow( tmc2_ctx* ctx ) {
  (void)hipSetDevice( ctx->device );
  for ( auto& s : ctx->stages ) ctx->foldStage( s );
  for ( auto e : ctx->freeEvents ) (void)hipEventDestroy( e );
  {
    ApiScope scope( ctx );
    ctx->gridTable.release();
    ctx->gridBits.release();
    ctx->scanState.release();
    ctx->voxelBitmap.release();
    ctx->constTables.clear();
    ctx->retiredTables.clear();
  }
  if ( ctx->stream ) (void)hipStreamSynchronize( ctx->stream );
  ctx->pool.drain();
  if ( ctx->stream ) (void)hipStreamDestroy( ctx->stream );
  if ( ctx->mailbox ) (void)hipHostFree( ctx->mailbox );
  delete ctx;
}

extern "C" {

int tmc2_host_alloc( size_t bytes, void** out ) {
  if ( !out || bytes == 0 ) return TMC2_E_INVALID;
  *out = nullptr;
  if ( hipHostMalloc( out, bytes, hipHostMallocPortable ) != hipSuccess ) {
    (void)hipGetLastError();
    setError( "host_alloc: %zu bytes of page-locked memory refused", bytes );
    return TMC2_E_HIP;
  }
  return TMC2_OK;
}
void tmc2_host_free( void* p ) {
  if ( p ) (void)hipHostFree( p );
}
int tmc2_host_register( void* p, size_t bytes ) {
  if ( !p || bytes == 0 ) return TMC2_E_INVALID;
  if ( hipHostRegister( p, bytes, hipHostRegisterPortable ) != hipSuccess ) {
    (void)hipGetLastError();
    setError( "host_register: %zu bytes at %p could not be page-locked", bytes, p );
    return TMC2_E_HIP;
  }
  return TMC2_OK;
}
int tmc2_host_unregister( void* p ) {
  if ( !p ) return TMC2_E_INVALID;
  if ( hipHostUnregister( p ) != hipSuccess ) {
    (void)hipGetLastError();
    return TMC2_E_HIP;
  }
  return TMC2_OK;
}

void tmc2_set_host_parallelism( int maxConcurrentHostSteps ) { tmc2::setHostParallelism( maxConcurrentHostSteps ); }
/* One encoder's budget of concurrently running host-resident steps (the k-d tree builds of option KDTREE_HOST, the orientation
   walk): shared by the contexts it is set on, and by nobody else.  The handle is the creator's; the contexts keep the gate alive. */
namespace {
std::mutex                                                  g_gatesLock;
std::map<tmc2_host_gate*, std::shared_ptr<tmc2_host_gate>> g_gates;  // handles handed out -> the shared object
}  // namespace
int tmc2_host_gate_create( int maxConcurrentHostSteps, tmc2_host_gate** out ) {
  if ( !out || maxConcurrentHostSteps < 0 ) {
    tmc2::setError( "tmc2_host_gate_create: invalid argument" );
    return TMC2_E_INVALID;
  }
  auto g   = std::make_shared<tmc2_host_gate>();
  g->limit = maxConcurrentHostSteps;
  std::lock_guard<std::mutex> lk( g_gatesLock );
  g_gates[g.get()] = g;
  *out             = g.get();
  return TMC2_OK;
}
void tmc2_host_gate_destroy( tmc2_host_gate* gate ) {
  std::lock_guard<std::mutex> lk( g_gatesLock );
  g_gates.erase( gate );  // (contexts that still use it keep it alive)
}
int tmc2_ctx_set_host_gate( tmc2_ctx* ctx, tmc2_host_gate* gate ) {
  if ( !ctx ) {
    tmc2::setError( "tmc2_ctx_set_host_gate: invalid argument" );
    return TMC2_E_INVALID;
  }
  std::shared_ptr<tmc2_host_gate> g;
  if ( gate ) {
    std::lock_guard<std::mutex> lk( g_gatesLock );
    const auto                  it = g_gates.find( gate );
    if ( it == g_gates.end() ) {
      tmc2::setError( "tmc2_ctx_set_host_gate: not a live gate" );
      return TMC2_E_INVALID;
    }
    g = it->second;
  }
  std::lock_guard<std::mutex> lk( ctx->optionsLock );
  ctx->hostGate = g;
  return TMC2_OK;
}

int tmc2_ctx_synchronize( tmc2_ctx* ctx ) {
  if ( !ctx ) return TMC2_E_INVALID;
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  return TMC2_OK;
}

int tmc2_ctx_stage_count( tmc2_ctx* ctx ) { return ctx ? int( ctx->stages.size() ) : 0; }
const char* tmc2_ctx_stage_name( tmc2_ctx* ctx, int i ) {
  return ( ctx && i >= 0 && i < int( ctx->stages.size() ) ) ? ctx->stages[i].name.c_str() : "";
}
double tmc2_ctx_stage_ms( tmc2_ctx* ctx, int i ) {
  if ( !ctx || i < 0 || i >= int( ctx->stages.size() ) ) return 0.0;
  auto& s = ctx->stages[i];
  ctx->foldStage( s );
  return s.ms;
}
void tmc2_ctx_stage_reset( tmc2_ctx* ctx ) {
  if ( !ctx ) return;
  for ( auto& s : ctx->stages ) {
    ctx->foldStage( s );
    s.ms    = 0.0;
    s.calls = 0;
  }
}
long tmc2_ctx_stage_calls( tmc2_ctx* ctx, int i ) {
  return ( ctx && i >= 0 && i < int( ctx->stages.size() ) ) ? ctx->stages[i].calls : 0;
}
void tmc2_ctx_set_timing( tmc2_ctx* ctx, int enabled ) {
  if ( ctx ) ctx->timing = enabled != 0;
}

int tmc2_frame_create( tmc2_ctx* ctx, const int16_t* xyz, const uint8_t* rgb, uint64_t n, tmc2_frame** out ) {
  if ( !ctx || !xyz || !out || n == 0 || n > 0x7FFFFFF0ull ) {
    setError( "frame_create: invalid argument" );
    return TMC2_E_INVALID;
  }
  *out = nullptr;
  ApiScope                    scope( ctx );
  std::unique_ptr<tmc2_frame> f( new tmc2_frame() );
  f->ticket.bind( ctx );
  f->ctx = ctx;
  f->n   = n;
  f->h_xyz.assign( xyz, xyz + 3 * n );
  for ( uint64_t i = 0; i < 3 * n; ++i ) f->geoMax = std::max( f->geoMax, xyz[i] );
  if ( rgb ) f->h_rgb.assign( rgb, rgb + 3 * n );
  // original-order points (one 8-byte record per point) go to HBM now; the k-d tree is built on first use
  std::vector<Pt> pts( n );
  for ( uint64_t i = 0; i < n; ++i ) pts[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0};
  TMC2_TRY( f->d_pts.alloc( n ) );
  hipStream_t s = ctx->stream;
  TMC2_HIP( hipMemcpyAsync( f->d_pts.p, pts.data(), n * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  if ( rgb ) {
    std::vector<uint8_t> c4( n * 4 );
    for ( uint64_t i = 0; i < n; ++i ) {
      c4[4 * i]     = rgb[3 * i];
      c4[4 * i + 1] = rgb[3 * i + 1];
      c4[4 * i + 2] = rgb[3 * i + 2];
      c4[4 * i + 3] = 0;
    }
    TMC2_TRY( f->d_rgb.alloc( n * 4 ) );
    TMC2_HIP( hipMemcpyAsync( f->d_rgb.p, c4.data(), n * 4, hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
  }
  TMC2_HIP( hipStreamSynchronize( s ) );
  *out = f.release();
  return TMC2_OK;
}

}  // extern "C"

namespace tmc2 {
// whether tmc2_segmenter_compute queues the refine step's geometry ahead of the orientation's host walk
// (tmc2_set_refine_overlap; unset: the environment variable TMC2_REFINE_OVERLAP decides, off without it)
// Options are per context (tmc2_ctx_set_option; defaults = the TMC2_* environment at tmc2_ctx_create).  The two process-wide
// setters of rounds 2-4 (tmc2_set_refine_overlap, tmc2_set_kdtree_placement) only preset what a context WITHOUT the option does.
const char* ctxOption( const tmc2_ctx* ctx, const char* key ) {
  if ( !ctx ) {
    const std::string name = std::string( "TMC2_" ) + key;
    return getenv( name.c_str() );
  }
  // a COPY of the value, in a slot of the calling thread (eight slots, used in turn): the map may change under a reader's feet --
  // options are set from the caller's thread while stage code of the context's worker thread looks them up -- and what
  // tmc2_ctx_get_option hands out must outlive the next set / unset of the key
  static thread_local std::string slots[8];
  static thread_local unsigned    turn = 0;
  std::lock_guard<std::mutex>     g( ctx->optionsLock );
  const auto                      it = ctx->options.find( key );
  if ( it == ctx->options.end() ) return nullptr;
  std::string& slot = slots[turn++ & 7u];
  slot              = it->second;
  return slot.c_str();
}
static std::atomic<int> g_refineOverlap{-1};
bool refineOverlap( const tmc2_ctx* ctx ) {
  if ( const char* e = ctxOption( ctx, "REFINE_OVERLAP" ) ) return e[0] == '1';
  return g_refineOverlap.load( std::memory_order_relaxed ) > 0;
}
// where the k-d trees are built (tmc2_set_kdtree_placement; TMC2_KDTREE_HOST=1 presets "host")
static std::atomic<int> g_kdtreeOnHost{-1};
int kdtreePlacement( const tmc2_ctx* ctx ) {
  if ( const char* e = ctxOption( ctx, "KDTREE_HOST" ) )
    if ( e[0] >= '0' && e[0] <= '2' ) return e[0] - '0';
  const int v = g_kdtreeOnHost.load( std::memory_order_relaxed );
  return v < 0 ? 0 : v;
}
int unionPrecheck( const tmc2_ctx* ctx ) {
  const char* e = ctxOption( ctx, "UF_PRECHECK" );
  return e ? ( e[0] != '0' ) : 1;
}
bool unionAgentScope( const tmc2_ctx* ctx ) {
  const char* e = ctxOption( ctx, "UF_SCOPE" );
  return e && e[0] == 'a';
}
bool unionCheck( const tmc2_ctx* ctx ) {
  const char* e = ctxOption( ctx, "UF_CHECK" );
  return e && e[0] == '1';
}
void setRefineOverlapDefault( int on ) { g_refineOverlap.store( on ? 1 : 0, std::memory_order_relaxed ); }
void setKdtreePlacement( int mode ) { g_kdtreeOnHost.store( mode < 0 || mode > 2 ? 0 : mode, std::memory_order_relaxed ); }
}  // namespace tmc2

const int* tmc2_ctx::constTable( uint64_t key, const std::vector<int>& host ) {
  auto it = constTables.find( key );
  if ( it != constTables.end() && it->second->host == host ) return it->second->dev.p;
  auto entry = std::make_unique<ConstTable>();
  if ( entry->dev.alloc( std::max<size_t>( host.size(), 1 ) ) != TMC2_OK ) return nullptr;
  entry->host = host;  // (kept: the key's identity, and the source of the copy below for as long as the runtime may read it)
  // The block comes from the context's pool, whose blocks may still be in use by work QUEUED on the context's stream (a stage hands
  // its buffers back without waiting): the upload has to be ordered on that stream too.  (Round 6's first version copied with
  // hipMemcpy -- the null stream, at once -- into a block the refinement's queued sweeps were still reading: a frame in a few
  // thousand went wrong, tests/test_gpu_fuzz.py seed 6.)
  if ( !host.empty() && hipMemcpyAsync( entry->dev.p, entry->host.data(), host.size() * sizeof( int ), hipMemcpyHostToDevice, stream ) != hipSuccess ) {
    tmc2::setError( "constTable: upload failed" );
    return nullptr;
  }
  const int* p = entry->dev.p;
  if ( it != constTables.end() ) retiredTables.push_back( std::move( it->second ) );  // (its block may still be read by queued work: freed with the context)
  constTables[key] = std::move( entry );
  return p;
}

int tmc2_frame::ensureTree() {
  if ( haveTree ) return TMC2_OK;
  const int placement = tmc2::kdtreePlacement( ctx );
  // adaptive: take a host slot if one is free right now, otherwise the device builds it (same tree either way)
  tmc2::HostGate gate( ctx, placement == 1 );
  if ( placement == 0 || !gate.held ) {
    const int sid = ctx->stageBegin( "kdtree_build" );
    TMC2_TRY( tmc2::buildKdTreeDevice( ctx, d_pts.p, n, d_ptsTree, d_perm, d_nodes, tree.lo, tree.hi, tree.depth ) );
    ctx->stageEnd( sid );
    haveTree = true;
    return TMC2_OK;
  }
  // built in page-locked staging (the orientation's row / sign staging is idle at this point) and uploaded as is
  Pt*       hp = ctx->hostD.get<Pt>( n );
  uint32_t* hi = ctx->hostA.get<uint32_t>( n );
  if ( !hp || !hi ) {
    setError( "kdtree: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  {
    const auto t0 = std::chrono::steady_clock::now();
    for ( uint64_t i = 0; i < n; ++i ) hp[i] = Pt{h_xyz[3 * size_t( i )], h_xyz[3 * size_t( i ) + 1], h_xyz[3 * size_t( i ) + 2], 0};
    tree.buildInPlace( hp, hi, n );
    const auto t1 = std::chrono::steady_clock::now();
    ctx->stageAddHostMs( "kdtree_build_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
  }
  gate.release();
  TMC2_TRY( d_ptsTree.alloc( n ) );
  TMC2_TRY( d_perm.alloc( n ) );
  TMC2_TRY( d_nodes.alloc( tree.nodes.size() ) );
  hipStream_t s = ctx->stream;
  TMC2_HIP( hipMemcpyAsync( d_ptsTree.p, hp, n * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( d_perm.p, hi, n * sizeof( uint32_t ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( d_nodes.p, tree.nodes.data(), tree.nodes.size() * sizeof( KdNode ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  haveTree = true;
  return TMC2_OK;
}

extern "C" {

/* replaces PCCKdTree::init explicitly (otherwise built on first use) */
int tmc2_kdtree_build( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return f->ensureTree();
}

int tmc2_frame_get_kdtree_order( tmc2_frame* f, uint32_t* perm, int32_t* depth ) {
  if ( !f || !perm ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( f->ensureTree() );
  TMC2_HIP( hipMemcpyAsync( perm, f->d_perm.p, f->n * sizeof( uint32_t ), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  if ( depth ) *depth = f->tree.depth;
  return TMC2_OK;
}

void tmc2_set_kdtree_placement( int mode ) { tmc2::setKdtreePlacement( mode ); }
void tmc2_set_refine_overlap( int on ) { tmc2::setRefineOverlapDefault( on ); }

void tmc2_frame_destroy( tmc2_frame* f ) {
  if ( !f ) return;
  ApiScope scope( f->ctx );
  (void)hipStreamSynchronize( f->ctx->stream );
  delete f;
}

uint64_t tmc2_frame_point_count( const tmc2_frame* f ) { return f ? f->n : 0; }

int tmc2_frame_reset( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  f->haveTree = f->haveKnn = f->haveNormals = f->havePartition = f->haveMutual = false;
  f->havePatches = f->havePacking = f->haveGeometryImages = f->haveAttributeImages = f->haveReconstruction = false;
  f->haveBoundaryTypes = f->haveColors16 = f->haveSmoothed = f->haveRgbPost = f->haveAttr16 = false;
  f->patches.clear();
  f->packOrder.clear();
  f->packMatch.clear();
  f->depthCount = f->occCount = 0;
  f->rounds = f->packedHeight = f->packedWidth = 0;
  {
    tmc2::ApiScope scope( f->ctx );
    f->refineJob.reset();
  }
  return TMC2_OK;
}

int tmc2_kdtree_search( tmc2_frame* f, const int16_t* queries, uint64_t nq, int k, uint32_t* idx, uint32_t* dist2 ) {
  if ( !f || !queries || !idx || nq == 0 || nq > ( 1ull << 32 ) ) {
    setError( "kdtree_search: invalid argument" );
    return TMC2_E_INVALID;
  }
  if ( k != 1 && k != 4 && k != 8 && k != 16 ) {  // (the kernels are instantiated for these; checked before anything is sized by k)
    setError( "kdtree_search: k = %d unsupported (1, 4, 8 or 16)", k );
    return TMC2_E_UNSUPPORTED;
  }
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( f->ensureTree() );
  std::vector<Pt> q( nq );
  bool bounded = true;  // every coordinate in [-4096, 12287]: the packed LDS-stack traversal applies
  for ( uint64_t i = 0; i < nq; ++i ) {
    q[i] = Pt{queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], 0};
    for ( int d = 0; d < 3; ++d ) bounded = bounded && queries[3 * i + d] >= -4096 && queries[3 * i + d] <= 12287;
  }
  DevBuf<Pt>       d_q;
  DevBuf<uint32_t> d_idx, d_dist;
  TMC2_TRY( d_q.alloc( nq ) );
  TMC2_TRY( d_idx.alloc( nq * size_t( k ) ) );
  if ( dist2 ) TMC2_TRY( d_dist.alloc( nq * size_t( k ) ) );
  hipStream_t s = f->ctx->stream;
  TMC2_HIP( hipMemcpyAsync( d_q.p, q.data(), nq * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_TRY( launchKnnQueries( f, d_q.p, nq, k, d_idx.p, dist2 ? d_dist.p : nullptr, bounded ) );
  TMC2_HIP( hipMemcpyAsync( idx, d_idx.p, nq * size_t( k ) * 4, hipMemcpyDeviceToHost, s ) );
  if ( dist2 ) TMC2_HIP( hipMemcpyAsync( dist2, d_dist.p, nq * size_t( k ) * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

int tmc2_normals_compute_normals( tmc2_frame* f, int k ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( f->ensureTree() );
  if ( !f->haveKnn || f->k != k ) TMC2_TRY( launchKnnSelf( f, k ) );
  return launchNormals( f );
}

int tmc2_normals_orient( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return orientNormalsHost( f );
}

int tmc2_normals_compute( tmc2_frame* f, int k, int orientation ) {
  TMC2_TRY( tmc2_normals_compute_normals( f, k ) );
  if ( orientation == 1 ) return tmc2_normals_orient( f );
  if ( orientation == 0 ) return TMC2_OK;
  setError( "normalOrientation=%d unsupported (0 none, 1 spanning tree)", orientation );
  return TMC2_E_UNSUPPORTED;
}

int tmc2_frame_get_normals( tmc2_frame* f, double* normals ) {
  if ( !f || !normals || !f->haveNormals ) {
    setError( "get_normals: no normals" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  TMC2_HIP( hipMemcpyAsync( normals, f->d_normals.p, f->n * 3 * sizeof( double ), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}

int tmc2_frame_set_normals( tmc2_frame* f, const double* normals ) {
  if ( !f || !normals ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( f->d_normals.alloc( f->n * 3 ) );
  TMC2_HIP( hipMemcpyAsync( f->d_normals.p, normals, f->n * 3 * sizeof( double ), hipMemcpyHostToDevice, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  f->haveNormals = true;
  return TMC2_OK;
}

int tmc2_frame_get_adjacency( tmc2_frame* f, uint32_t* adj ) {
  if ( !f || !adj || !f->haveKnn ) {
    setError( "get_adjacency: no adjacency" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  TMC2_HIP( hipMemcpyAsync( adj, f->d_knn.p, f->n * size_t( f->k ) * 4, hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}

int tmc2_weight_normal( tmc2_frame* f, int geometryBitDepth3D, double minWeightEPP, double weight[3] ) {
  if ( !f || !weight ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return weightNormal( f, geometryBitDepth3D, minWeightEPP, weight );
}

int tmc2_segmenter_initial_segmentation( tmc2_frame* f, const double weight[3] ) {
  if ( !f || !weight ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return launchInitialSegmentation( f, weight );
}

int tmc2_segmenter_refine_grid_based( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim,
                                      int searchRadius ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return refineGridBased( f, maxNNCount, lambda, iterationCount, voxDim, searchRadius );
}

int tmc2_frame_get_partition( tmc2_frame* f, uint32_t* partition ) {
  if ( !f || !partition || !f->havePartition ) {
    setError( "get_partition: no partition" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  std::vector<uint8_t> tmp( f->n );
  TMC2_HIP( hipMemcpyAsync( tmp.data(), f->d_partition.p, f->n, hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  for ( uint64_t i = 0; i < f->n; ++i ) partition[i] = tmp[i];
  return TMC2_OK;
}

int tmc2_frame_set_partition( tmc2_frame* f, const uint32_t* partition ) {
  if ( !f || !partition ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  std::vector<uint8_t> tmp( f->n );
  for ( uint64_t i = 0; i < f->n; ++i ) {
    if ( partition[i] > 5 ) {  // six projection planes: the labels index 6-bin histograms and the orientation table
      setError( "set_partition: label %u of point %llu out of range (0..5)", partition[i], (unsigned long long)i );
      return TMC2_E_INVALID;
    }
    tmp[i] = uint8_t( partition[i] );
  }
  TMC2_TRY( f->d_partition.alloc( f->n ) );
  TMC2_HIP( hipMemcpyAsync( f->d_partition.p, tmp.data(), f->n, hipMemcpyHostToDevice, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  f->havePartition = true;
  return TMC2_OK;
}

/* ---- host-only pieces of the path, callable without a device (exercised by the CPU test tier) ---- */
int tmc2_host_kdtree_build( const int16_t* xyz, uint64_t n, uint32_t* perm, uint64_t* nodeCount, int32_t* depth ) {
  if ( !xyz || !perm || n == 0 ) return TMC2_E_INVALID;
  KdTreeHost t;
  t.build( xyz, n );
  memcpy( perm, t.perm.data(), n * sizeof( uint32_t ) );
  if ( nodeCount ) *nodeCount = t.nodes.size();
  if ( depth ) *depth = t.depth;
  return TMC2_OK;
}

int tmc2_host_orient_normals( const int16_t* xyz, uint64_t n, const uint32_t* knn, int k, double* normals ) {
  if ( !xyz || !knn || !normals || k < 1 ) return TMC2_E_INVALID;
  orientNormalsSpanningTree( xyz, n, knn, k, normals, nullptr );
  return TMC2_OK;
}

}  // extern "C"
