// internal.h -- shared internals of libtmc2hip.so (MI355X / gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <functional>
#include <vector>

#include "tmc2hip.h"

namespace tmc2 {

void setError( const char* fmt, ... );

#define TMC2_HIP( expr )                                                                        \
  do {                                                                                          \
    hipError_t e_ = ( expr );                                                                   \
    if ( e_ != hipSuccess ) {                                                                   \
      tmc2::setError( "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString( e_ ) ); \
      return TMC2_E_HIP;                                                                        \
    }                                                                                           \
  } while ( 0 )

#define TMC2_TRY( expr )          \
  do {                            \
    int r_ = ( expr );            \
    if ( r_ != TMC2_OK ) return r_; \
  } while ( 0 )

// Caching device allocator, one per context.  hipMalloc / hipFree are slow and hipFree synchronises the
// whole device, which would serialise the per-frame host threads; blocks are therefore recycled by size
// class (powers of two) and only returned to the driver when the context is destroyed.
struct DevicePool {
  std::mutex                                 lock;
  std::map<size_t, std::vector<void*>>       freeBlocks;  // size class -> blocks
  size_t                                     bytesHeld = 0;
  // tmc2_ctx_reserve: slabs allocated up front; a miss of the free lists is carved from them (no hipMalloc -- which
  // synchronises the device under every frame in flight -- while they last)
  struct Slab {
    char*  base;
    size_t size, used;
  };
  std::vector<Slab>                          slabs;
  size_t                                     mallocCalls = 0, carved = 0;  // hipMalloc calls on a miss; blocks carved from a slab
  double                                     mallocMs = 0.0;               // host time spent in those hipMalloc calls
  int  reserve( size_t bytes );
  static size_t sizeClass( size_t bytes ) {
    size_t c = 256;
    while ( c < bytes ) c <<= 1;
    return c;
  }
  int  acquire( size_t bytes, void** out, size_t* got );
  void recycle( void* p, size_t cls );
  void drain();
};
DevicePool* currentPool();  // pool of the context the calling thread entered (see ApiScope), or nullptr

// Owning device buffer; memory comes from the current context's pool.
template <typename T>
struct DevBuf {
  T*          p = nullptr;
  size_t      count = 0;
  size_t      cls   = 0;
  DevicePool* pool  = nullptr;
  DevBuf() = default;
  DevBuf( const DevBuf& ) = delete;
  DevBuf& operator=( const DevBuf& ) = delete;
  ~DevBuf() { release(); }
  void release() {
    if ( p ) {
      if ( pool )
        pool->recycle( p, cls );
      else
        (void)hipFree( p );
    }
    p     = nullptr;
    count = 0;
    pool  = nullptr;
  }
  int alloc( size_t n ) {
    if ( n <= count && p ) return TMC2_OK;
    release();
    if ( n == 0 ) n = 1;
    DevicePool* pl = currentPool();
    if ( pl ) {
      void*  q   = nullptr;
      size_t got = 0;
      if ( n > ( size_t( 1 ) << 40 ) / sizeof( T ) ) {
        setError( "device allocation of %zu elements refused", n );
        return TMC2_E_INVALID;
      }
      TMC2_TRY( pl->acquire( n * sizeof( T ), &q, &got ) );
      p     = reinterpret_cast<T*>( q );
      cls   = got;
      pool  = pl;
      count = got / sizeof( T );
    } else {
      TMC2_HIP( hipMalloc( reinterpret_cast<void**>( &p ), n * sizeof( T ) ) );
      count = n;
    }
    return TMC2_OK;
  }
  size_t bytes() const { return count * sizeof( T ); }
};

// ---- k-d tree, flattened for the device (one 16-byte record per node, pre-order: left child = id+1)
struct alignas( 16 ) KdNode {
  int32_t a;       // leaf: first point (tree order)   inner: left child id (== own id + 1)
  int32_t b;       // leaf: one-past-last point        inner: right child id
  int16_t divlow;  // inner: upper bound of the left child's box on dim
  int16_t divhigh; // inner: lower bound of the right child's box on dim
  int32_t dim;     // -1 for a leaf
};

// AoS point with padding: one 8-byte load per point
struct alignas( 8 ) Pt {
  int16_t x, y, z, w;
};
struct KdTreeHost {
  std::vector<uint32_t> perm;     // tree order -> original index
  std::vector<Pt>       ptsTree;  // the points in tree order
  std::vector<KdNode>   nodes;
  int32_t               lo[3], hi[3];
  int                   depth = 0;
  void                  build( const int16_t* xyz, size_t n );              // fills perm / ptsTree
  void                  buildInPlace( Pt* pts, uint32_t* ind, size_t n );   // caller-owned storage; perm / ptsTree stay empty
};


// ---- loads that tolerate a stale value ------------------------------------------------------------------------------
// Words that only move one way (union-find links: always to a member of the same set of smaller priority; running minima /
// maxima; relaxed labels) may be read from a view that lags behind the other XCDs': a stale value costs a detour or a
// redundant atomic, never a wrong answer, as long as the view is no older than the kernel's start (tools/gpu/stale_view.hip
// measures exactly that on the device; the soak test re-runs the GOF with every such load at agent scope).  Workgroup scope
// = served by the CU's L1 / the XCD's L2; agent scope = a trip past the L2 on every use (gfx942 / gfx950: the L2s of the
// XCDs are not coherent with each other, caches are written back / invalidated at kernel boundaries).
// This is the ONE place the scope is chosen.  agent = true: the formally clean form (TMC2_UF_SCOPE=agent).
#if defined( __HIP_DEVICE_COMPILE__ ) && !defined( __gfx950__ ) && !defined( __gfx942__ )
#error "loadStaleOk: the stale-view argument is written for the gfx942 / gfx950 cache hierarchy"
#endif
#if defined( __HIPCC__ )  // (device helper; the host-only translation units also build with a plain host compiler: tools/asan_host_gcc.sh)
template <typename T>
__device__ __forceinline__ T loadStaleOk( const T* p, bool agent = false ) {
  return agent ? __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT )
               : __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
}
#endif

// The element of this lane in a one-element-per-lane pass whose grid is a multiple of 8 blocks: XCD x works through the x-th eighth
// of the blocks (block b runs on XCD b % 8 -- observed, not promised: only speed depends on it), so that neighbouring blocks --
// neighbouring regions of the cloud in every array that is in scan order -- share an L2 instead of being dealt round-robin over
// the eight.  Any other grid: the blocks as they come.  (chunkedGrid rounds a grid up; the surplus blocks find no element.)
#if defined( __HIPCC__ )
__device__ __forceinline__ uint32_t chunkedIndex() {
  const uint32_t b = ( gridDim.x & 7u ) ? blockIdx.x : ( blockIdx.x & 7u ) * ( gridDim.x >> 3 ) + ( blockIdx.x >> 3 );
  return b * blockDim.x + threadIdx.x;
}
#endif
inline uint32_t chunkedGrid( uint32_t blocks, bool chunked = true ) {
  return chunked ? ( blocks + 7u ) & ~7u : ( ( blocks & 7u ) ? blocks : blocks + 1u );
}

// pointToPixel of a reconstructed point in one word: canvas x, y (15 bits each: canvases up to kMaxCanvasDim pixels a side,
// enforced where a canvas size enters -- generateGeometryImages, the decoder frame), map layer, "a D1 point follows"
constexpr int kMaxCanvasDim = 32767;
__host__ __device__ __forceinline__ uint32_t packPixel( int x, int y, int layer, bool hasD1 ) {
  return uint32_t( x ) | ( uint32_t( y ) << 15 ) | ( uint32_t( layer ) << 30 ) | ( hasD1 ? 1u << 31 : 0u );
}
__host__ __device__ __forceinline__ uint32_t pixelX( uint32_t p ) { return p & 0x7FFFu; }
__host__ __device__ __forceinline__ uint32_t pixelY( uint32_t p ) { return ( p >> 15 ) & 0x7FFFu; }
__host__ __device__ __forceinline__ uint32_t pixelLayer( uint32_t p ) { return ( p >> 30 ) & 1u; }
__host__ __device__ __forceinline__ bool     pixelHasD1( uint32_t p ) { return ( p >> 31 ) != 0u; }

// placement + projection of one patch on the device, in packing order (shared by the image kernels)
struct PlaceDev {
  int32_t u0, v0, orient;
  int32_t sizeU, sizeV, sizeU0, sizeV0;
  int32_t tileBase;
  int32_t u1, v1, d1;
  int32_t axN, axT, axB, mode;
  int32_t pad;
  int64_t depthOff;
};

// a k-d tree resident on the device (tree-order points, permutation, nodes) -- what the k-NN kernels traverse
struct TreeDev {
  const Pt*       ptsTree = nullptr;
  const uint32_t* perm    = nullptr;
  const KdNode*   nodes   = nullptr;
  int32_t         lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int             depth = 0;
  uint64_t        n     = 0;
  bool            queriesBounded = false;  // the caller vouches: every query coordinate lies in [-4096, 12287]
  bool            queriesTight   = false;  // ... and even in [0, 8191] (a frame's own points / reconstruction): larger trees keep the LDS stack
};

// GPU time per named stage / kernel: hipEvent pairs recorded on the context's stream, folded lazily when the
// totals are queried (never blocks the host while work is being queued).
struct StageTimer {
  std::string                                    name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double                                         ms    = 0.0;  // folded GPU (or host) milliseconds
  long                                           calls = 0;
};

}  // namespace tmc2

namespace tmc2 {
// Page-locked host staging buffer that only ever grows; lives in the context so that the per-frame host steps
// (k-d tree build, normal orientation) neither re-fault ~100 MB of fresh pages nor copy through pageable memory.
struct PinnedBuf {
  void*  p     = nullptr;
  size_t bytes = 0;
  PinnedBuf() = default;
  PinnedBuf( const PinnedBuf& ) = delete;
  PinnedBuf& operator=( const PinnedBuf& ) = delete;
  ~PinnedBuf() {
    if ( p ) (void)hipHostFree( p );
  }
  template <typename T>
  T* get( size_t count ) {
    const size_t need = count * sizeof( T );
    if ( need > bytes ) {
      if ( p ) (void)hipHostFree( p );
      p     = nullptr;
      bytes = 0;
      const size_t cap = need + need / 4;
      if ( hipHostMalloc( &p, cap, hipHostMallocDefault ) != hipSuccess ) {
        p = nullptr;
        return nullptr;
      }
      bytes = cap;
    }
    return reinterpret_cast<T*>( p );
  }
};
}  // namespace tmc2

struct tmc2_ctx {
  // A context outlives its frames: tmc2_ctx_destroy on a context that still has frames only asks for the destruction, which the
  // last frame to go carries out (a frame's device buffers return to the context's pool when the frame is destroyed).
  std::atomic<int>              liveFrames{0};
  std::atomic<bool>             destroyRequested{false}, destroyClaimed{false};
  int                           device = 0;
  tmc2::DevicePool              pool;
  tmc2::PinnedBuf               hostA, hostB, hostC, hostD, hostE;  // staging for the host-side steps
  // small tables of a stage's host decisions, page-locked so that their copies are plain DMA with no staging copy behind them:
  // hostTables host -> device (patch records and tile lists of a patch round, the placement table), hostRecords device -> host (a
  // round's boxes and counters).  Reused from round to round: every user synchronises the stream before the next one refills them.
  tmc2::PinnedBuf               hostTables, hostRecords;
  std::vector<int32_t>          orientScratch;                      // per-vertex state of the orientation walk
  tmc2::DevBuf<uint32_t>        gridTable;       // persistent dense voxel-key table (kept all-ones between uses)
  tmc2::DevBuf<uint2>           gridBits;        // its occupancy, .x: one bit per key (kept all-zero between uses: S5 probes ball rows in it), .y: occupied keys below the word
  int                           refineCapTier = 0;  // the smallest neighbourhood-kernel instantiation that held this context's last frames
  uint32_t                      refineHitsPerVoxel = 640;  // room per voxel for the balls' hits kept between S5's two passes over the balls (grows when a frame runs out)
  tmc2::DevBuf<unsigned long long> scanState;    // look-back state of exclusiveScanU32: [0] tile tickets, [1 + t] tile t (epoch-tagged)
  uint32_t                      scanEpoch = 0, scanTickets = 0;
  tmc2::DevBuf<uint32_t>        voxelBitmap;     // dense 3-D occupancy bitmap of the resampled cloud (S9)
  std::map<uint32_t, int>       kdLevelHint;     // levels of level passes the last device k-d tree of ~ this size took (by n >> 15)
  // small tables that depend on the parameters only (the probe offsets of S9, the ball rows / cells of S5): uploaded once per context
  // and key, not once per frame (round 6: each was a pageable host-to-device copy -- a staging copy and a blit -- on every frame's chain)
  struct ConstTable {
    tmc2::DevBuf<int> dev;
    std::vector<int>  host;
  };
  std::map<uint64_t, std::unique_ptr<ConstTable>> constTables;
  std::vector<std::unique_ptr<ConstTable>>        retiredTables;
  const int* constTable( uint64_t key, const std::vector<int>& host );  // nullptr on failure (the error is set); the upload is ordered on the context's stream
  // Page-locked words the device writes and the host reads without a copy (allocated with the context; kernels take the pointer as
  // it is: page-locked host memory is mapped into the device's address space): the answers of the stages' host round trips, one
  // 64-byte line each (kAnswer*: which stage owns which line).  A kernel stores its answer with ONE plain store from its last
  // workgroup; the host reads it after the hipStreamSynchronize it needed anyway -- no staging copy, no blit kernel on the chain.
  uint32_t* mailbox = nullptr;
  static constexpr size_t kMailboxWords = 1024;
  volatile uint32_t* answerLine( int line ) const { return mailbox + size_t( line ) * 16; }
  enum { kAnswerRefineVoxels = 1, kAnswerPatchRound = 2, kAnswerRecon = 3, kAnswerOrientHead = 4, kAnswerTreeDepth = 5, kAnswerGeoError = 6, kAnswerAttrError = 7, kAnswerTreeLevels = 8 /* .. 12 */ };
  // Per-context options (tmc2_ctx_set_option): key = the name of the knob without its TMC2_ prefix.  Filled ONCE, when the
  // context is created, from the process environment (every TMC2_* variable: the defaults); nothing in the library reads the
  // environment after that, and nothing is process-wide: two encoders of one process can run with different settings.
  std::map<std::string, std::string> options;
  mutable std::mutex                 optionsLock;
  std::shared_ptr<tmc2_host_gate>    hostGate;  // this encoder's budget of host-resident steps (guarded by optionsLock); null: the process default
  hipStream_t                   stream = nullptr;
  std::vector<tmc2::StageTimer> stages;
  std::vector<hipEvent_t>       freeEvents;
  bool                          timing = true;
  int                           cuCount = 256;
  void foldStage( tmc2::StageTimer& t );
  int  stageBegin( const char* name );
  void stageEnd( int id );
  void stageAddHostMs( const char* name, double ms );
};

namespace tmc2 {
void destroyContextNow( tmc2_ctx* ctx );
// first member of a frame, so destroyed last: counts the frame in its context and, when it was the last one of a context whose
// destruction has been asked for, destroys the context
struct FrameTicket {
  tmc2_ctx* ctx = nullptr;
  void      bind( tmc2_ctx* c ) {
    ctx = c;
    c->liveFrames.fetch_add( 1 );
  }
  ~FrameTicket() {
    if ( ctx && ctx->liveFrames.fetch_sub( 1 ) == 1 && ctx->destroyRequested.load() && !ctx->destroyClaimed.exchange( true ) )
      destroyContextNow( ctx );
  }
};
}  // namespace tmc2

struct tmc2_frame {
  tmc2::FrameTicket ticket;
  tmc2_ctx* ctx = nullptr;
  uint64_t  n   = 0;
  int       k   = 0;  // k of the resident adjacency
  // host side
  std::vector<int16_t> h_xyz;
  std::vector<uint8_t> h_rgb;
  tmc2::KdTreeHost     tree;
  bool                 haveTree = false;
  int                  ensureTree();  // builds + uploads the k-d tree on first use (S1 belongs to the timed path)
  // device work that needs the points only and may run while the host walks the orientation graph (S3): called by the
  // orientation step right before its sequential host part, when set (tmc2_segmenter_compute: the refine step's geometry)
  std::function<int()>  beforeHostWalk;
  std::shared_ptr<void> refineJob;  // the refine step's geometry, prepared ahead (refine.hip)
  // device side
  tmc2::DevBuf<tmc2::Pt>     d_pts;       // original order
  tmc2::DevBuf<tmc2::Pt>     d_ptsTree;   // tree order
  tmc2::DevBuf<uint32_t>     d_perm;      // tree order -> original index
  tmc2::DevBuf<tmc2::KdNode> d_nodes;
  tmc2::DevBuf<uint8_t>      d_rgb;       // [n][4] (rgb + pad)
  tmc2::DevBuf<uint32_t>     d_knn;       // [n][k]
  tmc2::DevBuf<double>       d_normals;   // [n][3]
  tmc2::DevBuf<uint8_t>      d_partition; // [n]
  bool haveKnn = false, haveNormals = false, havePartition = false;
  tmc2::DevBuf<uint16_t>     d_mutual;    // [n] bit j: knn[i][j] lists i in its own row (k = 16); shared by S3 and S7
  bool haveMutual = false;
  int16_t geoMax = 0;  // largest coordinate (grid geometry of S5)
  // patches: records on the host, depth / occupancy pools resident on the device
  std::vector<tmc2_patch> patches;
  tmc2::DevBuf<int16_t>   d_depth0, d_depth1;  // patch-local depth maps, all patches back to back
  tmc2::DevBuf<uint8_t>   d_occupancy;         // per-block occupancy, all patches back to back
  int64_t                 depthCount = 0, occCount = 0;
  int                     rounds     = 0;
  std::vector<int32_t>    packMatch;           // S10': per list position the matched position in the previous frame, -1
  bool                    havePatches = false;
  int                     growPools();         // make the pools hold depthCount / occCount entries (keeps content)
  // packing + canvases (phase A images)
  std::vector<int32_t>    packOrder;            // packing order: list position -> patch index
  int                     packedHeight = 0, packedWidth = 0;  // tile size as the reference's packers leave it
  bool                    havePacking = false, haveGeometryImages = false;
  int                     canvasW = 0, canvasH = 0, occPrecision = 0;
  tmc2::DevBuf<uint8_t>   d_occMap;             // W*H precise occupancy
  tmc2::DevBuf<uint8_t>   d_occVideo;           // (W/p)*(H/p)
  tmc2::DevBuf<uint32_t>  d_blockToPatch;       // (W/16)*(H/16), list position + 1
  tmc2::DevBuf<uint16_t>  d_geo;                // 2 maps * W*H (D0 then D1), luma only (chroma planes are all-zero)
  tmc2::DevBuf<tmc2::PlaceDev> d_place;         // patches in packing order
  tmc2::DevBuf<uint32_t>  d_tilePatch;          // 16x16 patch blocks -> list position
  uint32_t                tileCount = 0;
  // phase B: reconstruction + attribute images
  uint64_t                reconCount = 0;
  bool                    haveAttributeImages = false, haveReconstruction = false;
  tmc2::DevBuf<tmc2::Pt>  d_recon;              // reconstructed points (generatePointCloud order)
  tmc2::DevBuf<uint32_t>  d_pointToPixel;       // tmc2::packPixel: x | y << 15 | layer << 30 | hasD1 << 31
  tmc2::DevBuf<uint8_t>   d_reconRgb;           // [M][4]
  tmc2::DevBuf<uint8_t>   d_attr;               // [2 maps][3 channels][H][W]
  // post-reconstruction tail (post_reconstruct.hip): per reconstructed point
  tmc2::DevBuf<uint8_t>   d_boundaryType;       // 0 inner, 1 boundary, 3 moved by the geometry smoothing
  tmc2::DevBuf<uint64_t>  d_colors16;           // 16-bit colours as (c0, c1, c2, 0) packed in 8 bytes
  tmc2::DevBuf<tmc2::Pt>  d_reconSmoothed;      // positions after the geometry smoothing
  tmc2::DevBuf<uint8_t>   d_rgbPost;            // [M][4] 8-bit RGB of the finished cloud
  tmc2::DevBuf<uint16_t>  d_attr16;             // decoded attribute frames, 16-bit 4:4:4: [2 maps][3 channels][H][W]
  bool                    haveAttr16 = false;
  bool                    haveBoundaryTypes = false, haveColors16 = false, haveSmoothed = false, haveRgbPost = false;
  tmc2::KdTreeHost        reconTree;
  tmc2::DevBuf<tmc2::Pt>  d_reconTreePts;
  tmc2::DevBuf<uint32_t>  d_reconPerm;
  tmc2::DevBuf<tmc2::KdNode> d_reconNodes;
};

namespace tmc2 {
// Gate around the host-resident, cache-hungry steps (k-d tree build, normal orientation): each walks a ~100 MB working set, so
// running more of them at once than there are last-level-cache domains makes all of them slower.  Frames beyond the limit wait here
// while their siblings' GPU phases proceed.  The gate is the context's own (tmc2_ctx_set_host_gate: one per encoder) or, without
// one, the process' default (tmc2_set_host_parallelism).  0 = unlimited.
struct HostGate {
  explicit HostGate( const tmc2_ctx* ctx, bool wait = true );  // wait = false: take a slot only if one is free right now
  ~HostGate();
  void release();
  bool held = false;
  std::shared_ptr<tmc2_host_gate> gate;
};
void setHostParallelism( int n );
// RAII guard of every extern "C" entry: selects the device and makes the context's pool current
struct ApiScope {
  tmc2_ctx* prev;
  explicit ApiScope( tmc2_ctx* ctx );
  ~ApiScope();
};
// kernels / stage launchers (each returns TMC2_OK or an error code; all work is queued on ctx->stream)
int launchKnnSelf( tmc2_frame* f, int k );
int launchKnnQueries( tmc2_frame* f, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx, uint32_t* d_dist,
                      bool queriesBounded );
int launchKnnTree( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx,
                   uint32_t* d_dist, const char* stage );
// the same search in two launches: the queries with an identical point in the tree first (d_easy[j]: the first result, or
// 0xFFFFFFFF), then the compacted rest through the ordinary kernel, rows in place (knn.hip: what a caller may do with d_easy)
int launchKnnSplit( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_easy, uint32_t* d_idx,
                    uint32_t* d_dist, const char* stage, bool uniqueTreeRows = false );
TreeDev frameTree( const tmc2_frame* f );
int generateAttributeImages( tmc2_frame* f );
int reconstructPointCloud( tmc2_frame* f );
inline void invalidateReconstruction( tmc2_frame* f ) {  // new or replaced canvases: everything derived from them is stale
  f->haveAttributeImages = f->haveReconstruction = false;
  f->haveBoundaryTypes = f->haveColors16 = f->haveSmoothed = f->haveRgbPost = false;
}
int uploadPlacement( tmc2_frame* f );
int launchNormals( tmc2_frame* f );
int orientNormalsHost( tmc2_frame* f );
int ensureMutualMask( tmc2_frame* f );  // k = 16 only
int launchEdgeDots( tmc2_frame* f, double* d_edgeDot );
constexpr size_t kOrientNegCountWords = 64 * 32;  // d_negCount: 64 counters, one per 128 bytes
int launchApplyOrientation( tmc2_frame* f, const int8_t* d_sign, uint32_t* d_negCount );
// contracted orientation graph (orient_host.cpp): clusters of mutual strong edges, their cross edges grouped by source
struct OrientCrossEdge {
  uint32_t u, v;  // start / end vertex (original indices)
  double   d;     // n_u . n_v on the original normals
};
struct OrientContraction {
  const uint32_t*        root;    // [n]   cluster (= root vertex id) of every vertex
  const uint8_t*         parity;  // [n]   1: the vertex' sign is the opposite of its cluster's
  const uint32_t*        off;     // [n+1] cross edges of cluster c: edges[off[c] .. off[c+1])
  const OrientCrossEdge* edges;
};
bool orientContractedSigns( size_t n, const OrientContraction& g, double tau, int8_t* clusterSign, uint32_t* component,
                            std::vector<uint32_t>& seeds, void* scratch );
void resolveSeedSigns( size_t n, const OrientContraction& g, int kNN, const std::vector<uint32_t>& seeds,
                       const uint32_t* component, const std::function<const uint32_t*( size_t )>& rowOf,
                       const std::function<const double*( size_t, int )>& normalOf, const int16_t* xyz0,
                       int8_t* clusterSign );
// the contracted graph in compact form (device contraction, orient_contract.hip): clusters numbered by first member, per
// ordered pair of clusters the one light cross edge that can be accepted + the strong one-way edges
struct OrientCompactEdge {
  uint32_t u, v, c2, pad;  // start / end vertex (original indices: the tie order of the reference's queue), cluster of v
  double   d;              // n_u . n_v on the original normals, negated if the parities of u and v differ
};
struct OrientClusterRec {
  uint32_t off;        // edges of cluster c: edges[rec[c].off .. rec[c + 1].off)
  uint32_t seedPoint;  // first member (smallest index) of the cluster
  uint32_t seedParity; // its parity
};
struct OrientCompact {
  uint32_t                 clusters;
  const OrientClusterRec*  rec;  // [clusters + 1]
  const OrientCompactEdge* edges;
};
bool orientCompactSigns( const OrientCompact& g, double tau, int8_t* clusterSign, uint32_t* component, std::vector<uint32_t>& seeds,
                         std::vector<uint32_t>& seedClusters );
void resolveSeedSignsCompact( const OrientCompact& g, int kNN, const std::vector<uint32_t>& seeds, const std::vector<uint32_t>& seedClusters,
                              const uint32_t* component, const uint32_t* who, const double* normals, const int16_t* xyz0, int8_t* clusterSign );
int gatherSeedTables( tmc2_frame* f, const uint32_t* d_cid, const uint8_t* d_parity, const std::vector<uint32_t>& seeds,
                      std::vector<uint32_t>& who, std::vector<double>& normals );
int orientSpanningTreeSigns( const int16_t* xyz, size_t n, const uint32_t* knn, int k, const double* normals,
                             const double* edgeDot, int8_t* sign, void* scratch, bool tryContraction, const tmc2_ctx* ctx );
double orientFirstTau( const tmc2_ctx* ctx );
int contractOrientationDevice( tmc2_frame* f, double tau, DevBuf<uint32_t>& d_cid,
                               DevBuf<uint8_t>& d_parity, OrientCompact& g, bool& ok );
int launchClusterSigns( tmc2_frame* f, const uint32_t* d_cid, const uint8_t* d_parity, const int8_t* d_clusterSign,
                        int8_t* d_sign );
int launchInitialSegmentation( tmc2_frame* f, const double weight[3] );
int weightNormal( tmc2_frame* f, int bits, double minWeightEPP, double w[3] );
int segmentPatches( tmc2_frame* f, const tmc2_segmenter_params* sp );
int packFlexibleHost( tmc2_frame* f, int presetWidth, int occRes, int numTilesHor, double tileHeightToWidthRatio );
// S10' second half (gpa.cpp): one frame of the GOF as the global patch allocation sees it
struct GpaFrameIO {
  std::vector<tmc2_patch> list;   // patches in list order (occOffset into occ)
  std::vector<uint8_t>    occ;    // block-occupancy pool
  std::vector<int32_t>    match;  // per list position: matched list position in the previous frame, -1
  int                     width = 0, height = 0;  // tile size, pixels
};
int globalPatchAllocationCore( std::vector<GpaFrameIO>& frames, int minW, int minH, int occRes );
int installPacking( tmc2_frame* f, const GpaFrameIO& g );
int generateGeometryImages( tmc2_frame* f, int W, int H, int occRes, int occPrecision );
int rgb444ToYuv420Device( tmc2_ctx* ctx, const uint8_t* d_rgb, int W, int H, int filter, uint8_t* d_yuv );
int yuv420ToYuv444Device( tmc2_ctx* ctx, const uint8_t* d_yuv, int W, int H, int filter, uint16_t* d_out );
int refineGridBased( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim, int searchRadius );
// the point-only half of it ahead of time (voxels, neighbourhood rows): queued, not waited for; refineGridBased picks it up
void setRefineOverlapDefault( int on );
bool refineOverlap( const tmc2_ctx* ctx );  // option REFINE_OVERLAP; unset: the process default of tmc2_set_refine_overlap
int refinePrepareGeometry( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim, int searchRadius );
// exclusive prefix sum of n uint32 (in -> out, may alias); returns the total through *d_total (device) if non-null
// Grid of a kernel that walks its items with a stride loop: at most eight 256-thread workgroups per CU -- one full set of
// resident waves.  A kernel of 50 K trivial workgroups is bound by workgroup dispatch (~ 500 per microsecond chip-wide), and
// sixteen frames in flight queue their dispatches behind each other.
inline uint32_t cappedBlocks( const tmc2_ctx* ctx, size_t wanted ) {
  return uint32_t( std::max<size_t>( 1, std::min<size_t>( wanted, size_t( 8 ) * size_t( ctx->cuCount ) ) ) );
}
// The value of a context's option (key without the TMC2_ prefix), nullptr if unset.  ctx == nullptr (the host-only entry points
// have no context): the process environment.
const char* ctxOption( const tmc2_ctx* ctx, const char* key );
int  kdtreePlacement( const tmc2_ctx* ctx );  // 0 device, 1 host, 2 adaptive (host while a host slot is free, else device)
// union passes (S3 contraction, S7 components): answer "same set already?" from the CU's possibly stale view before any
// find / compare-and-swap (TMC2_UF_PRECHECK=0 switches it off); TMC2_UF_CHECK=1: debug invariants after every union pass
int  unionPrecheck( const tmc2_ctx* ctx );
bool unionCheck( const tmc2_ctx* ctx );
bool unionAgentScope( const tmc2_ctx* ctx );  // TMC2_UF_SCOPE=agent: every load of the union passes at agent scope (the formally clean form)
int buildKdTreeDevice( tmc2_ctx* ctx, const Pt* d_pts, uint64_t n, DevBuf<Pt>& d_ptsTree, DevBuf<uint32_t>& d_perm,
                       DevBuf<KdNode>& d_nodes, int32_t lo[3], int32_t hi[3], int& depth );
// opt-in to more than 48 KB of dynamic LDS for a kernel (once per device and kernel, serialised)
int allowLargeLds( const void* kernel, size_t bytes, int device, size_t staticBytes = 64 );
// answer (optional): the total ALSO goes to host[0] of a page-locked answer line of the context (tmc2_ctx::answerLine), stored by the
// scan's last tile, with carryWords <= 7 device words behind it (host[1 ..]: written by earlier launches of the stream -- the flags
// and counters the host wants to read in the same round trip).  The host reads the line after the hipStreamSynchronize it needs
// anyway: no hipMemcpyAsync, no staging copy, no blit kernel on the frame's chain.
struct ScanAnswer {
  volatile uint32_t* host       = nullptr;
  const uint32_t*    carry      = nullptr;
  int                carryWords = 0;
};
int exclusiveScanU32( tmc2_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, uint32_t* d_total, ScanAnswer answer = ScanAnswer() );
// several device regions set to a byte value each in ONE launch (instead of one hipMemsetAsync per buffer)
struct FillRegion {
  void*   p;
  size_t  bytes;
  uint8_t value;
};
int fillRegions( tmc2_ctx* ctx, std::initializer_list<FillRegion> regions );
}  // namespace tmc2
