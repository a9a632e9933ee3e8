// api.cpp -- extern "C" surface of libtmc2hip.so (see include/tmc2hip.h for the reference seams).
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <memory>

#include "internal.h"

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

void orientNormalsSpanningTree( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals );

}  // namespace tmc2

using namespace tmc2;

int tmc2_ctx::stageBegin( const char* name ) {
  int id = -1;
  for ( size_t i = 0; i < stages.size(); ++i )
    if ( stages[i].name == name ) id = int( i );
  if ( id < 0 ) {
    StageTimer t;
    t.name = name;
    (void)hipEventCreate( &t.e0 );
    (void)hipEventCreate( &t.e1 );
    stages.push_back( t );
    id = int( stages.size() ) - 1;
  } else if ( stages[id].calls > 0 && stages[id].e0 ) {
    // fold the previous interval of this stage into the running total before reusing the events
    float ms = 0.f;
    if ( hipEventSynchronize( stages[id].e1 ) == hipSuccess &&
         hipEventElapsedTime( &ms, stages[id].e0, stages[id].e1 ) == hipSuccess )
      stages[id].ms += ms;
    stages[id].calls = 0;
  }
  if ( !stages[id].e0 ) {
    (void)hipEventCreate( &stages[id].e0 );
    (void)hipEventCreate( &stages[id].e1 );
  }
  (void)hipEventRecord( stages[id].e0, stream );
  return id;
}
void tmc2_ctx::stageEnd( int id ) {
  if ( id < 0 ) return;
  (void)hipEventRecord( stages[id].e1, stream );
  stages[id].calls = 1;
}
void tmc2_ctx::stageAddHostMs( const char* name, double ms ) {
  for ( auto& s : stages )
    if ( s.name == name ) {
      s.ms += ms;
      return;
    }
  StageTimer t;
  t.name = name;
  t.ms   = ms;
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
  if ( hipStreamCreateWithFlags( &c->stream, hipStreamNonBlocking ) != hipSuccess ) {
    setError( "hipStreamCreate failed" );
    delete c;
    return TMC2_E_HIP;
  }
  *out = c;
  return TMC2_OK;
}

void tmc2_ctx_destroy( tmc2_ctx* ctx ) {
  if ( !ctx ) return;
  (void)hipSetDevice( ctx->device );
  for ( auto& s : ctx->stages ) {
    if ( s.e0 ) (void)hipEventDestroy( s.e0 );
    if ( s.e1 ) (void)hipEventDestroy( s.e1 );
  }
  if ( ctx->stream ) (void)hipStreamDestroy( ctx->stream );
  delete ctx;
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
  if ( s.calls > 0 && s.e0 ) {
    float ms = 0.f;
    if ( hipEventSynchronize( s.e1 ) == hipSuccess && hipEventElapsedTime( &ms, s.e0, s.e1 ) == hipSuccess ) s.ms += ms;
    s.calls = 0;
  }
  return s.ms;
}
void tmc2_ctx_stage_reset( tmc2_ctx* ctx ) {
  if ( !ctx ) return;
  for ( auto& s : ctx->stages ) {
    s.ms    = 0.0;
    s.calls = 0;
  }
}

int tmc2_frame_create( tmc2_ctx* ctx, const int16_t* xyz, const uint8_t* rgb, uint64_t n, tmc2_frame** out ) {
  if ( !ctx || !xyz || !out || n == 0 || n > 0x7FFFFFF0ull ) {
    setError( "frame_create: invalid argument" );
    return TMC2_E_INVALID;
  }
  *out = nullptr;
  TMC2_HIP( hipSetDevice( ctx->device ) );
  std::unique_ptr<tmc2_frame> f( new tmc2_frame() );
  f->ctx = ctx;
  f->n   = n;
  f->h_xyz.assign( xyz, xyz + 3 * n );
  for ( uint64_t i = 0; i < 3 * n; ++i ) f->geoMax = std::max( f->geoMax, xyz[i] );
  if ( rgb ) f->h_rgb.assign( rgb, rgb + 3 * n );
  const auto t0 = std::chrono::steady_clock::now();
  f->tree.build( xyz, n );
  const auto t1 = std::chrono::steady_clock::now();
  ctx->stageAddHostMs( "kdtree_build_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
  // stage AoS-with-padding copies of the points in original and in tree order
  std::vector<Pt> pts( n ), ptsTree( n );
  for ( uint64_t i = 0; i < n; ++i ) pts[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0};
  for ( uint64_t i = 0; i < n; ++i ) ptsTree[i] = pts[f->tree.perm[i]];
  TMC2_TRY( f->d_pts.alloc( n ) );
  TMC2_TRY( f->d_ptsTree.alloc( n ) );
  TMC2_TRY( f->d_perm.alloc( n ) );
  TMC2_TRY( f->d_nodes.alloc( f->tree.nodes.size() ) );
  hipStream_t s = ctx->stream;
  TMC2_HIP( hipMemcpyAsync( f->d_pts.p, pts.data(), n * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( f->d_ptsTree.p, ptsTree.data(), n * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( f->d_perm.p, f->tree.perm.data(), n * sizeof( uint32_t ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( f->d_nodes.p, f->tree.nodes.data(), f->tree.nodes.size() * sizeof( KdNode ),
                            hipMemcpyHostToDevice, s ) );
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

void tmc2_frame_destroy( tmc2_frame* f ) {
  if ( !f ) return;
  (void)hipSetDevice( f->ctx->device );
  delete f;
}

uint64_t tmc2_frame_point_count( const tmc2_frame* f ) { return f ? f->n : 0; }

int tmc2_kdtree_search( tmc2_frame* f, const int16_t* queries, uint64_t nq, int k, uint32_t* idx, uint32_t* dist2 ) {
  if ( !f || !queries || !idx || nq == 0 ) {
    setError( "kdtree_search: invalid argument" );
    return TMC2_E_INVALID;
  }
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  std::vector<Pt> q( nq );
  for ( uint64_t i = 0; i < nq; ++i ) q[i] = Pt{queries[3 * i], queries[3 * i + 1], queries[3 * i + 2], 0};
  DevBuf<Pt>       d_q;
  DevBuf<uint32_t> d_idx, d_dist;
  TMC2_TRY( d_q.alloc( nq ) );
  TMC2_TRY( d_idx.alloc( nq * size_t( k ) ) );
  if ( dist2 ) TMC2_TRY( d_dist.alloc( nq * size_t( k ) ) );
  hipStream_t s = f->ctx->stream;
  TMC2_HIP( hipMemcpyAsync( d_q.p, q.data(), nq * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
  TMC2_TRY( launchKnnQueries( f, d_q.p, nq, k, d_idx.p, dist2 ? d_dist.p : nullptr ) );
  TMC2_HIP( hipMemcpyAsync( idx, d_idx.p, nq * size_t( k ) * 4, hipMemcpyDeviceToHost, s ) );
  if ( dist2 ) TMC2_HIP( hipMemcpyAsync( dist2, d_dist.p, nq * size_t( k ) * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

int tmc2_normals_compute_normals( tmc2_frame* f, int k ) {
  if ( !f ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  if ( !f->haveKnn || f->k != k ) TMC2_TRY( launchKnnSelf( f, k ) );
  return launchNormals( f );
}

int tmc2_normals_orient( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
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
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  TMC2_HIP( hipMemcpyAsync( normals, f->d_normals.p, f->n * 3 * sizeof( double ), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}

int tmc2_frame_set_normals( tmc2_frame* f, const double* normals ) {
  if ( !f || !normals ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
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
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  TMC2_HIP( hipMemcpyAsync( adj, f->d_knn.p, f->n * size_t( f->k ) * 4, hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}

int tmc2_weight_normal( tmc2_frame* f, int geometryBitDepth3D, double minWeightEPP, double weight[3] ) {
  if ( !f || !weight ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  return weightNormal( f, geometryBitDepth3D, minWeightEPP, weight );
}

int tmc2_segmenter_initial_segmentation( tmc2_frame* f, const double weight[3] ) {
  if ( !f || !weight ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  return launchInitialSegmentation( f, weight );
}

int tmc2_segmenter_refine_grid_based( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim,
                                      int searchRadius ) {
  if ( !f ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  return refineGridBased( f, maxNNCount, lambda, iterationCount, voxDim, searchRadius );
}

int tmc2_frame_get_partition( tmc2_frame* f, uint32_t* partition ) {
  if ( !f || !partition || !f->havePartition ) {
    setError( "get_partition: no partition" );
    return TMC2_E_STATE;
  }
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  std::vector<uint8_t> tmp( f->n );
  TMC2_HIP( hipMemcpyAsync( tmp.data(), f->d_partition.p, f->n, hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  for ( uint64_t i = 0; i < f->n; ++i ) partition[i] = tmp[i];
  return TMC2_OK;
}

int tmc2_frame_set_partition( tmc2_frame* f, const uint32_t* partition ) {
  if ( !f || !partition ) return TMC2_E_INVALID;
  TMC2_HIP( hipSetDevice( f->ctx->device ) );
  std::vector<uint8_t> tmp( f->n );
  for ( uint64_t i = 0; i < f->n; ++i ) tmp[i] = uint8_t( partition[i] );
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
  orientNormalsSpanningTree( xyz, n, knn, k, normals );
  return TMC2_OK;
}

}  // extern "C"
