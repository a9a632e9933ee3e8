// segmenter_api.cpp -- PCCPatchSegmenter3::compute chain, parameter validation and patch accessors.
#include <algorithm>

#include <algorithm>
#include <chrono>
#include <thread>

#include "internal.h"
using namespace tmc2;

template <typename T>
static int growKeep( DevBuf<T>& b, size_t need, hipStream_t s ) {
  if ( need <= b.count && b.p ) return TMC2_OK;
  size_t cap = std::max<size_t>( need + need / 2, 1 << 16 );
  T*     np  = nullptr;
  TMC2_HIP( hipMalloc( reinterpret_cast<void**>( &np ), cap * sizeof( T ) ) );
  if ( b.p && b.count ) {
    TMC2_HIP( hipMemcpyAsync( np, b.p, b.count * sizeof( T ), hipMemcpyDeviceToDevice, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    (void)hipFree( b.p );
  }
  b.p     = np;
  b.count = cap;
  return TMC2_OK;
}

int tmc2_frame::growPools() {
  TMC2_TRY( growKeep( d_depth0, size_t( depthCount ), ctx->stream ) );
  TMC2_TRY( growKeep( d_depth1, size_t( depthCount ), ctx->stream ) );
  TMC2_TRY( growKeep( d_occupancy, size_t( occCount ), ctx->stream ) );
  return TMC2_OK;
}

extern "C" {

int tmc2_segmenter_params_check( const tmc2_segmenter_params* p ) {
  if ( !p ) return TMC2_E_INVALID;
  if ( p->nnNormalEstimation != 16 || p->maxNNCountPatchSegmentation != 16 ) {
    setError( "params: nnNormalEstimation / maxNNCountPatchSegmentation must be 16 (one shared k-NN self-join)" );
    return TMC2_E_UNSUPPORTED;
  }
  if ( p->normalOrientation != 1 && p->normalOrientation != 0 ) {
    setError( "params: normalOrientation %d unsupported (0 none, 1 spanning tree)", p->normalOrientation );
    return TMC2_E_UNSUPPORTED;
  }
  if ( !p->gridBasedRefineSegmentation ) {
    setError( "params: only gridBasedRefineSegmentation=1 is implemented" );
    return TMC2_E_UNSUPPORTED;
  }
  if ( p->occupancyResolution != 16 ) {
    setError( "params: occupancyResolution must be 16" );
    return TMC2_E_UNSUPPORTED;
  }
  if ( p->mapCountMinus1 != 1 ) {
    setError( "params: mapCountMinus1 must be 1 (two maps, absoluteD1)" );
    return TMC2_E_UNSUPPORTED;
  }
  return TMC2_OK;
}

int tmc2_segmenter_segment_patches( tmc2_frame* f, const tmc2_segmenter_params* p ) {
  if ( !f || !p ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( tmc2_segmenter_params_check( p ) );
  return segmentPatches( f, p );
}

int tmc2_segmenter_compute( tmc2_frame* f, const tmc2_segmenter_params* p ) {
  if ( !f || !p ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( tmc2_segmenter_params_check( p ) );
  // tmc2_set_refine_overlap( 1 ) / TMC2_REFINE_OVERLAP=1: the refine step's geometry (voxels, neighbourhood rows forward and reverse:
  // points only) is queued right before the orientation's host walk and built while the host walks.  It shortens a frame's
  // chain and costs throughput when the chip is full: round 4, four frames in flight (one rank's share of an 8-GPU run):
  // longdress 30.8 -> 30.4 ms, loot (voxels of 2: 5 ms of geometry) 57.8 -> 52.0 ms; sixteen in flight: 173.5 -> 174.0 and
  // 91.5 -> 90.1 frames/s.  The GOF host (tmc2_amd/gof.py, integration/tmc2_encode_gof.cpp) turns it on for <= 4 frames in flight.
  struct HookGuard {  // on every way out: no hook left behind, and no half-used refine job (it holds the context's dense voxel table
    tmc2_frame* f;    // filled: another frame's refinement on this context would look its cells up in a dirty table)
    bool        done = false;
    ~HookGuard() {
      f->beforeHostWalk = nullptr;
      if ( !done ) f->refineJob.reset();
    }
  } guard{f};
  if ( p->gridBasedRefineSegmentation && tmc2::refineOverlap( f->ctx ) )
    f->beforeHostWalk = [f, p]() {
      return tmc2::refinePrepareGeometry( f, p->maxNNCountRefineSegmentation, p->lambdaRefineSegmentation,
                                          p->iterationCountRefineSegmentation, p->voxelDimensionRefineSegmentation,
                                          p->searchRadiusRefineSegmentation );
    };
  // Option FRAME_START_DELAY_US (few frames in flight: a rank of the 8-GPU run has four).  Frames that start together reach their
  // host-resident step -- S3's walk, ~ 3 ms -- together, and the GPU has nothing to do meanwhile (profiles/r06_rank_concurrency.txt:
  // a hole of ~ 3 ms in every 27 ms step).  A host that delays the start of half of its frames by about that long has one half
  // walking while the other half's kernels run.
  if ( const char* delay = tmc2::ctxOption( f->ctx, "FRAME_START_DELAY_US" ) ) {
    const int us = atoi( delay );
    if ( us > 0 ) std::this_thread::sleep_for( std::chrono::microseconds( std::min( us, 100000 ) ) );
  }
  TMC2_TRY( tmc2_normals_compute( f, p->nnNormalEstimation, p->normalOrientation ) );
  f->beforeHostWalk = nullptr;
  TMC2_TRY( tmc2_segmenter_initial_segmentation( f, p->weightNormal ) );
  TMC2_TRY( tmc2_segmenter_refine_grid_based( f, p->maxNNCountRefineSegmentation, p->lambdaRefineSegmentation,
                                              p->iterationCountRefineSegmentation, p->voxelDimensionRefineSegmentation,
                                              p->searchRadiusRefineSegmentation ) );
  guard.done = true;  // (the refinement consumed its job)
  return segmentPatches( f, p );
}

int tmc2_frame_patch_count( tmc2_frame* f ) { return f ? int( f->patches.size() ) : 0; }

int tmc2_frame_patch_pool_sizes( tmc2_frame* f, int64_t* d, int64_t* o ) {
  if ( !f || !d || !o ) return TMC2_E_INVALID;
  *d = f->depthCount;
  *o = f->occCount;
  return TMC2_OK;
}

int tmc2_frame_get_patches( tmc2_frame* f, tmc2_patch* patches, int16_t* depth0, int16_t* depth1, uint8_t* occ ) {
  if ( !f || !f->havePatches ) {
    setError( "get_patches: no patches" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  hipStream_t s = f->ctx->stream;
  if ( patches && !f->patches.empty() ) memcpy( patches, f->patches.data(), f->patches.size() * sizeof( tmc2_patch ) );
  if ( depth0 && f->depthCount )
    TMC2_HIP( hipMemcpyAsync( depth0, f->d_depth0.p, size_t( f->depthCount ) * 2, hipMemcpyDeviceToHost, s ) );
  if ( depth1 && f->depthCount )
    TMC2_HIP( hipMemcpyAsync( depth1, f->d_depth1.p, size_t( f->depthCount ) * 2, hipMemcpyDeviceToHost, s ) );
  if ( occ && f->occCount ) TMC2_HIP( hipMemcpyAsync( occ, f->d_occupancy.p, size_t( f->occCount ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}
}
