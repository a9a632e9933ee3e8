// integration/tmc2hip_adaptor.cpp -- see tmc2hip_adaptor.h: the adaptor bodies that call into libtmc2hip.so.  Flatten, call,
// write back; no algorithm lives here.
#include "tmc2hip_adaptor.h"

namespace tmc2hip {
using namespace pcc;

bool toParams( const PCCPatchSegmenter3Parameters& params, tmc2_segmenter_params& p ) {
  p                                      = tmc2_segmenter_params{};
  p.nnNormalEstimation                   = int( params.nnNormalEstimation_ );
  p.normalOrientation                    = int( params.normalOrientation_ );
  p.gridBasedRefineSegmentation          = params.gridBasedRefineSegmentation_;
  p.maxNNCountRefineSegmentation         = int( params.maxNNCountRefineSegmentation_ );
  p.iterationCountRefineSegmentation     = int( params.iterationCountRefineSegmentation_ );
  p.voxelDimensionRefineSegmentation     = int( params.voxelDimensionRefineSegmentation_ );
  p.searchRadiusRefineSegmentation       = int( params.searchRadiusRefineSegmentation_ );
  p.occupancyResolution                  = int( params.occupancyResolution_ );
  p.enablePatchSplitting                 = params.enablePatchSplitting_;
  p.maxPatchSize                         = int( params.maxPatchSize_ );
  p.quantizerSizeX                       = int( params.quantizerSizeX_ );
  p.quantizerSizeY                       = int( params.quantizerSizeY_ );
  p.minPointCountPerCCPatchSegmentation  = int( params.minPointCountPerCCPatchSegmentation_ );
  p.maxNNCountPatchSegmentation          = int( params.maxNNCountPatchSegmentation_ );
  p.surfaceThickness                     = int( params.surfaceThickness_ );
  p.mapCountMinus1                       = int( params.mapCountMinus1_ );
  p.minLevel                             = int( params.minLevel_ );
  p.maxAllowedDepth                      = int( params.maxAllowedDepth_ );
  p.geometryBitDepth2D                   = int( params.geometryBitDepth2D_ );
  p.geometryBitDepth3D                   = int( params.geometryBitDepth3D_ );
  p.maxAllowedDist2RawPointsDetection    = params.maxAllowedDist2RawPointsDetection_;
  p.maxAllowedDist2RawPointsSelection    = params.maxAllowedDist2RawPointsSelection_;
  p.lambdaRefineSegmentation             = params.lambdaRefineSegmentation_;
  for ( int c = 0; c < 3; ++c ) p.weightNormal[c] = params.weightNormal_[c];
  // what the library does not mirror stays with the reference's own body
  if ( params.gridBasedSegmentation_ || params.useEnhancedOccupancyMapCode_ || params.createSubPointCloud_ ||
       params.additionalProjectionPlaneMode_ != 0 || !params.absoluteD1_ || params.patchExpansion_ || params.surfaceSeparation_ ||
       params.highGradientSeparation_ || params.enablePointCloudPartitioning_ )
    return false;
  return tmc2_segmenter_params_check( &p ) == TMC2_OK;
}

int segmenterCompute( tmc2_ctx* ctx, const PCCPointSet3& geometry, size_t frameIndex, const PCCPatchSegmenter3Parameters& params,
                      std::vector<PCCPatch>& patches, tmc2_frame** keep ) {
  tmc2_segmenter_params p;
  if ( !toParams( params, p ) ) return TMC2_E_UNSUPPORTED;  // the caller runs the reference's own body
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  flatten( geometry, xyz, rgb );
  tmc2_frame* f = nullptr;
  int         r = tmc2_frame_create( ctx, xyz.data(), rgb.empty() ? nullptr : rgb.data(), geometry.getPointCount(), &f );
  if ( r != TMC2_OK ) return r;
  r = tmc2_segmenter_compute( f, &p );  // S1..S9
  int64_t depthCount = 0, occCount = 0;
  if ( r == TMC2_OK ) r = tmc2_frame_patch_pool_sizes( f, &depthCount, &occCount );
  if ( r != TMC2_OK ) {
    tmc2_frame_destroy( f );
    return r;
  }
  const int               count = tmc2_frame_patch_count( f );
  std::vector<tmc2_patch> rec( static_cast<size_t>( count ) );
  std::vector<int16_t>    d0( static_cast<size_t>( depthCount ) ), d1( static_cast<size_t>( depthCount ) );
  std::vector<uint8_t>    occ( static_cast<size_t>( occCount ) );
  r = tmc2_frame_get_patches( f, rec.data(), d0.data(), d1.data(), occ.data() );
  if ( r != TMC2_OK ) {
    tmc2_frame_destroy( f );
    return r;
  }
  toPCCPatches( rec.data(), count, d0.data(), d1.data(), occ.data(), params.occupancyResolution_, frameIndex, patches );
  if ( keep )
    *keep = f;
  else
    tmc2_frame_destroy( f );
  return TMC2_OK;
}

}  // namespace tmc2hip
