// integration/tmc2hip_adaptor.cpp -- see tmc2hip_adaptor.h.  Flatten, call, write back; no algorithm lives here.
#include "tmc2hip_adaptor.h"

#include <algorithm>

namespace tmc2hip {
using namespace pcc;

void flatten( const PCCPointSet3& cloud, std::vector<int16_t>& xyz, std::vector<uint8_t>& rgb ) {
  const size_t n = cloud.getPointCount();
  xyz.resize( 3 * n );
  rgb.resize( cloud.hasColors() ? 3 * n : 0 );
  for ( size_t i = 0; i < n; ++i )
    for ( int c = 0; c < 3; ++c ) {
      xyz[3 * i + c] = cloud[i][c];
      if ( cloud.hasColors() ) rgb[3 * i + c] = cloud.getColor( i )[c];
    }
}

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

void toPCCPatches( const tmc2_patch* records, int count, const int16_t* depth0, const int16_t* depth1, const uint8_t* occupancy,
                   size_t occupancyResolution, size_t frameIndex, std::vector<PCCPatch>& patches ) {
  const size_t base = patches.size();  // compute() appends
  patches.resize( base + size_t( count ) );
  for ( int i = 0; i < count; ++i ) {
    const tmc2_patch& r = records[i];
    PCCPatch&         q = patches[base + size_t( i )];
    q.setIndex( size_t( r.index ) );
    (void)frameIndex;  // compute() receives it but leaves PCCPatch::frameIndex_ alone (checked against the reference)
    q.setViewId( size_t( r.viewId ) );  // normal / tangent / bitangent axes and the projection mode follow from the view
    q.setU1( size_t( r.u1 ) ), q.setV1( size_t( r.v1 ) ), q.setD1( size_t( r.d1 ) );
    q.setSizeU( size_t( r.sizeU ) ), q.setSizeV( size_t( r.sizeV ) ), q.setSizeD( size_t( r.sizeD ) );
    q.setSizeDPixel( size_t( r.sizeDPixel ) );
    q.setSizeU0( size_t( r.sizeU0 ) ), q.setSizeV0( size_t( r.sizeV0 ) );
    q.setPatchSize2DXInPixel( size_t( r.size2DXInPixel ) ), q.setPatchSize2DYInPixel( size_t( r.size2DYInPixel ) );
    q.setOccupancyResolution( occupancyResolution );
    q.setD0Count( size_t( r.d0Count ) ), q.setEOMandD1Count( size_t( r.eomAndD1Count ) ), q.setEOMCount( 0 );
    const size_t px = size_t( r.sizeU ) * size_t( r.sizeV ), bl = size_t( r.sizeU0 ) * size_t( r.sizeV0 );
    q.setDepth( 0, std::vector<int16_t>( depth0 + r.depthOffset, depth0 + r.depthOffset + px ) );
    q.setDepth( 1, std::vector<int16_t>( depth1 + r.depthOffset, depth1 + r.depthOffset + px ) );
    std::vector<bool> occ( bl );
    for ( size_t k = 0; k < bl; ++k ) occ[k] = occupancy[r.occOffset + int64_t( k )] != 0;
    q.setOccupancy( occ );
  }
}

void toRecords( const std::vector<PCCPatch>& patches, std::vector<tmc2_patch>& records ) {
  records.assign( patches.size(), tmc2_patch{} );
  int64_t depthOffset = 0, occOffset = 0;
  for ( size_t i = 0; i < patches.size(); ++i ) {
    const PCCPatch& q = patches[i];
    tmc2_patch&     r = records[i];
    r.index           = int32_t( q.getIndex() );
    r.viewId          = int32_t( q.getViewId() );
    r.normalAxis = int32_t( q.getNormalAxis() ), r.tangentAxis = int32_t( q.getTangentAxis() );
    r.bitangentAxis = int32_t( q.getBitangentAxis() ), r.projectionMode = int32_t( q.getProjectionMode() );
    r.u1 = int32_t( q.getU1() ), r.v1 = int32_t( q.getV1() ), r.d1 = int32_t( q.getD1() );
    r.sizeU = int32_t( q.getSizeU() ), r.sizeV = int32_t( q.getSizeV() ), r.sizeD = int32_t( q.getSizeD() );
    r.sizeDPixel = int32_t( q.getSizeDPixel() );
    r.sizeU0 = int32_t( q.getSizeU0() ), r.sizeV0 = int32_t( q.getSizeV0() );
    r.size2DXInPixel = int32_t( q.getPatchSize2DXInPixel() ), r.size2DYInPixel = int32_t( q.getPatchSize2DYInPixel() );
    r.d0Count = int32_t( q.getD0Count() ), r.eomAndD1Count = int32_t( q.getEOMandD1Count() );
    r.u0 = int32_t( q.getU0() ), r.v0 = int32_t( q.getV0() ), r.patchOrientation = int32_t( q.getPatchOrientation() );
    r.depthOffset = depthOffset, r.occOffset = occOffset;
    depthOffset += int64_t( q.getSizeU() * q.getSizeV() );
    occOffset += int64_t( q.getSizeU0() * q.getSizeV0() );
  }
}

void applyPacking( const tmc2_patch* recordsByIndex, const int32_t* order, const int32_t* matches, int count,
                   std::vector<PCCPatch>& patches ) {
  std::vector<PCCPatch> byIndex;
  byIndex.swap( patches );
  patches.reserve( size_t( count ) );
  for ( int k = 0; k < count; ++k ) {
    const tmc2_patch& r = recordsByIndex[order[k]];
    patches.push_back( byIndex[size_t( order[k] )] );
    PCCPatch& q = patches.back();
    q.setU0( size_t( r.u0 ) ), q.setV0( size_t( r.v0 ) ), q.setPatchOrientation( size_t( r.patchOrientation ) );
    q.setBestMatchIdx( matches ? matches[k] : -1 );
  }
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
