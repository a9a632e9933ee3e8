// integration/tmc2hip_adaptor.cpp -- see tmc2hip_adaptor.h: the adaptor bodies that call into libtmc2hip.so.  Flatten, call,
// write back; no algorithm lives here.
#include "tmc2hip_adaptor.h"

#include <algorithm>

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

// ---- the seams of PCCEncoder::encode ---------------------------------------------------------------------------------------
bool toParams( const PCCEncoderParameters& e, tmc2_segmenter_params& p ) {
  p                                     = tmc2_segmenter_params{};
  p.nnNormalEstimation                  = int( e.nnNormalEstimation_ );
  p.normalOrientation                   = int( e.normalOrientation_ );
  p.gridBasedRefineSegmentation         = e.gridBasedRefineSegmentation_;
  p.maxNNCountRefineSegmentation        = int( e.maxNNCountRefineSegmentation_ );
  p.iterationCountRefineSegmentation    = int( e.iterationCountRefineSegmentation_ );
  p.voxelDimensionRefineSegmentation    = int( e.voxelDimensionRefineSegmentation_ );
  p.searchRadiusRefineSegmentation      = int( e.searchRadiusRefineSegmentation_ );
  p.occupancyResolution                 = int( e.occupancyResolution_ );
  p.enablePatchSplitting                = e.enablePatchSplitting_;
  p.maxPatchSize                        = int( e.maxPatchSize_ );
  p.quantizerSizeX                      = 1 << e.log2QuantizerSizeX_;
  p.quantizerSizeY                      = 1 << e.log2QuantizerSizeY_;
  p.minPointCountPerCCPatchSegmentation = int( e.minPointCountPerCCPatchSegmentation_ );
  p.maxNNCountPatchSegmentation         = int( e.maxNNCountPatchSegmentation_ );
  p.surfaceThickness                    = int( e.surfaceThickness_ );
  p.mapCountMinus1                      = int( e.mapCountMinus1_ );
  p.minLevel                            = int( e.minLevel_ );
  p.maxAllowedDepth                     = ( 1 << e.geometryNominal2dBitdepth_ ) - 1;
  p.geometryBitDepth2D                  = int( e.geometryNominal2dBitdepth_ );
  p.geometryBitDepth3D                  = int( e.geometry3dCoordinatesBitdepth_ ) + 1;
  p.maxAllowedDist2RawPointsDetection   = e.maxAllowedDist2RawPointsDetection_;
  p.maxAllowedDist2RawPointsSelection   = e.maxAllowedDist2RawPointsSelection_;
  p.lambdaRefineSegmentation            = e.lambdaRefineSegmentation_;
  p.weightNormal[0] = p.weightNormal[1] = p.weightNormal[2] = 1.0;
  // everything of the reference the path does not mirror keeps the reference's own bodies
  if ( e.gridBasedSegmentation_ || e.enhancedOccupancyMapCode_ || e.pointLocalReconstruction_ || e.singleMapPixelInterleaving_ ||
       e.additionalProjectionPlaneMode_ != 0 || !e.absoluteD1_ || e.patchExpansion_ || e.surfaceSeparation_ ||
       e.highGradientSeparation_ || e.enablePointCloudPartitioning_ || e.rawPointsPatch_ || e.lossyRawPointsPatch_ ||
       e.multipleStreams_ || e.useEightOrientations_ || e.packingStrategy_ != 1 || e.safeGuardDistance_ != 0 ||
       e.numMaxTilePerFrame_ != 1 || e.tileSegmentationType_ != 0 || e.levelOfDetailX_ > 1 || e.levelOfDetailY_ > 1 ||
       e.occupancyMapRefinement_ || e.geometryPadding_ != 0 || e.attributeBGFill_ != 1 || !e.groupDilation_ ||
       e.mapCountMinus1_ != 1 || e.globalPatchAllocation_ > 1 || ( e.globalPatchAllocation_ == 1 && !e.constrainedPack_ ) )
    return false;
  return tmc2_segmenter_params_check( &p ) == TMC2_OK;
}

EncoderDropIn::EncoderDropIn( int device ) {
  if ( tmc2_ctx_create( device, &ctx_ ) != TMC2_OK ) ctx_ = nullptr;
}
EncoderDropIn::~EncoderDropIn() {
  release();
  if ( ctx_ ) tmc2_ctx_destroy( ctx_ );
}
void EncoderDropIn::release() {
  for ( tmc2_frame* f : frames_ )
    if ( f ) tmc2_frame_destroy( f );
  frames_.clear();
}
bool EncoderDropIn::accepts( const PCCEncoderParameters& params ) {
  tmc2_segmenter_params p;
  return ctx_ != nullptr && toParams( params, p );
}

#define TMC2HIP_TRY( call )             \
  do {                                  \
    const int status_ = ( call );       \
    if ( status_ != TMC2_OK ) return status_; \
  } while ( 0 )

int EncoderDropIn::generateSegments( const PCCGroupOfFrames& sources, PCCContext& context, const PCCEncoderParameters& params ) {
  tmc2_segmenter_params p;
  if ( !ctx_ || !toParams( params, p ) ) return TMC2_E_UNSUPPORTED;
  release();
  auto& frames = context.getFrames();
  frames_.assign( frames.size(), nullptr );
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  for ( size_t i = 0; i < frames.size(); ++i ) {
    if ( sources[i].getPointCount() == 0u ) continue;  // (generateSegments returns true for an empty frame and leaves it alone)
    flatten( sources[i], xyz, rgb );
    TMC2HIP_TRY( tmc2_frame_create( ctx_, xyz.data(), rgb.empty() ? nullptr : rgb.data(), sources[i].getPointCount(), &frames_[i] ) );
    // calculateWeightNormal( geometryBitDepth3D, sources[0] ) :3569-3626: the axis weights of the whole GOF come from frame 0
    if ( i == 0 ) TMC2HIP_TRY( tmc2_weight_normal( frames_[0], p.geometryBitDepth3D, params.minWeightEPP_, p.weightNormal ) );
    TMC2HIP_TRY( tmc2_segmenter_compute( frames_[i], &p ) );
    int64_t depthCount = 0, occCount = 0;
    TMC2HIP_TRY( tmc2_frame_patch_pool_sizes( frames_[i], &depthCount, &occCount ) );
    const int               count = tmc2_frame_patch_count( frames_[i] );
    std::vector<tmc2_patch> rec( static_cast<size_t>( count ) );
    std::vector<int16_t>    d0( static_cast<size_t>( depthCount ) ), d1( static_cast<size_t>( depthCount ) );
    std::vector<uint8_t>    occ( static_cast<size_t>( occCount ) );
    TMC2HIP_TRY( tmc2_frame_get_patches( frames_[i], rec.data(), d0.data(), d1.data(), occ.data() ) );
    auto& tile = frames[i].getTitleFrameContext();
    tile.getPatches().reserve( 256 );
    toPCCPatches( rec.data(), count, d0.data(), d1.data(), occ.data(), params.occupancyResolution_, tile.getFrameIndex(),
                  tile.getPatches() );
  }
  return TMC2_OK;
}

int EncoderDropIn::placeSegments( PCCContext& context, const PCCEncoderParameters& params ) {
  auto&     frames = context.getFrames();
  const int count = int( frames.size() ), minW = int( params.minimumImageWidth_ ), minH = int( params.minimumImageHeight_ );
  const int tilesHor = int( params.numTilesHor_ );
  const double ratio = params.tileHeightToWidthRatio_;
  const bool   chain = params.constrainedPack_, gpa = chain && params.globalPatchAllocation_ == 1;
  std::vector<int32_t> widths( size_t( count ), minW ), heights( size_t( count ), 0 );
  std::vector<size_t>  matched( size_t( count ), 0 );  // what the chained packer leaves in setNumMatchedPatches (before any re-allocation)
  for ( int i = 0; i < count; ++i ) {
    if ( !frames_[size_t( i )] ) return TMC2_E_STATE;  // (an empty frame: the reference's own body handles the GOF)
    if ( i == 0 || !chain ) {
      TMC2HIP_TRY( tmc2_encoder_pack_flexible( frames_[size_t( i )], minW, tilesHor, ratio, &heights[size_t( i )] ) );
    } else {
      TMC2HIP_TRY( tmc2_encoder_pack_spatial_consistency( frames_[size_t( i )], frames_[size_t( i - 1 )], minW, tilesHor, ratio,
                                                          &heights[size_t( i )] ) );
      std::vector<int32_t> m( size_t( tmc2_frame_patch_count( frames_[size_t( i )] ) ) );
      TMC2HIP_TRY( tmc2_frame_get_patch_matches( frames_[size_t( i )], m.data() ) );
      for ( int32_t v : m ) matched[size_t( i )] += v >= 0 ? 1 : 0;
    }
  }
  if ( gpa ) TMC2HIP_TRY( tmc2_encoder_global_patch_allocation( frames_.data(), count, minW, minH, widths.data(), heights.data() ) );
  int32_t tileW = minW, tileH = 0;
  for ( int i = 0; i < count; ++i ) {
    tmc2_frame* f  = frames_[size_t( i )];
    const int   np = tmc2_frame_patch_count( f );
    int64_t     depthCount = 0, occCount = 0;
    TMC2HIP_TRY( tmc2_frame_patch_pool_sizes( f, &depthCount, &occCount ) );
    std::vector<tmc2_patch> rec( static_cast<size_t>( np ) );
    std::vector<int32_t>    order( static_cast<size_t>( np ) ), matches( static_cast<size_t>( np ) );
    std::vector<uint8_t>    occ( static_cast<size_t>( occCount ) );
    TMC2HIP_TRY( tmc2_frame_get_patches( f, rec.data(), nullptr, nullptr, occ.data() ) );
    TMC2HIP_TRY( tmc2_frame_get_patch_order( f, order.data() ) );
    TMC2HIP_TRY( tmc2_frame_get_patch_matches( f, matches.data() ) );
    auto& tile = frames[size_t( i )].getTitleFrameContext();
    if ( gpa )  // the allocation rewrote the lists: records come back in list order, with their block boxes and occupancy
      applyPackedList( rec.data(), matches.data(), occ.data(), np, tile.getPatches() );
    else
      applyPacking( rec.data(), order.data(), matches.data(), np, tile.getPatches() );
    // (PCCPatch::patchType_ is left alone: the packers write it -- with a side effect on the PREVIOUS frame's list,
    //  PCCEncoder.cpp:1254 -- but nothing in the reference reads it)
    tile.setNumMatchedPatches( matched[size_t( i )] );
    if ( !gpa ) TMC2HIP_TRY( tmc2_frame_get_packed_size( f, &widths[size_t( i )], &heights[size_t( i )] ) );
    tileW = std::max( tileW, widths[size_t( i )] );
    tileH = std::max( tileH, heights[size_t( i )] );
  }
  // resizeTileGeometryVideo + resizeGeometryVideo: one canvas for the GOF
  if ( gpa ) tileH = std::max( tileH, int32_t( minH ) );
  int32_t W = 0, H = 0;
  TMC2HIP_TRY( tmc2_encoder_canvas_size( &tileH, 1, tileW, minW, minH, &W, &H ) );
  width_ = W, height_ = H;
  for ( auto& frame : frames ) {
    frame.setAtlasFrameWidth( size_t( W ) );  // (also the title frame context's = the single tile's size)
    frame.setAtlasFrameHeight( size_t( H ) );
    if ( frame.getNumTilesInAtlasFrame() == 1 && frame.getNumPartitionWidth() > 0 && frame.getNumPartitionHeight() > 0 ) {
      frame.setPartitionWidth( size_t( W ), 0 );
      frame.setPartitionHeight( size_t( H ), 0 );
    }
  }
  return TMC2_OK;
}

int EncoderDropIn::generateGeometryVideo( PCCContext& context, const PCCEncoderParameters& params ) {
  auto&        frames = context.getFrames();
  const size_t W = size_t( width_ ), H = size_t( height_ ), prec = params.occupancyPrecision_;
  auto&        videoOcc = context.getVideoOccupancyMap();
  auto&        videoGeo = context.getVideoGeometryMultiple()[0];
  videoOcc.resize( frames.size() );
  videoGeo.resize( 2 * frames.size() );
  std::vector<uint8_t>  occ( W * H ), occVideo( ( W / prec ) * ( H / prec ) );
  std::vector<uint32_t> blockToPatch( ( W / 16 ) * ( H / 16 ) );
  std::vector<uint16_t> d0( W * H ), d1( W * H );
  for ( size_t i = 0; i < frames.size(); ++i ) {
    TMC2HIP_TRY( tmc2_encoder_generate_geometry_images( frames_[i], int( W ), int( H ), int( prec ) ) );
    TMC2HIP_TRY( tmc2_frame_get_geometry_images( frames_[i], occ.data(), occVideo.data(), blockToPatch.data(), d0.data(), d1.data() ) );
    auto& tile = frames[i].getTitleFrameContext();
    toFrameImages( occ.data(), occVideo.data(), blockToPatch.data(), d0.data(), d1.data(), W, H, prec, tile.getOccupancyMap(),
                   tile.getBlockToPatch(), videoOcc.getFrame( i ), videoGeo.getFrame( 2 * i ), videoGeo.getFrame( 2 * i + 1 ) );
  }
  return TMC2_OK;
}

int EncoderDropIn::generateAttributeVideo( PCCContext& context, PCCGroupOfFrames& reconstructs, const PCCEncoderParameters& ) {
  auto&        frames = context.getFrames();
  const size_t W = size_t( width_ ), H = size_t( height_ );
  auto&        videoAttribute = context.getVideoAttributesMultiple()[0];
  videoAttribute.resize( 2 * frames.size() );
  reconstructs.setFrameCount( frames.size() );
  std::vector<uint8_t> attribute( 6 * W * H );
  for ( size_t i = 0; i < frames.size(); ++i ) {
    TMC2HIP_TRY( tmc2_encoder_generate_attribute_images( frames_[i] ) );
    const size_t          M = size_t( tmc2_frame_recon_count( frames_[i] ) );
    std::vector<int16_t>  xyz( 3 * M );
    std::vector<uint8_t>  rgb( 3 * M );
    std::vector<uint32_t> p2p( 3 * M );
    TMC2HIP_TRY( tmc2_frame_get_reconstruction( frames_[i], xyz.data(), rgb.data(), p2p.data() ) );
    TMC2HIP_TRY( tmc2_frame_get_attribute_images( frames_[i], attribute.data() ) );
    toReconstruction( xyz.data(), rgb.data(), p2p.data(), M, reconstructs[i], frames[i].getTitleFrameContext().getPointToPixel() );
    toAttributeFrames( attribute.data(), W, H, videoAttribute.getFrame( 2 * i ), videoAttribute.getFrame( 2 * i + 1 ) );
  }
  return TMC2_OK;
}
#undef TMC2HIP_TRY

}  // namespace tmc2hip
