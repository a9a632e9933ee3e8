// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE.
//
// A thin C-ABI driver around the UNMODIFIED reference classes (compiled in place from
// /root/reference by oracle/Makefile, target `ref`).  Nothing here re-implements reference
// behaviour: every function builds the reference's own containers from flat arrays, calls the
// reference's own entry point, and copies the result back out.  It is used to
//   * pin our restatement (oracle/port_*.cpp) bit-for-bit,
//   * generate the golden vectors under tests/golden/ (tests/golden/make_golden.py),
//   * serve as bench.py's cpu_baseline of kind "reference".
// The product never links this file.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include <unistd.h>

// The image-generation helpers of PCCEncoder are private members; the harness needs to call them
// one by one (SURVEY.md §8c "library-level oracle").
#define private public
#define protected public
#include "PCCCommon.h"
#include "PCCPointSet.h"
#include "PCCKdTree.h"
#include "PCCPatch.h"
#include "PCCNormalsGenerator.h"
#include "PCCPatchSegmenter.h"
#include "PCCEncoderParameters.h"
#include "PCCEncoder.h"
#include "PCCMetrics.h"
#include "PCCMetricsParameters.h"
#include "PCCContext.h"
#include "PCCFrameContext.h"
#include "PCCGroupOfFrames.h"
#include "PCCImage.h"
#include "PCCInternalColorConverter.h"
#include "PCCChecksum.h"
#include "tmc2hip_adaptor.h"  // integration/: the reference-side conversions, checked below against the reference's own containers
#include "PCCVideo.h"
#include "PCCBitstream.h"
#undef private
#undef protected
// The reference's PCCEncoder.cpp itself, compiled in this translation unit (see oracle/Makefile): gives the
// harness the member templates dilateSmoothedPushPull<T> / pushPullMip<T> / pushPullFill<T> defined there.
#include "PCCEncoder.cpp"

#include "oracle.h"

using namespace pcc;

namespace {

struct Quiet {  // the reference prints progress to std::cout; silence it while we drive it
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old( std::cout.rdbuf() ) {
    if ( !getenv( "TMC2_REF_VERBOSE" ) ) std::cout.rdbuf( sink.rdbuf() );
  }
  ~Quiet() { std::cout.rdbuf( old ); }
};

void makeCloud( PCCPointSet3& pc, const int16_t* xyz, const uint8_t* rgb, size_t n ) {
  pc.resize( n );
  if ( rgb ) pc.addColors();
  for ( size_t i = 0; i < n; ++i ) {
    pc[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    if ( rgb ) pc.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
  }
}

PCCNormalsGenerator3Parameters normalsParams( int k, int orientation ) {
  // same aggregate as PCCPatchSegmenter3::compute builds (PCCPatchSegmenter.cpp:93-105)
  const double mx = ( std::numeric_limits<double>::max )();
  PCCNormalsGenerator3Parameters p = {PCCVector3D( 0.0 ), mx, mx, mx, mx, size_t( k ), size_t( k ), size_t( k ), 0,
                                      static_cast<PCCNormalsGeneratorOrientation>( orientation ), false, false, false};
  return p;
}

void toSegParams( const orc_seg_params& s, PCCPatchSegmenter3Parameters& p ) {
  p.gridBasedSegmentation_               = false;
  p.voxelDimensionGridBasedSegmentation_ = 2;
  p.nnNormalEstimation_                  = s.nnNormalEstimation;
  p.normalOrientation_                   = s.normalOrientation;
  p.gridBasedRefineSegmentation_         = s.gridBasedRefineSegmentation != 0;
  p.maxNNCountRefineSegmentation_        = s.maxNNCountRefineSegmentation;
  p.iterationCountRefineSegmentation_    = s.iterationCountRefineSegmentation;
  p.voxelDimensionRefineSegmentation_    = s.voxelDimensionRefineSegmentation;
  p.searchRadiusRefineSegmentation_      = s.searchRadiusRefineSegmentation;
  p.occupancyResolution_                 = s.occupancyResolution;
  p.enablePatchSplitting_                = s.enablePatchSplitting != 0;
  p.maxPatchSize_                        = s.maxPatchSize;
  p.quantizerSizeX_                      = s.quantizerSizeX;
  p.quantizerSizeY_                      = s.quantizerSizeY;
  p.minPointCountPerCCPatchSegmentation_ = s.minPointCountPerCC;
  p.maxNNCountPatchSegmentation_         = s.maxNNCountPatchSegmentation;
  p.surfaceThickness_                    = s.surfaceThickness;
  p.EOMFixBitCount_                      = 2;
  p.EOMSingleLayerMode_                  = false;
  p.mapCountMinus1_                      = s.mapCountMinus1;
  p.minLevel_                            = s.minLevel;
  p.maxAllowedDepth_                     = s.maxAllowedDepth;
  p.maxAllowedDist2RawPointsDetection_   = s.maxAllowedDist2RawPointsDetection;
  p.maxAllowedDist2RawPointsSelection_   = s.maxAllowedDist2RawPointsSelection;
  p.lambdaRefineSegmentation_            = s.lambdaRefineSegmentation;
  p.useEnhancedOccupancyMapCode_         = false;
  p.absoluteD1_                          = true;
  p.createSubPointCloud_                 = false;
  p.surfaceSeparation_                   = false;
  p.weightNormal_                        = PCCVector3D( s.weightNormal[0], s.weightNormal[1], s.weightNormal[2] );
  p.additionalProjectionPlaneMode_       = 0;
  p.partialAdditionalProjectionPlane_    = 0.0;
  p.geometryBitDepth2D_                  = s.geometryBitDepth2D;
  p.geometryBitDepth3D_                  = s.geometryBitDepth3D;
  p.patchExpansion_                      = false;
  p.highGradientSeparation_              = false;
  p.minGradient_                         = 15.0;
  p.minNumHighGradientPoints_            = 256;
  p.enablePointCloudPartitioning_        = false;
  p.numTilesHor_                         = 2;
  p.tileHeightToWidthRatio_              = 1.0;
  p.numCutsAlong1stLongestAxis_          = 1;
  p.numCutsAlong2ndLongestAxis_          = 1;
  p.numCutsAlong3rdLongestAxis_          = 1;
}

// last segmentation result (patch list) kept for the accessor calls
std::vector<PCCPatch> g_patches;

void fillPatch( const PCCPatch& p, orc_patch& o, int64_t depthOff, int64_t occOff ) {
  o.index            = int32_t( p.getIndex() );
  o.viewId           = int32_t( p.getViewId() );
  o.normalAxis       = int32_t( p.getNormalAxis() );
  o.tangentAxis      = int32_t( p.getTangentAxis() );
  o.bitangentAxis    = int32_t( p.getBitangentAxis() );
  o.projectionMode   = int32_t( p.getProjectionMode() );
  o.u1               = int32_t( p.getU1() );
  o.v1               = int32_t( p.getV1() );
  o.d1               = int32_t( p.getD1() );
  o.sizeU            = int32_t( p.getSizeU() );
  o.sizeV            = int32_t( p.getSizeV() );
  o.sizeD            = int32_t( p.getSizeD() );
  o.sizeDPixel       = int32_t( p.getSizeDPixel() );
  o.sizeU0           = int32_t( p.getSizeU0() );
  o.sizeV0           = int32_t( p.getSizeV0() );
  o.size2DXInPixel   = int32_t( p.getPatchSize2DXInPixel() );
  o.size2DYInPixel   = int32_t( p.getPatchSize2DYInPixel() );
  o.d0Count          = int32_t( p.getD0Count() );
  o.eomAndD1Count    = int32_t( p.getEOMandD1Count() );
  o.u0               = int32_t( p.getU0() );
  o.v0               = int32_t( p.getV0() );
  o.patchOrientation = int32_t( p.getPatchOrientation() );
  o.depthOffset      = depthOff;
  o.occOffset        = occOff;
}

}  // namespace

extern "C" {

int ref_version() { return TMC2_VERSION_MAJOR; }

// S1 + nanoflann kNN (PCCKdTree.cpp:56-66).  idx: nq*k u32, dist: nq*k f64 (may be NULL).
int ref_knn( const int16_t* xyz, size_t n, const int16_t* q, size_t nq, int k, uint32_t* idx, double* dist ) {
  PCCPointSet3 pc;
  makeCloud( pc, xyz, nullptr, n );
  PCCKdTree   tree( pc );
  PCCNNResult res;
  for ( size_t i = 0; i < nq; ++i ) {
    tree.search( PCCPoint3D( q[3 * i], q[3 * i + 1], q[3 * i + 2] ), size_t( k ), res );
    for ( int j = 0; j < k; ++j ) {
      idx[i * k + j] = uint32_t( res.indices( j ) );
      if ( dist ) dist[i * k + j] = res.dist( j );
    }
  }
  return 0;
}

// canonical radius search (PCCKdTree.cpp:68-79): returns count per query, results capped at cap.
int ref_radius( const int16_t* xyz, size_t n, const int16_t* q, size_t nq, double radius2, int cap, int32_t* count,
                uint32_t* idx ) {
  PCCPointSet3 pc;
  makeCloud( pc, xyz, nullptr, n );
  PCCKdTree tree( pc );
  for ( size_t i = 0; i < nq; ++i ) {
    PCCNNResult res;
    tree.searchRadius( PCCPoint3D( q[3 * i], q[3 * i + 1], q[3 * i + 2] ), size_t( cap ), radius2, res );
    count[i] = int32_t( res.count() );
    for ( size_t j = 0; j < res.count(); ++j ) idx[i * cap + j] = uint32_t( res.indices( j ) );
  }
  return 0;
}

// S2 (+S3 when orientation==1): PCCNormalsGenerator3 (PCCNormalsGenerator.cpp:61-70).
// stage 0 = computeNormals only (un-oriented), stage 1 = full compute().
int ref_normals( const int16_t* xyz, size_t n, int k, int orientation, int stage, double* normals ) {
  Quiet        quiet;
  PCCPointSet3 pc;
  makeCloud( pc, xyz, nullptr, n );
  PCCKdTree                            tree( pc );
  PCCNormalsGenerator3                 gen;
  const PCCNormalsGenerator3Parameters p = normalsParams( k, orientation );
  if ( stage == 0 ) {
    gen.nbThread_ = 1;
    gen.init( n, p );
    gen.computeNormals( pc, tree, p );
  } else {
    gen.compute( pc, tree, p, 1 );
  }
  for ( size_t i = 0; i < n; ++i ) {
    const auto nm      = gen.getNormal( i );
    normals[3 * i]     = nm[0];
    normals[3 * i + 1] = nm[1];
    normals[3 * i + 2] = nm[2];
  }
  return 0;
}

// S0: PCCEncoder::calculateWeightNormal (PCCEncoder.cpp:3569-3626), enhancedPP=1.
int ref_weight_normal( const int16_t* xyz, size_t n, int geometryBitDepth3D, double minWeightEPP, double* w ) {
  Quiet        quiet;
  PCCPointSet3 pc;
  makeCloud( pc, xyz, nullptr, n );
  PCCEncoder enc;
  enc.params_.enhancedPP_   = true;
  enc.params_.minWeightEPP_ = minWeightEPP;
  PCCVector3D v             = enc.calculateWeightNormal( size_t( geometryBitDepth3D ), pc );
  w[0]                      = v[0];
  w[1]                      = v[1];
  w[2]                      = v[2];
  return 0;
}

// S4: initialSegmentation (PCCPatchSegmenter.cpp:226-265), 6 planes.
int ref_initial_segmentation( const double* normals, size_t n, const double* weight, uint32_t* partition ) {
  Quiet                quiet;
  PCCPointSet3         pc;
  pc.resize( n );
  PCCNormalsGenerator3 gen;
  gen.getNormals().resize( n );
  for ( size_t i = 0; i < n; ++i )
    gen.getNormals()[i] = PCCVector3D( normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] );
  PCCPatchSegmenter3  seg;
  std::vector<size_t> part;
  seg.initialSegmentation( pc, gen, seg.orientations6, seg.orientationCount6, part,
                           PCCVector3D( weight[0], weight[1], weight[2] ) );
  for ( size_t i = 0; i < n; ++i ) partition[i] = uint32_t( part[i] );
  return 0;
}

// S5: refineSegmentationGridBased (PCCPatchSegmenter.cpp:1386-1561), 6 planes, partition in/out.
int ref_refine_grid( const int16_t* xyz, const double* normals, size_t n, uint32_t* partition, int maxNNCount,
                     double lambda, int iterationCount, int voxDim, int searchRadius ) {
  Quiet        quiet;
  PCCPointSet3 pc;
  makeCloud( pc, xyz, nullptr, n );
  PCCNormalsGenerator3 gen;
  gen.getNormals().resize( n );
  for ( size_t i = 0; i < n; ++i )
    gen.getNormals()[i] = PCCVector3D( normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] );
  std::vector<size_t> part( n );
  for ( size_t i = 0; i < n; ++i ) part[i] = partition[i];
  PCCPatchSegmenter3 seg;
  seg.refineSegmentationGridBased( pc, gen, seg.orientations6, seg.orientationCount6, size_t( maxNNCount ), lambda,
                                   size_t( iterationCount ), size_t( voxDim ), size_t( searchRadius ), part );
  for ( size_t i = 0; i < n; ++i ) partition[i] = uint32_t( part[i] );
  return 0;
}

// S1-S9: the whole PCCPatchSegmenter3::compute (PCCPatchSegmenter.cpp:53-150).
// Returns the number of patches; fetch them with ref_get_patches().
int ref_segment( const int16_t* xyz, const uint8_t* rgb, size_t n, const orc_seg_params* sp ) {
  Quiet        quiet;
  PCCPointSet3 pc;
  makeCloud( pc, xyz, rgb, n );
  PCCPatchSegmenter3Parameters p;
  toSegParams( *sp, p );
  PCCPatchSegmenter3 seg;
  seg.setNbThread( 1 );
  g_patches.clear();
  g_patches.reserve( 256 );
  std::vector<PCCPointSet3> sub;
  float                     dist = 0;
  seg.compute( pc, 0, p, g_patches, sub, dist );
  return int( g_patches.size() );
}

// sizes of the pools needed by ref_get_patches
int ref_patch_pool_sizes( int64_t* depthCount, int64_t* occCount ) {
  int64_t d = 0, o = 0;
  for ( auto& p : g_patches ) {
    d += int64_t( p.getSizeU() * p.getSizeV() );
    o += int64_t( p.getSizeU0() * p.getSizeV0() );
  }
  *depthCount = d;
  *occCount   = o;
  return 0;
}

int ref_get_patches( orc_patch* out, int16_t* depth0, int16_t* depth1, uint8_t* occ ) {
  int64_t d = 0, o = 0;
  for ( size_t i = 0; i < g_patches.size(); ++i ) {
    auto& p = g_patches[i];
    fillPatch( p, out[i], d, o );
    const size_t nd = p.getSizeU() * p.getSizeV();
    for ( size_t j = 0; j < nd; ++j ) {
      depth0[d + j] = p.getDepth( 0 )[j];
      depth1[d + j] = p.getDepth( 1 ).size() == nd ? p.getDepth( 1 )[j] : p.getDepth( 0 )[j];
    }
    const size_t no = p.getSizeU0() * p.getSizeV0();
    for ( size_t j = 0; j < no; ++j ) occ[o + j] = p.getOccupancy()[j] ? 1 : 0;
    d += int64_t( nd );
    o += int64_t( no );
  }
  return 0;
}

}  // extern "C"

// ================================================================================================
// GOF-level driver: the image-generation half of PCCEncoder::encode (PCCEncoder.cpp:71-730) with the
// video codec taken as the identity (compress() calls skipped: "decoded" video == generated video).
// The call ORDER below is the reference's; every call is a reference member function.
// ================================================================================================
namespace {
struct Gof {
  PCCEncoderParameters params;
  PCCEncoder           encoder;
  PCCContext           context;
  PCCGroupOfFrames     sources, reconstructs;
  PCCLogger            logger;
  std::vector<std::vector<uint32_t>> partitions;
  std::vector<std::vector<uint32_t>> occupancyAfterPhaseA;  // (phase B overwrites the tiles' occupancy maps)
};
std::unique_ptr<Gof> g_gof;

void setCtcParams( PCCEncoderParameters& p, int iterations, int bits3dMinus1, int occPrecision, int minW, int minH ) {
  // cfg/common/ctc-common.cfg
  p.nnNormalEstimation_                     = 16;
  p.maxNNCountRefineSegmentation_           = 1024;
  p.iterationCountRefineSegmentation_       = iterations;  // sequence cfg overrides the common 10
  p.voxelDimensionRefineSegmentation_       = 4;
  p.searchRadiusRefineSegmentation_         = 192;
  p.occupancyResolution_                    = 16;
  p.minPointCountPerCCPatchSegmentation_    = 16;
  p.maxNNCountPatchSegmentation_            = 16;
  p.surfaceThickness_                       = 4;
  p.maxAllowedDist2RawPointsDetection_      = 9;
  p.maxAllowedDist2RawPointsSelection_      = 1;
  p.lambdaRefineSegmentation_               = 3;
  p.minimumImageWidth_                      = minW;
  p.minimumImageHeight_                     = minH;
  p.bestColorSearchRange_                   = 0;
  p.numNeighborsColorTransferFwd_           = 8;
  p.numNeighborsColorTransferBwd_           = 1;
  p.useDistWeightedAverageFwd_              = true;
  p.useDistWeightedAverageBwd_              = true;
  p.skipAvgIfIdenticalSourcePointPresentFwd_ = true;
  p.skipAvgIfIdenticalSourcePointPresentBwd_ = true;
  p.distOffsetFwd_                          = 4;
  p.distOffsetBwd_                          = 4;
  p.maxGeometryDist2Fwd_                    = 1000;
  p.maxGeometryDist2Bwd_                    = 1000;
  p.maxColorDist2Fwd_                       = 1000;
  p.maxColorDist2Bwd_                       = 1000;
  p.maxCandidateCount_                      = 4;
  p.flagGeometrySmoothing_                  = true;
  p.gridSmoothing_                          = true;
  p.gridSize_                               = 8;
  p.thresholdSmoothing_                     = 64;
  p.thresholdColorPreSmoothing_             = 10.0;
  p.thresholdColorPreSmoothingLocalEntropy_ = 4.5;
  p.radius2ColorPreSmoothing_               = 64;
  p.neighborCountColorPreSmoothing_         = 64;
  p.flagColorPreSmoothing_                  = true;
  p.enablePointCloudPartitioning_           = false;
  p.enhancedOccupancyMapCode_               = false;
  p.profileReconstructionIdc_               = 1;
  // cfg/condition/ctc-all-intra.cfg
  p.constrainedPack_       = false;
  p.globalPatchAllocation_ = 0;
  // cfg/sequence/*.cfg
  p.geometry3dCoordinatesBitdepth_    = bits3dMinus1;
  p.geometryNominal2dBitdepth_        = 8;
  p.minNormSumOfInvDist4MPSelection_  = 0.33;
  p.partialAdditionalProjectionPlane_ = 0.17;
  p.maxPatchSize_                     = 1024;
  p.numTilesHor_                      = 2;
  p.tileHeightToWidthRatio_           = 1;
  // cfg/rate/ctc-rN.cfg
  p.occupancyPrecision_ = occPrecision;
  // what PccAppEncoder always supplies
  p.compressedStreamPath_ = "/tmp/tmc2_ref_harness.bin";
  p.uncompressedDataPath_ = "unused_%04d.ply";
  p.nbThread_             = 1;
  p.videoEncoderOccupancyCodecId_ = p.videoEncoderGeometryCodecId_ = p.videoEncoderAttributeCodecId_ = HMAPP;
  p.videoEncoderOccupancyPath_ = p.videoEncoderGeometryPath_ = p.videoEncoderAttributePath_ = "/bin/true";
  {
    std::ostringstream  sink;
    std::streambuf*     old = std::cerr.rdbuf();
    if ( !getenv( "TMC2_REF_VERBOSE" ) ) std::cerr.rdbuf( sink.rdbuf() );
    p.check();  // the reference's own parameter normalisation (forces absoluteD1/T1 for single-stream, ...)
    std::cerr.rdbuf( old );
  }
}
}  // namespace

extern "C" {

// constrainedPack != 0: cfg/condition/ctc-low-delay.cfg (the program default: frames after the first are packed by
// spatialConsistencyPackFlexible against their predecessor)
// constrainedPack == 2: additionally globalPatchAllocation = 1 (cfg/condition/ctc-random-access.cfg)
int ref_gof_begin2( int frameCount, int iterations, int bits3dMinus1, int occPrecision, int minW, int minH,
                    int constrainedPack ) {
  Quiet quiet;
  g_gof.reset( new Gof() );
  setCtcParams( g_gof->params, iterations, bits3dMinus1, occPrecision, minW, minH );
  g_gof->params.constrainedPack_       = constrainedPack != 0;
  g_gof->params.globalPatchAllocation_ = constrainedPack == 2 ? 1 : 0;
  g_gof->sources.setFrameCount( size_t( frameCount ) );
  return 0;
}

// sequence-level setting of cfg/sequence/{loot,redandblack,soldier}_vox10.cfg (2; the common value is 4); after ref_gof_begin*
int ref_gof_set_voxel_dimension_refine( int voxDim ) {
  g_gof->params.voxelDimensionRefineSegmentation_ = size_t( voxDim );
  return 0;
}

// --nbThread of PccAppEncoder: the width of the reference's TBB arenas (frames of a GOF in generateSegments, points and
// voxels inside normal estimation / segmentation, ...).  Only the ENABLE_TBB build (libtmc2ref_tbb.so) runs anything in
// parallel; results do not depend on it.  After ref_gof_begin*.
int ref_gof_set_nb_thread( int nbThread ) {
  g_gof->params.nbThread_ = size_t( nbThread );
  return 0;
}
int ref_built_with_tbb() {
#if defined( ENABLE_TBB )
  return 1;
#else
  return 0;
#endif
}

// per list position: position of the matched patch in the previous frame's list, or -1
int ref_gof_get_patch_matches( int frame, int32_t* out ) {
  auto& patches = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches();
  for ( size_t i = 0; i < patches.size(); ++i ) out[i] = int32_t( patches[i].getBestMatchIdx() );
  return 0;
}

int ref_gof_begin( int frameCount, int iterations, int bits3dMinus1, int occPrecision, int minW, int minH ) {
  Quiet quiet;
  g_gof.reset( new Gof() );
  setCtcParams( g_gof->params, iterations, bits3dMinus1, occPrecision, minW, minH );
  g_gof->sources.setFrameCount( size_t( frameCount ) );
  return 0;
}

int ref_gof_set_frame( int i, const int16_t* xyz, const uint8_t* rgb, size_t n ) {
  makeCloud( g_gof->sources[size_t( i )], xyz, rgb, n );
  return 0;
}

// what PccAppEncoder::compressVideo (PccAppEncoder.cpp:1042-1044) and the head of PCCEncoder::encode (:85-130) do before the
// first seam
static void gofPreEncode( Gof& G ) {
  PCCEncoder& E = G.encoder;
  E.setLogger( G.logger );
  E.setParameters( G.params );
  PCCContext& context = G.context;
  auto&       sources = G.sources;
  static PCCBitstreamStat bitstreamStat;
  context.setBitstreamStat( bitstreamStat );
  context.addV3CParameterSet( 0 );
  context.setActiveVpsId( 0 );
  G.reconstructs.setFrameCount( sources.getFrameCount() );
  context.resizeAtlas( 1 );
  context.setAtlasIndex( 0 );
  context.resize( sources.getFrameCount() );
  auto& frames = context.getFrames();
  for ( size_t i = 0; i < frames.size(); i++ ) {
    auto& fc = frames[i].getTitleFrameContext();
    fc.setFrameIndex( i );
    fc.setRawPatchEnabledFlag( E.params_.rawPointsPatch_ || E.params_.lossyRawPointsPatch_ );
    fc.setUseRawPointsSeparateVideo( E.params_.useRawPointsSeparateVideo_ );
    fc.setGeometry3dCoordinatesBitdepth( E.params_.geometry3dCoordinatesBitdepth_ + 1 );
    fc.setGeometry2dBitdepth( E.params_.geometryNominal2dBitdepth_ );
    fc.setMaxDepth( ( 1 << E.params_.geometryNominal2dBitdepth_ ) - 1 );
    fc.setLog2PatchQuantizerSizeX( E.params_.log2QuantizerSizeX_ );
    fc.setLog2PatchQuantizerSizeY( E.params_.log2QuantizerSizeY_ );
  }
}
// encode() between placeSegments and generateOccupancyMap
static void gofAfterPlacement( Gof& G ) {
  PCCContext& context    = G.context;
  auto&       frames     = context.getFrames();
  size_t      atlasIndex = context.getAtlasIndex();
  auto&       sps        = context.getVps();
  sps.setFrameWidth( atlasIndex, static_cast<uint16_t>( frames[0].getAtlasFrameWidth() ) );
  sps.setFrameHeight( atlasIndex, static_cast<uint16_t>( frames[0].getAtlasFrameHeight() ) );
  for ( auto& asps : context.getAtlasSequenceParameterSetList() ) {
    asps.setFrameHeight( sps.getFrameHeight( atlasIndex ) );
    asps.setFrameWidth( sps.getFrameWidth( atlasIndex ) );
  }
}

// S0-S16 in the order of PCCEncoder::encode :85-172
int ref_gof_phase_a() {
  Quiet quiet;
  Gof&  G = *g_gof;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );  // the reference also uses printf
  PCCEncoder& E = G.encoder;
  gofPreEncode( G );
  PCCContext& context = G.context;
  auto&       sources = G.sources;
  auto&       frames  = context.getFrames();
  E.generateSegments( sources, context );
  E.params_.initializeContext( context );
  E.placeSegments( sources, context );
  gofAfterPlacement( G );
  E.generateOccupancyMap( context, true );
  E.generateOccupancyMapVideo( sources, context );
  // identity codec: videoOccupancyMap stays as generated
  E.generateBlockToPatchFromOccupancyMapVideo( context, E.params_.occupancyResolution_, E.params_.occupancyPrecision_ );
  E.generateGeometryVideo( sources, context );
  G.occupancyAfterPhaseA.clear();
  for ( auto& fr : frames ) G.occupancyAfterPhaseA.push_back( fr.getTitleFrameContext().getOccupancyMap() );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}

int ref_gof_frame_size( int* width, int* height ) {
  auto& f = g_gof->context.getFrames()[0].getTitleFrameContext();
  *width  = int( f.getWidth() );
  *height = int( f.getHeight() );
  return 0;
}

int ref_gof_patch_count( int frame ) {
  return int( g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches().size() );
}

// patches of a frame in the (packing-sorted) list order, with u0/v0/orientation; pools optional
int ref_gof_get_patches( int frame, orc_patch* out ) {
  auto&   patches = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches();
  int64_t d = 0, o = 0;
  for ( size_t i = 0; i < patches.size(); ++i ) {
    fillPatch( patches[i], out[i], d, o );
    d += int64_t( patches[i].getSizeU() * patches[i].getSizeV() );
    o += int64_t( patches[i].getSizeU0() * patches[i].getSizeV0() );
  }
  return 0;
}

// occupancy map (u8 W*H), occupancy video luma (u8 (W/p)*(H/p)), blockToPatch (u32 (W/16)*(H/16)),
// geometry D0 / D1 luma (u16 W*H each).  Any pointer may be NULL.
int ref_gof_get_images( int frame, uint8_t* occupancy, uint8_t* occVideo, uint32_t* blockToPatch, uint16_t* geo0,
                        uint16_t* geo1 ) {
  Gof&  G  = *g_gof;
  auto& fc = G.context.getFrames()[size_t( frame )].getTitleFrameContext();
  if ( occupancy ) {
    auto& om = fc.getOccupancyMap();
    for ( size_t i = 0; i < om.size(); ++i ) occupancy[i] = uint8_t( om[i] );
  }
  if ( occVideo ) {
    auto& img = G.context.getVideoOccupancyMap().getFrame( size_t( frame ) );
    auto& ch  = img.getChannel( 0 );
    for ( size_t i = 0; i < img.getWidth() * img.getHeight(); ++i ) occVideo[i] = ch[i];
  }
  if ( blockToPatch ) {
    auto& b = fc.getBlockToPatch();
    for ( size_t i = 0; i < b.size(); ++i ) blockToPatch[i] = uint32_t( b[i] );
  }
  auto& vg = G.context.getVideoGeometryMultiple()[0];
  if ( geo0 ) {
    auto& ch = vg.getFrame( 2 * size_t( frame ) ).getChannel( 0 );
    std::copy( ch.begin(), ch.end(), geo0 );
  }
  if ( geo1 ) {
    auto& ch = vg.getFrame( 2 * size_t( frame ) + 1 ).getChannel( 0 );
    std::copy( ch.begin(), ch.end(), geo1 );
  }
  return 0;
}

// S17-S22 in the order of PCCEncoder::encode :313-424 (identity codec: the "decoded" geometry / occupancy videos
// are the generated ones).  The per-frame padding dispatch and the attribute group dilation are INLINE code of
// encode() (:342-424), not callable members; the two loops below transcribe that control flow and call the
// reference members (dilateSmoothedPushPull) for the actual work.
int ref_gof_phase_b() {
  Quiet quiet;
  Gof&  G = *g_gof;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  PCCEncoder& E       = G.encoder;
  PCCContext& context = G.context;
  auto&       frames  = context.getFrames();
  GeneratePointCloudParameters gpcParams;
  E.setGeneratePointCloudParameters( gpcParams, context );
  context.allocOneLayerData();
  G.partitions.assign( context.size(), std::vector<uint32_t>() );
  for ( size_t frameIdx = 0; frameIdx < context.size(); frameIdx++ ) {
    auto& frame = context[frameIdx];
    for ( size_t tileIdx = 0; tileIdx < frame.getNumTilesInAtlasFrame(); tileIdx++ ) {
      PCCPointSet3 tileReconstrct;
      E.generatePointCloud( tileReconstrct, context, frameIdx, tileIdx, gpcParams, G.partitions[frameIdx], false );
      G.reconstructs[frameIdx].appendPointSet( tileReconstrct );
    }
  }
  E.generateAttributeVideo( G.sources, G.reconstructs, context, E.params_ );
  const size_t mapCount = E.params_.mapCountMinus1_ + 1;
  for ( size_t f = 0; f < frames.size(); f++ ) {                       // encode() :349-423, attributeBGFill_ == 1
    for ( size_t mapIdx = 0; mapIdx < mapCount; mapIdx++ ) {
      auto& videoAttribute = context.getVideoAttributesMultiple()[0];
      E.dilateSmoothedPushPull( frames[f].getTitleFrameContext(), videoAttribute.getFrame( f * mapCount + mapIdx ) );
    }
    if ( mapCount > 1 && E.params_.groupDilation_ ) {                  // encode() :380-402
      auto& frame        = frames[f].getTitleFrameContext();
      auto& occupancyMap = frame.getOccupancyMap();
      auto& frame1       = context.getVideoAttributesMultiple()[0].getFrame( f * mapCount );
      auto& frame2       = context.getVideoAttributesMultiple()[0].getFrame( f * mapCount + 1 );
      for ( size_t y = 0; y < frame.getHeight(); y++ )
        for ( size_t x = 0; x < frame.getWidth(); x++ )
          if ( occupancyMap[y * frame.getWidth() + x] == 0 )
            for ( size_t c = 0; c < 3; c++ ) {
              uint8_t  d0  = frame1.getValue( c, x, y );
              uint8_t  d1  = frame2.getValue( c, x, y );
              uint32_t avg = ( static_cast<uint32_t>( d0 ) + static_cast<uint32_t>( d1 ) + 1 ) >> 1;
              frame1.setValue( c, x, y, static_cast<uint8_t>( avg ) );
              frame2.setValue( c, x, y, static_cast<uint8_t>( avg ) );
            }
    }
  }
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}

int64_t ref_gof_recon_count( int frame ) { return int64_t( g_gof->reconstructs[size_t( frame )].getPointCount() ); }

// reconstructed cloud of a frame: xyz int16[M][3], rgb u8[M][3] (after colour transfer), pointToPixel u32[M][3]
int ref_gof_get_recon( int frame, int16_t* xyz, uint8_t* rgb, uint32_t* pointToPixel ) {
  auto& rec = g_gof->reconstructs[size_t( frame )];
  auto& p2p = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPointToPixel();
  for ( size_t i = 0; i < rec.getPointCount(); ++i ) {
    if ( xyz ) {
      xyz[3 * i]     = rec[i][0];
      xyz[3 * i + 1] = rec[i][1];
      xyz[3 * i + 2] = rec[i][2];
    }
    if ( rgb ) {
      const auto c   = rec.getColor( i );
      rgb[3 * i]     = c[0];
      rgb[3 * i + 1] = c[1];
      rgb[3 * i + 2] = c[2];
    }
    if ( pointToPixel ) {
      pointToPixel[3 * i]     = uint32_t( p2p[i][0] );
      pointToPixel[3 * i + 1] = uint32_t( p2p[i][1] );
      pointToPixel[3 * i + 2] = uint32_t( p2p[i][2] );
    }
  }
  return 0;
}

// attribute images of a frame: u8 [2 maps][3 channels][H][W] (values are <= 255 in the reference's uint16 planes)
int ref_gof_get_attribute_images( int frame, uint8_t* out ) {
  auto&  va = g_gof->context.getVideoAttributesMultiple()[0];
  size_t o  = 0;
  for ( size_t m = 0; m < 2; ++m ) {
    auto& img = va.getFrame( 2 * size_t( frame ) + m );
    for ( size_t c = 0; c < 3; ++c )
      for ( auto v : img.getChannel( c ) ) {
        if ( v > 255 ) return -1;
        out[o++] = uint8_t( v );
      }
  }
  return 0;
}

// ---- PCCEncoder::placeSegments (:4762-4835) on patch RECORDS instead of segmented clouds: the packers (packFlexible,
// spatialConsistencyPackFlexible, resizeTileGeometryVideo, performDataAdaptiveGPAMethod, resizeGeometryVideo) see exactly the
// fields they read -- block sizes, block occupancy, the (u1, v1, sizeU, sizeV) box and the view of every patch.  records:
// all frames back to back, counts[f] each, in creation order; occupancy pool of frame f at occ + occBase[f].
// constrainedPack as in ref_gof_begin2.  Read the result with ref_gof_get_patches / _patch_matches / _frame_size.
int ref_place_records( int frames, const int32_t* counts, const orc_patch* records, const uint8_t* occ, const int64_t* occBase, int minW,
                       int minH, int constrainedPack ) {
  Quiet quiet;
  g_gof.reset( new Gof() );
  Gof& G = *g_gof;
  setCtcParams( G.params, 10, 10, 4, minW, minH );
  G.params.constrainedPack_       = constrainedPack != 0;
  G.params.globalPatchAllocation_ = constrainedPack == 2 ? 1 : 0;
  G.sources.setFrameCount( size_t( frames ) );
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  PCCEncoder& E = G.encoder;
  E.setLogger( G.logger );
  E.setParameters( G.params );
  PCCContext&             context = G.context;
  static PCCBitstreamStat bitstreamStat;
  context.setBitstreamStat( bitstreamStat );
  context.addV3CParameterSet( 0 );
  context.setActiveVpsId( 0 );
  context.resizeAtlas( 1 );
  context.setAtlasIndex( 0 );
  context.resize( size_t( frames ) );
  size_t at = 0;
  for ( int f = 0; f < frames; ++f ) {
    auto& fc = context.getFrames()[size_t( f )].getTitleFrameContext();
    fc.setFrameIndex( size_t( f ) );
    auto& patches = fc.getPatches();
    patches.resize( size_t( counts[f] ) );
    for ( int i = 0; i < counts[f]; ++i, ++at ) {
      const orc_patch& r = records[at];
      PCCPatch&        p = patches[size_t( i )];
      p.setIndex( size_t( r.index ) );
      p.setViewId( size_t( r.viewId ) );
      p.setU1( size_t( r.u1 ) ), p.setV1( size_t( r.v1 ) ), p.setD1( size_t( r.d1 ) );
      p.setSizeU( size_t( r.sizeU ) ), p.setSizeV( size_t( r.sizeV ) );
      p.setSizeU0( size_t( r.sizeU0 ) ), p.setSizeV0( size_t( r.sizeV0 ) );
      p.setOccupancyResolution( 16 );
      p.setBestMatchIdx( -1 );
      std::vector<bool> o( size_t( r.sizeU0 ) * size_t( r.sizeV0 ) );
      for ( size_t k = 0; k < o.size(); ++k ) o[k] = occ[occBase[f] + r.occOffset + int64_t( k )] != 0;
      p.setOccupancy( o );
    }
  }
  E.params_.initializeContext( context );
  E.placeSegments( G.sources, context );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}
// block occupancy of the patches of a frame after placeSegments, list order, back to back
int64_t ref_gof_get_patch_occupancy( int frame, uint8_t* out ) {
  auto&   patches = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches();
  int64_t o       = 0;
  for ( auto& p : patches )
    for ( bool b : p.getOccupancy() ) {
      if ( out ) out[o] = b ? 1 : 0;
      ++o;
    }
  return o;
}
int ref_gof_tile_size( int frame, int* width, int* height ) {
  auto& f = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext();
  *width  = int( f.getWidth() );
  *height = int( f.getHeight() );
  return 0;
}

// ---- post-reconstruction tail of PCCEncoder::encode (:571-719; the decoder runs the same members, PCCDecoder.cpp:330-470) ----
// boundary point types as generatePointCloud left them (identifyBoundaryPoints, PCCCodec.cpp:268-327)
int ref_gof_get_boundary_types( int frame, uint16_t* out ) {
  auto& rec = g_gof->reconstructs[size_t( frame )];
  for ( size_t i = 0; i < rec.getPointCount(); ++i ) out[i] = rec.getBoundaryPointType( i );
  return 0;
}
int ref_gof_get_partition( int frame, uint32_t* out ) {
  auto& part = g_gof->partitions[size_t( frame )];
  std::copy( part.begin(), part.end(), out );
  return int( part.size() );
}
// stand-in for the attribute video codec + colour conversion: the "decoded" attribute frames of a point-cloud frame,
// u16 [2 maps][3 channels][H][W]
int ref_gof_set_decoded_attribute( int frame, const uint16_t* planes ) {
  auto&  va = g_gof->context.getVideoAttributesMultiple()[0];
  size_t o  = 0;
  for ( size_t m = 0; m < 2; ++m ) {
    auto& img = va.getFrame( 2 * size_t( frame ) + m );
    for ( size_t c = 0; c < 3; ++c )
      for ( auto& v : img.getChannel( c ) ) v = planes[o++];
  }
  return 0;
}
// colorPointCloud for every frame, then the post-processing loop (grid geometry smoothing, transferColors16bitBP with
// attrTransferFilterType 1, convertYUV16ToRGB8), in the order and with the arguments of encode() :571-719
int ref_gof_phase_c() {
  Quiet quiet;
  Gof&  G = *g_gof;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  PCCEncoder& E       = G.encoder;
  PCCContext& context = G.context;
  GeneratePointCloudParameters gpcParams;
  E.setGeneratePointCloudParameters( gpcParams, context );
  std::vector<bool> absoluteT1List( E.params_.mapCountMinus1_ + 1, E.params_.absoluteT1_ );
  for ( size_t frameIdx = 0; frameIdx < context.size(); frameIdx++ ) {
    G.reconstructs[frameIdx].addColors();
    G.reconstructs[frameIdx].addColors16bit();
    size_t accTilePointCount = 0;
    for ( size_t tileIdx = 0; tileIdx < context[frameIdx].getNumTilesInAtlasFrame(); tileIdx++ ) {
      auto& tile        = context[frameIdx].getTile( tileIdx );
      accTilePointCount = E.colorPointCloud( G.reconstructs[frameIdx], context, tile, absoluteT1List, 0, 1, accTilePointCount, gpcParams );
    }
  }
  const bool isAttributes444 = static_cast<int>( E.params_.rawPointsPatch_ ) == 1;
  for ( size_t frameIdx = 0; frameIdx < G.sources.getFrameCount(); frameIdx++ ) {
    GeneratePointCloudParameters ppSEIParams;
    E.setPostProcessingSeiParameters( ppSEIParams, context );
    auto& reconstruct = G.reconstructs[frameIdx];
    auto& partition   = G.partitions[frameIdx];
    if ( E.params_.applyGeoSmoothingType_ != 0 && ppSEIParams.flagGeometrySmoothing_ ) {
      PCCPointSet3 tempFrameBuffer = reconstruct;
      if ( ppSEIParams.gridSmoothing_ ) E.smoothPointCloudPostprocess( reconstruct, E.params_.colorTransform_, ppSEIParams, partition );
      if ( !ppSEIParams.pbfEnableFlag_ && E.params_.attrTransferFilterType_ == 1 )
        tempFrameBuffer.transferColors16bitBP( reconstruct, E.params_.attrTransferFilterType_, int32_t( 0 ), isAttributes444, 8, 1,
                                               true, true, true, false, 4, 4, 1000, 1000, 1000 * 256, 1000 * 256 );
      else if ( !ppSEIParams.pbfEnableFlag_ )
        return -2;  // the other transfer filters are not part of the CTC path
    }
    if ( E.params_.applyAttrSmoothingType_ != 0 && ppSEIParams.flagColorSmoothing_ ) return -3;  // off under the CTC
    if ( !isAttributes444 )
      reconstruct.convertYUV16ToRGB8();
    else
      reconstruct.copyRGB16ToRGB8();
  }
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}
// after phase_c: positions int16[M][3], 16-bit colours u16[M][3], 8-bit colours u8[M][3], boundary types u16[M]
int ref_gof_get_post( int frame, int16_t* xyz, uint16_t* c16, uint8_t* rgb, uint16_t* btype ) {
  auto& rec = g_gof->reconstructs[size_t( frame )];
  for ( size_t i = 0; i < rec.getPointCount(); ++i ) {
    const auto c = rec.getColor( i );
    const auto d = rec.getColor16bit( i );
    for ( int k = 0; k < 3; ++k ) {
      xyz[3 * i + k] = rec[i][k];
      c16[3 * i + k] = d[k];
      rgb[3 * i + k] = c[k];
    }
    btype[i] = rec.getBoundaryPointType( i );
  }
  return 0;
}

// T3 + T4 on an arbitrary cloud (not one the pipeline produced): PCCCodec::smoothPointCloudPostprocess and
// PCCPointSet3::transferColors16bitBP with the encoder's arguments (PCCEncoder.cpp:646-672).  xyz / btype / colors16 in and out.
int ref_smooth_and_transfer( int16_t* xyz, uint16_t* btype, const uint32_t* partition, uint16_t* colors16, size_t M, int gridSize,
                             double thresholdSmoothing ) {
  Quiet        quiet;
  PCCEncoder   E;
  PCCPointSet3 rec;
  rec.addColors();
  rec.addColors16bit();
  rec.resize( M );
  std::vector<uint32_t> part( partition, partition + M );
  for ( size_t i = 0; i < M; ++i ) {
    rec[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    rec.setColor16bit( i, PCCColor16bit( colors16[3 * i], colors16[3 * i + 1], colors16[3 * i + 2] ) );
    rec.setBoundaryPointType( i, btype[i] );
  }
  GeneratePointCloudParameters pp;
  pp.flagGeometrySmoothing_ = true;
  pp.gridSmoothing_         = true;
  pp.gridSize_              = size_t( gridSize );
  pp.thresholdSmoothing_    = thresholdSmoothing;
  pp.pbfEnableFlag_         = false;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  dup2( fileno( devnull ), 1 );
  PCCPointSet3 temp = rec;
  E.smoothPointCloudPostprocess( rec, COLOR_TRANSFORM_NONE, pp, part );
  temp.transferColors16bitBP( rec, 1, int32_t( 0 ), false, 8, 1, true, true, true, false, 4, 4, 1000, 1000, 1000 * 256, 1000 * 256 );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  for ( size_t i = 0; i < M; ++i ) {
    const auto d = rec.getColor16bit( i );
    for ( int k = 0; k < 3; ++k ) xyz[3 * i + k] = rec[i][k], colors16[3 * i + k] = d[k];
    btype[i] = rec.getBoundaryPointType( i );
  }
  return 0;
}

// ---- colour-space conversion around the attribute video codec (PCCVideoEncoder::compress, PCCVideoEncoder.cpp:326-413, with
// the internal converter: "RGB444ToYUV420_8_<downsamplingFilter>" before the codec, "YUV420ToYUV444_8_<upsamplingFilter>"
// after it).  Attribute videos are PCCVideo<uint16_t, 3>.
int ref_convert_rgb444_to_yuv420( const uint8_t* rgb, int W, int H, int filter, uint8_t* y, uint8_t* u, uint8_t* v ) {
  Quiet                               quiet;
  PCCVideo<uint16_t, 3>               video;
  PCCInternalColorConverter<uint16_t> converter;
  video.resize( 1 );
  video[0].resize( size_t( W ), size_t( H ), PCCCOLORFORMAT::RGB444 );
  for ( int c = 0; c < 3; ++c )
    for ( size_t i = 0; i < size_t( W ) * H; ++i ) video[0][c][i] = rgb[size_t( c ) * W * H + i];
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  dup2( fileno( devnull ), 1 );
  converter.convert( "RGB444ToYUV420_8_" + std::to_string( filter ), video );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  if ( video[0][0].size() != size_t( W ) * H || video[0][1].size() != size_t( W / 2 ) * ( H / 2 ) ) return -1;
  for ( size_t i = 0; i < video[0][0].size(); ++i ) y[i] = uint8_t( video[0][0][i] );
  for ( size_t i = 0; i < video[0][1].size(); ++i ) u[i] = uint8_t( video[0][1][i] ), v[i] = uint8_t( video[0][2][i] );
  return 0;
}
int ref_convert_yuv420_to_yuv444( const uint8_t* y, const uint8_t* u, const uint8_t* v, int W, int H, int filter, uint16_t* out ) {
  Quiet                               quiet;
  PCCVideo<uint16_t, 3>               video, dst;
  PCCInternalColorConverter<uint16_t> converter;
  video.resize( 1 );
  video[0].resize( size_t( W ), size_t( H ), PCCCOLORFORMAT::YUV420 );
  if ( video[0][1].size() != size_t( W / 2 ) * ( H / 2 ) ) return -1;
  for ( size_t i = 0; i < size_t( W ) * H; ++i ) video[0][0][i] = y[i];
  for ( size_t i = 0; i < video[0][1].size(); ++i ) video[0][1][i] = u[i], video[0][2][i] = v[i];
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  dup2( fileno( devnull ), 1 );
  converter.convert( "YUV420ToYUV444_8_" + std::to_string( filter ), video, dst );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  for ( int c = 0; c < 3; ++c ) {
    if ( dst[0][c].size() != size_t( W ) * H ) return -1;
    std::copy( dst[0][c].begin(), dst[0][c].end(), out + size_t( c ) * W * H );
  }
  return 0;
}

// ---- integration/tmc2hip_convert.cpp against the containers the reference itself filled (after ref_gof_phase_a / _b) ----
// The caller passes what the C-ABI getters return for frame `frame` (tmc2_frame_get_geometry_images,
// tmc2_frame_get_attribute_images, tmc2_frame_get_reconstruction); the adaptor's conversions rebuild the reference's
// containers from them, and every element / format / plane is compared with the GOF context.  Returns the number of
// containers that differ (0 = a maintainer's write-back with these conversions leaves the context as the reference does).
int ref_adaptor_check_frame( int frame, const uint8_t* occupancy, const uint8_t* occVideo, const uint32_t* blockToPatch,
                             const uint16_t* geo0, const uint16_t* geo1, const uint8_t* attribute, const int16_t* recXyz,
                             const uint8_t* recRgb, const uint32_t* pointToPixel, size_t recCount ) {
  Gof&   G  = *g_gof;
  auto&  fc = G.context.getFrames()[size_t( frame )].getTitleFrameContext();
  size_t W = fc.getWidth(), H = fc.getHeight(), p = G.encoder.params_.occupancyPrecision_;
  int    bad = 0;
  auto   sameImage = []( auto& a, auto& b ) {
    if ( a.getWidth() != b.getWidth() || a.getHeight() != b.getHeight() || a.getColorFormat() != b.getColorFormat() ) return false;
    for ( size_t c = 0; c < 3; ++c )
      if ( a.getChannel( c ) != b.getChannel( c ) ) return false;
    return true;
  };
  {
    std::vector<uint32_t>  om;
    std::vector<size_t>    b2p;
    PCCImage<uint8_t, 3>   ov;
    PCCImage<uint16_t, 3>  d0, d1;
    tmc2hip::toFrameImages( occupancy, occVideo, blockToPatch, geo0, geo1, W, H, p, om, b2p, ov, d0, d1 );
    auto& vg = G.context.getVideoGeometryMultiple()[0];
    // (phase B overwrites the tile's occupancy map with the upsampled occupancy video, PCCCodec.cpp:559-572: compare phase A's)
    if ( !G.occupancyAfterPhaseA.empty() && om != G.occupancyAfterPhaseA[size_t( frame )] ) bad |= 1;
    if ( b2p != fc.getBlockToPatch() ) bad |= 2;
    if ( !sameImage( ov, G.context.getVideoOccupancyMap().getFrame( size_t( frame ) ) ) ) bad |= 4;
    if ( !sameImage( d0, vg.getFrame( 2 * size_t( frame ) ) ) ) bad |= 8;
    if ( !sameImage( d1, vg.getFrame( 2 * size_t( frame ) + 1 ) ) ) bad |= 16;
  }
  if ( attribute ) {
    PCCImage<uint16_t, 3> t0, t1;
    tmc2hip::toAttributeFrames( attribute, W, H, t0, t1 );
    auto& va = G.context.getVideoAttributesMultiple()[0];
    if ( !sameImage( t0, va.getFrame( 2 * size_t( frame ) ) ) ) bad |= 32;
    if ( !sameImage( t1, va.getFrame( 2 * size_t( frame ) + 1 ) ) ) bad |= 64;
  }
  if ( recXyz ) {
    PCCPointSet3                    cloud;
    std::vector<PCCVector3<size_t>> p2p;
    tmc2hip::toReconstruction( recXyz, recRgb, pointToPixel, recCount, cloud, p2p );
    auto& rec = G.reconstructs[size_t( frame )];
    if ( cloud.getPointCount() != rec.getPointCount() ) bad |= 128;
    else
      for ( size_t i = 0; i < rec.getPointCount(); ++i )
        if ( cloud[i] != rec[i] || cloud.getColor( i ) != rec.getColor( i ) ) {
          bad |= 128;
          break;
        }
    auto& refP2p = fc.getPointToPixel();
    if ( p2p.size() != refP2p.size() ) bad |= 256;
    else
      for ( size_t i = 0; i < p2p.size(); ++i )
        if ( p2p[i][0] != refP2p[i][0] || p2p[i][1] != refP2p[i][1] || p2p[i][2] != refP2p[i][2] ) {
          bad |= 256;
          break;
        }
  }
  return bad;
}

#ifdef TMC2_WITH_DROPIN
// ---- needs an MI355X (only built into oracle/_ref/libtmc2adaptor.so, which links the product library) --------------------------
// The GOF of g_gof once more, over a second PCCContext, in encode()'s order -- but with every seam of the hot path answered by
// integration/tmc2hip_adaptor.cpp's EncoderDropIn (the product, through the C-ABI) instead of the reference's member:
// generateSegments, placeSegments, generateOccupancyMap .. generateGeometryVideo, generatePointCloud .. the attribute padding.
// What encode() does in between (parameter set / context initialisation, sizes into the VPS / ASPS, allocOneLayerData) is the
// reference's own code on both sides.  Compared with what the reference's members left in g_gof (ref_gof_phase_a and
// ref_gof_phase_b must have run): bit mask of the containers that differ; negative = a status of the library.
//   1 patch lists (every getter the packers, the image generation and the bitstream writer read), 2 tile / atlas frame sizes,
//   4 occupancy maps, 8 blockToPatch, 16 occupancy video, 32 geometry video, 64 attribute video, 128 reconstructed clouds,
//   256 pointToPixel, 512 sizes written to the VPS / ASPS, 1024 matched-patch counts
// (PCCPatch::patchType_ is not compared: the packers write it, nothing in the reference reads it)
int ref_gof_dropin_check( int device ) {
  Quiet quiet;
  Gof&  A = *g_gof;
  Gof   B;
  B.params  = A.params;
  B.sources = A.sources;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  int status = 0;
  {
    gofPreEncode( B );
    PCCEncoder&            E = B.encoder;
    tmc2hip::EncoderDropIn D( device );
    if ( !D.accepts( E.params_ ) ) status = -1000;
    if ( status == 0 ) status = D.generateSegments( B.sources, B.context, E.params_ );
    if ( status == 0 ) {
      E.params_.initializeContext( B.context );
      status = D.placeSegments( B.context, E.params_ );
    }
    if ( status == 0 ) {
      gofAfterPlacement( B );
      status = D.generateGeometryVideo( B.context, E.params_ );
    }
    if ( status == 0 ) {
      B.context.allocOneLayerData();
      status = D.generateAttributeVideo( B.context, B.reconstructs, E.params_ );
    }
    if ( status != 0 ) fprintf( stderr, "ref_gof_dropin_check: status %d: %s\n", status, D.lastError() );
  }
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  if ( status != 0 ) return status < 0 ? status : -status;
  int   bad = 0;
  auto& fa = A.context.getFrames();
  auto& fb = B.context.getFrames();
  auto  sameImage = []( auto& a, auto& b ) {
    if ( a.getWidth() != b.getWidth() || a.getHeight() != b.getHeight() || a.getColorFormat() != b.getColorFormat() ) return false;
    for ( size_t c = 0; c < 3; ++c )
      if ( a.getChannel( c ) != b.getChannel( c ) ) return false;
    return true;
  };
  if ( fa.size() != fb.size() ) return 1 << 20;
  for ( size_t f = 0; f < fa.size(); ++f ) {
    auto& ta = fa[f].getTitleFrameContext();
    auto& tb = fb[f].getTitleFrameContext();
    auto& pa = ta.getPatches();
    auto& pb = tb.getPatches();
    if ( pa.size() != pb.size() ) {
      bad |= 1;
    } else {
      for ( size_t i = 0; i < pa.size(); ++i ) {
        const PCCPatch &a = pa[i], &b = pb[i];
#define SAME( getter ) ( a.getter() == b.getter() )
        if ( !( SAME( getIndex ) && SAME( getViewId ) && SAME( getNormalAxis ) && SAME( getTangentAxis ) && SAME( getBitangentAxis ) &&
                SAME( getProjectionMode ) && SAME( getU1 ) && SAME( getV1 ) && SAME( getD1 ) && SAME( getSizeU ) && SAME( getSizeV ) &&
                SAME( getSizeD ) && SAME( getSizeDPixel ) && SAME( getSizeU0 ) && SAME( getSizeV0 ) && SAME( getPatchSize2DXInPixel ) &&
                SAME( getPatchSize2DYInPixel ) && SAME( getOccupancyResolution ) && SAME( getD0Count ) && SAME( getEOMandD1Count ) &&
                SAME( getEOMCount ) && SAME( getLodScaleX ) && SAME( getLodScaleY ) && SAME( getU0 ) && SAME( getV0 ) &&
                SAME( getPatchOrientation ) && SAME( getBestMatchIdx ) && SAME( getAxisOfAdditionalPlane ) &&
                a.getDepth( 0 ) == b.getDepth( 0 ) && a.getDepth( 1 ) == b.getDepth( 1 ) && a.getOccupancy() == b.getOccupancy() ) )
          bad |= 1;
#undef SAME
      }
    }
    if ( ta.getNumMatchedPatches() != tb.getNumMatchedPatches() ) bad |= 1024;
    if ( ta.getWidth() != tb.getWidth() || ta.getHeight() != tb.getHeight() ||
         fa[f].getAtlasFrameWidth() != fb[f].getAtlasFrameWidth() || fa[f].getAtlasFrameHeight() != fb[f].getAtlasFrameHeight() ||
         fa[f].getNumPartitionWidth() != fb[f].getNumPartitionWidth() ||
         ( fa[f].getNumPartitionWidth() > 0 && ( fa[f].getPartitionWidth( 0 ) != fb[f].getPartitionWidth( 0 ) ||
                                                 fa[f].getPartitionHeight( 0 ) != fb[f].getPartitionHeight( 0 ) ) ) )
      bad |= 2;
    // (phase B overwrites the tile's occupancy map with the upsampled occupancy video: the reference side kept phase A's)
    if ( A.occupancyAfterPhaseA[f] != tb.getOccupancyMap() ) bad |= 4;
    if ( ta.getBlockToPatch() != tb.getBlockToPatch() ) bad |= 8;
    if ( !sameImage( A.context.getVideoOccupancyMap().getFrame( f ), B.context.getVideoOccupancyMap().getFrame( f ) ) ) bad |= 16;
    for ( size_t m = 0; m < 2; ++m ) {
      if ( !sameImage( A.context.getVideoGeometryMultiple()[0].getFrame( 2 * f + m ), B.context.getVideoGeometryMultiple()[0].getFrame( 2 * f + m ) ) )
        bad |= 32;
      if ( !sameImage( A.context.getVideoAttributesMultiple()[0].getFrame( 2 * f + m ), B.context.getVideoAttributesMultiple()[0].getFrame( 2 * f + m ) ) )
        bad |= 64;
    }
    auto& ra = A.reconstructs[f];
    auto& rb = B.reconstructs[f];
    if ( ra.getPointCount() != rb.getPointCount() ) {
      bad |= 128;
    } else {
      for ( size_t i = 0; i < ra.getPointCount(); ++i )
        if ( ra[i] != rb[i] || ra.getColor( i ) != rb.getColor( i ) ) {
          bad |= 128;
          break;
        }
    }
    auto& qa = ta.getPointToPixel();
    auto& qb = tb.getPointToPixel();
    if ( qa.size() != qb.size() ) {
      bad |= 256;
    } else {
      for ( size_t i = 0; i < qa.size(); ++i )
        if ( qa[i][0] != qb[i][0] || qa[i][1] != qb[i][1] || qa[i][2] != qb[i][2] ) {
          bad |= 256;
          break;
        }
    }
  }
  const size_t atlas = A.context.getAtlasIndex();
  if ( A.context.getVps().getFrameWidth( atlas ) != B.context.getVps().getFrameWidth( atlas ) ||
       A.context.getVps().getFrameHeight( atlas ) != B.context.getVps().getFrameHeight( atlas ) )
    bad |= 512;
  return bad;
}

// needs an MI355X, after ref_gof_phase_a / _b / (ref_gof_set_decoded_attribute) / _c: tmc2hip::DecoderDropIn::reconstructFrame --
// the per-frame finish of PCCDecoder::decode -- from the context's own containers (patch list, decoded occupancy / geometry /
// attribute frames) against the clouds the reference's members finished for the same GOF: positions, 16-bit colours, 8-bit
// colours, boundary point types.  Bit mask per kind of difference (1, 2, 4, 8); negative: the drop-in failed.
int ref_gof_decoder_dropin_check( int device ) {
  Gof&                   G = *g_gof;
  tmc2hip::DecoderDropIn D( device );
  GeneratePointCloudParameters pp;
  G.encoder.setPostProcessingSeiParameters( pp, G.context );
  int bad = 0;
  for ( size_t f = 0; f < G.context.size(); ++f ) {
    PCCPointSet3 got;
    const int    rc = D.reconstructFrame( G.context, f, G.encoder.params_.occupancyPrecision_, pp.gridSize_, pp.thresholdSmoothing_, got );
    if ( rc != 0 ) {
      fprintf( stderr, "ref_gof_decoder_dropin_check: frame %zu: status %d: %s\n", f, rc, D.lastError() );
      return rc < 0 ? rc : -rc;
    }
    const PCCPointSet3& want = G.reconstructs[f];
    if ( got.getPointCount() != want.getPointCount() ) return 1 << 20;
    for ( size_t i = 0; i < want.getPointCount(); ++i ) {
      if ( got[i] != want[i] ) bad |= 1;
      if ( got.getColor16bit( i ) != want.getColor16bit( i ) ) bad |= 2;
      if ( got.getColor( i ) != want.getColor( i ) ) bad |= 4;
      if ( got.getBoundaryPointType( i ) != want.getBoundaryPointType( i ) ) bad |= 8;
    }
  }
  return bad;
}
#endif

// integration/tmc2hip_convert.cpp, applyPacking: the records the product's packers return for a frame (by index, with
// placements), its list order and matches, applied to a PCCPatch vector in creation order -- against the vector the
// reference's own placeSegments left for that frame (after ref_place_records on the same records, all-intra or low-delay
// condition: the global patch allocation also rewrites block boxes, which is more than a placement).  Bit mask of what differs.
int ref_adaptor_check_packing( int frame, const tmc2_patch* recordsByIndex, const int32_t* order, const int32_t* matches, int count ) {
  auto& theirs = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches();
  if ( size_t( count ) != theirs.size() ) return 1;
  std::vector<PCCPatch> mine( static_cast<size_t>( count ) );
  for ( int i = 0; i < count; ++i ) {  // the vector as the segmenter leaves it: creation order, nothing placed
    const tmc2_patch& r = recordsByIndex[i];
    PCCPatch&         p = mine[size_t( i )];
    p.setIndex( size_t( r.index ) );
    p.setViewId( size_t( r.viewId ) );
    p.setU1( size_t( r.u1 ) ), p.setV1( size_t( r.v1 ) ), p.setD1( size_t( r.d1 ) );
    p.setSizeU( size_t( r.sizeU ) ), p.setSizeV( size_t( r.sizeV ) );
    p.setSizeU0( size_t( r.sizeU0 ) ), p.setSizeV0( size_t( r.sizeV0 ) );
    p.setBestMatchIdx( -1 );
  }
  tmc2hip::applyPacking( recordsByIndex, order, matches, count, mine );
  int bad = 0;
  for ( int k = 0; k < count; ++k ) {
    const PCCPatch &a = mine[size_t( k )], &b = theirs[size_t( k )];
    if ( a.getIndex() != b.getIndex() || a.getViewId() != b.getViewId() || a.getU1() != b.getU1() || a.getV1() != b.getV1() ||
         a.getSizeU0() != b.getSizeU0() || a.getSizeV0() != b.getSizeV0() )
      bad |= 2;  // list order
    if ( a.getU0() != b.getU0() || a.getV0() != b.getV0() ) bad |= 4;
    if ( a.getPatchOrientation() != b.getPatchOrientation() ) bad |= 8;
    if ( a.getBestMatchIdx() != b.getBestMatchIdx() ) bad |= 16;
  }
  return bad;
}

// the same for applyPackedList and the random-access condition (ref_place_records with constrainedPack = 2): index, block
// box, block occupancy, placement, orientation, best-match index of every list position
int ref_adaptor_check_packed_list( int frame, const tmc2_patch* createdRecords, int created, const tmc2_patch* list,
                                   const int32_t* matches, const uint8_t* occupancy, int count ) {
  auto& theirs = g_gof->context.getFrames()[size_t( frame )].getTitleFrameContext().getPatches();
  if ( size_t( count ) != theirs.size() || count != created ) return 1;
  std::vector<PCCPatch> mine( static_cast<size_t>( created ) );
  for ( int i = 0; i < created; ++i ) {
    const tmc2_patch& r = createdRecords[i];
    PCCPatch&         p = mine[size_t( i )];
    p.setIndex( size_t( r.index ) );
    p.setViewId( size_t( r.viewId ) );
    p.setU1( size_t( r.u1 ) ), p.setV1( size_t( r.v1 ) ), p.setD1( size_t( r.d1 ) );
    p.setSizeU( size_t( r.sizeU ) ), p.setSizeV( size_t( r.sizeV ) );
    p.setSizeU0( size_t( r.sizeU0 ) ), p.setSizeV0( size_t( r.sizeV0 ) );
    p.setBestMatchIdx( -1 );
  }
  tmc2hip::applyPackedList( list, matches, occupancy, count, mine );
  int bad = 0;
  for ( int k = 0; k < count; ++k ) {
    const PCCPatch &a = mine[size_t( k )], &b = theirs[size_t( k )];
    if ( a.getViewId() != b.getViewId() || a.getU1() != b.getU1() || a.getV1() != b.getV1() || a.getSizeU() != b.getSizeU() ||
         a.getSizeV() != b.getSizeV() )
      bad |= 2;  // which patch sits at this list position
    if ( a.getIndex() != b.getIndex() ) bad |= 32;
    if ( a.getSizeU0() != b.getSizeU0() || a.getSizeV0() != b.getSizeV0() ) bad |= 64;
    if ( a.getU0() != b.getU0() || a.getV0() != b.getV0() ) bad |= 4;
    if ( a.getPatchOrientation() != b.getPatchOrientation() ) bad |= 8;
    if ( a.getBestMatchIdx() != b.getBestMatchIdx() ) bad |= 16;
    if ( a.getOccupancy() != b.getOccupancy() ) bad |= 128;
  }
  return bad;
}

// ---- ingest and checksums: PCCPointSet3::read (PCCPointSet.cpp:464-757), computeChecksum (:222-243) ----
namespace {
PCCPointSet3 g_ply;
}
// returns the point count (-1: read failed); flags bit 0 colours, bit 1 normals
int64_t ref_ply_read( const char* path, int readNormals, int* flags ) {
  Quiet quiet;
  g_ply = PCCPointSet3();
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  dup2( fileno( devnull ), 1 );
  const bool ok = g_ply.read( path, readNormals != 0 );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  if ( !ok ) return -1;
  if ( flags ) *flags = ( g_ply.hasColors() ? 1 : 0 ) | ( g_ply.hasNormals() ? 2 : 0 );
  return int64_t( g_ply.getPointCount() );
}
int ref_ply_get( int16_t* xyz, uint8_t* rgb, double* normals ) {
  for ( size_t i = 0; i < g_ply.getPointCount(); ++i )
    for ( int k = 0; k < 3; ++k ) {
      xyz[3 * i + k] = g_ply[i][k];
      if ( rgb && g_ply.hasColors() ) rgb[3 * i + k] = g_ply.getColor( i )[k];
      if ( normals && g_ply.hasNormals() ) normals[3 * i + k] = g_ply.getNormals()[i][k];
    }
  return 0;
}
int ref_ply_write( const char* path, const int16_t* xyz, const uint8_t* rgb, const double* normals, size_t n, int asAscii ) {
  PCCPointSet3 pc;
  if ( rgb ) pc.addColors();
  if ( normals ) pc.addNormals();
  pc.resize( n );
  for ( size_t i = 0; i < n; ++i ) {
    pc[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    if ( rgb ) pc.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
    if ( normals ) pc.setNormal( i, PCCNormal3D( normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] ) );
  }
  return pc.write( path, asAscii != 0 ) ? 0 : -1;
}
// PCCChecksum::computeReconstructed + write for `frames` copies of clouds (xyz / rgb back to back, counts[f] points each);
// the file lands at <streamPath without extension>.checksum
int ref_checksum_file_write( const char* streamPath, const int16_t* xyz, const uint8_t* rgb, const int64_t* counts, int frames ) {
  Quiet            quiet;
  PCCGroupOfFrames gof;
  gof.setFrameCount( size_t( frames ) );
  size_t at = 0;
  for ( int f = 0; f < frames; ++f ) {
    makeCloud( gof[size_t( f )], xyz + 3 * at, rgb + 3 * at, size_t( counts[f] ) );
    at += size_t( counts[f] );
  }
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  dup2( fileno( devnull ), 1 );
  PCCChecksum checksum;
  checksum.computeReconstructed( gof );
  checksum.write( streamPath );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}
int ref_checksum( const int16_t* xyz, const uint8_t* rgb, size_t n, int reorderPoints, uint8_t* digest16 ) {
  PCCPointSet3 pc;
  if ( rgb ) pc.addColors();
  pc.resize( n );
  for ( size_t i = 0; i < n; ++i ) {
    pc[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    if ( rgb ) pc.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
  }
  const auto d = pc.computeChecksum( reorderPoints != 0 );
  std::copy( d.begin(), d.begin() + 16, digest16 );
  return 0;
}

// S18 alone: PCCPointSet3::transferColors (PCCPointSet.cpp:807-1124) with the arguments PCCEncoder::generateAttributeVideo
// passes under the CTC (PCCEncoder.cpp:6679-6697)
int ref_transfer_colors( const int16_t* srcXyz, const uint8_t* srcRgb, size_t n, const int16_t* tgtXyz, size_t m, uint8_t* tgtRgb ) {
  Quiet quiet;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  PCCPointSet3 src, tgt;
  makeCloud( src, srcXyz, srcRgb, n );
  makeCloud( tgt, tgtXyz, nullptr, m );
  src.transferColors( tgt, 0, false, 8, 1, true, true, true, true, 4, 4, 1000, 1000, 1000, 1000, false, 10.0 );
  for ( size_t i = 0; i < m; ++i ) {
    const auto c = tgt.getColor( i );
    tgtRgb[3 * i] = c[0], tgtRgb[3 * i + 1] = c[1], tgtRgb[3 * i + 2] = c[2];
  }
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}

// S23: PCCMetrics::compute (PCCMetrics.cpp:324-375) on one frame.  normals may be NULL (then no D2).
// out[3][8] = for q1 (A->B), q2 (B->A), final: c2cMse, c2cPsnr, c2pMse, c2pPsnr, colorMse[3] (Y,U,V) + colorPsnr[0].
// counts[2] = point counts of source / reconstruction after duplicate removal.
int ref_metrics( const int16_t* srcXyz, const uint8_t* srcRgb, size_t n, const int16_t* recXyz, const uint8_t* recRgb,
                 size_t m, const double* srcNormals, double resolution, double* out, int64_t* counts ) {
  Quiet quiet;
  fflush( stdout );
  FILE* devnull  = fopen( "/dev/null", "w" );
  int   savedOut = dup( 1 );
  if ( !getenv( "TMC2_REF_VERBOSE" ) ) dup2( fileno( devnull ), 1 );
  PCCGroupOfFrames sources, recs, normals;
  sources.setFrameCount( 1 );
  recs.setFrameCount( 1 );
  makeCloud( sources[0], srcXyz, srcRgb, n );
  makeCloud( recs[0], recXyz, recRgb, m );
  if ( srcNormals ) {
    normals.setFrameCount( 1 );
    makeCloud( normals[0], srcXyz, srcRgb, n );
    normals[0].addNormals();
    for ( size_t i = 0; i < n; ++i )
      normals[0].setNormal( i, PCCNormal3D( srcNormals[3 * i], srcNormals[3 * i + 1], srcNormals[3 * i + 2] ) );
  }
  PCCMetricsParameters mp;
  mp.resolution_ = size_t( resolution );
  mp.computeC2p_ = srcNormals != nullptr;
  PCCMetrics metrics;
  metrics.setParameters( mp );
  metrics.compute( sources, recs, normals );
  QualityMetrics* q[3] = {&metrics.quality1_[0], &metrics.quality2_[0], &metrics.qualityF_[0]};
  for ( int i = 0; i < 3; ++i ) {
    out[8 * i + 0] = q[i]->c2cMse_;
    out[8 * i + 1] = q[i]->c2cPsnr_;
    out[8 * i + 2] = q[i]->c2pMse_;
    out[8 * i + 3] = q[i]->c2pPsnr_;
    out[8 * i + 4] = q[i]->colorMse_[0];
    out[8 * i + 5] = q[i]->colorMse_[1];
    out[8 * i + 6] = q[i]->colorMse_[2];
    out[8 * i + 7] = q[i]->colorPsnr_[0];
  }
  counts[0] = int64_t( metrics.sourceDuplicates_[0] );
  counts[1] = int64_t( metrics.reconstructDuplicates_[0] );
  fflush( stdout );
  dup2( savedOut, 1 );
  close( savedOut );
  fclose( devnull );
  return 0;
}

// the text PCCMetrics::display() prints for the same inputs, at the precision the applications give std::cout (9)
int64_t ref_metrics_display( const int16_t* srcXyz, const uint8_t* srcRgb, size_t n, const int16_t* recXyz, const uint8_t* recRgb,
                             size_t m, const double* srcNormals, double resolution, char* text, int64_t capacity ) {
  PCCGroupOfFrames sources, recs, normals;
  sources.setFrameCount( 1 );
  recs.setFrameCount( 1 );
  makeCloud( sources[0], srcXyz, srcRgb, n );
  makeCloud( recs[0], recXyz, recRgb, m );
  if ( srcNormals ) {
    normals.setFrameCount( 1 );
    makeCloud( normals[0], srcXyz, srcRgb, n );
    normals[0].addNormals();
    for ( size_t i = 0; i < n; ++i )
      normals[0].setNormal( i, PCCNormal3D( srcNormals[3 * i], srcNormals[3 * i + 1], srcNormals[3 * i + 2] ) );
  }
  PCCMetricsParameters mp;
  mp.resolution_ = size_t( resolution );
  mp.computeC2p_ = srcNormals != nullptr;
  PCCMetrics metrics;
  metrics.setParameters( mp );
  char  path[] = "/tmp/tmc2_ref_display_XXXXXX";
  int   fd     = mkstemp( path );
  fflush( stdout );
  int savedOut = dup( 1 );
  dup2( fd, 1 );
  metrics.compute( sources, recs, normals );
  fflush( stdout );
  std::cout.flush();
  ftruncate( fd, 0 );
  lseek( fd, 0, SEEK_SET );
  const auto oldPrecision = std::cout.precision( std::numeric_limits<float>::max_digits10 );
  metrics.display();
  std::cout.flush();
  fflush( stdout );
  std::cout.precision( oldPrecision );
  dup2( savedOut, 1 );
  close( savedOut );
  const int64_t size = lseek( fd, 0, SEEK_END );
  lseek( fd, 0, SEEK_SET );
  int64_t got = 0;
  if ( size < capacity ) {
    got       = read( fd, text, size_t( size ) );
    text[got] = 0;
  }
  close( fd );
  unlink( path );
  return size;
}

// chroma planes of the geometry frames must stay zero (generateIntraImage :3932, dilate3DPadding on 3 channels)
int ref_gof_geometry_chroma_nonzero( int frame ) {
  auto&  vg = g_gof->context.getVideoGeometryMultiple()[0];
  size_t nz = 0;
  for ( size_t m = 0; m < 2; ++m )
    for ( size_t c = 1; c < 3; ++c )
      for ( auto v : vg.getFrame( 2 * size_t( frame ) + m ).getChannel( c ) ) nz += v != 0;
  return int( nz );
}

}  // extern "C"
