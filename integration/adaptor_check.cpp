// integration/adaptor_check.cpp -- TEST INFRASTRUCTURE: runs the reference's own PCCPatchSegmenter3 next to the adaptor's
// conversions (and, on a machine with an MI355X, next to the adaptor's drop-in segmenterCompute) and counts differences
// between the two PCCPatch lists, getter by getter.  Built into oracle/_ref/libtmc2adaptor.so together with the unmodified
// reference objects; never part of the product.
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#define private public  // (QualityMetrics keeps its numbers private: the check reads them next to the drop-in's)
#include "PCCMetrics.h"
#undef private
#include "tmc2hip_adaptor.h"

using namespace pcc;

namespace {
void referenceParams( const tmc2_segmenter_params& s, PCCPatchSegmenter3Parameters& p ) {
  p.gridBasedSegmentation_               = false;
  p.voxelDimensionGridBasedSegmentation_ = 2;
  p.nnNormalEstimation_                  = size_t( s.nnNormalEstimation );
  p.normalOrientation_                   = size_t( s.normalOrientation );
  p.gridBasedRefineSegmentation_         = s.gridBasedRefineSegmentation != 0;
  p.maxNNCountRefineSegmentation_        = size_t( s.maxNNCountRefineSegmentation );
  p.iterationCountRefineSegmentation_    = size_t( s.iterationCountRefineSegmentation );
  p.voxelDimensionRefineSegmentation_    = size_t( s.voxelDimensionRefineSegmentation );
  p.searchRadiusRefineSegmentation_      = size_t( s.searchRadiusRefineSegmentation );
  p.occupancyResolution_                 = size_t( s.occupancyResolution );
  p.enablePatchSplitting_                = s.enablePatchSplitting != 0;
  p.maxPatchSize_                        = size_t( s.maxPatchSize );
  p.quantizerSizeX_                      = size_t( s.quantizerSizeX );
  p.quantizerSizeY_                      = size_t( s.quantizerSizeY );
  p.minPointCountPerCCPatchSegmentation_ = size_t( s.minPointCountPerCCPatchSegmentation );
  p.maxNNCountPatchSegmentation_         = size_t( s.maxNNCountPatchSegmentation );
  p.surfaceThickness_                    = size_t( s.surfaceThickness );
  p.EOMFixBitCount_                      = 2;
  p.EOMSingleLayerMode_                  = false;
  p.mapCountMinus1_                      = size_t( s.mapCountMinus1 );
  p.minLevel_                            = size_t( s.minLevel );
  p.maxAllowedDepth_                     = size_t( s.maxAllowedDepth );
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
  p.geometryBitDepth2D_                  = size_t( s.geometryBitDepth2D );
  p.geometryBitDepth3D_                  = size_t( s.geometryBitDepth3D );
  p.patchExpansion_                      = false;
  p.highGradientSeparation_              = false;
  p.minGradient_                         = 15.0;
  p.minNumHighGradientPoints_            = 256;
  p.enablePointCloudPartitioning_        = false;
  p.numTilesHor_                         = 2;
  p.tileHeightToWidthRatio_              = 1.0;
  p.numCutsAlong1stLongestAxis_ = p.numCutsAlong2ndLongestAxis_ = p.numCutsAlong3rdLongestAxis_ = 1;
}
void makeCloud( PCCPointSet3& pc, const int16_t* xyz, const uint8_t* rgb, size_t n ) {
  pc.addColors();
  pc.resize( n );
  for ( size_t i = 0; i < n; ++i ) {
    pc[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    pc.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
  }
}
struct Quiet {  // the reference prints progress to stdout
  FILE* devnull;
  int   saved;
  Quiet() {
    fflush( stdout );
    devnull = fopen( "/dev/null", "w" );
    saved   = dup( 1 );
    dup2( fileno( devnull ), 1 );
  }
  ~Quiet() {
    fflush( stdout );
    dup2( saved, 1 );
    close( saved );
    fclose( devnull );
  }
};
int differ( const char* what, size_t patch, long long a, long long b, int& shown ) {
  if ( a == b ) return 0;
  if ( shown++ < 8 ) fprintf( stderr, "adaptor_check: patch %zu %s: reference %lld, adaptor %lld\n", patch, what, a, b );
  return 1;
}
// every getter the encoder reads from a segmented patch before packing
int comparePatchLists( const std::vector<PCCPatch>& ref, const std::vector<PCCPatch>& got ) {
  int bad = 0, shown = 0;
  if ( ref.size() != got.size() ) {
    fprintf( stderr, "adaptor_check: %zu reference patches, %zu from the adaptor\n", ref.size(), got.size() );
    return 1 << 20;
  }
  for ( size_t i = 0; i < ref.size(); ++i ) {
    const PCCPatch &a = ref[i], &b = got[i];
#define CMP( getter ) bad += differ( #getter, i, (long long)a.getter(), (long long)b.getter(), shown )
    CMP( getIndex );
    CMP( getFrameIndex );
    CMP( getViewId );
    CMP( getNormalAxis );
    CMP( getTangentAxis );
    CMP( getBitangentAxis );
    CMP( getProjectionMode );
    CMP( getAxisOfAdditionalPlane );
    CMP( getU1 );
    CMP( getV1 );
    CMP( getD1 );
    CMP( getSizeU );
    CMP( getSizeV );
    CMP( getSizeD );
    CMP( getSizeDPixel );
    CMP( getSizeU0 );
    CMP( getSizeV0 );
    CMP( getPatchSize2DXInPixel );
    CMP( getPatchSize2DYInPixel );
    CMP( getOccupancyResolution );
    CMP( getD0Count );
    CMP( getEOMandD1Count );
    CMP( getEOMCount );
    CMP( getLodScaleX );
    CMP( getLodScaleY );
#undef CMP
    for ( int m = 0; m < 2; ++m ) {
      const auto &da = a.getDepth( m ), &db = b.getDepth( m );
      // (the reference leaves the second map empty when a single map is coded; with two maps both are full)
      if ( da.size() != db.size() ) bad += differ( m ? "depth1 size" : "depth0 size", i, (long long)da.size(), (long long)db.size(), shown );
      else
        for ( size_t k = 0; k < da.size(); ++k )
          if ( da[k] != db[k] ) {
            bad += differ( m ? "depth1 value" : "depth0 value", i, da[k], db[k], shown );
            break;
          }
    }
    const auto &oa = a.getOccupancy(), &ob = b.getOccupancy();
    if ( oa.size() != ob.size() ) bad += differ( "occupancy size", i, (long long)oa.size(), (long long)ob.size(), shown );
    else
      for ( size_t k = 0; k < oa.size(); ++k )
        if ( oa[k] != ob[k] ) {
          bad += differ( "occupancy value", i, oa[k], ob[k], shown );
          break;
        }
  }
  return bad;
}
}  // namespace

extern "C" {

// the reference's segmenter on (xyz, rgb) next to toPCCPatches( records the caller obtained for the same cloud ); returns the
// number of differing getters (0 = the adaptor rebuilds exactly what PCCPatchSegmenter3::compute appends)
int adaptor_check_patches( const int16_t* xyz, const uint8_t* rgb, size_t n, const tmc2_segmenter_params* sp, const tmc2_patch* records,
                           int count, const int16_t* depth0, const int16_t* depth1, const uint8_t* occupancy ) {
  PCCPatchSegmenter3Parameters params;
  referenceParams( *sp, params );
  std::vector<PCCPatch> ref;
  {
    Quiet        quiet;
    PCCPointSet3 cloud;
    makeCloud( cloud, xyz, rgb, n );
    PCCPatchSegmenter3 seg;
    seg.setNbThread( 1 );
    ref.reserve( 256 );
    std::vector<PCCPointSet3> sub;
    float                     dist = 0;
    seg.compute( cloud, 3, params, ref, sub, dist );
    // flatten() must hand the library the same cloud
    std::vector<int16_t> fx;
    std::vector<uint8_t> fc;
    tmc2hip::flatten( cloud, fx, fc );
    if ( fx.size() != 3 * n || memcmp( fx.data(), xyz, 6 * n ) != 0 || fc.size() != 3 * n || memcmp( fc.data(), rgb, 3 * n ) != 0 ) return -1;
  }
  // the parameter mapping, there and back
  tmc2_segmenter_params back;
  if ( !tmc2hip::toParams( params, back ) ) return -2;
  if ( memcmp( &back, sp, sizeof( back ) ) != 0 ) return -3;
  std::vector<PCCPatch> got;
  tmc2hip::toPCCPatches( records, count, depth0, depth1, occupancy, params.occupancyResolution_, 3, got );
  int bad = comparePatchLists( ref, got );
  // and back to records (what a decoder-side caller does with its own patch list)
  std::vector<tmc2_patch> rec2;
  tmc2hip::toRecords( got, rec2 );
  for ( int i = 0; i < count && bad == 0; ++i ) {
    tmc2_patch a = records[i], b = rec2[size_t( i )];
    a.u0 = b.u0, a.v0 = b.v0, a.patchOrientation = b.patchOrientation;  // not set before packing
    if ( memcmp( &a, &b, sizeof( a ) ) != 0 ) bad = 1 << 21;
  }
  return bad;
}

// needs an MI355X: the adaptor's drop-in body of PCCPatchSegmenter3::compute against the reference's own on the same cloud
int adaptor_check_segmenter_compute( int device, const int16_t* xyz, const uint8_t* rgb, size_t n, const tmc2_segmenter_params* sp ) {
  PCCPatchSegmenter3Parameters params;
  referenceParams( *sp, params );
  PCCPointSet3 cloud;
  makeCloud( cloud, xyz, rgb, n );
  std::vector<PCCPatch> ref, got;
  {
    Quiet              quiet;
    PCCPatchSegmenter3 seg;
    seg.setNbThread( 1 );
    ref.reserve( 256 );
    std::vector<PCCPointSet3> sub;
    float                     dist = 0;
    seg.compute( cloud, 3, params, ref, sub, dist );
  }
  tmc2_ctx* ctx = nullptr;
  if ( tmc2_ctx_create( device, &ctx ) != TMC2_OK ) return -10;
  const int r = tmc2hip::segmenterCompute( ctx, cloud, 3, params, got, nullptr );
  tmc2_ctx_destroy( ctx );
  if ( r != TMC2_OK ) return -20 + r;
  return comparePatchLists( ref, got );
}

// needs an MI355X: tmc2hip::MetricsDropIn in the place of the PCCMetrics object -- setParameters, compute( sources, reconstructs,
// normals ), display() -- against the reference's own object on the same groups of frames: the 24 numbers of every frame (raw
// doubles) and the text display() prints, at the precision the applications set.  0 = identical; bit 0: numbers, bit 1: counts,
// bit 2: text.
int adaptor_check_metrics( int device, int frames, const int16_t* srcXyz, const uint8_t* srcRgb, const int64_t* n, const int16_t* recXyz,
                           const uint8_t* recRgb, const int64_t* m, const double* srcNormals, double resolution ) {
  PCCGroupOfFrames sources, recs, normals;
  sources.setFrameCount( size_t( frames ) );
  recs.setFrameCount( size_t( frames ) );
  if ( srcNormals ) normals.setFrameCount( size_t( frames ) );
  size_t so = 0, ro = 0;
  for ( int f = 0; f < frames; ++f ) {
    makeCloud( sources[size_t( f )], srcXyz + 3 * so, srcRgb + 3 * so, size_t( n[f] ) );
    makeCloud( recs[size_t( f )], recXyz + 3 * ro, recRgb + 3 * ro, size_t( m[f] ) );
    if ( srcNormals ) {
      auto& nc = normals[size_t( f )];
      makeCloud( nc, srcXyz + 3 * so, srcRgb + 3 * so, size_t( n[f] ) );
      nc.addNormals();
      for ( size_t i = 0; i < size_t( n[f] ); ++i )
        nc.setNormal( i, PCCNormal3D( srcNormals[3 * ( so + i )], srcNormals[3 * ( so + i ) + 1], srcNormals[3 * ( so + i ) + 2] ) );
    }
    so += size_t( n[f] ), ro += size_t( m[f] );
  }
  PCCMetricsParameters mp;
  mp.computeMetrics_ = true;
  mp.resolution_     = size_t( resolution );
  mp.computeC2p_     = srcNormals != nullptr;
  auto captured = [&]( auto&& body ) {  // what `body` writes to stdout
    char path[] = "/tmp/tmc2_adaptor_metrics_XXXXXX";
    int  fd     = mkstemp( path );
    fflush( stdout );
    std::cout.flush();
    int saved = dup( 1 );
    dup2( fd, 1 );
    const auto old = std::cout.precision( std::numeric_limits<float>::max_digits10 );
    body();
    std::cout.flush();
    fflush( stdout );
    std::cout.precision( old );
    dup2( saved, 1 );
    close( saved );
    std::string text( size_t( lseek( fd, 0, SEEK_END ) ), '\0' );
    lseek( fd, 0, SEEK_SET );
    if ( !text.empty() && read( fd, &text[0], text.size() ) != ssize_t( text.size() ) ) text.clear();
    close( fd );
    unlink( path );
    return text;
  };
  PCCMetrics ref;
  ref.setParameters( mp );
  {
    Quiet quiet;
    ref.compute( sources, recs, normals );
  }
  const std::string refText = captured( [&] { ref.display(); } );
  tmc2hip::MetricsDropIn got( device );
  if ( !got.accepts( mp ) ) return -10;
  got.setParameters( mp );
  const int rc = got.compute( sources, recs, normals );
  if ( rc != TMC2_OK ) return -20 + rc;
  const std::string gotText = captured( [&] { got.display(); } );
  int bad = 0;
  if ( got.results().size() != size_t( frames ) ) return 1 << 20;
  for ( int f = 0; f < frames; ++f ) {
    const QualityMetrics* q[3] = {&ref.quality1_[size_t( f )], &ref.quality2_[size_t( f )], &ref.qualityF_[size_t( f )]};
    for ( int i = 0; i < 3; ++i ) {
      const double want[8] = {q[i]->c2cMse_,      q[i]->c2cPsnr_,     q[i]->c2pMse_,      q[i]->c2pPsnr_,
                              q[i]->colorMse_[0], q[i]->colorMse_[1], q[i]->colorMse_[2], q[i]->colorPsnr_[0]};
      if ( memcmp( want, got.results()[size_t( f )].data() + 8 * i, sizeof( want ) ) != 0 ) bad |= 1;
    }
  }
  if ( refText != gotText ) {
    bad |= 4;
    fprintf( stderr, "adaptor_check_metrics: display() differs\n--- reference\n%s--- drop-in\n%s", refText.c_str(), gotText.c_str() );
  }
  return bad;
}

// needs an MI355X: tmc2hip::KdTreeDropIn against PCCKdTree on the same cloud and queries: indices (incl. the order among
// equidistant neighbours) and squared distances of every result.  Returns the number of queries whose result rows differ.
int adaptor_check_kdtree( int device, const int16_t* xyz, size_t n, const int16_t* queries, size_t nq, int k ) {
  PCCPointSet3 cloud;
  cloud.resize( n );
  for ( size_t i = 0; i < n; ++i ) cloud[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
  PCCKdTree             ref( cloud );
  tmc2hip::KdTreeDropIn got( device );
  if ( got.init( cloud ) != TMC2_OK ) return -10;
  std::vector<PCCPoint3D> q( nq );
  for ( size_t i = 0; i < nq; ++i ) q[i] = PCCPoint3D( queries[3 * i], queries[3 * i + 1], queries[3 * i + 2] );
  std::vector<PCCNNResult> rows;
  if ( got.searchBatch( q, size_t( k ), rows ) != TMC2_OK ) return -20;
  int         bad = 0;
  PCCNNResult want, one;
  for ( size_t i = 0; i < nq; ++i ) {
    ref.search( q[i], size_t( k ), want );
    bool same = rows[i].size() == want.size();
    for ( size_t j = 0; same && j < want.size(); ++j ) same = rows[i].indices( j ) == want.indices( j ) && rows[i].dist( j ) == want.dist( j );
    if ( i < 4 ) {  // and the one-point signature
      if ( got.search( q[i], size_t( k ), one ) != TMC2_OK ) return -30;
      for ( size_t j = 0; same && j < want.size(); ++j ) same = one.indices( j ) == want.indices( j ) && one.dist( j ) == want.dist( j );
    }
    bad += same ? 0 : 1;
  }
  return bad;
}
}
