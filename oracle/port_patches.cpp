// oracle/port_patches.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S7  connected components on the directed 16-NN graph, same plane, seeded in index order
//     (PCCPatchSegmenter3::segmentPatches, PCCPatchSegmenter.cpp:804-841)
// S8  per-component patch: bounding box, D0 map, 64-quantised depth origin, per-block peak filter,
//     D1 map, block occupancy, resampled points, depth-range quantisation (:910-1290, resampledPointcloud :362-470)
// S9  raw-point update: distance of every input point to the resampled cloud (:1291-1298)
// plus orc_segment = the whole PCCPatchSegmenter3::compute (:53-150) chained from the other port files.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "oracle.h"

struct orc_kdtree;
extern "C" {
orc_kdtree* orc_kdtree_build( const int16_t* xyz, size_t n );
void        orc_kdtree_free( orc_kdtree* t );
int         orc_knn( const orc_kdtree* t, const int16_t* q, size_t nq, int k, uint32_t* idx, double* dist );
int         orc_compute_normals( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals );
int         orc_orient_normals( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals );
int         orc_initial_segmentation( const double* normals, size_t n, const double* weight, uint32_t* partition );
int         orc_refine_grid( const int16_t* xyz, const double* normals, size_t n, uint32_t* partition, int maxNNCount,
                             double lambda, int iterationCount, int voxDim, int searchRadius );
}

struct orc_seg_result {
  std::vector<orc_patch> patches;
  std::vector<int16_t>   depth0, depth1;
  std::vector<uint8_t>   occupancy;
  std::vector<int16_t>   resampled;  // xyz triples
  std::vector<uint32_t>  rawPoints;
  std::vector<int32_t>   roundRaw;   // raw-point count after each round
};

namespace {
const int16_t INF_DEPTH = 32767;
// viewId -> (normal, tangent, bitangent) axes; projection mode = viewId / 3   (PCCPatch::setViewId)
const int AXES[3][3] = {{0, 2, 1}, {1, 2, 0}, {2, 0, 1}};
}  // namespace

extern "C" {

orc_seg_result* orc_segment_patches( const int16_t* xyz, const uint8_t* rgb, size_t n, const uint32_t* knn, int K,
                                     const uint32_t* partition, const orc_seg_params* sp ) {
  orc_seg_result* R = new orc_seg_result();
  std::vector<uint32_t> raw( n );
  std::vector<double>   rawDist( n, 1.7976931348623157e308 );
  for ( size_t i = 0; i < n; ++i ) raw[i] = uint32_t( i );
  const int occRes = sp->occupancyResolution;
  while ( !raw.empty() ) {
    // ---- S7: components in seed order -------------------------------------------------------------
    std::vector<std::vector<uint32_t>> ccs;
    {
      std::vector<uint8_t>  flag( n, 0 );
      std::vector<uint32_t> stack;
      for ( uint32_t i : raw ) flag[i] = 1;
      for ( uint32_t i : raw ) {
        if ( !flag[i] || !( rawDist[i] > sp->maxAllowedDist2RawPointsDetection ) ) continue;
        flag[i]            = 0;
        const uint32_t pl  = partition[i];
        std::vector<uint32_t> cc;
        stack.push_back( i );
        cc.push_back( i );
        while ( !stack.empty() ) {
          const uint32_t cur = stack.back();
          stack.pop_back();
          for ( int j = 0; j < K; ++j ) {
            const uint32_t v = knn[size_t( cur ) * K + j];
            if ( partition[v] == pl && flag[v] ) {
              flag[v] = 0;
              stack.push_back( v );
              cc.push_back( v );
            }
          }
        }
        if ( cc.size() >= size_t( sp->minPointCountPerCC ) ) ccs.push_back( std::move( cc ) );
      }
    }
    if ( ccs.empty() ) break;
    // ---- S8: one patch per component --------------------------------------------------------------
    for ( auto& cc : ccs ) {
      orc_patch P{};
      P.index          = int32_t( R->patches.size() );
      P.viewId         = int32_t( partition[cc[0]] );
      const int* ax    = AXES[P.viewId % 3];
      P.normalAxis     = ax[0];
      P.tangentAxis    = ax[1];
      P.bitangentAxis  = ax[2];
      P.projectionMode = P.viewId / 3;
      const int dir    = 1 - 2 * P.projectionMode;  // +1: keep nearest (min depth), -1: keep farthest
      if ( sp->enablePatchSplitting ) {
        int minU = 32767, minV = 32767;
        for ( uint32_t i : cc ) {
          minU = std::min<int>( minU, xyz[3 * size_t( i ) + ax[1]] );
          minV = std::min<int>( minV, xyz[3 * size_t( i ) + ax[2]] );
        }
        std::vector<uint32_t> kept;
        for ( uint32_t i : cc )
          if ( xyz[3 * size_t( i ) + ax[1]] - minU < sp->maxPatchSize && xyz[3 * size_t( i ) + ax[2]] - minV < sp->maxPatchSize )
            kept.push_back( i );
        cc.swap( kept );
      }
      int bbMin[3] = {1 << 30, 1 << 30, 1 << 30}, bbMax[3] = {0, 0, 0};
      for ( uint32_t i : cc )
        for ( int d = 0; d < 3; ++d ) {
          bbMin[d] = std::min<int>( bbMin[d], xyz[3 * size_t( i ) + d] );
          bbMax[d] = std::max<int>( bbMax[d], xyz[3 * size_t( i ) + d] );
        }
      P.sizeU = 1 + bbMax[ax[1]] - bbMin[ax[1]];
      P.sizeV = 1 + bbMax[ax[2]] - bbMin[ax[2]];
      P.u1    = bbMin[ax[1]];
      P.v1    = bbMin[ax[2]];
      const size_t         area = size_t( P.sizeU ) * P.sizeV;
      std::vector<int16_t> d0( area, INF_DEPTH ), d1;
      std::vector<int64_t> d0idx( area, -1 );
      int                  maxU = 0, maxV = 0, dmin = 1 << 30, dmax = -( 1 << 30 );
      for ( uint32_t i : cc ) {
        const int    d = xyz[3 * size_t( i ) + ax[0]];
        const int    u = xyz[3 * size_t( i ) + ax[1]] - P.u1;
        const int    v = xyz[3 * size_t( i ) + ax[2]] - P.v1;
        const size_t p = size_t( v ) * P.sizeU + u;
        const bool   better = ( P.projectionMode == 0 ) ? ( d0[p] > d ) : ( d0[p] == INF_DEPTH || d0[p] < d );
        if ( better ) {
          d0[p]    = int16_t( d );
          d0idx[p] = i;
        }
        maxU = std::max( maxU, u );
        maxV = std::max( maxV, v );
        dmin = std::min( dmin, d );
        dmax = std::max( dmax, d );
      }
      // depth origin, quantised to minLevel (running min/max in the reference == global min/max)
      const int L = sp->minLevel;
      P.d1        = ( P.projectionMode == 0 ) ? ( dmin / L ) * L : int( std::ceil( double( dmax ) / double( L ) ) ) * L;
      P.sizeU0    = maxU / occRes + 1;
      P.sizeV0    = maxV / occRes + 1;
      P.size2DXInPixel = maxU + 1;
      P.size2DYInPixel = maxV + 1;
      if ( sp->quantizerSizeX )
        P.size2DXInPixel = int( std::ceil( double( P.size2DXInPixel ) / double( sp->quantizerSizeX ) ) * sp->quantizerSizeX );
      if ( sp->quantizerSizeY )
        P.size2DYInPixel = int( std::ceil( double( P.size2DYInPixel ) / double( sp->quantizerSizeY ) ) * sp->quantizerSizeY );
      // per-block peak filter
      std::vector<int16_t> peak( size_t( P.sizeU0 ) * P.sizeV0, P.projectionMode == 0 ? INF_DEPTH : int16_t( 0 ) );
      for ( int v = 0; v < P.sizeV; ++v )
        for ( int u = 0; u < P.sizeU; ++u ) {
          const int16_t d = d0[size_t( v ) * P.sizeU + u];
          if ( d == INF_DEPTH ) continue;
          int16_t& pk = peak[size_t( v / occRes ) * P.sizeU0 + u / occRes];
          pk          = ( P.projectionMode == 0 ) ? std::min( pk, d ) : std::max( pk, d );
        }
      for ( int v = 0; v < P.sizeV; ++v )
        for ( int u = 0; u < P.sizeU; ++u ) {
          const size_t  p = size_t( v ) * P.sizeU + u;
          const int16_t d = d0[p];
          if ( d == INF_DEPTH ) continue;
          const int16_t a = int16_t( std::abs( d - peak[size_t( v / occRes ) * P.sizeU0 + u / occRes] ) );
          const int16_t b = int16_t( int16_t( sp->surfaceThickness ) + dir * d );
          const int16_t c = int16_t( dir * P.d1 + int16_t( sp->maxAllowedDepth ) );
          if ( a > 32 || b > c ) {
            d0[p]    = INF_DEPTH;
            d0idx[p] = -1;
          }
        }
      // D1: farthest same-pixel point within surfaceThickness of D0 and colour-similar to it
      d1 = d0;
      if ( sp->surfaceThickness > 0 ) {
        for ( uint32_t i : cc ) {
          const int     d = xyz[3 * size_t( i ) + ax[0]];
          const size_t  p = size_t( xyz[3 * size_t( i ) + ax[2]] - P.v1 ) * P.sizeU + ( xyz[3 * size_t( i ) + ax[1]] - P.u1 );
          const int16_t z0 = d0[p];
          if ( !( z0 < INF_DEPTH ) ) continue;
          const int16_t  delta = int16_t( dir * ( d - z0 ) );
          const uint8_t* ci    = rgb + 3 * size_t( i );
          const uint8_t* c0    = rgb + 3 * size_t( d0idx[p] );
          const bool similar = std::abs( c0[0] - ci[0] ) < 128 && std::abs( c0[1] - ci[1] ) < 128 && std::abs( c0[2] - ci[2] ) < 128;
          if ( delta <= int16_t( sp->surfaceThickness ) && delta >= 0 && similar )
            if ( dir * ( d - d1[p] ) > 0 ) d1[p] = int16_t( d );
        }
      }
      // resample: raster order, D0 point then D1 point (always both), block occupancy, local depths
      std::vector<uint8_t> occ( size_t( P.sizeU0 ) * P.sizeV0, 0 );
      int                  sizeD = 0, d0Count = 0, d1Count = 0;
      for ( int v = 0; v < P.sizeV; ++v )
        for ( int u = 0; u < P.sizeU; ++u ) {
          const size_t p = size_t( v ) * P.sizeU + u;
          if ( !( d0[p] < INF_DEPTH ) ) continue;
          occ[size_t( v / occRes ) * P.sizeU0 + u / occRes] = 1;
          int16_t pt[3];
          pt[ax[0]] = d0[p];
          pt[ax[1]] = int16_t( u + P.u1 );
          pt[ax[2]] = int16_t( v + P.v1 );
          R->resampled.insert( R->resampled.end(), pt, pt + 3 );
          ++d0Count;
          pt[ax[0]] = d1[p];
          if ( d0[p] != d1[p] ) ++d1Count;
          R->resampled.insert( R->resampled.end(), pt, pt + 3 );
          d0[p] = int16_t( dir * ( d0[p] - int16_t( P.d1 ) ) );
          sizeD = std::max( sizeD, int( d0[p] ) );
          d1[p] = int16_t( dir * ( d1[p] - int16_t( P.d1 ) ) );
          sizeD = std::max( sizeD, int( d1[p] ) );
        }
      P.sizeDPixel = sizeD;
      {
        const int bits  = std::min( sp->geometryBitDepth3D, sp->geometryBitDepth2D );
        int       sd    = std::min( ( 1 << bits ) - 1, sizeD );
        const int bitsD = bits - int( std::log2( L ) );
        const int maxDD = 1 << bitsD;
        int       q     = sd == 0 ? 0 : ( ( sd - 1 ) / L + 1 );
        q               = std::min( q, maxDD - 1 );
        P.sizeD         = q == 0 ? 0 : ( q * L - 1 );
      }
      P.d0Count       = d0Count;
      P.eomAndD1Count = 0;
      (void)d1Count;
      P.depthOffset = int64_t( R->depth0.size() );
      P.occOffset   = int64_t( R->occupancy.size() );
      R->depth0.insert( R->depth0.end(), d0.begin(), d0.end() );
      R->depth1.insert( R->depth1.end(), d1.begin(), d1.end() );
      R->occupancy.insert( R->occupancy.end(), occ.begin(), occ.end() );
      R->patches.push_back( P );
    }
    // ---- S9: distance of every input point to the resampled cloud --------------------------------
    const size_t          m = R->resampled.size() / 3;
    orc_kdtree*           t = orc_kdtree_build( R->resampled.data(), m );
    std::vector<uint32_t> nn( n );
    std::vector<double>   dist( n );
    orc_knn( t, xyz, n, 1, nn.data(), dist.data() );
    orc_kdtree_free( t );
    raw.clear();
    for ( size_t i = 0; i < n; ++i ) {
      rawDist[i] = dist[i];
      if ( dist[i] > sp->maxAllowedDist2RawPointsSelection ) raw.push_back( uint32_t( i ) );
    }
    R->roundRaw.push_back( int32_t( raw.size() ) );
  }
  R->rawPoints = raw;
  return R;
}

void orc_seg_result_free( orc_seg_result* r ) { delete r; }
int  orc_seg_result_sizes( const orc_seg_result* r, int32_t* patches, int64_t* depthCount, int64_t* occCount,
                           int64_t* resampledCount, int32_t* rounds ) {
  *patches        = int32_t( r->patches.size() );
  *depthCount     = int64_t( r->depth0.size() );
  *occCount       = int64_t( r->occupancy.size() );
  *resampledCount = int64_t( r->resampled.size() / 3 );
  *rounds         = int32_t( r->roundRaw.size() );
  return 0;
}
int orc_seg_result_copy( const orc_seg_result* r, orc_patch* patches, int16_t* depth0, int16_t* depth1, uint8_t* occ,
                         int16_t* resampled, int32_t* roundRaw ) {
  std::copy( r->patches.begin(), r->patches.end(), patches );
  std::copy( r->depth0.begin(), r->depth0.end(), depth0 );
  std::copy( r->depth1.begin(), r->depth1.end(), depth1 );
  std::copy( r->occupancy.begin(), r->occupancy.end(), occ );
  if ( resampled ) std::copy( r->resampled.begin(), r->resampled.end(), resampled );
  if ( roundRaw ) std::copy( r->roundRaw.begin(), r->roundRaw.end(), roundRaw );
  return 0;
}

// whole segmenter: S1..S9
orc_seg_result* orc_segment( const int16_t* xyz, const uint8_t* rgb, size_t n, const orc_seg_params* sp ) {
  const int             K = sp->nnNormalEstimation;
  orc_kdtree*           t = orc_kdtree_build( xyz, n );
  std::vector<uint32_t> knn( n * size_t( K ) );
  orc_knn( t, xyz, n, K, knn.data(), nullptr );
  std::vector<double> normals( 3 * n );
  orc_compute_normals( xyz, n, knn.data(), K, normals.data() );
  if ( sp->normalOrientation == 1 ) orc_orient_normals( xyz, n, knn.data(), K, normals.data() );
  std::vector<uint32_t> part( n );
  orc_initial_segmentation( normals.data(), n, sp->weightNormal, part.data() );
  orc_refine_grid( xyz, normals.data(), n, part.data(), sp->maxNNCountRefineSegmentation, sp->lambdaRefineSegmentation,
                   sp->iterationCountRefineSegmentation, sp->voxelDimensionRefineSegmentation,
                   sp->searchRadiusRefineSegmentation );
  std::vector<uint32_t> adj;
  const uint32_t*       adjp = knn.data();
  int                   KA   = K;
  if ( sp->maxNNCountPatchSegmentation != K ) {
    KA = sp->maxNNCountPatchSegmentation;
    adj.resize( n * size_t( KA ) );
    orc_knn( t, xyz, n, KA, adj.data(), nullptr );
    adjp = adj.data();
  }
  orc_kdtree_free( t );
  return orc_segment_patches( xyz, rgb, n, adjp, KA, part.data(), sp );
}

}  // extern "C"
