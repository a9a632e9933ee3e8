// oracle/port_post.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// The post-reconstruction tail that PCCEncoder::encode runs after the attribute video
// (PccLibEncoder/source/PCCEncoder.cpp:571-719) and PCCDecoder::decode runs for every frame (PccLibDecoder/source/
// PCCDecoder.cpp:330-470), under the CTC settings (flagGeometrySmoothing 1, gridSmoothing 1, gridSize 8,
// thresholdSmoothing 64, attrTransferFilterType 1, flagColorSmoothing 0, two maps in one stream, lossy attributes):
//   T1  PCCCodec::identifyBoundaryPoints           (PccLibCommon/source/PCCCodec.cpp:268-327, called from :955-976)
//   T2  PCCCodec::colorPointCloud                   (:1319-1460, the "f < mapCount" branch)
//   T3  PCCCodec::smoothPointCloudPostprocess       (:54-148) + addGridCentroid (:982-1000) + gridFiltering (:1002-1065)
//       + smoothPointCloudGrid (:1067-1106)
//   T4  PCCPointSet3::transferColors16bitBP         (PccLibCommon/source/PCCPointSet.cpp:1126-1470) with filterType 1 and
//       the arguments of PCCEncoder.cpp:657-672
//   T5  PCCPointSet3::convertYUV16ToRGB8            (PccLibCommon/include/PCCPointSet.h:133-166)
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
}

extern "C" {

// T1.  occupancy = the occupancy video upsampled by occPrecision (what generatePointCloud leaves in the tile's
// occupancy map).  pointToPixel u32[M][3] = (x, y, layer).  btype u16[M] out (0 or 1).
int orc_identify_boundary_points( const uint32_t* pointToPixel, int64_t M, const uint8_t* occVideo, int W, int H, int occPrecision,
                                  uint16_t* btype ) {
  const int Wv  = W / occPrecision;
  auto      occ = [&]( size_t x, size_t y ) { return occVideo[( y / occPrecision ) * Wv + x / occPrecision] != 0; };
  for ( int64_t i = 0; i < M; ++i ) {
    const size_t x = pointToPixel[3 * i], y = pointToPixel[3 * i + 1];
    const size_t w = size_t( W ), h = size_t( H );
    uint16_t     t = 0;
    if ( !occ( x, y ) ) {
      btype[i] = 0;
      continue;
    }
    if ( y > 0 && y < h - 1 && ( !occ( x, y - 1 ) || !occ( x, y + 1 ) ) ) t = 1;
    if ( x > 0 && x < w - 1 && t != 1 && ( !occ( x + 1, y ) || !occ( x - 1, y ) ) ) t = 1;
    if ( y > 0 && y < h - 1 && x > 0 && t != 1 && ( !occ( x - 1, y - 1 ) || !occ( x - 1, y + 1 ) ) ) t = 1;
    if ( y > 0 && y < h - 1 && x < w - 1 && t != 1 && ( !occ( x + 1, y - 1 ) || !occ( x + 1, y + 1 ) ) ) t = 1;
    if ( y == 0 || y == h - 1 || x == 0 || x == w - 1 ) t = 1;
    if ( t != 1 ) {  // second layer: the ring at distance two
      for ( int ix = -2; ix <= 2; ++ix )
        for ( int iy = -2; iy <= 2; ++iy )
          if ( std::abs( ix ) > 1 || std::abs( iy ) > 1 ) {
            const size_t yy = y + size_t( int64_t( iy ) ), xx = x + size_t( int64_t( ix ) );  // unsigned wrap = out of range
            if ( yy < h && xx < w && !occ( xx, yy ) ) {
              t  = 1;
              ix = 4;
              iy = 4;
            }
          }
      if ( y == 1 || y == h - 2 || x == 1 || x == w - 2 ) t = 1;
    }
    btype[i] = t;
  }
  return 0;
}

// T2.  attribute: the decoded attribute frames of this point-cloud frame, u16 [2 maps][3 channels][H][W].
int orc_color_point_cloud( const uint32_t* pointToPixel, int64_t M, const uint16_t* attribute, int W, int H, uint16_t* colors16 ) {
  const size_t plane = size_t( W ) * H;
  for ( int64_t i = 0; i < M; ++i ) {
    const size_t x = pointToPixel[3 * i], y = pointToPixel[3 * i + 1], f = pointToPixel[3 * i + 2];
    for ( int c = 0; c < 3; ++c ) colors16[3 * i + c] = attribute[( f * 3 + size_t( c ) ) * plane + y * W + x];
  }
  return 0;
}

// T3.  xyz in/out, btype in/out (moved points become 3); partition = patch index (list position) of every point.
int orc_smooth_point_cloud_grid( int16_t* xyz, uint16_t* btype, const uint32_t* partition, int64_t M, int gridSize,
                                 double thresholdSmoothing ) {
  if ( M == 0 ) return 0;
  int maxSize = 0;
  {
    int mx[3] = {xyz[0], xyz[1], xyz[2]};
    for ( int64_t j = 0; j < M; ++j )
      for ( int k = 0; k < 3; ++k ) mx[k] = std::max<int>( mx[k], xyz[3 * j + k] );
    maxSize = std::max( std::max( mx[0], mx[1] ), mx[2] );
  }
  const int        w = ( maxSize + gridSize - 1 ) / gridSize;
  std::vector<int> cellIndex( size_t( w ) * w * w, -1 );
  int              cells = 0;
  const int        disth = std::max( gridSize / 2, 1 ), th = gridSize * w;
  auto             outside = [&]( const int* P ) {
    return P[0] < disth || P[1] < disth || P[2] < disth || th <= P[0] + disth || th <= P[1] + disth || th <= P[2] + disth;
  };
  for ( int64_t n = 0; n < M; ++n ) {
    if ( btype[n] != 1 ) continue;
    const int P[3] = {xyz[3 * n], xyz[3 * n + 1], xyz[3 * n + 2]};
    if ( outside( P ) ) continue;
    int Q[3];
    for ( int k = 0; k < 3; ++k ) Q[k] = P[k] / gridSize + ( ( P[k] % gridSize < gridSize / 2 ) ? -1 : 0 );
    for ( int ix = 0; ix < 2; ++ix )
      for ( int iy = 0; iy < 2; ++iy )
        for ( int iz = 0; iz < 2; ++iz ) {
          const int cellId = ( Q[0] + ix ) + ( Q[1] + iy ) * w + ( Q[2] + iz ) * w * w;
          if ( cellIndex[size_t( cellId )] == -1 ) cellIndex[size_t( cellId )] = cells++;
        }
  }
  struct F3 {
    float v[3];
  };
  std::vector<F3>       center( size_t( cells ), F3{{0.f, 0.f, 0.f}} );
  std::vector<uint16_t> count( size_t( cells ), 0 );
  std::vector<uint32_t> owner( size_t( cells ), 0 );
  std::vector<uint8_t>  doSmooth( size_t( cells ), 0 );
  for ( int64_t j = 0; j < M; ++j ) {
    const int P[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
    if ( outside( P ) ) continue;
    const int cellId = P[0] / gridSize + ( P[1] / gridSize ) * w + ( P[2] / gridSize ) * w * w;
    const int cell   = cellIndex[size_t( cellId )];
    if ( cell == -1 ) continue;
    const uint32_t patch = partition[j] + 1;
    if ( count[size_t( cell )] == 0 ) {
      owner[size_t( cell )]    = patch;
      center[size_t( cell )]   = F3{{0.f, 0.f, 0.f}};
      doSmooth[size_t( cell )] = 0;
    } else if ( !doSmooth[size_t( cell )] && owner[size_t( cell )] != patch ) {
      doSmooth[size_t( cell )] = 1;
    }
    for ( int k = 0; k < 3; ++k ) center[size_t( cell )].v[k] += float( P[k] );
    count[size_t( cell )]++;
  }
  for ( int i = 0; i < cells; ++i )
    if ( count[size_t( i )] != 0 )
      for ( int k = 0; k < 3; ++k ) center[size_t( i )].v[k] /= float( count[size_t( i )] );
  // smoothPointCloudGrid
  const int half = gridSize / 2, gridSize2 = gridSize * 2, norm = gridSize2 * gridSize2 * gridSize2;
  for ( int64_t c = 0; c < M; ++c ) {
    const int P[3] = {xyz[3 * c], xyz[3 * c + 1], xyz[3 * c + 2]};
    if ( outside( P ) ) continue;
    if ( btype[c] != 1 ) continue;
    int S[3];
    for ( int k = 0; k < 3; ++k ) {
      const int P2 = P[k] / gridSize, P3 = P[k] - P2 * gridSize;
      S[k]         = P2 + ( ( P3 < half ) ? -1 : 0 );
    }
    int  idx[2][2][2];
    bool other = false;
    for ( int dz = 0; dz < 2; ++dz )
      for ( int dy = 0; dy < 2; ++dy )
        for ( int dx = 0; dx < 2; ++dx ) {
          const int tmp   = ( S[0] + dx ) + ( S[1] + dy ) * w + ( S[2] + dz ) * w * w;
          idx[dz][dy][dx] = tmp;
          const int cell  = cellIndex[size_t( tmp )];
          if ( doSmooth[size_t( cell )] && count[size_t( cell )] != 0 ) other = true;
        }
    if ( !other ) continue;
    const double cur[3] = {double( P[0] ), double( P[1] ), double( P[2] )};
    double       centroid3[2][2][2][3];
    for ( int dz = 0; dz < 2; ++dz )
      for ( int dy = 0; dy < 2; ++dy )
        for ( int dx = 0; dx < 2; ++dx ) {
          const int cell = cellIndex[size_t( idx[dz][dy][dx] )];
          for ( int k = 0; k < 3; ++k )
            centroid3[dz][dy][dx][k] = count[size_t( cell )] > 0 ? double( center[size_t( cell )].v[k] ) : cur[k];
        }
    int Wt[3], Q[3];
    for ( int k = 0; k < 3; ++k ) {
      Wt[k] = ( P[k] - S[k] * gridSize - half ) * 2 + 1;
      Q[k]  = gridSize2 - Wt[k];
    }
    int    cnt         = 0;
    double centroid4[3] = {0.0, 0.0, 0.0};
    for ( int dz = 0, cz = Q[2]; dz < 2; ++dz, cz = Wt[2] )
      for ( int dy = 0, b = Q[1]; dy < 2; ++dy, b = Wt[1] )
        for ( int dx = 0, a = Q[0]; dx < 2; ++dx, a = Wt[0] ) {
          const int wgt = a * b * cz;
          for ( int k = 0; k < 3; ++k ) {
            centroid3[dz][dy][dx][k] *= double( wgt );
            centroid4[k] += centroid3[dz][dy][dx][k];
          }
          cnt += wgt * int( count[size_t( cellIndex[size_t( idx[dz][dy][dx] )] )] );
        }
    for ( int k = 0; k < 3; ++k ) centroid4[k] /= double( norm );
    cnt /= norm;
    double centroid[3];
    for ( int k = 0; k < 3; ++k ) centroid[k] = centroid4[k] * double( cnt );
    double d[3];
    for ( int k = 0; k < 3; ++k ) d[k] = cur[k] * double( cnt ) - centroid[k];
    const double dist2 = ( d[0] * d[0] + d[1] * d[1] + d[2] * d[2] ) / double( cnt ) + 0.5;
    if ( dist2 >= double( std::max( int( thresholdSmoothing ), cnt ) * 2 ) ) {
      for ( int k = 0; k < 3; ++k ) xyz[3 * c + k] = int16_t( double( int64_t( centroid[k] / double( cnt ) + 0.5 ) ) );
      btype[c] = 3;
    }
  }
  return 0;
}

// T4.  source = the cloud before smoothing (positions + 16-bit colours), target = the smoothed cloud whose 16-bit
// colours (tgtColors16, in/out) still are the source's; only points of boundary type 3 are recoloured.
int orc_transfer_colors16_bp( const int16_t* srcXyz, const uint16_t* srcColors16, const int16_t* tgtXyz, const uint16_t* tgtBtype,
                              int64_t M, uint16_t* tgtColors16 ) {
  if ( M == 0 ) return 0;
  orc_kdtree* tgtTree = orc_kdtree_build( tgtXyz, size_t( M ) );
  orc_kdtree* srcTree = orc_kdtree_build( srcXyz, size_t( M ) );
  std::vector<uint16_t> refined( tgtColors16, tgtColors16 + 3 * M );
  std::vector<uint32_t> part;  // the forward neighbours of every moved point, in order: indices into the source
  uint32_t              idx[8];
  double                dist[8];
  for ( int64_t t = 0; t < M; ++t ) {
    if ( tgtBtype[t] != 3 ) continue;
    orc_knn( srcTree, tgtXyz + 3 * t, 1, 8, idx, dist );
    const int found = int( std::min<int64_t>( 8, M ) );
    for ( int i = 0; i < found; ++i ) part.push_back( idx[i] );
    if ( dist[0] < 0.0001 || found == 1 ) {
      for ( int k = 0; k < 3; ++k ) refined[3 * t + k] = srcColors16[3 * size_t( idx[0] ) + k];
      continue;
    }
    // (the colour-spread test is disabled: 1000 * 256 >= 131072 turns the bound into DBL_MAX)
    double c[3] = {0.0, 0.0, 0.0}, sumW = 0.0;
    for ( int i = 0; i < found; ++i ) {
      const double w = 1 / ( dist[i] + 4.0 );
      for ( int k = 0; k < 3; ++k ) c[k] += srcColors16[3 * size_t( idx[i] ) + k] * w;
      sumW += w;
    }
    for ( int k = 0; k < 3; ++k ) refined[3 * t + k] = uint16_t( std::max( 0.0, std::min( std::round( c[k] / sumW ), 65535.0 ) ) );
  }
  // backward: every collected neighbour votes for its nearest target if their colours are close
  struct Cand {
    double   d;
    uint16_t c[3];
  };
  std::vector<std::vector<Cand>> cand;
  cand.resize( size_t( M ) );
  for ( size_t i = 0; i < part.size(); ++i ) {
    const uint16_t* color = srcColors16 + 3 * size_t( part[i] );
    uint32_t        t;
    double          d;
    orc_knn( tgtTree, srcXyz + 3 * size_t( part[i] ), 1, 1, &t, &d );
    const uint16_t* tc = tgtColors16 + 3 * size_t( t );
    if ( std::abs( int( color[0] ) - int( tc[0] ) ) < 40 && std::abs( int( color[1] ) - int( tc[1] ) ) < 40 &&
         std::abs( int( color[2] ) - int( tc[2] ) ) < 40 )
      cand[t].push_back( Cand{d, {color[0], color[1], color[2]}} );
  }
  orc_kdtree_free( srcTree );
  orc_kdtree_free( tgtTree );
  std::vector<uint16_t> out( tgtColors16, tgtColors16 + 3 * M );
  for ( int64_t t = 0; t < M; ++t ) {
    if ( tgtBtype[t] != 3 ) continue;
    auto& L = cand[size_t( t )];
    // the reference's std::sort by distance only (libstdc++ introsort; the oracle is built with the same library)
    std::sort( L.begin(), L.end(), []( const Cand& a, const Cand& b ) { return a.d < b.d; } );
    if ( L.empty() ) {
      for ( int k = 0; k < 3; ++k ) out[3 * t + k] = refined[3 * t + k];
      continue;
    }
    double c2[3] = {0.0, 0.0, 0.0};
    if ( L.size() == 1 ) {
      for ( int k = 0; k < 3; ++k ) c2[k] = L[0].c[k];
    } else {
      double sumW = 0.0;
      for ( auto& e : L ) {
        const double w = 1 / ( std::sqrt( e.d ) + 4.0 );
        for ( int k = 0; k < 3; ++k ) c2[k] += e.c[k] * w;
        sumW += w;
      }
      for ( int k = 0; k < 3; ++k ) c2[k] /= sumW;
    }
    // fixWeight: w = 0; searchRange 0: the single candidate colour is taken
    for ( int k = 0; k < 3; ++k ) {
      const double color0 = std::max( 0.0, std::min( std::round( 0.0 * double( refined[3 * t + k] ) + 1.0 * c2[k] ), 65535.0 ) );
      out[3 * t + k]      = uint16_t( std::max( 0.0, std::min( color0 + 0, 65535.0 ) ) );
    }
  }
  std::copy( out.begin(), out.end(), tgtColors16 );
  return 0;
}

// T5.
int orc_convert_yuv16_to_rgb8( const uint16_t* colors16, int64_t M, uint8_t* rgb ) {
  for ( int64_t k = 0; k < M; ++k ) {
    double       y1 = colors16[3 * k], u1 = colors16[3 * k + 1], v1 = colors16[3 * k + 2];
    const double offset = 32768.0, scale = 65535.0, weight = 1.0 / scale;
    y1 = weight * y1;
    u1 = weight * ( u1 - offset );
    v1 = weight * ( v1 - offset );
    y1 = std::min( std::max( y1, 0.0 ), 1.0 );
    u1 = std::min( std::max( u1, -0.5 ), 0.5 );
    v1 = std::min( std::max( v1, -0.5 ), 0.5 );
    double r = y1 + 1.57480 * v1;
    double g = y1 - 0.18733 * u1 - 0.46813 * v1;
    double b = y1 + 1.85563 * u1;
    r = std::max( 0.0, std::min( std::round( r * 255 ), 255.0 ) );
    g = std::max( 0.0, std::min( std::round( g * 255 ), 255.0 ) );
    b = std::max( 0.0, std::min( std::round( b * 255 ), 255.0 ) );
    rgb[3 * k] = uint8_t( r ), rgb[3 * k + 1] = uint8_t( g ), rgb[3 * k + 2] = uint8_t( b );
  }
  return 0;
}
}
