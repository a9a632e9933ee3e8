// oracle/port_metrics.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S23  D1 (point-to-point), D2 (point-to-plane) and colour distortion between a source and a reconstructed cloud:
//      PCCMetrics::compute (PccLibMetrics/source/PCCMetrics.cpp:324-375), QualityMetrics::compute (:73-229),
//      QualityMetrics::operator+ (:289-322), getPSNR (:42-46), convertRGBtoYUVBT709 (:48-53);
//      PCCPointSet3::removeDuplicate (PccLibCommon/source/PCCPointSet.cpp:169-220, dropDuplicates = 2),
//      copyNormals (:2282-2320), scaleNormals (:2322-2380).   Defaults of PCCMetricsParameters.cpp:39-57.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "oracle.h"

struct orc_kdtree;
extern "C" {
orc_kdtree* orc_kdtree_build( const int16_t* xyz, size_t n );
void        orc_kdtree_free( orc_kdtree* t );
int         orc_knn( const orc_kdtree* t, const int16_t* q, size_t nq, int k, uint32_t* idx, double* dist );
}

namespace {

struct Cloud {
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  std::vector<double>  nrm;  // optional
  size_t               size() const { return xyz.size() / 3; }
};

// lexicographic (x,y,z) order, one point per position; colour = integer mean of the duplicates
Cloud dedup( const int16_t* xyz, const uint8_t* rgb, size_t n ) {
  std::vector<uint32_t> order( n );
  std::iota( order.begin(), order.end(), 0u );
  std::stable_sort( order.begin(), order.end(), [&]( uint32_t a, uint32_t b ) {
    for ( int d = 0; d < 3; ++d )
      if ( xyz[3 * size_t( a ) + d] != xyz[3 * size_t( b ) + d] ) return xyz[3 * size_t( a ) + d] < xyz[3 * size_t( b ) + d];
    return false;
  } );
  Cloud c;
  for ( size_t i = 0; i < n; ) {
    size_t j = i + 1;
    while ( j < n && xyz[3 * size_t( order[j] )] == xyz[3 * size_t( order[i] )] &&
            xyz[3 * size_t( order[j] ) + 1] == xyz[3 * size_t( order[i] ) + 1] &&
            xyz[3 * size_t( order[j] ) + 2] == xyz[3 * size_t( order[i] ) + 2] )
      ++j;
    for ( int d = 0; d < 3; ++d ) c.xyz.push_back( xyz[3 * size_t( order[i] ) + d] );
    size_t s[3] = {0, 0, 0};
    for ( size_t k = i; k < j; ++k )
      for ( int d = 0; d < 3; ++d ) s[d] += rgb[3 * size_t( order[k] ) + d];
    for ( int d = 0; d < 3; ++d ) c.rgb.push_back( uint8_t( s[d] / ( j - i ) ) );
    i = j;
  }
  return c;
}

// nearest group of every query: kNN with k = 5, 10, ... 30 until the k-th result is farther than the first
struct Nearest {
  std::vector<uint32_t> off, idx;  // CSR of the equal-distance group, in RESULT order
  std::vector<double>   d0;
};
Nearest nearestGroups( const orc_kdtree* t, const int16_t* q, size_t nq ) {
  Nearest  r;
  uint32_t id[30];
  double   d[30];
  r.off.push_back( 0 );
  for ( size_t i = 0; i < nq; ++i ) {
    int k = 0;
    do {
      k += 5;
      orc_knn( t, q + 3 * i, 1, k, id, d );
    } while ( d[0] == d[k - 1] && k + 5 <= 30 );
    for ( int j = 0; j < k && std::fabs( d[0] - d[j] ) < 1e-8; ++j ) r.idx.push_back( id[j] );
    r.off.push_back( uint32_t( r.idx.size() ) );
    r.d0.push_back( d[0] );
  }
  return r;
}

void yuv709( const uint8_t* c, float* yuv ) {
  yuv[0] = float( ( 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2] ) / 255.0 );
  yuv[1] = float( ( -0.1146 * c[0] - 0.3854 * c[1] + 0.5000 * c[2] ) / 255.0 + 0.5000 );
  yuv[2] = float( ( 0.5000 * c[0] - 0.4542 * c[1] - 0.0458 * c[2] ) / 255.0 + 0.5000 );
}

double psnr( double dist, double p, double factor ) { return 10 * std::log10( ( factor * p * p ) / dist ); }

// out[8]: c2cMse, c2cPsnr, c2pMse, c2pPsnr, colorMse[3], colorPsnr[0]
void quality( const Cloud& A, const Cloud& B, bool withNormals, double resolution, double* out ) {
  orc_kdtree*   t = orc_kdtree_build( B.xyz.data(), B.size() );
  const Nearest g = nearestGroups( t, A.xyz.data(), A.size() );
  orc_kdtree_free( t );
  double sseC2c = 0, sseC2p = 0, sseCol[3] = {0, 0, 0};
  for ( size_t a = 0; a < A.size(); ++a ) {
    std::vector<uint32_t> same( g.idx.begin() + g.off[a], g.idx.begin() + g.off[a + 1] );
    std::sort( same.begin(), same.end() );
    double c2p = 0.0;
    if ( withNormals ) {
      for ( uint32_t b : same ) {
        double e[3];
        for ( int d = 0; d < 3; ++d ) e[d] = A.xyz[3 * a + d] - B.xyz[3 * size_t( b ) + d];
        const double dp = e[0] * B.nrm[3 * size_t( b )] + e[1] * B.nrm[3 * size_t( b ) + 1] + e[2] * B.nrm[3 * size_t( b ) + 2];
        c2p += dp * dp;
      }
      c2p /= double( same.size() );
    }
    float         ya[3], yb[3];
    unsigned      r = 0, gg = 0, bb = 0;
    yuv709( &A.rgb[3 * a], ya );
    for ( uint32_t b : same ) {
      r += B.rgb[3 * size_t( b )];
      gg += B.rgb[3 * size_t( b ) + 1];
      bb += B.rgb[3 * size_t( b ) + 2];
    }
    const int     cnt    = int( same.size() );
    const uint8_t avg[3] = {(unsigned char)std::round( double( r ) / cnt ), (unsigned char)std::round( double( gg ) / cnt ),
                            (unsigned char)std::round( double( bb ) / cnt )};
    yuv709( avg, yb );
    sseC2c += g.d0[a];
    sseC2p += c2p;
    for ( int i = 0; i < 3; ++i ) {
      const float df = ya[i] - yb[i];
      sseCol[i] += double( float( df * df ) );
    }
  }
  const double num = double( A.size() );
  out[0]           = sseC2c / num;
  out[1]           = psnr( out[0], resolution, 3 );
  out[2]           = withNormals ? sseC2p / num : 0.0;
  out[3]           = withNormals ? psnr( out[2], resolution, 3 ) : 0.0;
  for ( int i = 0; i < 3; ++i ) out[4 + i] = sseCol[i] / num;
  out[7] = psnr( out[4], 1.0, 1.0 );
}

}  // namespace

extern "C" int orc_metrics( const int16_t* srcXyz, const uint8_t* srcRgb, size_t n, const int16_t* recXyz,
                            const uint8_t* recRgb, size_t m, const double* srcNormals, double resolution, double* out,
                            int64_t* counts ) {
  Cloud S = dedup( srcXyz, srcRgb, n ), R = dedup( recXyz, recRgb, m );
  counts[0] = int64_t( S.size() );
  counts[1] = int64_t( R.size() );
  const bool withNormals = srcNormals != nullptr;
  if ( withNormals ) {
    if ( S.size() != n ) return -1;  // the reference exits: normal cloud and deduplicated source must match 1:1
    // copyNormals: by position
    std::vector<uint32_t> order( n );
    std::iota( order.begin(), order.end(), 0u );
    std::sort( order.begin(), order.end(), [&]( uint32_t a, uint32_t b ) {
      for ( int d = 0; d < 3; ++d )
        if ( srcXyz[3 * size_t( a ) + d] != srcXyz[3 * size_t( b ) + d] ) return srcXyz[3 * size_t( a ) + d] < srcXyz[3 * size_t( b ) + d];
      return false;
    } );
    S.nrm.resize( 3 * n );
    for ( size_t i = 0; i < n; ++i )
      for ( int d = 0; d < 3; ++d ) S.nrm[3 * i + d] = srcNormals[3 * size_t( order[i] ) + d];
    // scaleNormals: every source point (original order) adds its normal to its nearest reconstructed points
    R.nrm.assign( 3 * R.size(), 0.0 );
    std::vector<size_t> cnt( R.size(), 0 );
    orc_kdtree*         tR = orc_kdtree_build( R.xyz.data(), R.size() );
    {
      uint32_t id[30];
      double   d[30];
      for ( size_t i = 0; i < n; ++i ) {
        int k = 0;
        do {
          k += 5;
          orc_knn( tR, srcXyz + 3 * i, 1, k, id, d );
        } while ( d[0] == d[k - 1] && k + 5 <= 30 );
        for ( int j = 0; j < k; ++j )
          if ( d[0] == d[j] ) {
            for ( int c = 0; c < 3; ++c ) R.nrm[3 * size_t( id[j] ) + c] += srcNormals[3 * i + c];
            cnt[id[j]]++;
          }
      }
    }
    orc_kdtree_free( tR );
    orc_kdtree* tS = orc_kdtree_build( srcXyz, n );
    for ( size_t i = 0; i < R.size(); ++i ) {
      if ( cnt[i] > 0 ) {
        for ( int c = 0; c < 3; ++c ) R.nrm[3 * i + c] /= double( cnt[i] );
      } else {
        uint32_t id[30];
        double   d[30];
        int      k = 0;
        do {
          k += 5;
          orc_knn( tS, &R.xyz[3 * i], 1, k, id, d );
        } while ( d[0] == d[k - 1] && k + 5 <= 30 );
        size_t num = 0;
        for ( int j = 0; j < k; ++j )
          if ( d[0] == d[j] ) {
            for ( int c = 0; c < 3; ++c ) R.nrm[3 * i + c] += srcNormals[3 * size_t( id[j] ) + c];
            num++;
          }
        for ( int c = 0; c < 3; ++c ) R.nrm[3 * i + c] /= double( num );
      }
    }
    orc_kdtree_free( tS );
  }
  quality( S, R, withNormals, resolution, out );
  quality( R, S, withNormals, resolution, out + 8 );
  for ( int i = 0; i < 8; ++i ) {  // symmetric result: max of the MSEs, min of the PSNRs
    const bool isPsnr = ( i == 1 || i == 3 || i == 7 );
    out[16 + i]       = isPsnr ? std::min( out[i], out[8 + i] ) : std::max( out[i], out[8 + i] );
  }
  return 0;
}
