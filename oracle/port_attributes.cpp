// oracle/port_attributes.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S17  point-cloud reconstruction from the (decoded) geometry images
//      PCCCodec::generatePointCloud, lossy CTC branch (PccLibCommon/source/PCCCodec.cpp:519-980; generatePoints
//      :329-517 absolute-D1 branch :499-515; PCCPatch::generatePoint PCCPatch.h:177-207)
// S18  colour transfer source -> reconstruction: PCCPointSet3::transferColors (PCCPointSet.cpp:807-1124) with the
//      CTC settings (fwd k=8, bwd k=1, distance-weighted, skip-if-identical, offsets 4, thresholds off, searchRange 0)
// S19  colour pre-smoothing: a no-op in the reference build (boundary type 2 is never assigned)
// S20  attribute scatter: PCCEncoder::generateAttributeVideo (PCCEncoder.cpp:6736-6794)
// S21  push-pull background fill: dilateSmoothedPushPull / pushPullMip / pushPullFill / mean4w (:6357-6591)
// S22  attribute group dilation (inline in PCCEncoder::encode :380-402)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

struct orc_kdtree;
extern "C" {
orc_kdtree* orc_kdtree_build( const int16_t* xyz, size_t n );
void        orc_kdtree_free( orc_kdtree* t );
int         orc_knn( const orc_kdtree* t, const int16_t* q, size_t nq, int k, uint32_t* idx, double* dist );
}

namespace {
inline double clip255( double v ) { return std::max( 0.0, std::min( v, 255.0 ) ); }

struct Plane3 {  // three u8 planes of one image
  int                  w = 0, h = 0;
  std::vector<uint8_t> c[3];
  void                 resize( int W, int H ) {
    w = W, h = H;
    for ( auto& p : c ) p.assign( size_t( W ) * H, 0 );
  }
};

int mean4w( int p1, int w1, int p2, int w2, int p3, int w3, int p4, int w4 ) {
  return ( p1 * w1 + p2 * w2 + p3 * w3 + p4 * w4 ) / ( w1 + w2 + w3 + w4 );
}

void mip( const Plane3& img, const std::vector<uint8_t>& occ, Plane3& out, std::vector<uint8_t>& occOut ) {
  const int W = img.w, H = img.h, w = ( W + 1 ) >> 1, h = ( H + 1 ) >> 1;
  out.resize( w, h );
  occOut.assign( size_t( w ) * h, 0 );
  for ( int y = 0; y < h; ++y )
    for ( int x = 0; x < w; ++x ) {
      const int  X = 2 * x, Y = 2 * y;
      const bool i2 = X + 1 < W, i3 = Y + 1 < H;
      const int  w1 = occ[size_t( Y ) * W + X] ? 255 : 0;
      const int  w2 = ( i2 && occ[size_t( Y ) * W + X + 1] ) ? 255 : 0;
      const int  w3 = ( i3 && occ[size_t( Y + 1 ) * W + X] ) ? 255 : 0;
      const int  w4 = ( i2 && i3 && occ[size_t( Y + 1 ) * W + X + 1] ) ? 255 : 0;
      if ( w1 + w2 + w3 + w4 == 0 ) continue;
      for ( int k = 0; k < 3; ++k ) {
        const auto& p = img.c[k];
        const int   v1 = p[size_t( Y ) * W + X], v2 = i2 ? p[size_t( Y ) * W + X + 1] : 0;
        const int   v3 = i3 ? p[size_t( Y + 1 ) * W + X] : 0, v4 = ( i2 && i3 ) ? p[size_t( Y + 1 ) * W + X + 1] : 0;
        out.c[k][size_t( y ) * w + x] = uint8_t( mean4w( v1, w1, v2, w2, v3, w3, v4, w4 ) );
      }
      occOut[size_t( y ) * w + x] = 1;
    }
}

void fill( Plane3& img, const Plane3& m, const std::vector<uint8_t>& occ, int numIters ) {
  const int W = img.w, H = img.h, w = m.w, h = m.h;
  for ( int Y = 0; Y < H; ++Y )
    for ( int X = 0; X < W; ++X ) {
      if ( occ[size_t( Y ) * W + X] ) continue;
      const int  x = X >> 1, y = Y >> 1;
      const int  dx = ( X & 1 ) ? 1 : -1, dy = ( Y & 1 ) ? 1 : -1;  // odd pixels lean right/down, even ones left/up
      const bool hx = ( dx < 0 ) ? x > 0 : x < w - 1, hy = ( dy < 0 ) ? y > 0 : y < h - 1;
      for ( int k = 0; k < 3; ++k ) {
        const auto& p  = m.c[k];
        const int   v  = p[size_t( y ) * w + x];
        const int   vx = hx ? p[size_t( y ) * w + x + dx] : 0;
        const int   vy = hy ? p[size_t( y + dy ) * w + x] : 0;
        const int   vd = ( hx && hy ) ? p[size_t( y + dy ) * w + x + dx] : 0;
        img.c[k][size_t( Y ) * W + X] = uint8_t( mean4w( v, 144, vx, hx ? 48 : 0, vy, hy ? 48 : 0, vd, ( hx && hy ) ? 16 : 0 ) );
      }
    }
  Plane3 tmp = img;
  for ( int n = 0; n < numIters; ++n ) {
    for ( int y = 0; y < H; ++y )
      for ( int x = 0; x < W; ++x ) {
        if ( occ[size_t( y ) * W + x] ) continue;
        const int x1 = x > 0 ? x - 1 : x, y1 = y > 0 ? y - 1 : y, x2 = x < W - 1 ? x + 1 : x, y2 = y < H - 1 ? y + 1 : y;
        for ( int k = 0; k < 3; ++k ) {
          const auto& p = img.c[k];
          const int   s = p[size_t( y1 ) * W + x1] + p[size_t( y1 ) * W + x2] + p[size_t( y2 ) * W + x1] +
                        p[size_t( y2 ) * W + x2] + p[size_t( y ) * W + x1] + p[size_t( y ) * W + x2] +
                        p[size_t( y1 ) * W + x] + p[size_t( y2 ) * W + x];
          tmp.c[k][size_t( y ) * W + x] = uint8_t( ( s + 4 ) >> 3 );
        }
      }
    std::swap( img, tmp );
  }
}

void pushPull( Plane3& img, const std::vector<uint8_t>& occ ) {
  std::vector<Plane3>               mips;
  std::vector<std::vector<uint8_t>> occs;
  for ( ;; ) {
    mips.emplace_back();
    occs.emplace_back();
    const size_t l = mips.size() - 1;
    if ( l > 0 )
      mip( mips[l - 1], occs[l - 1], mips[l], occs[l] );
    else
      mip( img, occ, mips[0], occs[0] );
    if ( mips[l].w <= 4 || mips[l].h <= 4 ) break;
  }
  int iters = 4;
  for ( int i = int( mips.size() ) - 1; i >= 0; --i ) {
    if ( i > 0 )
      fill( mips[i - 1], mips[i], occs[i - 1], iters );
    else
      fill( img, mips[0], occ, iters );
    iters = std::min( iters + 1, 16 );
  }
}
}  // namespace

extern "C" {

// S17.  patches in packing order (order[k] -> patch record).  Upper bound of the output is 2*W*H points.
// pointToPixel: u32[M][3] = (x, y, layer).  Returns M.
int64_t orc_generate_point_cloud( const orc_patch* patches, const int32_t* order, int P, const uint8_t* occVideo,
                                  const uint32_t* blockToPatch, const uint16_t* geo0, const uint16_t* geo1, int W, int H,
                                  int occRes, int occPrecision, int16_t* xyz, uint32_t* pointToPixel ) {
  const int Wv = W / occPrecision, Wb = W / occRes;
  int64_t   M  = 0;
  for ( int k = 0; k < P; ++k ) {
    const orc_patch& p = patches[order[k]];
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub ) {
        const int bx = p.patchOrientation == 0 ? ub + p.u0 : vb + p.u0;
        const int by = p.patchOrientation == 0 ? vb + p.v0 : ub + p.v0;
        if ( blockToPatch[size_t( by ) * Wb + bx] != uint32_t( k + 1 ) ) continue;
        for ( int j = 0; j < occRes; ++j )
          for ( int i = 0; i < occRes; ++i ) {
            const int u = ub * occRes + i, v = vb * occRes + j;
            const int x = p.patchOrientation == 0 ? u + p.u0 * occRes : v + p.u0 * occRes;
            const int y = p.patchOrientation == 0 ? v + p.v0 * occRes : u + p.v0 * occRes;
            if ( !occVideo[size_t( y / occPrecision ) * Wv + x / occPrecision] ) continue;
            for ( int layer = 0; layer < 2; ++layer ) {
              const int depth = ( layer ? geo1 : geo0 )[size_t( y ) * W + x];
              int       pt[3];
              if ( p.projectionMode == 0 )
                pt[p.normalAxis] = depth + p.d1;
              else
                pt[p.normalAxis] = std::max( 0, p.d1 - depth );
              pt[p.tangentAxis]   = u + p.u1;
              pt[p.bitangentAxis] = v + p.v1;
              if ( layer == 1 && pt[0] == xyz[3 * ( M - 1 )] && pt[1] == xyz[3 * ( M - 1 ) + 1] &&
                   pt[2] == xyz[3 * ( M - 1 ) + 2] )
                continue;  // removeDuplicatePoints: D1 equal to D0 is dropped
              xyz[3 * M] = int16_t( pt[0] ), xyz[3 * M + 1] = int16_t( pt[1] ), xyz[3 * M + 2] = int16_t( pt[2] );
              pointToPixel[3 * M] = uint32_t( x ), pointToPixel[3 * M + 1] = uint32_t( y ), pointToPixel[3 * M + 2] = uint32_t( layer );
              ++M;
            }
          }
      }
  }
  return M;
}

// S18 (+S19 no-op).
int orc_transfer_colors( const int16_t* srcXyz, const uint8_t* srcRgb, size_t n, const int16_t* tgtXyz, size_t m,
                         uint8_t* tgtRgb ) {
  orc_kdtree*           srcTree = orc_kdtree_build( srcXyz, n );
  orc_kdtree*           tgtTree = orc_kdtree_build( tgtXyz, m );
  std::vector<uint32_t> idx8( m * 8 ), idx1( n );
  std::vector<double>   dist8( m * 8 ), dist1( n );
  orc_knn( srcTree, tgtXyz, m, 8, idx8.data(), dist8.data() );
  orc_knn( tgtTree, srcXyz, n, 1, idx1.data(), dist1.data() );
  orc_kdtree_free( srcTree );
  orc_kdtree_free( tgtTree );
  // forward colour of every target point
  std::vector<uint8_t> fwd( 3 * m );
  for ( size_t t = 0; t < m; ++t ) {
    const uint32_t* id = &idx8[8 * t];
    const double*   d  = &dist8[8 * t];
    if ( d[0] < 0.0001 ) {
      for ( int k = 0; k < 3; ++k ) fwd[3 * t + k] = srcRgb[3 * size_t( id[0] ) + k];
      continue;
    }
    double c[3] = {0.0, 0.0, 0.0}, sumW = 0.0;
    for ( int i = 0; i < 8; ++i ) {
      const double w = 1 / ( d[i] + 4.0 );
      for ( int k = 0; k < 3; ++k ) c[k] += srcRgb[3 * size_t( id[i] ) + k] * w;
      sumW += w;
    }
    for ( int k = 0; k < 3; ++k ) fwd[3 * t + k] = uint8_t( clip255( std::round( c[k] / sumW ) ) );
  }
  // backward candidates: every source point votes for its nearest target, in source order
  struct Cand {
    double   d;
    uint32_t s;
  };
  std::vector<std::vector<Cand>> cand( m );
  for ( size_t s = 0; s < n; ++s ) cand[idx1[s]].push_back( Cand{dist1[s], uint32_t( s )} );
  for ( size_t t = 0; t < m; ++t ) {
    auto& L = cand[t];
    if ( L.empty() ) {
      for ( int k = 0; k < 3; ++k ) tgtRgb[3 * t + k] = fwd[3 * t + k];
      continue;
    }
    // the reference's std::sort by distance only: NOT stable beyond 16 elements, so equal distances come out in
    // libstdc++ introsort order; the oracle is built with the same libstdc++ and simply calls it
    std::sort( L.begin(), L.end(), []( const Cand& a, const Cand& b ) { return a.d < b.d; } );
    double c2[3] = {0.0, 0.0, 0.0};
    if ( L[0].d < 0.0001 || L.size() == 1 ) {
      for ( int k = 0; k < 3; ++k ) c2[k] = srcRgb[3 * size_t( L[0].s ) + k];
    } else {
      double sumW = 0.0;
      for ( auto& e : L ) {
        const double w = 1 / ( std::sqrt( e.d ) + 4.0 );
        for ( int k = 0; k < 3; ++k ) c2[k] += srcRgb[3 * size_t( e.s ) + k] * w;
        sumW += w;
      }
      for ( int k = 0; k < 3; ++k ) c2[k] /= sumW;
    }
    // fixWeight: w = 0, searchRange = 0  ->  round(0*centroid1 + 1*centroid2)
    for ( int k = 0; k < 3; ++k )
      tgtRgb[3 * t + k] = uint8_t( clip255( std::round( 0.0 * double( fwd[3 * t + k] ) + 1.0 * c2[k] ) ) );
  }
  return 0;
}

// S20-S22.  out: u8 [2 maps][3 channels][H][W]
int orc_attribute_images( const uint8_t* rgb, const uint32_t* pointToPixel, int64_t M, const uint8_t* occVideo, int W, int H,
                          int occPrecision, uint8_t* out ) {
  const int            Wv = W / occPrecision;
  std::vector<uint8_t> occ( size_t( W ) * H );
  for ( int y = 0; y < H; ++y )
    for ( int x = 0; x < W; ++x ) occ[size_t( y ) * W + x] = occVideo[size_t( y / occPrecision ) * Wv + x / occPrecision];
  Plane3 img[2];
  img[0].resize( W, H );
  img[1].resize( W, H );
  std::vector<uint8_t> markT1( size_t( W ) * H, 0 );
  for ( int64_t i = 0; i < M; ++i ) {
    const size_t px = size_t( pointToPixel[3 * i + 1] ) * W + pointToPixel[3 * i];
    const int    f  = int( pointToPixel[3 * i + 2] );
    for ( int k = 0; k < 3; ++k ) img[f].c[k][px] = rgb[3 * i + k];
    if ( f == 0 ) {
      if ( !markT1[px] )
        for ( int k = 0; k < 3; ++k ) img[1].c[k][px] = rgb[3 * i + k];
    } else {
      markT1[px] = 1;
    }
  }
  pushPull( img[0], occ );
  pushPull( img[1], occ );
  for ( size_t px = 0; px < size_t( W ) * H; ++px )
    if ( !occ[px] )
      for ( int k = 0; k < 3; ++k ) {
        const uint32_t avg = ( uint32_t( img[0].c[k][px] ) + uint32_t( img[1].c[k][px] ) + 1 ) >> 1;
        img[0].c[k][px] = img[1].c[k][px] = uint8_t( avg );
      }
  size_t o = 0;
  for ( int mI = 0; mI < 2; ++mI )
    for ( int k = 0; k < 3; ++k ) {
      std::memcpy( out + o, img[mI].c[k].data(), size_t( W ) * H );
      o += size_t( W ) * H;
    }
  return 0;
}

}  // extern "C"
