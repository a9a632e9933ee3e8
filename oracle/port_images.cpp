// oracle/port_images.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S10  patch packing, all-intra: PCCEncoder::packFlexible (PCCEncoder.cpp:2306-2449) with
//      PCCPatch::gt / checkFitPatchCanvas / patchBlock2CanvasBlock (PCCPatch.cpp:253-371), and the GOF canvas
//      size rule of resizeTileGeometryVideo / resizeGeometryVideo (:5593-5632, :5546-5591)
// S11  occupancy map                    PCCEncoder::generateOccupancyMap (:3767-3784), patch2Canvas (PCCPatch.cpp:192-251)
// S12  occupancy video (precision p)    PCCEncoder::generateOccupancyMapVideo (:806-861)
// S13  block-to-patch map               PCCCodec::generateBlockToPatchFromOccupancyMapVideo (PCCCodec.cpp:1736-1775)
// S14  geometry images D0 / D1          PCCEncoder::generateIntraImage (:3929-3992)
// S15  block dilation                   PCCEncoder::dilate3DPadding, geometryPadding=0 (:5951-6130)
// S16  group dilation of D0 / D1        PCCEncoder::dilateGroupGeometryVideo (:3717-3739)
// Two patch orientations (DEFAULT, SWAP) as under the CTC (packingStrategy=1, useEightOrientations=0).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {
enum { ORIENT_DEFAULT = 0, ORIENT_SWAP = 1 };

// canvas position of patch pixel (u,v)
inline void toCanvas( const orc_patch& p, int occRes, int u, int v, int& x, int& y ) {
  if ( p.patchOrientation == ORIENT_DEFAULT ) {
    x = u + p.u0 * occRes;
    y = v + p.v0 * occRes;
  } else {
    x = v + p.u0 * occRes;
    y = u + p.v0 * occRes;
  }
}
}  // namespace

extern "C" {

// patches: in/out, indexed by patch index; order: out, the packing order (a permutation of 0..P-1).
// Returns the frame height in pixels through *height.
int orc_pack_flexible( orc_patch* patches, int P, const uint8_t* occupancy, int presetWidth, int occRes, int numTilesHor,
                       double tileHeightToWidthRatio, int32_t* order, int32_t* height ) {
  for ( int i = 0; i < P; ++i ) order[i] = i;
  if ( P == 0 ) {
    *height = 0;
    return 0;
  }
  std::sort( order, order + P, [&]( int a, int b ) {
    const orc_patch &A = patches[a], &B = patches[b];
    const int        aMax = std::max( A.sizeU0, A.sizeV0 ), aMin = std::min( A.sizeU0, A.sizeV0 );
    const int        bMax = std::max( B.sizeU0, B.sizeV0 ), bMin = std::min( B.sizeU0, B.sizeV0 );
    return aMax != bMax ? aMax > bMax : ( aMin != bMin ? aMin > bMin : A.index < B.index );
  } );
  size_t sizeU = size_t( presetWidth / occRes );
  size_t sizeV = size_t( std::max( patches[order[0]].sizeV0, patches[order[0]].sizeU0 ) );
  for ( int i = 0; i < P; ++i ) sizeU = std::max( sizeU, size_t( patches[i].sizeU0 + 1 ) );
  const int tileW = int( sizeU ) / numTilesHor;
  const int tileH = int( tileW * tileHeightToWidthRatio );
  if ( int( sizeV ) < tileH ) sizeV = size_t( tileH );
  size_t               h = sizeV * occRes;
  std::vector<uint8_t> canvas( sizeU * sizeV, 0 );
  for ( int k = 0; k < P; ++k ) {
    orc_patch& p = patches[order[k]];
    bool       found = false;
    while ( !found ) {
      for ( size_t v = 0; v < sizeV && !found; ++v )
        for ( size_t u = 0; u < sizeU && !found; ++u )
          for ( int o = 0; o < 2 && !found; ++o ) {
            const int orient = ( p.sizeU0 > p.sizeV0 ) ? ( o == 0 ? ORIENT_SWAP : ORIENT_DEFAULT )
                                                       : ( o == 0 ? ORIENT_DEFAULT : ORIENT_SWAP );
            const size_t w  = orient == ORIENT_DEFAULT ? p.sizeU0 : p.sizeV0;
            const size_t hh = orient == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
            bool         fits = u + w <= sizeU && v + hh <= sizeV;
            for ( size_t y = v; fits && y < v + hh; ++y )
              for ( size_t x = u; x < u + w; ++x )
                if ( canvas[y * sizeU + x] ) {
                  fits = false;
                  break;
                }
            // the reference leaves u0/v0/orientation at the last probed values even when nothing fits
            p.u0               = int32_t( u );
            p.v0               = int32_t( v );
            p.patchOrientation = orient;
            if ( fits ) found = true;
          }
      if ( !found ) {
        if ( sizeV > ( size_t( 1 ) << 20 ) ) return -2;  // nothing can be placed: the reference spins here
        sizeV *= 2;
        canvas.resize( sizeU * sizeV, 0 );
      }
    }
    const uint8_t* occ = occupancy + p.occOffset;
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub ) {
        const size_t x = p.patchOrientation == ORIENT_DEFAULT ? ub + p.u0 : vb + p.u0;
        const size_t y = p.patchOrientation == ORIENT_DEFAULT ? vb + p.v0 : ub + p.v0;
        canvas[y * sizeU + x] = canvas[y * sizeU + x] || occ[vb * p.sizeU0 + ub];
      }
    const int span = p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
    h              = std::max( h, size_t( p.v0 + span ) * occRes );
  }
  *height = int32_t( h );
  return 0;
}

// S10' (low-delay condition: constrainedPack = 1, globalPatchAllocation = 0)
//      PCCEncoder::spatialConsistencyPackFlexible (PCCEncoder.cpp:1183-1412, packingStrategy = 1, safeguard 0, two
//      orientations, lowDelayEncoding off), pcc::computeIOU (PCCPatchSegmenter.cpp:1563-1570), Rect::operator&
//      (PCCPatchSegmenter.h:407-417): patches of the previous frame (in ITS list order) look, one after the other, for
//      the still unmatched patch of this frame (sorted by gt) with the same view and the largest bounding-box IoU
//      (float, > 0.2); matched patches come first, in the previous frame's order, and are tried at their match's
//      position and orientation before a raster scan that keeps the orientation; the others follow as in packFlexible.
// patches: in/out, indexed by patch index; prev: the previous frame's patches in list order (u0, v0, orientation set).
// order: out, list order; bestMatch: out, per LIST position the matched position in prev or -1.
int orc_pack_spatial_consistency( orc_patch* patches, int P, const uint8_t* occupancy, const orc_patch* prev, int Pprev,
                                  int presetWidth, int occRes, int numTilesHor, double tileHeightToWidthRatio,
                                  int32_t* order, int32_t* bestMatch, int32_t* height ) {
  if ( P == 0 ) {
    *height = 0;
    return 0;
  }
  std::vector<int> sorted( P );
  for ( int i = 0; i < P; ++i ) sorted[i] = i;
  std::sort( sorted.begin(), sorted.end(), [&]( int a, int b ) {
    const orc_patch &A = patches[a], &B = patches[b];
    const int        aMax = std::max( A.sizeU0, A.sizeV0 ), aMin = std::min( A.sizeU0, A.sizeV0 );
    const int        bMax = std::max( B.sizeU0, B.sizeV0 ), bMin = std::min( B.sizeU0, B.sizeV0 );
    return aMax != bMax ? aMax > bMax : ( aMin != bMin ? aMin > bMin : A.index < B.index );
  } );
  size_t sizeU = size_t( presetWidth / occRes );
  size_t sizeV = size_t( std::max( patches[sorted[0]].sizeU0, patches[sorted[0]].sizeV0 ) );
  std::vector<int> match( P, -1 );  // per patch index
  std::vector<int> list;
  for ( int id = 0; id < Pprev; ++id ) {
    const orc_patch& q       = prev[id];
    float            maxIou  = 0.0F;
    int              bestIdx = -1;
    for ( int c = 0; c < P; ++c ) {
      const orc_patch& r = patches[sorted[c]];
      if ( q.viewId != r.viewId || match[sorted[c]] != -1 ) continue;
      const int x1 = std::max( q.u1, r.u1 ), y1 = std::max( q.v1, r.v1 );
      int       w  = std::min( q.u1 + q.sizeU, r.u1 + r.sizeU ) - x1, h = std::min( q.v1 + q.sizeV, r.v1 + r.sizeV ) - y1;
      if ( w <= 0 || h <= 0 ) w = h = 0;
      const int   inter = w * h, uni = q.sizeU * q.sizeV + r.sizeU * r.sizeV - inter;
      const float iou   = static_cast<float>( inter ) / uni;
      if ( iou > maxIou ) {
        maxIou  = iou;
        bestIdx = c;
      }
    }
    if ( maxIou > 0.2F ) {
      match[sorted[bestIdx]] = id;
      list.push_back( sorted[bestIdx] );
    }
  }
  for ( int c = 0; c < P; ++c )
    if ( match[sorted[c]] == -1 ) list.push_back( sorted[c] );
  for ( int k = 0; k < P; ++k ) {
    order[k]     = list[k];
    bestMatch[k] = match[list[k]];
    sizeU        = std::max( sizeU, size_t( patches[list[k]].sizeU0 + 1 ) );
  }
  const int tileW = int( sizeU ) / numTilesHor;
  const int tileH = int( tileW * tileHeightToWidthRatio );
  if ( int( sizeV ) < tileH ) sizeV = size_t( tileH );
  size_t               hPix = sizeV * occRes;
  std::vector<uint8_t> canvas( sizeU * sizeV, 0 );
  auto fits = [&]( const orc_patch& p ) {
    const size_t w = p.patchOrientation == ORIENT_DEFAULT ? p.sizeU0 : p.sizeV0;
    const size_t h = p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
    if ( size_t( p.u0 ) + w > sizeU || size_t( p.v0 ) + h > sizeV ) return false;
    for ( size_t y = p.v0; y < p.v0 + h; ++y )
      for ( size_t x = p.u0; x < p.u0 + w; ++x )
        if ( canvas[y * sizeU + x] ) return false;
    return true;
  };
  for ( int k = 0; k < P; ++k ) {
    orc_patch& p     = patches[list[k]];
    bool       found = false;
    while ( !found ) {
      if ( match[list[k]] != -1 ) {
        const orc_patch& q = prev[match[list[k]]];
        p.patchOrientation = q.patchOrientation;
        p.u0               = q.u0;
        p.v0               = q.v0;
        found              = fits( p );
        for ( size_t v = 0; v <= sizeV && !found; ++v )
          for ( size_t u = 0; u <= sizeU && !found; ++u ) {
            p.u0  = int32_t( u );
            p.v0  = int32_t( v );
            found = fits( p );
          }
      } else {
        for ( size_t v = 0; v < sizeV && !found; ++v )
          for ( size_t u = 0; u < sizeU && !found; ++u )
            for ( int o = 0; o < 2 && !found; ++o ) {
              p.u0               = int32_t( u );
              p.v0               = int32_t( v );
              p.patchOrientation = ( p.sizeU0 > p.sizeV0 ) ? ( o == 0 ? ORIENT_SWAP : ORIENT_DEFAULT )
                                                           : ( o == 0 ? ORIENT_DEFAULT : ORIENT_SWAP );
              found              = fits( p );
            }
      }
      if ( !found ) {
        if ( sizeV > ( size_t( 1 ) << 20 ) ) return -2;  // nothing can be placed: the reference spins here
        sizeV *= 2;
        canvas.resize( sizeU * sizeV, 0 );
      }
    }
    const uint8_t* occ = occupancy + p.occOffset;
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub ) {
        const size_t x = p.patchOrientation == ORIENT_DEFAULT ? ub + p.u0 : vb + p.u0;
        const size_t y = p.patchOrientation == ORIENT_DEFAULT ? vb + p.v0 : ub + p.v0;
        canvas[y * sizeU + x] = canvas[y * sizeU + x] || occ[vb * p.sizeU0 + ub];
      }
    const int span = p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
    hPix           = std::max( hPix, size_t( p.v0 + span ) * occRes );
  }
  *height = int32_t( hPix );
  return 0;
}

// common canvas of a GOF: max over frames and the configured minimum, rounded up to 64
int orc_gof_canvas_size( const int32_t* frameHeights, int frames, int tileWidth, int minWidth, int minHeight,
                         int32_t* W, int32_t* H ) {
  size_t w = size_t( std::max( tileWidth, minWidth ) ), h = size_t( minHeight );
  for ( int i = 0; i < frames; ++i ) h = std::max( h, size_t( frameHeights[i] ) );
  *W = int32_t( std::ceil( double( w ) / 64.0 ) * 64 );
  *H = int32_t( std::ceil( double( h ) / 64.0 ) * 64 );
  return 0;
}

// S11-S16 for one frame.  Outputs (caller-allocated): occMap u8[W*H], occVideo u8[(W/p)*(H/p)],
// blockToPatch u32[(W/16)*(H/16)], geo0/geo1 u16[W*H].
int orc_generate_geometry_images( const orc_patch* patches, const int32_t* order, int P, const int16_t* depth0,
                                  const int16_t* depth1, int W, int H, int occRes, int occPrecision, uint8_t* occMap,
                                  uint8_t* occVideo, uint32_t* blockToPatch, uint16_t* geo0, uint16_t* geo1 ) {
  const size_t area = size_t( W ) * H;
  std::memset( occMap, 0, area );
  std::memset( geo0, 0, area * 2 );
  std::memset( geo1, 0, area * 2 );
  // S11 + S14
  for ( int k = 0; k < P; ++k ) {
    const orc_patch& p = patches[order[k]];
    for ( int v = 0; v < p.sizeV; ++v )
      for ( int u = 0; u < p.sizeU; ++u ) {
        const size_t  q = size_t( p.depthOffset ) + size_t( v ) * p.sizeU + u;
        const int16_t d = depth0[q];
        if ( d < 32767 ) {
          int x, y;
          toCanvas( p, occRes, u, v, x, y );
          if ( x >= W || y >= H ) return -180;  // the reference exit(180)s here
          occMap[size_t( y ) * W + x] = 1;
          geo0[size_t( y ) * W + x]   = uint16_t( d );
          geo1[size_t( y ) * W + x]   = uint16_t( depth1[q] );
        }
      }
  }
  // S12
  const int Wv = W / occPrecision, Hv = H / occPrecision;
  for ( int yv = 0; yv < Hv; ++yv )
    for ( int xv = 0; xv < Wv; ++xv ) {
      uint8_t full = 0;
      for ( int j = 0; j < occPrecision && !full; ++j )
        for ( int i = 0; i < occPrecision && !full; ++i )
          full = occMap[size_t( yv * occPrecision + j ) * W + xv * occPrecision + i] > 0;
      occVideo[size_t( yv ) * Wv + xv] = full;
    }
  // S13: later patches (in packing order) overwrite earlier ones
  const int Wb = W / occRes, Hb = H / occRes;
  std::fill( blockToPatch, blockToPatch + size_t( Wb ) * Hb, 0u );
  for ( int k = 0; k < P; ++k ) {
    const orc_patch& p = patches[order[k]];
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub ) {
        const int bx = p.patchOrientation == ORIENT_DEFAULT ? ub + p.u0 : vb + p.u0;
        const int by = p.patchOrientation == ORIENT_DEFAULT ? vb + p.v0 : ub + p.v0;
        bool      any = false;
        for ( int j = 0; j < occRes && !any; ++j )
          for ( int i = 0; i < occRes && !any; ++i )
            any = occVideo[size_t( ( by * occRes + j ) / occPrecision ) * Wv + ( bx * occRes + i ) / occPrecision] != 0;
        if ( any ) blockToPatch[size_t( by ) * Wb + bx] = uint32_t( k + 1 );
      }
  }
  // S15 per map, raster block order
  std::vector<uint32_t> level( size_t( occRes ) * occRes );
  std::vector<int32_t>  sum( level.size() );
  std::vector<uint32_t> cnt( level.size() );
  uint16_t*             maps[2] = {geo0, geo1};
  for ( int m = 0; m < 2; ++m ) {
    uint16_t* img = maps[m];
    for ( int by = 0; by < Hb; ++by )
      for ( int bx = 0; bx < Wb; ++bx ) {
        const int x0 = bx * occRes, y0 = by * occRes;
        size_t    filled = 0;
        for ( int j = 0; j < occRes; ++j )
          for ( int i = 0; i < occRes; ++i ) {
            level[j * occRes + i] = occMap[size_t( y0 + j ) * W + x0 + i] ? 1 : 0;
            filled += level[j * occRes + i];
          }
        if ( filled == 0 ) {
          if ( bx > 0 ) {
            for ( int j = 0; j < occRes; ++j )
              for ( int i = 0; i < occRes; ++i ) img[size_t( y0 + j ) * W + x0 + i] = img[size_t( y0 + j ) * W + x0 + i - 1];
          } else if ( by > 0 ) {
            for ( int j = 0; j < occRes; ++j )
              for ( int i = 0; i < occRes; ++i ) img[size_t( y0 + j ) * W + x0 + i] = img[size_t( y0 + j - 1 ) * W + x0 + i];
          }
          continue;
        }
        uint32_t it = 1;
        while ( filled < level.size() ) {
          std::fill( sum.begin(), sum.end(), 0 );
          std::fill( cnt.begin(), cnt.end(), 0u );
          for ( int j = 0; j < occRes; ++j )
            for ( int i = 0; i < occRes; ++i ) {
              if ( level[j * occRes + i] != it ) continue;
              static const int nb[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
              for ( auto& d : nb ) {
                const int i2 = i + d[0], j2 = j + d[1];
                if ( i2 < 0 || j2 < 0 || i2 >= occRes || j2 >= occRes || level[j2 * occRes + i2] != 0 ) continue;
                sum[j2 * occRes + i2] += img[size_t( y0 + j ) * W + x0 + i];
                ++cnt[j2 * occRes + i2];
              }
            }
          for ( int j = 0; j < occRes; ++j )
            for ( int i = 0; i < occRes; ++i ) {
              const uint32_t c = cnt[j * occRes + i];
              if ( !c ) continue;
              ++filled;
              level[j * occRes + i]              = it + 1;
              img[size_t( y0 + j ) * W + x0 + i] = uint16_t( ( uint32_t( sum[j * occRes + i] ) + c / 2 ) / c );
            }
          ++it;
        }
      }
  }
  // S16
  for ( int y = 0; y < H; ++y )
    for ( int x = 0; x < W; ++x )
      if ( occVideo[size_t( y / occPrecision ) * Wv + x / occPrecision] == 0 ) {
        const uint32_t avg = ( uint32_t( geo0[size_t( y ) * W + x] ) + uint32_t( geo1[size_t( y ) * W + x] ) + 1 ) >> 1;
        geo0[size_t( y ) * W + x] = geo1[size_t( y ) * W + x] = uint16_t( avg );
      }
  return 0;
}

}  // extern "C"
