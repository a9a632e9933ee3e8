// packing.cpp -- patch placement on the block canvas (S10), host side.
//
// Replaces PCCEncoder::packFlexible (reference: source/lib/PccLibEncoder/source/PCCEncoder.cpp:2306-2449) as used
// by placeSegments for the all-intra CTC condition (constrainedPack=0, packingStrategy=1, safeguard 0, two
// orientations), PCCPatch::gt / checkFitPatchCanvas / patchBlock2CanvasBlock (PccLibCommon/source/PCCPatch.cpp:253-371)
// and the canvas-size rule of resizeTileGeometryVideo / resizeGeometryVideo (PCCEncoder.cpp:5593-5632, 5546-5591).
//
// S10' (low-delay condition, constrainedPack = 1): PCCEncoder::spatialConsistencyPackFlexible (:1183-1412) with
// pcc::computeIOU (PCCPatchSegmenter.cpp:1563-1570) -- frames after the first are packed against their predecessor.
//
// Inherently sequential first-fit over a few hundred patches on an 80-block-wide canvas -- microseconds of
// work on a few KB of data -- so it runs on the host between the segmentation kernels and the raster kernels.
// The reference copies the whole canvas by value for every probe (PCCPatch.h:219); here each canvas row is a
// bit mask, and a candidate position is rejected with one AND per row of the patch's box.
#include <algorithm>
#include <cmath>

#include "internal.h"

namespace tmc2 {

namespace {
enum { ORIENT_DEFAULT = 0, ORIENT_SWAP = 1 };

// canvas rows as arrays of 64-bit words; bit x of row y = block (x,y) occupied
struct BlockCanvas {
  size_t                width, height, words;
  std::vector<uint64_t> rows;
  BlockCanvas( size_t w, size_t h ) : width( w ), height( h ), words( ( w + 63 ) / 64 ), rows( words * h, 0 ) {}
  void grow( size_t h ) {
    rows.resize( words * h, 0 );
    height = h;
  }
  bool boxFree( size_t x, size_t y, size_t w, size_t h ) const {
    if ( x + w > width || y + h > height ) return false;
    for ( size_t r = y; r < y + h; ++r ) {
      const uint64_t* row = &rows[r * words];
      for ( size_t c = x; c < x + w; ) {
        const size_t   word = c >> 6, bit = c & 63;
        const size_t   span = std::min<size_t>( 64 - bit, x + w - c );
        const uint64_t mask = ( span == 64 ? ~0ull : ( ( 1ull << span ) - 1ull ) ) << bit;
        if ( row[word] & mask ) return false;
        c += span;
      }
    }
    return true;
  }
  void set( size_t x, size_t y ) { rows[y * words + ( x >> 6 )] |= 1ull << ( x & 63 ); }
};
}  // namespace

// Core of S10 on plain records (no device involved).  pt: the frame's patches by index (u0 / v0 / orientation out); occ: their
// block-occupancy pool; order: list order out (PCCPatch::gt).  Returns the frame height in pixels, -1 if a patch fits at no
// canvas height.
int packFlexibleCore( tmc2_patch* pt, int P, const uint8_t* occ, int presetWidth, int occRes, int numTilesHor, double ratio,
                      int32_t* order ) {
  for ( int i = 0; i < P; ++i ) order[i] = i;
  if ( P == 0 ) return 0;
  // largest block dimension first, then the other dimension, then creation order (a total order)
  std::sort( order, order + P, [&]( int a, int b ) {
    const int aMax = std::max( pt[a].sizeU0, pt[a].sizeV0 ), aMin = std::min( pt[a].sizeU0, pt[a].sizeV0 );
    const int bMax = std::max( pt[b].sizeU0, pt[b].sizeV0 ), bMin = std::min( pt[b].sizeU0, pt[b].sizeV0 );
    if ( aMax != bMax ) return aMax > bMax;
    if ( aMin != bMin ) return aMin > bMin;
    return pt[a].index < pt[b].index;
  } );
  size_t sizeU = size_t( presetWidth / occRes );
  for ( int i = 0; i < P; ++i ) sizeU = std::max( sizeU, size_t( pt[i].sizeU0 + 1 ) );
  size_t    sizeV = size_t( std::max( pt[order[0]].sizeU0, pt[order[0]].sizeV0 ) );
  const int tileH = int( ( int( sizeU ) / numTilesHor ) * ratio );
  sizeV           = std::max( sizeV, size_t( std::max( tileH, 0 ) ) );
  size_t      heightBlocks = sizeV;
  BlockCanvas canvas( sizeU, sizeV );
  for ( int k = 0; k < P; ++k ) {
    tmc2_patch& p      = pt[order[k]];
    const bool  wide   = p.sizeU0 > p.sizeV0;
    const int   first  = wide ? ORIENT_SWAP : ORIENT_DEFAULT;  // wide patches are tried upright first
    const int   second = wide ? ORIENT_DEFAULT : ORIENT_SWAP;
    bool        placed = false;
    while ( !placed ) {
      for ( size_t v = 0; v < canvas.height && !placed; ++v )
        for ( size_t u = 0; u < canvas.width && !placed; ++u )
          for ( int o = 0; o < 2 && !placed; ++o ) {
            const int    orient = o == 0 ? first : second;
            const size_t w = orient == ORIENT_DEFAULT ? p.sizeU0 : p.sizeV0, h = orient == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
            if ( canvas.boxFree( u, v, w, h ) ) {
              p.u0               = int32_t( u );
              p.v0               = int32_t( v );
              p.patchOrientation = orient;
              placed             = true;
            }
          }
      if ( !placed ) {
        if ( canvas.height > ( size_t( 1 ) << 20 ) ) return -1;
        canvas.grow( canvas.height * 2 );
      }
    }
    const uint8_t* o = occ + p.occOffset;
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub )
        if ( o[vb * p.sizeU0 + ub] ) {
          if ( p.patchOrientation == ORIENT_DEFAULT )
            canvas.set( size_t( p.u0 + ub ), size_t( p.v0 + vb ) );
          else
            canvas.set( size_t( p.u0 + vb ), size_t( p.v0 + ub ) );
        }
    heightBlocks = std::max( heightBlocks, size_t( p.v0 + ( p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0 ) ) );
  }
  return int( heightBlocks ) * occRes;
}

int packFlexibleHost( tmc2_frame* f, int presetWidth, int occRes, int numTilesHor, double ratio ) {
  if ( !f->havePatches ) {
    setError( "packFlexible: no patches" );
    return TMC2_E_STATE;
  }
  const int P = int( f->patches.size() );
  f->packOrder.resize( P );
  f->packMatch.assign( P, -1 );
  for ( int i = 0; i < P; ++i ) f->packOrder[i] = i;
  f->packedHeight = 0;
  f->packedWidth  = presetWidth;  // packFlexible works on a COPY of the tile width (PCCEncoder.cpp:2312): the tile keeps its own
  f->havePacking  = true;
  if ( P == 0 ) return TMC2_OK;
  // per-block occupancy of the patches comes back from the device (a few KB)
  std::vector<uint8_t> occ( size_t( f->occCount ) );
  TMC2_HIP( hipMemcpyAsync( occ.data(), f->d_occupancy.p, occ.size(), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  f->packedHeight = packFlexibleCore( f->patches.data(), P, occ.data(), presetWidth, occRes, numTilesHor, ratio, f->packOrder.data() );
  if ( f->packedHeight < 0 ) {
    f->havePacking = false;
    setError( "packFlexible: a patch fits at no canvas height" );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}


namespace {
bool gtOrder( const tmc2_patch& A, const tmc2_patch& B ) {  // PCCPatch::gt
  const int aMax = std::max( A.sizeU0, A.sizeV0 ), aMin = std::min( A.sizeU0, A.sizeV0 );
  const int bMax = std::max( B.sizeU0, B.sizeV0 ), bMin = std::min( B.sizeU0, B.sizeV0 );
  if ( aMax != bMax ) return aMax > bMax;
  if ( aMin != bMin ) return aMin > bMin;
  return A.index < B.index;
}
}  // namespace

// Core of S10' on plain records (no device involved).  pt: this frame's patches by index (u0 / v0 / orientation out);
// occ: their block occupancy pool; prev: the previous frame's patches IN LIST ORDER, packed.  order: list order out;
// match: per list position the matched position in prev, or -1.  Returns the frame height in pixels.
int packSpatialConsistencyCore( tmc2_patch* pt, int P, const uint8_t* occ, const tmc2_patch* prev, int Pprev, int presetWidth,
                                int occRes, int numTilesHor, double ratio, int32_t* order, int32_t* match ) {
  if ( P == 0 ) return 0;
  std::vector<int> sorted( P );
  for ( int i = 0; i < P; ++i ) sorted[i] = i;
  std::sort( sorted.begin(), sorted.end(), [&]( int a, int b ) { return gtOrder( pt[a], pt[b] ); } );
  size_t sizeU = size_t( presetWidth / occRes );
  size_t sizeV = size_t( std::max( pt[sorted[0]].sizeU0, pt[sorted[0]].sizeV0 ) );
  // matching: every patch of the previous frame, in its list order, claims the unmatched patch of the same view whose
  // bounding box overlaps its own most (intersection over union, float as in the reference, first maximum wins)
  std::vector<int> matchOf( P, -1 ), list;
  list.reserve( P );
  for ( int id = 0; id < Pprev; ++id ) {
    const tmc2_patch& q      = prev[id];
    float             maxIou = 0.0F;
    int               best   = -1;
    for ( int c = 0; c < P; ++c ) {
      const tmc2_patch& r = pt[sorted[c]];
      if ( q.viewId != r.viewId || matchOf[sorted[c]] != -1 ) continue;
      const int x1 = std::max( q.u1, r.u1 ), y1 = std::max( q.v1, r.v1 );
      int       w  = std::min( q.u1 + q.sizeU, r.u1 + r.sizeU ) - x1, h = std::min( q.v1 + q.sizeV, r.v1 + r.sizeV ) - y1;
      if ( w <= 0 || h <= 0 ) w = h = 0;
      const int   inter = w * h, uni = q.sizeU * q.sizeV + r.sizeU * r.sizeV - inter;
      const float iou   = static_cast<float>( inter ) / uni;
      if ( iou > maxIou ) {
        maxIou = iou;
        best   = c;
      }
    }
    if ( maxIou > 0.2F ) {
      matchOf[sorted[best]] = id;
      list.push_back( sorted[best] );
    }
  }
  for ( int c = 0; c < P; ++c )
    if ( matchOf[sorted[c]] == -1 ) list.push_back( sorted[c] );
  for ( int k = 0; k < P; ++k ) {
    order[k] = list[k];
    match[k] = matchOf[list[k]];
    sizeU    = std::max( sizeU, size_t( pt[list[k]].sizeU0 + 1 ) );
  }
  const int tileH = int( ( int( sizeU ) / numTilesHor ) * ratio );
  sizeV           = std::max( sizeV, size_t( std::max( tileH, 0 ) ) );
  size_t      heightBlocks = sizeV;
  BlockCanvas canvas( sizeU, sizeV );
  auto boxOf = [&]( const tmc2_patch& p, size_t& w, size_t& h ) {
    w = p.patchOrientation == ORIENT_DEFAULT ? p.sizeU0 : p.sizeV0;
    h = p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0;
  };
  for ( int k = 0; k < P; ++k ) {
    tmc2_patch& p      = pt[list[k]];
    bool        placed = false;
    size_t      w, h;
    while ( !placed ) {
      if ( match[k] != -1 ) {
        // same orientation as the match; its position first, then the first free position in raster order
        const tmc2_patch& q = prev[match[k]];
        p.patchOrientation  = q.patchOrientation;
        boxOf( p, w, h );
        if ( q.u0 >= 0 && q.v0 >= 0 && canvas.boxFree( size_t( q.u0 ), size_t( q.v0 ), w, h ) ) {
          p.u0   = q.u0;
          p.v0   = q.v0;
          placed = true;
        }
        for ( size_t v = 0; v < canvas.height && !placed; ++v )
          for ( size_t u = 0; u < canvas.width && !placed; ++u )
            if ( canvas.boxFree( u, v, w, h ) ) {
              p.u0   = int32_t( u );
              p.v0   = int32_t( v );
              placed = true;
            }
      } else {
        const bool wide   = p.sizeU0 > p.sizeV0;
        const int  first  = wide ? ORIENT_SWAP : ORIENT_DEFAULT;
        const int  second = wide ? ORIENT_DEFAULT : ORIENT_SWAP;
        for ( size_t v = 0; v < canvas.height && !placed; ++v )
          for ( size_t u = 0; u < canvas.width && !placed; ++u )
            for ( int o = 0; o < 2 && !placed; ++o ) {
              p.patchOrientation = o == 0 ? first : second;
              boxOf( p, w, h );
              if ( canvas.boxFree( u, v, w, h ) ) {
                p.u0   = int32_t( u );
                p.v0   = int32_t( v );
                placed = true;
              }
            }
      }
      if ( !placed ) {
        if ( canvas.height > ( size_t( 1 ) << 20 ) ) return -1;  // fits at no canvas height (e.g. the orientation inherited from the
                                                                  // match makes it wider than the canvas): the reference spins here
        canvas.grow( canvas.height * 2 );
      }
    }
    const uint8_t* o = occ + p.occOffset;
    for ( int vb = 0; vb < p.sizeV0; ++vb )
      for ( int ub = 0; ub < p.sizeU0; ++ub )
        if ( o[vb * p.sizeU0 + ub] ) {
          if ( p.patchOrientation == ORIENT_DEFAULT )
            canvas.set( size_t( p.u0 + ub ), size_t( p.v0 + vb ) );
          else
            canvas.set( size_t( p.u0 + vb ), size_t( p.v0 + ub ) );
        }
    heightBlocks = std::max( heightBlocks, size_t( p.v0 + ( p.patchOrientation == ORIENT_DEFAULT ? p.sizeV0 : p.sizeU0 ) ) );
  }
  return int( heightBlocks ) * occRes;
}

int packSpatialConsistencyHost( tmc2_frame* f, tmc2_frame* prevFrame, int presetWidth, int occRes, int numTilesHor, double ratio ) {
  if ( !f->havePatches || !prevFrame->havePacking ) {
    setError( "packSpatialConsistency: this frame has no patches or the previous frame is not packed" );
    return TMC2_E_STATE;
  }
  const int P = int( f->patches.size() );
  f->packOrder.assign( P, 0 );
  f->packMatch.assign( P, -1 );
  f->packedHeight = 0;
  f->packedWidth  = ( presetWidth / occRes ) * occRes;
  f->havePacking  = true;
  if ( P == 0 ) return TMC2_OK;
  std::vector<uint8_t> occ( size_t( f->occCount ) );
  TMC2_HIP( hipMemcpyAsync( occ.data(), f->d_occupancy.p, occ.size(), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  std::vector<tmc2_patch> prevList( prevFrame->patches.size() );
  for ( size_t k = 0; k < prevList.size(); ++k ) prevList[k] = prevFrame->patches[prevFrame->packOrder[k]];
  f->packedHeight = packSpatialConsistencyCore( f->patches.data(), P, occ.data(), prevList.data(), int( prevList.size() ),
                                                presetWidth, occRes, numTilesHor, ratio, f->packOrder.data(),
                                                f->packMatch.data() );
  {  // spatialConsistencyPackFlexible updates the tile width through a reference (:1190, :1308): the canvas it packed on
    int sizeU = presetWidth / occRes;
    for ( auto& p : f->patches ) sizeU = std::max( sizeU, p.sizeU0 + 1 );
    f->packedWidth = sizeU * occRes;
  }
  if ( f->packedHeight < 0 ) {
    f->havePacking = false;
    setError( "packSpatialConsistency: a patch fits at no canvas height" );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

int tmc2_encoder_pack_flexible( tmc2_frame* f, int presetWidth, int numTilesHor, double tileHeightToWidthRatio,
                                int32_t* height ) {
  if ( !f || presetWidth <= 0 || numTilesHor <= 0 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( tmc2::packFlexibleHost( f, presetWidth, 16, numTilesHor, tileHeightToWidthRatio ) );
  if ( height ) *height = f->packedHeight;
  return TMC2_OK;
}

int tmc2_encoder_pack_spatial_consistency( tmc2_frame* f, tmc2_frame* previous, int presetWidth, int numTilesHor,
                                           double tileHeightToWidthRatio, int32_t* height ) {
  if ( !f || !previous || presetWidth <= 0 || numTilesHor <= 0 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( tmc2::packSpatialConsistencyHost( f, previous, presetWidth, 16, numTilesHor, tileHeightToWidthRatio ) );
  if ( height ) *height = f->packedHeight;
  return TMC2_OK;
}

int tmc2_frame_get_packed_size( tmc2_frame* f, int32_t* width, int32_t* height ) {
  if ( !f || !f->havePacking ) {
    tmc2::setError( "get_packed_size: frame not packed" );
    return TMC2_E_STATE;
  }
  if ( width ) *width = f->packedWidth;
  if ( height ) *height = f->packedHeight;
  return TMC2_OK;
}

int tmc2_frame_get_patch_matches( tmc2_frame* f, int32_t* matches ) {
  if ( !f || !matches || !f->havePacking ) {
    tmc2::setError( "get_patch_matches: frame not packed" );
    return TMC2_E_STATE;
  }
  for ( size_t k = 0; k < f->packOrder.size(); ++k ) matches[k] = k < f->packMatch.size() ? f->packMatch[k] : -1;
  return TMC2_OK;
}

int tmc2_host_pack_flexible( tmc2_patch* patches, int count, const uint8_t* occupancy, int presetWidth, int numTilesHor,
                             double tileHeightToWidthRatio, int32_t* order, int32_t* height ) {
  if ( count < 0 || presetWidth <= 0 || numTilesHor <= 0 || !order || !height || ( count && ( !patches || !occupancy ) ) )
    return TMC2_E_INVALID;
  *height = tmc2::packFlexibleCore( patches, count, occupancy, presetWidth, 16, numTilesHor, tileHeightToWidthRatio, order );
  if ( *height < 0 ) {
    tmc2::setError( "host_pack_flexible: a patch fits at no canvas height" );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}

int tmc2_host_place_segments( int frames, const int32_t* counts, tmc2_patch* patches, const uint8_t* occupancy,
                              const int64_t* occupancyBase, int mode, int minimumImageWidth, int minimumImageHeight, int numTilesHor,
                              double tileHeightToWidthRatio, int32_t* matches, uint8_t* occupancyOut, int64_t occupancyOutCapacity,
                              int64_t* occupancyOutBase, int32_t* widths, int32_t* heights ) {
  if ( frames <= 0 || !counts || !occupancyBase || !occupancyOutBase || mode < 0 || mode > 2 || minimumImageWidth <= 0 ||
       minimumImageHeight <= 0 || numTilesHor <= 0 )
    return TMC2_E_INVALID;
  const int                     occRes = 16;
  std::vector<tmc2::GpaFrameIO> io;
  io.resize( size_t( frames ) );
  size_t at = 0;
  for ( int f = 0; f < frames; ++f ) {
    const int P = counts[f];
    if ( P < 0 || ( P && ( !patches || !occupancy || !matches ) ) ) return TMC2_E_INVALID;
    tmc2::GpaFrameIO&       g = io[size_t( f )];
    std::vector<tmc2_patch> pt( patches + at, patches + at + P );
    int64_t                 bytes = 0;
    for ( auto& p : pt ) {
      if ( p.sizeU0 <= 0 || p.sizeV0 <= 0 || p.occOffset < 0 ) {
        tmc2::setError( "host_place_segments: frame %d holds a record without a block box", f );
        return TMC2_E_INVALID;
      }
      bytes = std::max( bytes, p.occOffset + int64_t( p.sizeU0 ) * p.sizeV0 );
    }
    g.occ.assign( occupancy + occupancyBase[f], occupancy + occupancyBase[f] + bytes );
    std::vector<int32_t> order( size_t( P ), 0 );
    g.match.assign( size_t( P ), -1 );
    const bool chained = mode > 0 && f > 0;
    // packFlexible leaves the tile width alone (it works on a copy, PCCEncoder.cpp:2312); the chained packer writes the
    // width of the canvas it packed on back into the tile (:1190, :1308)
    g.width = minimumImageWidth;
    if ( chained ) {
      int sizeU = minimumImageWidth / occRes;
      for ( auto& p : pt ) sizeU = std::max( sizeU, p.sizeU0 + 1 );
      g.width = sizeU * occRes;
    }
    g.height = 0;
    if ( P ) {
      g.height = chained ? tmc2::packSpatialConsistencyCore( pt.data(), P, g.occ.data(), io[size_t( f ) - 1].list.data(),
                                                             int( io[size_t( f ) - 1].list.size() ), minimumImageWidth, occRes, numTilesHor,
                                                             tileHeightToWidthRatio, order.data(), g.match.data() )
                         : tmc2::packFlexibleCore( pt.data(), P, g.occ.data(), minimumImageWidth, occRes, numTilesHor,
                                                   tileHeightToWidthRatio, order.data() );
      if ( g.height < 0 ) {
        tmc2::setError( "host_place_segments: frame %d: a patch fits at no canvas height", f );
        return TMC2_E_INVALID;
      }
    }
    g.list.resize( size_t( P ) );
    for ( int k = 0; k < P; ++k ) g.list[size_t( k )] = pt[size_t( order[size_t( k )] )];
    at += size_t( P );
  }
  if ( mode == 2 ) {
    int tileW = minimumImageWidth, tileH = minimumImageHeight;  // resizeTileGeometryVideo ran before the allocation
    for ( auto& g : io ) tileW = std::max( tileW, g.width ), tileH = std::max( tileH, g.height );
    for ( auto& g : io ) g.width = tileW, g.height = tileH;
    TMC2_TRY( tmc2::globalPatchAllocationCore( io, minimumImageWidth, minimumImageHeight, occRes ) );
  }
  int64_t total = 0;
  for ( auto& g : io ) total += int64_t( g.occ.size() );
  occupancyOutBase[frames] = total;
  if ( total > occupancyOutCapacity || ( total && !occupancyOut ) ) {
    tmc2::setError( "host_place_segments: the occupancy pools need %lld bytes", (long long)total );
    return TMC2_E_INVALID;
  }
  at           = 0;
  int64_t base = 0;
  for ( int f = 0; f < frames; ++f ) {
    tmc2::GpaFrameIO& g = io[size_t( f )];
    occupancyOutBase[f] = base;
    std::copy( g.list.begin(), g.list.end(), patches + at );
    std::copy( g.match.begin(), g.match.end(), matches + at );
    if ( !g.occ.empty() ) std::memcpy( occupancyOut + base, g.occ.data(), g.occ.size() );
    base += int64_t( g.occ.size() );
    at += g.list.size();
    if ( widths ) widths[f] = g.width;
    if ( heights ) heights[f] = g.height;
  }
  return TMC2_OK;
}

int tmc2_host_pack_spatial_consistency( tmc2_patch* patches, int count, const uint8_t* occupancy, const tmc2_patch* previous,
                                        int previousCount, int presetWidth, int numTilesHor, double tileHeightToWidthRatio,
                                        int32_t* order, int32_t* matches, int32_t* height ) {
  if ( count < 0 || previousCount < 0 || presetWidth <= 0 || numTilesHor <= 0 || !order || !matches || !height ||
       ( count && ( !patches || !occupancy ) ) || ( previousCount && !previous ) )
    return TMC2_E_INVALID;
  *height = tmc2::packSpatialConsistencyCore( patches, count, occupancy, previous, previousCount, presetWidth, 16, numTilesHor,
                                              tileHeightToWidthRatio, order, matches );
  if ( *height < 0 ) {
    tmc2::setError( "host_pack_spatial_consistency: a patch fits at no canvas height" );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}

int tmc2_frame_get_patch_order( tmc2_frame* f, int32_t* order ) {
  if ( !f || !order || !f->havePacking ) {
    tmc2::setError( "get_patch_order: frame not packed" );
    return TMC2_E_STATE;
  }
  std::copy( f->packOrder.begin(), f->packOrder.end(), order );
  return TMC2_OK;
}

int tmc2_encoder_canvas_size( const int32_t* frameHeights, int frames, int tileWidth, int minimumImageWidth,
                              int minimumImageHeight, int32_t* width, int32_t* height ) {
  if ( !frameHeights || frames <= 0 || !width || !height ) return TMC2_E_INVALID;
  int w = std::max( tileWidth, minimumImageWidth ), h = minimumImageHeight;
  for ( int i = 0; i < frames; ++i ) h = std::max( h, frameHeights[i] );
  *width  = int32_t( std::ceil( double( w ) / 64.0 ) * 64 );
  *height = int32_t( std::ceil( double( h ) / 64.0 ) * 64 );
  return TMC2_OK;
}
}
