// attributes.hip -- phase B on gfx950: point-cloud reconstruction (S17), colour transfer (S18), attribute scatter
// (S20), push-pull background fill (S21) and attribute group dilation (S22).
//
// Replaces (reference: source/lib/...)
//   PCCCodec::generatePointCloud, lossy CTC branch      PccLibCommon/source/PCCCodec.cpp:519-980 (+ generatePoints :329-517,
//                                                       PCCPatch::generatePoint PccLibCommon/include/PCCPatch.h:177-207)
//   PCCPointSet3::transferColors (CTC settings)         PccLibCommon/source/PCCPointSet.cpp:807-1124
//   PCCEncoder::presmoothPointCloudColor                PccLibEncoder/source/PCCEncoder.cpp:6593-6655 -- a no-op in the reference
//                                                       build (boundary type 2 is never produced), therefore skipped
//   PCCEncoder::generateAttributeVideo (per tile)       PCCEncoder.cpp:6736-6794
//   PCCEncoder::dilateSmoothedPushPull / pushPullMip / pushPullFill / mean4w   PCCEncoder.cpp:6357-6591
//   attribute group dilation, inline in PCCEncoder::encode                      PCCEncoder.cpp:380-402
//
// S17 is a stream compaction whose ORDER is part of the contract (patch list order, block raster, pixel raster, D0
// before D1): one workgroup per 16x16 patch block counts its points (ballot + popcount), a prefix sum over the tile
// list gives each tile its output offset, and a second pass emits with an in-tile prefix (wave scan).
// S18 reuses the exact nanoflann-order k-NN kernel: 8-NN of every reconstructed point in the SOURCE tree (the frame's
// own tree) and 1-NN of every source point in a tree built over the reconstruction; the backward votes are bucketed
// per target (count, scan, fill), ordered by (distance, source index) -- the order the reference's insertion sort
// leaves for its <=16-element lists -- and reduced in fp64 in that order.
// S21: every pyramid level is an image-parallel kernel over all six planes (2 maps x RGB) at once.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "internal.h"
#include "cand_sort.h"

namespace tmc2 {
namespace {

__device__ __forceinline__ void toCanvasB( const PlaceDev& p, int u, int v, int& x, int& y ) {
  if ( p.orient == 0 ) {
    x = u + p.u0 * 16;
    y = v + p.v0 * 16;
  } else {
    x = v + p.u0 * 16;
    y = u + p.v0 * 16;
  }
}

__device__ __forceinline__ int normalCoord( const PlaceDev& p, int depth ) {
  return p.mode == 0 ? depth + p.d1 : max( 0, p.d1 - depth );
}

// EMIT = false: tileCount[tile] = number of points of the tile.  EMIT = true: write them at tileOffset[tile].
template <bool EMIT>
__global__ __launch_bounds__( 256 ) void reconTileKernel( const PlaceDev* __restrict__ place,
                                                           const uint32_t* __restrict__ tilePatch,
                                                           const uint8_t* __restrict__ occVideo,
                                                           const uint32_t* __restrict__ blockToPatch,
                                                           const uint16_t* __restrict__ geo, int W, int H, int prec,
                                                           uint32_t* __restrict__ tileCount,
                                                           const uint32_t* __restrict__ tileOffset, Pt* __restrict__ recon,
                                                           uint32_t* __restrict__ pointToPixel ) {
  __shared__ uint32_t waveTotal[4];
  const uint32_t      tile  = blockIdx.x;
  const uint32_t      k     = tilePatch[tile];
  const PlaceDev      p     = place[k];
  const int           local = int( tile ) - p.tileBase;
  const int           ub = local % p.sizeU0, vb = local / p.sizeU0;
  const int           bx = p.orient == 0 ? ub + p.u0 : vb + p.u0, by = p.orient == 0 ? vb + p.v0 : ub + p.v0;
  const bool          owned = blockToPatch[size_t( by ) * ( W / 16 ) + bx] == k + 1;
  const int           u = ub * 16 + int( threadIdx.x & 15 ), v = vb * 16 + int( threadIdx.x >> 4 );
  int                 x, y;
  toCanvasB( p, u, v, x, y );
  uint32_t cnt = 0;
  int      c0 = 0, c1 = 0;
  if ( owned && x < W && y < H && occVideo[size_t( y / prec ) * ( W / prec ) + x / prec] ) {
    c0  = normalCoord( p, geo[size_t( y ) * W + x] );
    c1  = normalCoord( p, geo[size_t( W ) * H + size_t( y ) * W + x] );
    cnt = c1 != c0 ? 2u : 1u;
  }
  // inclusive scan of cnt over the 256 lanes (pixel raster order == lane order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t  inc  = cnt;
#pragma unroll
  for ( int off = 1; off < 64; off <<= 1 ) {
    const uint32_t t = __shfl_up( inc, off, 64 );
    if ( lane >= off ) inc += t;
  }
  if ( lane == 63 ) waveTotal[wave] = inc;
  __syncthreads();
  if ( !EMIT ) {
    if ( threadIdx.x == 0 ) tileCount[tile] = waveTotal[0] + waveTotal[1] + waveTotal[2] + waveTotal[3];
    return;
  }
  if ( cnt == 0 ) return;
  uint32_t off = tileOffset[tile] + inc - cnt;
  for ( int w = 0; w < wave; ++w ) off += waveTotal[w];
  int c[3];
  c[p.axT] = u + p.u1;
  c[p.axB] = v + p.v1;
  c[p.axN] = c0;
  recon[off]        = Pt{int16_t( c[0] ), int16_t( c[1] ), int16_t( c[2] ), 0};
  pointToPixel[off] = packPixel( x, y, 0, cnt == 2 );
  if ( cnt == 2 ) {
    c[p.axN]              = c1;
    recon[off + 1]        = Pt{int16_t( c[0] ), int16_t( c[1] ), int16_t( c[2] ), 0};
    pointToPixel[off + 1] = packPixel( x, y, 1, false );
  }
}

// ---- S18 ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t toU8( double v ) { return uint8_t( fmax( 0.0, fmin( round( v ), 255.0 ) ) ); }

// easy: non-null = the first result of the queries that have an identical source point (0xFFFFFFFF for the others: only
// their rows of idx8 / dist8 are filled -- launchKnnSplit)
__global__ __launch_bounds__( 256 ) void forwardColorKernel( const uint32_t* __restrict__ idx8, const uint32_t* __restrict__ dist8,
                                                              const uint32_t* __restrict__ easy,
                                                              const uint8_t* __restrict__ srcRgb4, uint32_t m,
                                                              uint8_t* __restrict__ fwdRgb4 ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= m ) return;
  if ( easy && easy[t] != 0xFFFFFFFFu ) {  // "dist < 0.0001": an identical source point exists, take its colour
    reinterpret_cast<uchar4*>( fwdRgb4 )[t] = reinterpret_cast<const uchar4*>( srcRgb4 )[easy[t]];
    return;
  }
  const uint4* ir = reinterpret_cast<const uint4*>( idx8 + size_t( t ) * 8 );
  const uint4* dr = reinterpret_cast<const uint4*>( dist8 + size_t( t ) * 8 );
  const uint4  i0 = ir[0], i1 = ir[1], d0 = dr[0], d1 = dr[1];
  const uint32_t id[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
  const uint32_t ds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
  uchar4         out;
  if ( ds[0] == 0 ) {  // "dist < 0.0001": an identical source point exists, take its colour
    out = reinterpret_cast<const uchar4*>( srcRgb4 )[id[0]];
  } else {
    double r = 0.0, g = 0.0, b = 0.0, sw = 0.0;
#pragma unroll
    for ( int i = 0; i < 8; ++i ) {
      const double w = __ddiv_rn( 1.0, double( ds[i] ) + 4.0 );
      const uchar4 c = reinterpret_cast<const uchar4*>( srcRgb4 )[id[i]];
      r += double( c.x ) * w;
      g += double( c.y ) * w;
      b += double( c.z ) * w;
      sw += w;
    }
    out = make_uchar4( toU8( __ddiv_rn( r, sw ) ), toU8( __ddiv_rn( g, sw ) ), toU8( __ddiv_rn( b, sw ) ), 0 );
  }
  reinterpret_cast<uchar4*>( fwdRgb4 )[t] = out;
}

__global__ __launch_bounds__( 256 ) void backwardCountKernel( const uint32_t* __restrict__ idx1, uint32_t n,
                                                               uint32_t* __restrict__ count ) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if ( s < n ) atomicAdd( &count[idx1[s]], 1u );
}

__global__ __launch_bounds__( 256 ) void backwardFillKernel( const uint32_t* __restrict__ idx1, const uint32_t* __restrict__ dist1,
                                                              const uint32_t* __restrict__ offset, uint32_t n,
                                                              uint32_t* __restrict__ cursor, uint2* __restrict__ entries ) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if ( s >= n ) return;
  const uint32_t t          = idx1[s];
  const uint32_t slot       = atomicAdd( &cursor[t], 1u );
  entries[offset[t] + slot] = make_uint2( dist1[s], s );
}

__global__ __launch_bounds__( 256 ) void combineColorKernel( const uint32_t* __restrict__ count, const uint32_t* __restrict__ offset,
                                                              uint2* __restrict__ entries, const uint8_t* __restrict__ srcRgb4,
                                                              const uint8_t* __restrict__ fwdRgb4, uint32_t m,
                                                              uint8_t* __restrict__ outRgb4, uint32_t* __restrict__ error ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= m ) return;
  const int n = int( count[t] );
  if ( n == 0 ) {
    reinterpret_cast<uchar4*>( outRgb4 )[t] = reinterpret_cast<const uchar4*>( fwdRgb4 )[t];
    return;
  }
  uint2* e = entries + offset[t];
  // (a) source-index order (keys unique): plain insertion sort
  for ( int i = 1; i < n; ++i ) {
    const uint2 v = e[i];
    int         k = i - 1;
    while ( k >= 0 && e[k].y > v.y ) {
      e[k + 1] = e[k];
      --k;
    }
    e[k + 1] = v;
  }
  // (b) std::sort by distance
  const CandSort cs{e};
  if ( !cs.sort( n ) ) *error = 1;
  double r = 0.0, g = 0.0, b = 0.0, sw = 0.0;
  if ( e[0].x == 0 || n == 1 ) {  // an identical source point, or a single candidate: its colour, unweighted
    const uchar4 c = reinterpret_cast<const uchar4*>( srcRgb4 )[e[0].y];
    r = double( c.x ), g = double( c.y ), b = double( c.z );
  } else {
    for ( int k = 0; k < n; ++k ) {
      const uchar4 c = reinterpret_cast<const uchar4*>( srcRgb4 )[e[k].y];
      const double w = __ddiv_rn( 1.0, __dsqrt_rn( double( e[k].x ) ) + 4.0 );
      r += double( c.x ) * w;
      g += double( c.y ) * w;
      b += double( c.z ) * w;
      sw += w;
    }
    r = __ddiv_rn( r, sw );
    g = __ddiv_rn( g, sw );
    b = __ddiv_rn( b, sw );
  }
  const uchar4 f = reinterpret_cast<const uchar4*>( fwdRgb4 )[t];
  // fixWeight: w = 0  ->  round( 0 * centroid1 + 1 * centroid2 )
  reinterpret_cast<uchar4*>( outRgb4 )[t] = make_uchar4( toU8( 0.0 * double( f.x ) + 1.0 * r ), toU8( 0.0 * double( f.y ) + 1.0 * g ),
                                                         toU8( 0.0 * double( f.z ) + 1.0 * b ), 0 );
}

// ---- S20 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void attributeScatterKernel( const uint8_t* __restrict__ rgb4,
                                                                  const uint32_t* __restrict__ pointToPixel, uint32_t m,
                                                                  int W, int H, uint8_t* __restrict__ attr ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= m ) return;
  const uint32_t pp = pointToPixel[i];
  const size_t   px = size_t( pixelY( pp ) ) * W + pixelX( pp ), plane = size_t( W ) * H;
  const uchar4   c  = reinterpret_cast<const uchar4*>( rgb4 )[i];
  const bool     layer1 = pixelLayer( pp ), hasD1 = pixelHasD1( pp );
  if ( !layer1 ) {
    attr[px] = c.x, attr[plane + px] = c.y, attr[2 * plane + px] = c.z;
    if ( !hasD1 ) attr[3 * plane + px] = c.x, attr[4 * plane + px] = c.y, attr[5 * plane + px] = c.z;
  } else {
    attr[3 * plane + px] = c.x, attr[4 * plane + px] = c.y, attr[5 * plane + px] = c.z;
  }
}

__global__ __launch_bounds__( 256 ) void upsampleOccupancyKernel( const uint8_t* __restrict__ occVideo, int W, int H, int prec,
                                                                   uint8_t* __restrict__ occ ) {
  const size_t c = size_t( blockIdx.x ) * blockDim.x + threadIdx.x;
  if ( c >= size_t( W ) * H ) return;
  occ[c] = occVideo[size_t( ( c / W ) / prec ) * ( W / prec ) + ( c % W ) / prec];
}

// ---- S21 ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mean4w( int p1, int w1, int p2, int w2, int p3, int w3, int p4, int w4 ) {
  return ( p1 * w1 + p2 * w2 + p3 * w3 + p4 * w4 ) / ( w1 + w2 + w3 + w4 );
}

// six planes (2 maps x 3 channels) of size W*H each, plane stride = W*H
// (planes p0 .. p1 - 1 of the six: all of them, or one per workgroup in the coarse-levels kernel)
__device__ __forceinline__ void pushPullMipPixel( int i, const uint8_t* img, const uint8_t* occ, int W, int H, uint8_t* mip,
                                                  uint8_t* mipOcc, int w, int h, int p0 = 0, int p1 = 6 ) {
  const int  x = i % w, y = i / w, X = 2 * x, Y = 2 * y;
  const bool i2 = X + 1 < W, i3 = Y + 1 < H;
  const int  w1 = occ[size_t( Y ) * W + X] ? 255 : 0;
  const int  w2 = ( i2 && occ[size_t( Y ) * W + X + 1] ) ? 255 : 0;
  const int  w3 = ( i3 && occ[size_t( Y + 1 ) * W + X] ) ? 255 : 0;
  const int  w4 = ( i2 && i3 && occ[size_t( Y + 1 ) * W + X + 1] ) ? 255 : 0;
  const bool any = ( w1 + w2 + w3 + w4 ) > 0;
  mipOcc[i]      = any ? 1 : 0;
  for ( int p = p0; p < p1; ++p ) {
    const uint8_t* s = img + size_t( p ) * W * H;
    uint8_t        v = 0;
    if ( any ) {
      const int v1 = s[size_t( Y ) * W + X], v2 = i2 ? s[size_t( Y ) * W + X + 1] : 0;
      const int v3 = i3 ? s[size_t( Y + 1 ) * W + X] : 0, v4 = ( i2 && i3 ) ? s[size_t( Y + 1 ) * W + X + 1] : 0;
      v            = uint8_t( mean4w( v1, w1, v2, w2, v3, w3, v4, w4 ) );
    }
    mip[size_t( p ) * w * h + i] = v;
  }
}
__global__ __launch_bounds__( 256 ) void pushPullMipKernel( const uint8_t* __restrict__ img, const uint8_t* __restrict__ occ, int W,
                                                             int H, uint8_t* __restrict__ mip, uint8_t* __restrict__ mipOcc, int w,
                                                             int h ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < w * h ) pushPullMipPixel( i, img, occ, W, H, mip, mipOcc, w, h );
}

__device__ __forceinline__ void pushPullFillPixel( int i, uint8_t* img, const uint8_t* occ, int W, int H, const uint8_t* mip, int w,
                                                   int h, int p0 = 0, int p1 = 6 ) {
  if ( occ[i] ) return;
  const int  X = i % W, Y = i / W, x = X >> 1, y = Y >> 1;
  const int  dx = ( X & 1 ) ? 1 : -1, dy = ( Y & 1 ) ? 1 : -1;
  const bool hx = dx < 0 ? x > 0 : x < w - 1, hy = dy < 0 ? y > 0 : y < h - 1;
  for ( int p = p0; p < p1; ++p ) {
    const uint8_t* m  = mip + size_t( p ) * w * h;
    const int      v  = m[size_t( y ) * w + x];
    const int      vx = hx ? m[size_t( y ) * w + x + dx] : 0;
    const int      vy = hy ? m[size_t( y + dy ) * w + x] : 0;
    const int      vd = ( hx && hy ) ? m[size_t( y + dy ) * w + x + dx] : 0;
    img[size_t( p ) * W * H + i] = uint8_t( mean4w( v, 144, vx, hx ? 48 : 0, vy, hy ? 48 : 0, vd, ( hx && hy ) ? 16 : 0 ) );
  }
}
__global__ __launch_bounds__( 256 ) void pushPullFillKernel( uint8_t* __restrict__ img, const uint8_t* __restrict__ occ, int W, int H,
                                                              const uint8_t* __restrict__ mip, int w, int h ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < W * H ) pushPullFillPixel( i, img, occ, W, H, mip, w, h );
}

__device__ __forceinline__ void pushPullBlurPixel( int i, const uint8_t* src, uint8_t* dst, const uint8_t* occ, int W, int H, int p0 = 0,
                                                   int p1 = 6 ) {
  if ( occ[i] ) return;
  const int x = i % W, y = i / W;
  const int x1 = x > 0 ? x - 1 : x, y1 = y > 0 ? y - 1 : y, x2 = x < W - 1 ? x + 1 : x, y2 = y < H - 1 ? y + 1 : y;
  for ( int p = p0; p < p1; ++p ) {
    const uint8_t* s   = src + size_t( p ) * W * H;
    const int      sum = s[size_t( y1 ) * W + x1] + s[size_t( y1 ) * W + x2] + s[size_t( y2 ) * W + x1] + s[size_t( y2 ) * W + x2] +
                    s[size_t( y ) * W + x1] + s[size_t( y ) * W + x2] + s[size_t( y1 ) * W + x] + s[size_t( y2 ) * W + x];
    dst[size_t( p ) * W * H + i] = uint8_t( ( sum + 4 ) >> 3 );
  }
}
__global__ __launch_bounds__( 256 ) void pushPullBlurKernel( const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              const uint8_t* __restrict__ occ, int W, int H ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < W * H ) pushPullBlurPixel( i, src, dst, occ, W, H );
}

// One level of the way up in ONE launch: the fill from the coarser level and all `iters` blur passes, a 96 x 96 tile of one
// plane per workgroup, in LDS.  The tile is loaded with a margin of `iters` pixels (clipped to the image): a blur pass reads
// the 8 neighbours, so what a pass computes next to a cut edge of the region is wrong one pixel further in per pass -- and
// after `iters` passes has just not reached the tile.  Coordinates clamp at the region's edges, which are the image's edges
// wherever the clamp is the reference's.  Occupied pixels never change; only unoccupied pixels of the tile are written, and no
// workgroup reads a pixel another one writes (unoccupied pixels are computed from the coarser level, not loaded).
constexpr int kPushPullTile = 96, kPushPullSpan = 128, kPushPullMaxIters = ( kPushPullSpan - kPushPullTile ) / 2;
__global__ __launch_bounds__( 1024 ) void pushPullFillBlurKernel( uint8_t* __restrict__ img, const uint8_t* __restrict__ occ, int W,
                                                                   int H, const uint8_t* __restrict__ mip, int w, int h, int iters,
                                                                   int tilesX ) {
  __shared__ uint8_t sOcc[kPushPullSpan * kPushPullSpan];
  __shared__ uint8_t sBuf[2][kPushPullSpan * kPushPullSpan];
  const int      X0 = ( blockIdx.x % tilesX ) * kPushPullTile, Y0 = ( blockIdx.x / tilesX ) * kPushPullTile;
  const int      RX0 = max( X0 - iters, 0 ), RY0 = max( Y0 - iters, 0 );
  const int      RW = min( X0 + kPushPullTile + iters, W ) - RX0, RH = min( Y0 + kPushPullTile + iters, H ) - RY0;
  uint8_t*       plane = img + size_t( blockIdx.y ) * W * H;
  const uint8_t* m     = mip + size_t( blockIdx.y ) * w * h;
  for ( int i = threadIdx.x; i < kPushPullSpan * RH; i += 1024 ) {
    const int rx = i & ( kPushPullSpan - 1 ), ry = i >> 7;
    if ( rx >= RW ) continue;
    const int     X = RX0 + rx, Y = RY0 + ry;
    const size_t  g = size_t( Y ) * W + X;
    const uint8_t o = occ[g];
    uint8_t       v;
    if ( o ) {
      v = plane[g];
    } else {  // pushPullFillPixel, this plane
      const int  x = X >> 1, y = Y >> 1, dx = ( X & 1 ) ? 1 : -1, dy = ( Y & 1 ) ? 1 : -1;
      const bool hx = dx < 0 ? x > 0 : x < w - 1, hy = dy < 0 ? y > 0 : y < h - 1;
      const int  c  = m[size_t( y ) * w + x];
      const int  vx = hx ? m[size_t( y ) * w + x + dx] : 0;
      const int  vy = hy ? m[size_t( y + dy ) * w + x] : 0;
      const int  vd = ( hx && hy ) ? m[size_t( y + dy ) * w + x + dx] : 0;
      v             = uint8_t( mean4w( c, 144, vx, hx ? 48 : 0, vy, hy ? 48 : 0, vd, ( hx && hy ) ? 16 : 0 ) );
    }
    sOcc[i]    = o;
    sBuf[0][i] = v;
    sBuf[1][i] = v;
  }
  __syncthreads();
  int cur = 0;
  for ( int it = 0; it < iters; ++it ) {
    const uint8_t* src = sBuf[cur];
    uint8_t*       dst = sBuf[cur ^ 1];
    for ( int i = threadIdx.x; i < kPushPullSpan * RH; i += 1024 ) {
      const int rx = i & ( kPushPullSpan - 1 ), ry = i >> 7;
      if ( rx >= RW || sOcc[i] ) continue;
      const int l = rx > 0 ? -1 : 0, r = rx < RW - 1 ? 1 : 0;
      const int u = ry > 0 ? -kPushPullSpan : 0, d = ry < RH - 1 ? kPushPullSpan : 0;
      const int sum = src[i + u + l] + src[i + u + r] + src[i + d + l] + src[i + d + r] + src[i + l] + src[i + r] + src[i + u] +
                      src[i + d];
      dst[i] = uint8_t( ( sum + 4 ) >> 3 );
    }
    __syncthreads();
    cur ^= 1;
  }
  const uint8_t* res = sBuf[cur];
  for ( int i = threadIdx.x; i < kPushPullSpan * RH; i += 1024 ) {
    const int rx = i & ( kPushPullSpan - 1 ), ry = i >> 7;
    const int X = RX0 + rx, Y = RY0 + ry;
    if ( rx >= RW || sOcc[i] || X < X0 || X >= X0 + kPushPullTile || Y < Y0 || Y >= Y0 + kPushPullTile ) continue;
    plane[size_t( Y ) * W + X] = res[i];
  }
}

// The coarse end of the pyramid in ONE workgroup: the levels of at most kPushPullSmall pixels -- their mip maps on the way
// down, and on the way up the fill, the ping-pong partner's copy and the 4, 5, ... blur iterations of every level.  That is
// ~ 60 of the ~ 100 launches of the padding, each over a few hundred to a few thousand pixels; between the steps a
// workgroup barrier does what a kernel boundary did (the images are tiny: L1 / L2 resident, one CU).
constexpr int kPushPullSmall  = 160 * 160;
constexpr int kPushPullLevels = 16;
struct PushPullLevels {
  int      count;                   // levels of the pyramid, 0 = the canvas
  int      first;                   // the first (finest) level handled here: every level >= first has at most kPushPullSmall pixels
  int      w[kPushPullLevels], h[kPushPullLevels];
  uint8_t *img[kPushPullLevels], *tmp[kPushPullLevels], *occ[kPushPullLevels];
};
__global__ __launch_bounds__( 1024 ) void pushPullCoarseLevelsKernel( PushPullLevels L ) {
  // The six planes (2 maps x 3 channels) never read each other: one workgroup per plane, six CUs instead of one for the ~ 60
  // dependent steps (0.79 -> ~ 0.15 ms).  The occupancy of the coarse levels is common to the planes: every workgroup derives
  // it for itself (the same bytes from six writers).
  const int p0 = int( blockIdx.x ), p1 = p0 + 1;
  // down: mip maps of the levels first + 1 .. count - 1 (level `first` itself was produced by the launch before)
  for ( int l = L.first + 1; l < L.count; ++l ) {
    const int cnt = L.w[l] * L.h[l];
    for ( int i = threadIdx.x; i < cnt; i += blockDim.x )
      pushPullMipPixel( i, L.img[l - 1], L.occ[l - 1], L.w[l - 1], L.h[l - 1], L.img[l], L.occ[l], L.w[l], L.h[l], p0, p1 );
    __syncthreads();
  }
  // up: fill level l - 1 from level l, then iters blur passes between the level's two buffers (host loop of
  // generateAttributeImages, same order, same buffers)
  int iters = 4;
  for ( int l = L.count - 1; l > L.first; --l ) {
    const int fw = L.w[l - 1], fh = L.h[l - 1], cnt = fw * fh;
    uint8_t * img = L.img[l - 1], *tmp = L.tmp[l - 1];
    const uint8_t* occ = L.occ[l - 1];
    for ( int i = threadIdx.x; i < cnt; i += blockDim.x ) pushPullFillPixel( i, img, occ, fw, fh, L.img[l], L.w[l], L.h[l], p0, p1 );
    __syncthreads();
    for ( int i = threadIdx.x; i < cnt; i += blockDim.x ) tmp[size_t( p0 ) * cnt + i] = img[size_t( p0 ) * cnt + i];
    __syncthreads();
    uint8_t *src = img, *dst = tmp;
    for ( int it = 0; it < iters; ++it ) {
      for ( int i = threadIdx.x; i < cnt; i += blockDim.x ) pushPullBlurPixel( i, src, dst, occ, fw, fh, p0, p1 );
      __syncthreads();
      uint8_t* t = src;
      src        = dst;
      dst        = t;
    }
    if ( src != img ) {  // odd iteration count: the result sits in the partner buffer -- the level's image from now on
      L.img[l - 1] = src;
      L.tmp[l - 1] = img;
    }
    iters = min( iters + 1, 16 );
  }
}

// ---- S22 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void attributeGroupDilateKernel( const uint8_t* __restrict__ occ, int W, int H,
                                                                      uint8_t* __restrict__ attr ) {
  const size_t c = size_t( blockIdx.x ) * blockDim.x + threadIdx.x, plane = size_t( W ) * H;
  if ( c >= plane || occ[c] ) return;
#pragma unroll
  for ( int k = 0; k < 3; ++k ) {
    const uint32_t a = attr[k * plane + c], b = attr[( 3 + k ) * plane + c];
    const uint8_t  v = uint8_t( ( a + b + 1 ) >> 1 );
    attr[k * plane + c] = attr[( 3 + k ) * plane + c] = v;
  }
}

}  // namespace

// S18 on device-resident clouds: source (tree + original-order points + colours) -> target (tree + points)
int transferColorsDevice( tmc2_ctx* ctx, const TreeDev& srcTree, const Pt* d_srcPts, const uint8_t* d_srcRgb4, uint32_t n,
                          const TreeDev& tgtTree, const Pt* d_tgtPts, uint32_t M, uint8_t* d_outRgb4, uint32_t* d_error ) {
  hipStream_t      s = ctx->stream;
  const dim3       blk( 256 );
  DevBuf<uint32_t> d_idx8, d_dist8, d_idx1, d_dist1, d_count, d_offset, d_cursor;
  DevBuf<uint2>    d_entries;
  DevBuf<uint8_t>  d_fwd;
  TMC2_TRY( d_idx8.alloc( size_t( M ) * 8 ) );
  TMC2_TRY( d_dist8.alloc( size_t( M ) * 8 ) );
  TMC2_TRY( d_idx1.alloc( n ) );
  TMC2_TRY( d_dist1.alloc( n ) );
  TMC2_TRY( d_count.alloc( M ) );
  TMC2_TRY( d_offset.alloc( M ) );
  TMC2_TRY( d_cursor.alloc( M ) );
  TMC2_TRY( d_entries.alloc( n ) );
  TMC2_TRY( d_fwd.alloc( size_t( M ) * 4 ) );
  // (round 6: both searches in two launches -- the queries that have an identical point in the tree first, then the compacted
  //  rest: knn.hip, launchKnnSplit; option KNN_SPLIT=0: one launch each, as rounds 1-5)
  DevBuf<uint32_t> d_easy8;
  const char*      splitEnv = ctxOption( ctx, "KNN_SPLIT" );
  const bool       split    = !( splitEnv && splitEnv[0] == '0' );
  if ( split ) {
    TMC2_TRY( d_easy8.alloc( M ) );
    TMC2_TRY( launchKnnSplit( ctx, srcTree, d_tgtPts, M, 8, d_easy8.p, d_idx8.p, d_dist8.p, "knn8_recon_in_source" ) );
    TMC2_TRY( launchKnnSplit( ctx, tgtTree, d_srcPts, n, 1, d_idx1.p, d_idx1.p, d_dist1.p, "knn1_source_in_recon" ) );
  } else {
    TMC2_TRY( launchKnnTree( ctx, srcTree, d_tgtPts, M, 8, d_idx8.p, d_dist8.p, "knn8_recon_in_source" ) );
    TMC2_TRY( launchKnnTree( ctx, tgtTree, d_srcPts, n, 1, d_idx1.p, d_dist1.p, "knn1_source_in_recon" ) );
  }
  const int sid = ctx->stageBegin( "transfer_colors" );
  TMC2_TRY( fillRegions( ctx, {{d_count.p, size_t( M ) * 4, 0}, {d_cursor.p, size_t( M ) * 4, 0}, {d_error, 4, 0}} ) );
  const dim3 grdM( ( M + 255 ) / 256 ), grdN( ( n + 255 ) / 256 );
  hipLaunchKernelGGL( forwardColorKernel, grdM, blk, 0, s, d_idx8.p, d_dist8.p, split ? d_easy8.p : (const uint32_t*)nullptr, d_srcRgb4, M, d_fwd.p );
  hipLaunchKernelGGL( backwardCountKernel, grdN, blk, 0, s, d_idx1.p, n, d_count.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_offset.p, M, nullptr ) );
  hipLaunchKernelGGL( backwardFillKernel, grdN, blk, 0, s, d_idx1.p, d_dist1.p, d_offset.p, n, d_cursor.p, d_entries.p );
  hipLaunchKernelGGL( combineColorKernel, grdM, blk, 0, s, d_count.p, d_offset.p, d_entries.p, d_srcRgb4, d_fwd.p, M,
                      d_outRgb4, d_error );
  ctx->stageEnd( sid );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

// S17 + the k-d tree over the reconstruction: all the decoder needs before the post-reconstruction tail, and the first half
// of the encoder's phase B
int reconstructPointCloud( tmc2_frame* f ) {
  if ( !f->haveGeometryImages ) {
    setError( "generatePointCloud: geometry images missing" );
    return TMC2_E_STATE;
  }
  f->haveReconstruction = f->haveAttributeImages = false;
  f->haveBoundaryTypes = f->haveColors16 = f->haveSmoothed = f->haveRgbPost = false;
  tmc2_ctx*    ctx = f->ctx;
  hipStream_t  s   = ctx->stream;
  const int    W = f->canvasW, H = f->canvasH, prec = f->occPrecision;
  const dim3   blk( 256 );
  const uint32_t tiles = f->tileCount;
  // ---- S17 ----------------------------------------------------------------------------------------------
  int sid = ctx->stageBegin( "reconstruct" );
  DevBuf<uint32_t> d_tileCount, d_tileOffset, d_small;
  TMC2_TRY( d_tileCount.alloc( std::max( tiles, 1u ) ) );
  TMC2_TRY( d_tileOffset.alloc( std::max( tiles, 1u ) ) );
  TMC2_TRY( d_small.alloc( 8 ) );
  uint32_t M = 0;
  if ( tiles ) {
    hipLaunchKernelGGL( reconTileKernel<false>, dim3( tiles ), blk, 0, s, f->d_place.p, f->d_tilePatch.p, f->d_occVideo.p,
                        f->d_blockToPatch.p, f->d_geo.p, W, H, prec, d_tileCount.p, (const uint32_t*)nullptr, (Pt*)nullptr,
                        (uint32_t*)nullptr );
    volatile uint32_t* answer = ctx->answerLine( tmc2_ctx::kAnswerRecon );  // (the point count straight to a page-locked word: no copy)
    TMC2_TRY( exclusiveScanU32( ctx, d_tileCount.p, d_tileOffset.p, tiles, d_small.p, ScanAnswer{answer, nullptr, 0} ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    M = answer[0];
  }
  if ( M == 0 ) {
    ctx->stageEnd( sid );
    setError( "generatePointCloud: empty reconstruction" );
    return TMC2_E_STATE;
  }
  TMC2_TRY( f->d_recon.alloc( M ) );
  TMC2_TRY( f->d_pointToPixel.alloc( M ) );
  hipLaunchKernelGGL( reconTileKernel<true>, dim3( tiles ), blk, 0, s, f->d_place.p, f->d_tilePatch.p, f->d_occVideo.p,
                      f->d_blockToPatch.p, f->d_geo.p, W, H, prec, d_tileCount.p, d_tileOffset.p, f->d_recon.p,
                      f->d_pointToPixel.p );
  ctx->stageEnd( sid );
  f->reconCount = M;
  // ---- tree over the reconstruction (like S1) --------------------------------------------------------------
  const int placement = kdtreePlacement( ctx );
  HostGate  treeGate( ctx, placement == 1 );
  if ( placement == 0 || !treeGate.held ) {
    const int kt = ctx->stageBegin( "kdtree_build_recon" );
    TMC2_TRY( buildKdTreeDevice( ctx, f->d_recon.p, M, f->d_reconTreePts, f->d_reconPerm, f->d_reconNodes, f->reconTree.lo,
                                 f->reconTree.hi, f->reconTree.depth ) );
    ctx->stageEnd( kt );
  } else {
    Pt*       hp = ctx->hostD.get<Pt>( M );
    uint32_t* hi = ctx->hostA.get<uint32_t>( M );
    if ( !hp || !hi ) {
      setError( "generatePointCloud: hipHostMalloc failed" );
      return TMC2_E_HIP;
    }
    TMC2_HIP( hipMemcpyAsync( hp, f->d_recon.p, size_t( M ) * sizeof( Pt ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    {
      const auto t0 = std::chrono::steady_clock::now();
      f->reconTree.buildInPlace( hp, hi, M );
      const auto t1 = std::chrono::steady_clock::now();
      ctx->stageAddHostMs( "kdtree_build_recon_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
    }
    treeGate.release();
    TMC2_TRY( f->d_reconTreePts.alloc( M ) );
    TMC2_TRY( f->d_reconPerm.alloc( M ) );
    TMC2_TRY( f->d_reconNodes.alloc( f->reconTree.nodes.size() ) );
    TMC2_HIP( hipMemcpyAsync( f->d_reconTreePts.p, hp, size_t( M ) * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipMemcpyAsync( f->d_reconPerm.p, hi, size_t( M ) * 4, hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipMemcpyAsync( f->d_reconNodes.p, f->reconTree.nodes.data(), f->reconTree.nodes.size() * sizeof( KdNode ),
                              hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
  }
  f->haveReconstruction = true;
  return TMC2_OK;
}

int generateAttributeImages( tmc2_frame* f ) {
  if ( !f->haveGeometryImages ) {
    setError( "generateAttributeImages: geometry images missing" );
    return TMC2_E_STATE;
  }
  if ( f->d_rgb.count == 0 ) {
    setError( "generateAttributeImages: the frame has no colours" );
    return TMC2_E_STATE;
  }
  TMC2_TRY( f->ensureTree() );
  TMC2_TRY( reconstructPointCloud( f ) );
  tmc2_ctx*    ctx = f->ctx;
  hipStream_t  s   = ctx->stream;
  const int    W = f->canvasW, H = f->canvasH, prec = f->occPrecision;
  const size_t area = size_t( W ) * H;
  const dim3   blk( 256 );
  const uint32_t n = uint32_t( f->n ), M = uint32_t( f->reconCount );
  int          sid = 0;
  DevBuf<uint32_t> d_small;
  TMC2_TRY( d_small.alloc( 8 ) );
  TreeDev rt;
  rt.ptsTree = f->d_reconTreePts.p;
  rt.perm    = f->d_reconPerm.p;
  rt.nodes   = f->d_reconNodes.p;
  for ( int d = 0; d < 3; ++d ) rt.lo[d] = f->reconTree.lo[d], rt.hi[d] = f->reconTree.hi[d];
  rt.depth = f->reconTree.depth;
  rt.n     = M;
  rt.queriesBounded = true;  // queried with the frame's own points; dispatch() checks both boxes
  rt.queriesTight   = true;  // (non-negative and below 2^13, like the reconstruction the tree is built over)
  // ---- S18 ----------------------------------------------------------------------------------------------
  TMC2_TRY( f->d_reconRgb.alloc( size_t( M ) * 4 ) );
  const dim3 grdM( ( M + 255 ) / 256 );
  // (the sort-depth flag is a page-locked word of the context, read after the stage's last synchronisation: no copy)
  volatile uint32_t* h_err = ctx->answerLine( tmc2_ctx::kAnswerAttrError );
  TMC2_TRY( transferColorsDevice( ctx, frameTree( f ), f->d_pts.p, f->d_rgb.p, n, rt, f->d_recon.p, M, f->d_reconRgb.p,
                                  const_cast<uint32_t*>( h_err ) ) );
  // ---- S20 ----------------------------------------------------------------------------------------------
  sid = ctx->stageBegin( "attribute_images" );
  DevBuf<uint8_t> d_occ;
  TMC2_TRY( d_occ.alloc( area ) );
  TMC2_TRY( f->d_attr.alloc( 6 * area ) );
  TMC2_HIP( hipMemsetAsync( f->d_attr.p, 0, 6 * area, s ) );
  hipLaunchKernelGGL( upsampleOccupancyKernel, dim3( uint32_t( ( area + 255 ) / 256 ) ), blk, 0, s, f->d_occVideo.p, W, H, prec,
                      d_occ.p );
  hipLaunchKernelGGL( attributeScatterKernel, grdM, blk, 0, s, f->d_reconRgb.p, f->d_pointToPixel.p, M, W, H, f->d_attr.p );
  // ---- S21: pyramid ------------------------------------------------------------------------------------------
  struct Level {
    int      w, h;
    uint8_t *img, *tmp, *occ;
  };
  std::vector<Level> lv;
  lv.push_back( Level{W, H, f->d_attr.p, nullptr, d_occ.p} );
  size_t pyramidBytes = 6 * area;  // ping-pong partner of level 0
  {
    int w = W, h = H;
    for ( ;; ) {
      w = ( w + 1 ) >> 1;
      h = ( h + 1 ) >> 1;
      lv.push_back( Level{w, h, nullptr, nullptr, nullptr} );
      pyramidBytes += size_t( w ) * h * ( 6 + 6 + 1 );
      if ( w <= 4 || h <= 4 ) break;
    }
  }
  DevBuf<uint8_t> d_pyr;
  TMC2_TRY( d_pyr.alloc( pyramidBytes + 64 * lv.size() ) );
  {
    uint8_t* p = d_pyr.p;
    lv[0].tmp  = p;
    p += 6 * area;
    for ( size_t l = 1; l < lv.size(); ++l ) {
      const size_t a = size_t( lv[l].w ) * lv[l].h;
      lv[l].img = p, p += 6 * a;
      lv[l].tmp = p, p += 6 * a;
      lv[l].occ = p, p += ( a + 15 ) & ~size_t( 15 );
    }
  }
  // the coarse end of the pyramid (levels of at most kPushPullSmall pixels) goes through ONE launch
  size_t firstSmall = lv.size();
  if ( lv.size() <= size_t( kPushPullLevels ) )
    for ( size_t l = 1; l < lv.size(); ++l )
      if ( lv[l].w * lv[l].h <= kPushPullSmall ) {
        firstSmall = l;
        break;
      }
  for ( size_t l = 1; l < lv.size() && l <= firstSmall; ++l ) {
    const int cnt = lv[l].w * lv[l].h;
    hipLaunchKernelGGL( pushPullMipKernel, dim3( ( cnt + 255 ) / 256 ), blk, 0, s, lv[l - 1].img, lv[l - 1].occ, lv[l - 1].w,
                        lv[l - 1].h, lv[l].img, lv[l].occ, lv[l].w, lv[l].h );
  }
  int iters = 4;
  if ( firstSmall + 1 < lv.size() ) {
    PushPullLevels L;
    L.count = int( lv.size() ), L.first = int( firstSmall );
    for ( size_t l = 0; l < lv.size(); ++l ) L.w[l] = lv[l].w, L.h[l] = lv[l].h, L.img[l] = lv[l].img, L.tmp[l] = lv[l].tmp, L.occ[l] = lv[l].occ;
    hipLaunchKernelGGL( pushPullCoarseLevelsKernel, dim3( 6 ), dim3( 1024 ), 0, s, L );
    for ( size_t l = lv.size() - 1; l > firstSmall; --l ) {  // (what the kernel did to the buffers of the levels it filled)
      if ( iters & 1 ) std::swap( lv[l - 1].img, lv[l - 1].tmp );
      iters = std::min( iters + 1, 16 );
    }
  }
  for ( size_t l = std::min( firstSmall, lv.size() - 1 ); l >= 1; --l ) {
    Level&    fine = lv[l - 1];
    const int cnt  = fine.w * fine.h;
    const dim3 grd( ( cnt + 255 ) / 256 );
    if ( iters <= kPushPullMaxIters ) {  // fill + every blur pass of the level: one launch, tiles in LDS
      const int tilesX = ( fine.w + kPushPullTile - 1 ) / kPushPullTile, tilesY = ( fine.h + kPushPullTile - 1 ) / kPushPullTile;
      hipLaunchKernelGGL( pushPullFillBlurKernel, dim3( tilesX * tilesY, 6 ), dim3( 1024 ), 0, s, fine.img, fine.occ, fine.w, fine.h,
                          lv[l].img, lv[l].w, lv[l].h, iters, tilesX );
      iters = std::min( iters + 1, 16 );
      continue;
    }
    hipLaunchKernelGGL( pushPullFillKernel, grd, blk, 0, s, fine.img, fine.occ, fine.w, fine.h, lv[l].img, lv[l].w, lv[l].h );
    TMC2_HIP( hipMemcpyAsync( fine.tmp, fine.img, size_t( 6 ) * cnt, hipMemcpyDeviceToDevice, s ) );
    uint8_t *src = fine.img, *dst = fine.tmp;
    for ( int it = 0; it < iters; ++it ) {
      hipLaunchKernelGGL( pushPullBlurKernel, grd, blk, 0, s, src, dst, fine.occ, fine.w, fine.h );
      std::swap( src, dst );
    }
    if ( src != fine.img ) {  // odd iteration count: latest result sits in the partner buffer
      if ( l - 1 == 0 )
        TMC2_HIP( hipMemcpyAsync( fine.img, src, size_t( 6 ) * cnt, hipMemcpyDeviceToDevice, s ) );
      else
        std::swap( fine.img, fine.tmp );
    }
    iters = std::min( iters + 1, 16 );
  }
  // ---- S22 ----------------------------------------------------------------------------------------------
  hipLaunchKernelGGL( attributeGroupDilateKernel, dim3( uint32_t( ( area + 255 ) / 256 ) ), blk, 0, s, d_occ.p, W, H, f->d_attr.p );
  ctx->stageEnd( sid );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  const uint32_t err = *h_err;
  if ( err ) {
    setError( "transferColors: a backward candidate list hit std::sort's depth limit (heapsort fallback of libstdc++ "
              "introsort is not reproduced)" );
    return TMC2_E_UNSUPPORTED;
  }
  f->haveAttributeImages = true;
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

// replaces PCCPointSet3::transferColors on two host clouds (CTC settings); rgb out = uint8[m][3]
int tmc2_transfer_colors( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n, const int16_t* tgtXyz,
                          uint64_t m, uint8_t* tgtRgb ) {
  using namespace tmc2;
  if ( !ctx || !srcXyz || !srcRgb || !tgtXyz || !tgtRgb || n < 8 || m < 1 ) {
    setError( "transfer_colors: invalid argument (the source needs at least 8 points)" );
    return TMC2_E_INVALID;
  }
  ApiScope    scope( ctx );
  hipStream_t s = ctx->stream;
  struct Side {
    KdTreeHost       tree;
    DevBuf<Pt>       pts, ptsTree;
    DevBuf<uint32_t> perm;
    DevBuf<KdNode>   nodes;
    TreeDev          dev;
  } S, T;
  auto upload = [&]( Side& sd, const int16_t* xyz, uint64_t cnt ) -> int {
    std::vector<Pt> pts( cnt );
    for ( uint64_t i = 0; i < cnt; ++i ) pts[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0};
    TMC2_TRY( sd.pts.alloc( cnt ) );
    TMC2_HIP( hipMemcpyAsync( sd.pts.p, pts.data(), cnt * sizeof( Pt ), hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    TMC2_TRY( buildKdTreeDevice( ctx, sd.pts.p, cnt, sd.ptsTree, sd.perm, sd.nodes, sd.tree.lo, sd.tree.hi, sd.tree.depth ) );
    sd.dev.ptsTree = sd.ptsTree.p, sd.dev.perm = sd.perm.p, sd.dev.nodes = sd.nodes.p;
    for ( int d = 0; d < 3; ++d ) sd.dev.lo[d] = sd.tree.lo[d], sd.dev.hi[d] = sd.tree.hi[d];
    sd.dev.depth = sd.tree.depth, sd.dev.n = cnt;
    return TMC2_OK;
  };
  TMC2_TRY( upload( S, srcXyz, n ) );
  TMC2_TRY( upload( T, tgtXyz, m ) );
  std::vector<uint8_t> c4( 4 * n );
  for ( uint64_t i = 0; i < n; ++i ) c4[4 * i] = srcRgb[3 * i], c4[4 * i + 1] = srcRgb[3 * i + 1], c4[4 * i + 2] = srcRgb[3 * i + 2], c4[4 * i + 3] = 0;
  DevBuf<uint8_t>  d_rgb, d_out;
  DevBuf<uint32_t> d_err;
  TMC2_TRY( d_rgb.alloc( 4 * n ) );
  TMC2_TRY( d_out.alloc( 4 * m ) );
  TMC2_TRY( d_err.alloc( 1 ) );
  TMC2_HIP( hipMemcpyAsync( d_rgb.p, c4.data(), 4 * n, hipMemcpyHostToDevice, s ) );
  TMC2_TRY( transferColorsDevice( ctx, S.dev, S.pts.p, d_rgb.p, uint32_t( n ), T.dev, T.pts.p, uint32_t( m ), d_out.p, d_err.p ) );
  std::vector<uint8_t> o4( 4 * m );
  uint32_t             err = 0;
  TMC2_HIP( hipMemcpyAsync( o4.data(), d_out.p, 4 * m, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( &err, d_err.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  if ( err ) {
    setError( "transferColors: a backward candidate list hit std::sort's depth limit (not reproduced)" );
    return TMC2_E_UNSUPPORTED;
  }
  for ( uint64_t i = 0; i < m; ++i ) tgtRgb[3 * i] = o4[4 * i], tgtRgb[3 * i + 1] = o4[4 * i + 1], tgtRgb[3 * i + 2] = o4[4 * i + 2];
  return TMC2_OK;
}

int tmc2_codec_generate_point_cloud( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::reconstructPointCloud( f );
}

int tmc2_encoder_generate_attribute_images( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::generateAttributeImages( f );
}

int64_t tmc2_frame_recon_count( tmc2_frame* f ) { return ( f && f->haveReconstruction ) ? int64_t( f->reconCount ) : 0; }

int tmc2_frame_get_reconstruction( tmc2_frame* f, int16_t* xyz, uint8_t* rgb, uint32_t* pointToPixel ) {
  if ( !f || !f->haveReconstruction ) {
    tmc2::setError( "get_reconstruction: not generated" );
    return TMC2_E_STATE;
  }
  if ( rgb && !f->haveAttributeImages ) {
    tmc2::setError( "get_reconstruction: no transferred colours (tmc2_encoder_generate_attribute_images produces them)" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  hipStream_t    s = f->ctx->stream;
  const size_t   M = f->reconCount;
  std::vector<tmc2::Pt>  pts( M );
  std::vector<uint8_t>   c4( rgb ? 4 * M : 0 );
  std::vector<uint32_t>  pp( M );
  TMC2_HIP( hipMemcpyAsync( pts.data(), f->d_recon.p, M * sizeof( tmc2::Pt ), hipMemcpyDeviceToHost, s ) );
  if ( rgb ) TMC2_HIP( hipMemcpyAsync( c4.data(), f->d_reconRgb.p, 4 * M, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( pp.data(), f->d_pointToPixel.p, 4 * M, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  for ( size_t i = 0; i < M; ++i ) {
    if ( xyz ) xyz[3 * i] = pts[i].x, xyz[3 * i + 1] = pts[i].y, xyz[3 * i + 2] = pts[i].z;
    if ( rgb ) rgb[3 * i] = c4[4 * i], rgb[3 * i + 1] = c4[4 * i + 1], rgb[3 * i + 2] = c4[4 * i + 2];
    if ( pointToPixel )
      pointToPixel[3 * i] = tmc2::pixelX( pp[i] ), pointToPixel[3 * i + 1] = tmc2::pixelY( pp[i] ), pointToPixel[3 * i + 2] = tmc2::pixelLayer( pp[i] );
  }
  return TMC2_OK;
}

int tmc2_frame_device_attribute( tmc2_frame* f, void** attribute ) {
  if ( !f || !attribute || !f->haveAttributeImages ) return TMC2_E_STATE;
  *attribute = f->d_attr.p;
  return TMC2_OK;
}

int tmc2_frame_get_attribute_images( tmc2_frame* f, uint8_t* attribute ) {
  if ( !f || !attribute || !f->haveAttributeImages ) {
    tmc2::setError( "get_attribute_images: not generated" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  TMC2_HIP( hipMemcpyAsync( attribute, f->d_attr.p, size_t( 6 ) * f->canvasW * f->canvasH, hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}
}
