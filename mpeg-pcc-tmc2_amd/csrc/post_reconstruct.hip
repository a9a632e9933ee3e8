// post_reconstruct.hip -- the post-reconstruction tail on gfx950: what PCCEncoder::encode runs
// on every reconstructed frame after the attribute video (PccLibEncoder/source/PCCEncoder.cpp:571-719) and what
// PCCDecoder::decode runs to finish a frame (PccLibDecoder/source/PCCDecoder.cpp:330-470), under the CTC settings.
//
// Replaces (reference: source/lib/...)
//   T1  PCCCodec::identifyBoundaryPoints             PccLibCommon/source/PCCCodec.cpp:268-327 (called from generatePointCloud :955-976)
//   T2  PCCCodec::colorPointCloud                    PCCCodec.cpp:1319-1460, single-stream branch "f < mapCount"
//   T3  PCCCodec::smoothPointCloudPostprocess        PCCCodec.cpp:54-148 with addGridCentroid :982-1000, gridFiltering :1002-1065,
//                                                    smoothPointCloudGrid :1067-1106 (gridSmoothing = 1)
//   T4  PCCPointSet3::transferColors16bitBP          PccLibCommon/source/PCCPointSet.cpp:1126-1470, filterType 1, arguments of
//                                                    PCCEncoder.cpp:657-672 / PCCDecoder.cpp:416-431
//   T5  PCCPointSet3::convertYUV16ToRGB8             PccLibCommon/include/PCCPointSet.h:133-166
//
// All five are point-parallel once the sequential bookkeeping of the reference is taken apart:
//  * T1: the staged 3x3 / 5x5 tests collapse to "within two pixels of the canvas border, or an unoccupied pixel in the 5x5
//    window"; occupancy is the p x p-granular occupancy video, so a window is at most 3x3 cells.
//  * T3: the reference numbers the boundary cells in first-touch order and sums their points in float, in point order.
//    Cell numbers are only names (here: rank of the cell in raster order, by a prefix sum over the cell flags), and the
//    float sums are sums of integers below 2^24, hence exact and order-free: integer atomics give the same value
//    (checked: a cell whose sum or count leaves that range is reported, not guessed).  "doSmooth" (a second patch showed
//    up in the cell) is min(patch) != max(patch).  The filter itself reads only the cell table and the point's own
//    position, so moving points in place, as the reference does, does not couple the points.
//  * T4 with filterType 1 touches only the moved points (a few percent): their 8-NN in the cloud before smoothing
//    (the tree S18 built), the 1-NN of those neighbours in the smoothed cloud (one more tree), candidate lists bucketed
//    per moved target, ordered as libstdc++'s std::sort leaves them (cand_sort.h), reduced in fp64 in that order.
#include <algorithm>

#include "cand_sort.h"
#include "internal.h"

namespace tmc2 {
namespace {

__device__ __forceinline__ void unpackPixel( uint32_t p, int& x, int& y, int& layer ) {
  x     = int( pixelX( p ) );
  y     = int( pixelY( p ) );
  layer = int( pixelLayer( p ) );
}

// ---- T1 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void boundaryTypeKernel( const uint32_t* __restrict__ pointToPixel, uint32_t M,
                                                              const uint8_t* __restrict__ occVideo, int W, int H, int prec,
                                                              uint8_t* __restrict__ btype ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M ) return;
  int x, y, layer;
  unpackPixel( pointToPixel[i], x, y, layer );
  const int Wv = W / prec;
  uint8_t   t  = 0;
  if ( occVideo[size_t( y / prec ) * Wv + x / prec] ) {
    if ( x < 2 || y < 2 || x >= W - 2 || y >= H - 2 ) {
      t = 1;
    } else {
      const int cx0 = ( x - 2 ) / prec, cx1 = ( x + 2 ) / prec, cy0 = ( y - 2 ) / prec, cy1 = ( y + 2 ) / prec;
      for ( int cy = cy0; cy <= cy1; ++cy )
        for ( int cx = cx0; cx <= cx1; ++cx )
          if ( !occVideo[size_t( cy ) * Wv + cx] ) t = 1;
    }
  }
  btype[i] = t;
}

// ---- T2 ---------------------------------------------------------------------------------------------------
// attribute: six u16 planes (2 maps x 3 channels) of W*H
__global__ __launch_bounds__( 256 ) void colorGatherKernel( const uint32_t* __restrict__ pointToPixel, uint32_t M,
                                                             const uint16_t* __restrict__ attribute, int W, int H,
                                                             ushort4* __restrict__ colors16 ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M ) return;
  int x, y, layer;
  unpackPixel( pointToPixel[i], x, y, layer );
  const size_t    plane = size_t( W ) * H;
  const uint16_t* a     = attribute + size_t( layer ) * 3 * plane + size_t( y ) * W + x;
  colors16[i]           = make_ushort4( a[0], a[plane], a[2 * plane], 0 );
}

// ---- T3 ---------------------------------------------------------------------------------------------------
struct GridGeom {
  int gridSize, half, w, disth, th;
};
__device__ __forceinline__ bool outsideGrid( const GridGeom& g, int px, int py, int pz ) {
  return px < g.disth || py < g.disth || pz < g.disth || g.th <= px + g.disth || g.th <= py + g.disth || g.th <= pz + g.disth;
}
// the lower corner of the 2x2x2 cells around a point: the point's cell, or the one before it in every direction in
// which the point sits in the lower half of its cell
__device__ __forceinline__ int lowerCell( const GridGeom& g, int p ) {
  const int c = p / g.gridSize;
  return c + ( ( p - c * g.gridSize < g.half ) ? -1 : 0 );
}

__global__ __launch_bounds__( 256 ) void markCellsKernel( const Pt* __restrict__ pts, const uint8_t* __restrict__ btype, uint32_t M,
                                                           GridGeom g, uint32_t* __restrict__ cellFlag ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M || btype[i] != 1 ) return;
  const Pt p = pts[i];
  if ( outsideGrid( g, p.x, p.y, p.z ) ) return;
  const int qx = lowerCell( g, p.x ), qy = lowerCell( g, p.y ), qz = lowerCell( g, p.z );
  for ( int dz = 0; dz < 2; ++dz )
    for ( int dy = 0; dy < 2; ++dy )
      for ( int dx = 0; dx < 2; ++dx ) cellFlag[( size_t( qz + dz ) * g.w + ( qy + dy ) ) * g.w + ( qx + dx )] = 1u;
}

struct CellAcc {  // one boundary cell: integer sums of its points, the patches seen
  uint32_t count, sx, sy, sz, patchMin, patchMax;
};

__global__ __launch_bounds__( 256 ) void initCellsKernel( CellAcc* __restrict__ cells, uint32_t n ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) cells[i] = CellAcc{0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u};
}

__global__ __launch_bounds__( 256 ) void accumulateCellsKernel( const Pt* __restrict__ pts, const uint32_t* __restrict__ pointToPixel,
                                                                 uint32_t M, const uint32_t* __restrict__ blockToPatch, int Wb,
                                                                 GridGeom g, const uint32_t* __restrict__ cellFlag,
                                                                 const uint32_t* __restrict__ cellSlot, CellAcc* __restrict__ cells ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M ) return;
  const Pt p = pts[i];
  if ( outsideGrid( g, p.x, p.y, p.z ) ) return;
  const size_t cell = ( size_t( p.z / g.gridSize ) * g.w + p.y / g.gridSize ) * g.w + p.x / g.gridSize;
  if ( !cellFlag[cell] ) return;
  int x, y, layer;
  unpackPixel( pointToPixel[i], x, y, layer );
  const uint32_t patch = blockToPatch[size_t( y / 16 ) * Wb + x / 16];  // list position + 1 of the patch that emitted the point
  CellAcc*       c     = cells + cellSlot[cell];
  atomicAdd( &c->count, 1u );
  atomicAdd( &c->sx, uint32_t( p.x ) );
  atomicAdd( &c->sy, uint32_t( p.y ) );
  atomicAdd( &c->sz, uint32_t( p.z ) );
  if ( c->patchMin > patch ) atomicMin( &c->patchMin, patch );
  if ( c->patchMax < patch ) atomicMax( &c->patchMax, patch );
}

__global__ __launch_bounds__( 256 ) void smoothGridKernel( const Pt* __restrict__ pts, uint32_t M, GridGeom g,
                                                            const uint32_t* __restrict__ cellSlot, const CellAcc* __restrict__ cells,
                                                            int threshold, Pt* __restrict__ out, uint8_t* __restrict__ btype,
                                                            uint32_t* __restrict__ error ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M ) return;
  const Pt p = pts[i];
  out[i]     = p;
  if ( btype[i] != 1 || outsideGrid( g, p.x, p.y, p.z ) ) return;
  const int P[3] = {p.x, p.y, p.z};
  const int S[3] = {lowerCell( g, P[0] ), lowerCell( g, P[1] ), lowerCell( g, P[2] )};
  // the eight cells: count, centre (float division of the exact integer sums, as the reference's float centre)
  double c3[2][2][2][3];
  int    cnt8[2][2][2];
  bool   other = false;
  for ( int dz = 0; dz < 2; ++dz )
    for ( int dy = 0; dy < 2; ++dy )
      for ( int dx = 0; dx < 2; ++dx ) {
        const CellAcc c = cells[cellSlot[( size_t( S[2] + dz ) * g.w + ( S[1] + dy ) ) * g.w + ( S[0] + dx )]];
        cnt8[dz][dy][dx] = int( c.count );
        if ( c.count > 65535u || c.sx >= ( 1u << 24 ) || c.sy >= ( 1u << 24 ) || c.sz >= ( 1u << 24 ) ) *error = 1;
        if ( c.count != 0 && c.patchMin != c.patchMax ) other = true;
        if ( c.count > 0 ) {
          const float n       = float( c.count );
          c3[dz][dy][dx][0] = double( __fdiv_rn( float( c.sx ), n ) );
          c3[dz][dy][dx][1] = double( __fdiv_rn( float( c.sy ), n ) );
          c3[dz][dy][dx][2] = double( __fdiv_rn( float( c.sz ), n ) );
        } else {
          for ( int k = 0; k < 3; ++k ) c3[dz][dy][dx][k] = double( P[k] );
        }
      }
  if ( !other ) return;
  const int gridSize2 = g.gridSize * 2, norm = gridSize2 * gridSize2 * gridSize2;
  int       Wt[3], Q[3];
  for ( int k = 0; k < 3; ++k ) {
    Wt[k] = ( P[k] - S[k] * g.gridSize - g.half ) * 2 + 1;
    Q[k]  = gridSize2 - Wt[k];
  }
  int    count       = 0;
  double centroid[3] = {0.0, 0.0, 0.0};
  for ( int dz = 0; dz < 2; ++dz )
    for ( int dy = 0; dy < 2; ++dy )
      for ( int dx = 0; dx < 2; ++dx ) {
        const int    wgt = ( dx ? Wt[0] : Q[0] ) * ( dy ? Wt[1] : Q[1] ) * ( dz ? Wt[2] : Q[2] );
        const double wd  = double( wgt );
        for ( int k = 0; k < 3; ++k ) centroid[k] += c3[dz][dy][dx][k] * wd;
        count += wgt * cnt8[dz][dy][dx];
      }
  count /= norm;
  const double cd = double( count );
  double       d2 = 0.0;
  for ( int k = 0; k < 3; ++k ) {
    centroid[k]    = __ddiv_rn( centroid[k], double( norm ) ) * cd;
    const double d = double( P[k] ) * cd - centroid[k];
    d2 += d * d;
  }
  const double dist2 = __ddiv_rn( d2, cd ) + 0.5;  // count == 0: 0/0, NaN, compares false like the reference
  if ( dist2 >= double( max( threshold, count ) * 2 ) ) {
    Pt q;
    q.x      = int16_t( (long long)( __ddiv_rn( centroid[0], cd ) + 0.5 ) );
    q.y      = int16_t( (long long)( __ddiv_rn( centroid[1], cd ) + 0.5 ) );
    q.z      = int16_t( (long long)( __ddiv_rn( centroid[2], cd ) + 0.5 ) );
    q.w      = 0;
    out[i]   = q;
    btype[i] = 3;
  }
}

// ---- T4 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void movedFlagKernel( const uint8_t* __restrict__ btype, uint32_t M, uint32_t* __restrict__ flag ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < M ) flag[i] = btype[i] == 3 ? 1u : 0u;
}
__global__ __launch_bounds__( 256 ) void movedGatherKernel( const uint8_t* __restrict__ btype, const uint32_t* __restrict__ rank,
                                                             const Pt* __restrict__ pts, uint32_t M, uint32_t* __restrict__ moved,
                                                             Pt* __restrict__ queries ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M || btype[i] != 3 ) return;
  moved[rank[i]]   = i;
  queries[rank[i]] = pts[i];
}
__device__ __forceinline__ uint16_t toU16( double v ) { return uint16_t( fmax( 0.0, fmin( round( v ), 65535.0 ) ) ); }

// forward colour of every moved point from its 8 nearest points of the cloud before smoothing; the neighbours'
// positions become the queries of the backward search
__global__ __launch_bounds__( 256 ) void forwardColor16Kernel( const uint32_t* __restrict__ idx8, const uint32_t* __restrict__ dist8,
                                                                const ushort4* __restrict__ srcColors, const Pt* __restrict__ srcPts,
                                                                uint32_t K, ushort4* __restrict__ refined, Pt* __restrict__ partPts ) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if ( r >= K ) return;
  uint32_t id[8], ds[8];
  for ( int i = 0; i < 8; ++i ) {
    id[i]                      = idx8[size_t( r ) * 8 + i];
    ds[i]                      = dist8[size_t( r ) * 8 + i];
    partPts[size_t( r ) * 8 + i] = srcPts[id[i]];
  }
  if ( ds[0] == 0 ) {  // "dist < 0.0001"
    refined[r] = srcColors[id[0]];
    return;
  }
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, sw = 0.0;
  for ( int i = 0; i < 8; ++i ) {
    const double  w = __ddiv_rn( 1.0, double( ds[i] ) + 4.0 );
    const ushort4 c = srcColors[id[i]];
    c0 += double( c.x ) * w;
    c1 += double( c.y ) * w;
    c2 += double( c.z ) * w;
    sw += w;
  }
  refined[r] = make_ushort4( toU16( __ddiv_rn( c0, sw ) ), toU16( __ddiv_rn( c1, sw ) ), toU16( __ddiv_rn( c2, sw ) ), 0 );
}

__device__ __forceinline__ bool closeColors( const ushort4 a, const ushort4 b ) {
  return abs( int( a.x ) - int( b.x ) ) < 40 && abs( int( a.y ) - int( b.y ) ) < 40 && abs( int( a.z ) - int( b.z ) ) < 40;
}
// entry e (= 8 * moved rank + neighbour) votes for its nearest point of the smoothed cloud if that one was moved and the
// colours are close.  FILL = false: count per moved target; FILL = true: place (dist, e) in the target's bucket
template <bool FILL>
__global__ __launch_bounds__( 256 ) void backwardVoteKernel( const uint32_t* __restrict__ idx8, const uint32_t* __restrict__ nn1,
                                                              const uint32_t* __restrict__ nn1Dist, uint32_t entries,
                                                              const ushort4* __restrict__ srcColors, const uint8_t* __restrict__ btype,
                                                              const uint32_t* __restrict__ rank, uint32_t* __restrict__ count,
                                                              const uint32_t* __restrict__ offset, uint32_t* __restrict__ cursor,
                                                              uint2* __restrict__ bucket ) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if ( e >= entries ) return;
  const uint32_t t = nn1[e];
  if ( btype[t] != 3 ) return;  // only moved points are recoloured; the other lists are never read
  if ( !closeColors( srcColors[idx8[e]], srcColors[t] ) ) return;  // the target still carries the colour it had before
  const uint32_t r = rank[t];
  if ( !FILL ) {
    atomicAdd( &count[r], 1u );
  } else {
    bucket[offset[r] + atomicAdd( &cursor[r], 1u )] = make_uint2( nn1Dist[e], e );
  }
}

__global__ __launch_bounds__( 256 ) void combineColor16Kernel( const uint32_t* __restrict__ moved, const uint32_t* __restrict__ count,
                                                                const uint32_t* __restrict__ offset, uint2* __restrict__ bucket,
                                                                const uint32_t* __restrict__ idx8, const ushort4* __restrict__ srcColors,
                                                                ushort4* __restrict__ refined, uint32_t K,
                                                                uint32_t* __restrict__ error ) {
  // refined[r]: in the forward colour, out the final colour of moved point r.  The cloud's colours are NOT touched here:
  // moved points are sources of other moved points' candidates, and the reference reads those from its untouched copy
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if ( r >= K ) return;
  const int n = int( count[r] );
  if ( n == 0 ) return;
  uint2* e = bucket + offset[r];
  for ( int i = 1; i < n; ++i ) {  // the order in which the reference appended them: by entry number
    const uint2 v = e[i];
    int         k = i - 1;
    while ( k >= 0 && e[k].y > v.y ) {
      e[k + 1] = e[k];
      --k;
    }
    e[k + 1] = v;
  }
  const CandSort cs{e};
  if ( !cs.sort( n ) ) *error = 1;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  if ( n == 1 ) {
    const ushort4 c = srcColors[idx8[e[0].y]];
    c0 = double( c.x ), c1 = double( c.y ), c2 = double( c.z );
  } else {
    double sw = 0.0;
    for ( int k = 0; k < n; ++k ) {
      const ushort4 c = srcColors[idx8[e[k].y]];
      const double  w = __ddiv_rn( 1.0, __dsqrt_rn( double( e[k].x ) ) + 4.0 );
      c0 += double( c.x ) * w;
      c1 += double( c.y ) * w;
      c2 += double( c.z ) * w;
      sw += w;
    }
    c0 = __ddiv_rn( c0, sw ), c1 = __ddiv_rn( c1, sw ), c2 = __ddiv_rn( c2, sw );
  }
  const ushort4 f = refined[r];  // fixWeight: w = 0  ->  round( 0 * centroid1 + 1 * centroid2 )
  refined[r] = make_ushort4( toU16( 0.0 * double( f.x ) + 1.0 * c0 ), toU16( 0.0 * double( f.y ) + 1.0 * c1 ),
                             toU16( 0.0 * double( f.z ) + 1.0 * c2 ), 0 );
}
__global__ __launch_bounds__( 256 ) void scatterColor16Kernel( const uint32_t* __restrict__ moved, const ushort4* __restrict__ final16,
                                                                uint32_t K, ushort4* __restrict__ colors16 ) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if ( r < K ) colors16[moved[r]] = final16[r];
}

// ---- T5 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void yuv16ToRgb8Kernel( const ushort4* __restrict__ colors16, uint32_t M, uchar4* __restrict__ rgb ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= M ) return;
  const ushort4 c      = colors16[i];
  const double  weight = __ddiv_rn( 1.0, 65535.0 );
  double        y1 = weight * double( c.x ), u1 = weight * ( double( c.y ) - 32768.0 ), v1 = weight * ( double( c.z ) - 32768.0 );
  y1 = fmin( fmax( y1, 0.0 ), 1.0 );
  u1 = fmin( fmax( u1, -0.5 ), 0.5 );
  v1 = fmin( fmax( v1, -0.5 ), 0.5 );
  const double r = y1 + 1.57480 * v1;
  const double g = y1 - 0.18733 * u1 - 0.46813 * v1;
  const double b = y1 + 1.85563 * u1;
  rgb[i] = make_uchar4( uint8_t( fmax( 0.0, fmin( round( r * 255 ), 255.0 ) ) ), uint8_t( fmax( 0.0, fmin( round( g * 255 ), 255.0 ) ) ),
                        uint8_t( fmax( 0.0, fmin( round( b * 255 ), 255.0 ) ) ), 0 );
}

TreeDev reconTreeDev( const tmc2_frame* f ) {
  TreeDev rt;
  rt.ptsTree = f->d_reconTreePts.p;
  rt.perm    = f->d_reconPerm.p;
  rt.nodes   = f->d_reconNodes.p;
  for ( int d = 0; d < 3; ++d ) rt.lo[d] = f->reconTree.lo[d], rt.hi[d] = f->reconTree.hi[d];
  rt.depth          = f->reconTree.depth;
  rt.n              = f->reconCount;
  rt.queriesBounded = true;  // queries are points of the reconstruction before / after smoothing
  return rt;
}
int needReconstruction( tmc2_frame* f, const char* who ) {
  if ( !f->haveReconstruction || f->reconCount == 0 ) {
    setError( "%s: the frame has no reconstruction (tmc2_codec_generate_point_cloud or tmc2_encoder_generate_attribute_images first)", who );
    return TMC2_E_STATE;
  }
  return TMC2_OK;
}
}  // namespace

int identifyBoundaryPoints( tmc2_frame* f ) {
  TMC2_TRY( needReconstruction( f, "identifyBoundaryPoints" ) );
  tmc2_ctx*      ctx = f->ctx;
  const uint32_t M   = uint32_t( f->reconCount );
  TMC2_TRY( f->d_boundaryType.alloc( M ) );
  const int sid = ctx->stageBegin( "boundary_points" );
  hipLaunchKernelGGL( boundaryTypeKernel, dim3( ( M + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, f->d_pointToPixel.p, M,
                      f->d_occVideo.p, f->canvasW, f->canvasH, f->occPrecision, f->d_boundaryType.p );
  ctx->stageEnd( sid );
  TMC2_HIP( hipGetLastError() );
  f->haveBoundaryTypes = true;
  f->haveSmoothed      = false;
  return TMC2_OK;
}

// attribute: host, u16 [2 maps][3 channels][H][W] -- the decoded attribute frames of this point-cloud frame; NULL: the
// frames already on the device (tmc2_codec_set_decoded_attribute_yuv420)
int colorPointCloud( tmc2_frame* f, const uint16_t* attribute ) {
  TMC2_TRY( needReconstruction( f, "colorPointCloud" ) );
  tmc2_ctx*      ctx  = f->ctx;
  hipStream_t    s    = ctx->stream;
  const uint32_t M    = uint32_t( f->reconCount );
  const size_t   area = size_t( f->canvasW ) * f->canvasH;
  if ( attribute ) {
    TMC2_TRY( f->d_attr16.alloc( 6 * area ) );
    TMC2_HIP( hipMemcpyAsync( f->d_attr16.p, attribute, 6 * area * sizeof( uint16_t ), hipMemcpyHostToDevice, s ) );
    f->haveAttr16 = true;
  } else if ( !f->haveAttr16 ) {
    setError( "colorPointCloud: no decoded attribute frames (pass them, or tmc2_codec_set_decoded_attribute_yuv420 first)" );
    return TMC2_E_STATE;
  }
  TMC2_TRY( f->d_colors16.alloc( M ) );
  const int sid = ctx->stageBegin( "color_point_cloud" );
  hipLaunchKernelGGL( colorGatherKernel, dim3( ( M + 255 ) / 256 ), dim3( 256 ), 0, s, f->d_pointToPixel.p, M, f->d_attr16.p,
                      f->canvasW, f->canvasH, reinterpret_cast<ushort4*>( f->d_colors16.p ) );
  ctx->stageEnd( sid );
  TMC2_HIP( hipStreamSynchronize( s ) );  // the caller's buffer is free again
  TMC2_HIP( hipGetLastError() );
  f->haveColors16  = true;
  f->haveRgbPost   = false;
  return TMC2_OK;
}

int smoothPointCloudGrid( tmc2_frame* f, int gridSize, double thresholdSmoothing ) {
  TMC2_TRY( needReconstruction( f, "smoothPointCloudPostprocess" ) );
  if ( !f->haveBoundaryTypes || f->haveSmoothed ) TMC2_TRY( identifyBoundaryPoints( f ) );  // (a previous run left 3s behind)
  if ( gridSize < 2 || gridSize > 64 ) {
    setError( "smoothPointCloudPostprocess: gridSize %d unsupported", gridSize );
    return TMC2_E_UNSUPPORTED;
  }
  tmc2_ctx*      ctx = f->ctx;
  hipStream_t    s   = ctx->stream;
  const uint32_t M   = uint32_t( f->reconCount );
  const dim3     blk( 256 ), grdM( ( M + 255 ) / 256 );
  // the tree over the reconstruction knows the bounding box: the grid spans [0, max coordinate]
  const int maxSize = std::max( std::max( f->reconTree.hi[0], f->reconTree.hi[1] ), f->reconTree.hi[2] );
  GridGeom  g;
  g.gridSize = gridSize, g.half = gridSize / 2;
  g.w     = ( maxSize + gridSize - 1 ) / gridSize;
  g.disth = std::max( gridSize / 2, 1 );
  g.th    = gridSize * g.w;
  TMC2_TRY( f->d_reconSmoothed.alloc( M ) );
  const size_t cellCount = size_t( g.w ) * g.w * g.w;
  if ( cellCount == 0 || cellCount > ( size_t( 1 ) << 31 ) ) {
    setError( "smoothPointCloudPostprocess: grid of %d^3 cells unsupported", g.w );
    return TMC2_E_UNSUPPORTED;
  }
  DevBuf<uint32_t> d_flag, d_slot, d_small;
  TMC2_TRY( d_flag.alloc( cellCount ) );
  TMC2_TRY( d_slot.alloc( cellCount ) );
  TMC2_TRY( d_small.alloc( 4 ) );
  const int sid = ctx->stageBegin( "geometry_smoothing" );
  TMC2_HIP( hipMemsetAsync( d_flag.p, 0, cellCount * 4, s ) );
  TMC2_HIP( hipMemsetAsync( d_small.p, 0, 16, s ) );
  hipLaunchKernelGGL( markCellsKernel, grdM, blk, 0, s, f->d_recon.p, f->d_boundaryType.p, M, g, d_flag.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_slot.p, cellCount, d_small.p ) );
  uint32_t cells = 0;
  TMC2_HIP( hipMemcpyAsync( &cells, d_small.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  DevBuf<CellAcc> d_cells;
  TMC2_TRY( d_cells.alloc( std::max( cells, 1u ) ) );
  if ( cells ) hipLaunchKernelGGL( initCellsKernel, dim3( ( cells + 255 ) / 256 ), blk, 0, s, d_cells.p, cells );
  hipLaunchKernelGGL( accumulateCellsKernel, grdM, blk, 0, s, f->d_recon.p, f->d_pointToPixel.p, M, f->d_blockToPatch.p,
                      f->canvasW / 16, g, d_flag.p, d_slot.p, d_cells.p );
  hipLaunchKernelGGL( smoothGridKernel, grdM, blk, 0, s, f->d_recon.p, M, g, d_slot.p, d_cells.p, int( thresholdSmoothing ),
                      f->d_reconSmoothed.p, f->d_boundaryType.p, d_small.p + 1 );
  ctx->stageEnd( sid );
  uint32_t err = 0;
  TMC2_HIP( hipMemcpyAsync( &err, d_small.p + 1, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  if ( err ) {
    setError( "smoothPointCloudPostprocess: a grid cell holds more than 65535 points or a coordinate sum beyond 2^24 (the "
              "reference's uint16 count / float sum leave their exact range there)" );
    return TMC2_E_UNSUPPORTED;
  }
  f->haveSmoothed = true;
  return TMC2_OK;
}

int transferColors16bitBP( tmc2_frame* f ) {
  TMC2_TRY( needReconstruction( f, "transferColors16bitBP" ) );
  if ( !f->haveSmoothed || !f->haveColors16 ) {
    setError( "transferColors16bitBP: needs the smoothed cloud and its 16-bit colours (colorPointCloud, smoothPointCloudPostprocess)" );
    return TMC2_E_STATE;
  }
  tmc2_ctx*      ctx = f->ctx;
  hipStream_t    s   = ctx->stream;
  const uint32_t M   = uint32_t( f->reconCount );
  const dim3     blk( 256 ), grdM( ( M + 255 ) / 256 );
  DevBuf<uint32_t> d_flag, d_rank, d_small;
  TMC2_TRY( d_flag.alloc( M ) );
  TMC2_TRY( d_rank.alloc( M ) );
  TMC2_TRY( d_small.alloc( 4 ) );
  int sid = ctx->stageBegin( "transfer_colors16" );
  TMC2_HIP( hipMemsetAsync( d_small.p, 0, 16, s ) );
  hipLaunchKernelGGL( movedFlagKernel, grdM, blk, 0, s, f->d_boundaryType.p, M, d_flag.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_rank.p, M, d_small.p ) );
  uint32_t K = 0;
  TMC2_HIP( hipMemcpyAsync( &K, d_small.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  ctx->stageEnd( sid );
  if ( K == 0 ) return TMC2_OK;  // nothing moved: every point keeps its colour
  if ( M < 8 ) {
    setError( "transferColors16bitBP: fewer than 8 points" );
    return TMC2_E_UNSUPPORTED;
  }
  // the tree over the smoothed cloud (the one over the cloud before smoothing is S18's)
  DevBuf<Pt>       d_treePts;
  DevBuf<uint32_t> d_perm;
  DevBuf<KdNode>   d_nodes;
  TreeDev          tt;
  {
    const int kt = ctx->stageBegin( "kdtree_build_smoothed" );
    TMC2_TRY( buildKdTreeDevice( ctx, f->d_reconSmoothed.p, M, d_treePts, d_perm, d_nodes, tt.lo, tt.hi, tt.depth ) );
    ctx->stageEnd( kt );
  }
  tt.ptsTree = d_treePts.p, tt.perm = d_perm.p, tt.nodes = d_nodes.p, tt.n = M, tt.queriesBounded = true;
  const uint32_t entries = K * 8;
  DevBuf<uint32_t> d_moved, d_idx8, d_dist8, d_nn1, d_nn1Dist, d_count, d_offset, d_cursor;
  DevBuf<Pt>       d_queries, d_partPts;
  DevBuf<uint64_t> d_refined;
  DevBuf<uint2>    d_bucket;
  TMC2_TRY( d_moved.alloc( K ) );
  TMC2_TRY( d_queries.alloc( K ) );
  TMC2_TRY( d_idx8.alloc( entries ) );
  TMC2_TRY( d_dist8.alloc( entries ) );
  TMC2_TRY( d_partPts.alloc( entries ) );
  TMC2_TRY( d_nn1.alloc( entries ) );
  TMC2_TRY( d_nn1Dist.alloc( entries ) );
  TMC2_TRY( d_count.alloc( K ) );
  TMC2_TRY( d_offset.alloc( K ) );
  TMC2_TRY( d_cursor.alloc( K ) );
  TMC2_TRY( d_refined.alloc( K ) );
  TMC2_TRY( d_bucket.alloc( entries ) );
  const dim3     grdK( ( K + 255 ) / 256 ), grdE( ( entries + 255 ) / 256 );
  const ushort4* colors = reinterpret_cast<const ushort4*>( f->d_colors16.p );
  hipLaunchKernelGGL( movedGatherKernel, grdM, blk, 0, s, f->d_boundaryType.p, d_rank.p, f->d_reconSmoothed.p, M, d_moved.p,
                      d_queries.p );
  TMC2_TRY( launchKnnTree( ctx, reconTreeDev( f ), d_queries.p, K, 8, d_idx8.p, d_dist8.p, "knn8_moved_in_recon" ) );
  sid = ctx->stageBegin( "transfer_colors16" );
  hipLaunchKernelGGL( forwardColor16Kernel, grdK, blk, 0, s, d_idx8.p, d_dist8.p, colors, f->d_recon.p, K,
                      reinterpret_cast<ushort4*>( d_refined.p ), d_partPts.p );
  ctx->stageEnd( sid );
  TMC2_TRY( launchKnnTree( ctx, tt, d_partPts.p, entries, 1, d_nn1.p, d_nn1Dist.p, "knn1_neighbours_in_smoothed" ) );
  sid = ctx->stageBegin( "transfer_colors16" );
  TMC2_HIP( hipMemsetAsync( d_count.p, 0, size_t( K ) * 4, s ) );
  TMC2_HIP( hipMemsetAsync( d_cursor.p, 0, size_t( K ) * 4, s ) );
  hipLaunchKernelGGL( backwardVoteKernel<false>, grdE, blk, 0, s, d_idx8.p, d_nn1.p, d_nn1Dist.p, entries, colors,
                      f->d_boundaryType.p, d_rank.p, d_count.p, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint2*)nullptr );
  TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_offset.p, K, nullptr ) );
  hipLaunchKernelGGL( backwardVoteKernel<true>, grdE, blk, 0, s, d_idx8.p, d_nn1.p, d_nn1Dist.p, entries, colors,
                      f->d_boundaryType.p, d_rank.p, d_count.p, d_offset.p, d_cursor.p, d_bucket.p );
  // every vote reads the colours as they were before the transfer; only then are the moved points rewritten
  hipLaunchKernelGGL( combineColor16Kernel, grdK, blk, 0, s, d_moved.p, d_count.p, d_offset.p, d_bucket.p, d_idx8.p, colors,
                      reinterpret_cast<ushort4*>( d_refined.p ), K, d_small.p + 1 );
  hipLaunchKernelGGL( scatterColor16Kernel, grdK, blk, 0, s, d_moved.p, reinterpret_cast<const ushort4*>( d_refined.p ), K,
                      reinterpret_cast<ushort4*>( f->d_colors16.p ) );
  ctx->stageEnd( sid );
  uint32_t err = 0;
  TMC2_HIP( hipMemcpyAsync( &err, d_small.p + 1, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  if ( err ) {
    setError( "transferColors16bitBP: a candidate list hit std::sort's depth limit (heapsort fallback not reproduced)" );
    return TMC2_E_UNSUPPORTED;
  }
  f->haveRgbPost = false;
  return TMC2_OK;
}

int convertYuv16ToRgb8( tmc2_frame* f ) {
  TMC2_TRY( needReconstruction( f, "convertYUV16ToRGB8" ) );
  if ( !f->haveColors16 ) {
    setError( "convertYUV16ToRGB8: no 16-bit colours (colorPointCloud first)" );
    return TMC2_E_STATE;
  }
  tmc2_ctx*      ctx = f->ctx;
  const uint32_t M   = uint32_t( f->reconCount );
  TMC2_TRY( f->d_rgbPost.alloc( size_t( M ) * 4 ) );
  const int sid = ctx->stageBegin( "yuv16_to_rgb8" );
  hipLaunchKernelGGL( yuv16ToRgb8Kernel, dim3( ( M + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream,
                      reinterpret_cast<const ushort4*>( f->d_colors16.p ), M, reinterpret_cast<uchar4*>( f->d_rgbPost.p ) );
  ctx->stageEnd( sid );
  TMC2_HIP( hipGetLastError() );
  f->haveRgbPost = true;
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

int tmc2_codec_identify_boundary_points( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::identifyBoundaryPoints( f );
}
int tmc2_codec_color_point_cloud( tmc2_frame* f, const uint16_t* attribute ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::colorPointCloud( f, attribute );
}
int tmc2_codec_smooth_point_cloud_postprocess( tmc2_frame* f, int gridSize, double thresholdSmoothing ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::smoothPointCloudGrid( f, gridSize, thresholdSmoothing );
}
int tmc2_codec_transfer_colors_16bit_bp( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::transferColors16bitBP( f );
}
int tmc2_codec_convert_yuv16_to_rgb8( tmc2_frame* f ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::convertYuv16ToRgb8( f );
}

int tmc2_frame_get_post_reconstruction( tmc2_frame* f, int16_t* xyz, uint16_t* colors16, uint8_t* rgb, uint16_t* boundaryType ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  TMC2_TRY( tmc2::needReconstruction( f, "get_post_reconstruction" ) );
  const size_t M = size_t( f->reconCount );
  hipStream_t  s = f->ctx->stream;
  if ( xyz ) {
    std::vector<tmc2::Pt> h( M );
    const tmc2::Pt* src = f->haveSmoothed ? f->d_reconSmoothed.p : f->d_recon.p;
    TMC2_HIP( hipMemcpyAsync( h.data(), src, M * sizeof( tmc2::Pt ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    for ( size_t i = 0; i < M; ++i ) xyz[3 * i] = h[i].x, xyz[3 * i + 1] = h[i].y, xyz[3 * i + 2] = h[i].z;
  }
  if ( colors16 ) {
    if ( !f->haveColors16 ) {
      tmc2::setError( "get_post_reconstruction: no 16-bit colours" );
      return TMC2_E_STATE;
    }
    std::vector<uint64_t> h( M );
    TMC2_HIP( hipMemcpyAsync( h.data(), f->d_colors16.p, M * 8, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    for ( size_t i = 0; i < M; ++i )
      for ( int k = 0; k < 3; ++k ) colors16[3 * i + k] = uint16_t( h[i] >> ( 16 * k ) );
  }
  if ( rgb ) {
    if ( !f->haveRgbPost ) {
      tmc2::setError( "get_post_reconstruction: no 8-bit colours (tmc2_codec_convert_yuv16_to_rgb8 first)" );
      return TMC2_E_STATE;
    }
    std::vector<uint8_t> h( M * 4 );
    TMC2_HIP( hipMemcpyAsync( h.data(), f->d_rgbPost.p, M * 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    for ( size_t i = 0; i < M; ++i )
      for ( int k = 0; k < 3; ++k ) rgb[3 * i + k] = h[4 * i + k];
  }
  if ( boundaryType ) {
    if ( !f->haveBoundaryTypes ) {
      tmc2::setError( "get_post_reconstruction: no boundary types" );
      return TMC2_E_STATE;
    }
    std::vector<uint8_t> h( M );
    TMC2_HIP( hipMemcpyAsync( h.data(), f->d_boundaryType.p, M, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    for ( size_t i = 0; i < M; ++i ) boundaryType[i] = h[i];
  }
  return TMC2_OK;
}
}
