// images.hip -- occupancy / geometry image generation and padding (S11-S16) on gfx950.
//
// Replaces, for one frame (reference: source/lib/PccLibEncoder/source/PCCEncoder.cpp unless noted)
//   generateOccupancyMap(tile)                :3767-3784   + PCCPatch::patch2Canvas (PccLibCommon/source/PCCPatch.cpp:192-251)
//   generateOccupancyMapVideo                 :806-861
//   generateBlockToPatchFromOccupancyMapVideo  PccLibCommon/source/PCCCodec.cpp:1736-1775
//   generateIntraImage                        :3929-3992
//   dilate3DPadding (geometryPadding = 0)     :5951-6130
//   dilateGroupGeometryVideo                  :3717-3739
//
// Layout in HBM: the canvases are plain row-major planes (u8 occupancy, u16 depth); only the luma plane of
// the geometry images exists on the device -- the reference's two chroma planes stay identically zero
// through generateIntraImage and dilate3DPadding (rounded means of zeros), the host adaptor hands the
// video encoder zero planes.
//
//   * raster: one 256-lane workgroup per 16x16 patch block (the same tile list as the patch kernels),
//     coalesced reads of the patch-local depth pools, scatter through patch2Canvas.  Patches never overlap
//     on valid pixels (a patch box may only cover FREE blocks of earlier patches), so no ordering is needed.
//   * block dilation: one workgroup per canvas block, the 16x16 tile lives in LDS; the reference's
//     wavefront-by-wavefront 4-neighbour growth (new value = rounded mean of the neighbours filled in the
//     previous wavefront) is a level-synchronous loop with one barrier per level.
//   * empty blocks copy their left neighbour's last column (first block column: the bottom row of the block
//     above) IN RASTER BLOCK ORDER in the reference; that is a prefix "carry" along each pixel row, done by
//     one lane per pixel row after the first block column has been resolved top-down.
#include <memory>

#include "internal.h"

namespace tmc2 {
namespace {

__device__ __forceinline__ void toCanvas( const PlaceDev& p, int u, int v, int& x, int& y ) {
  if ( p.orient == 0 ) {
    x = u + p.u0 * 16;
    y = v + p.v0 * 16;
  } else {  // PATCH_ORIENTATION_SWAP
    x = v + p.u0 * 16;
    y = u + p.v0 * 16;
  }
}

__global__ __launch_bounds__( 256 ) void rasterTileKernel( const PlaceDev* __restrict__ place,
                                                            const uint32_t* __restrict__ tilePatch,
                                                            const int16_t* __restrict__ depth0,
                                                            const int16_t* __restrict__ depth1, int W, int H,
                                                            uint8_t* __restrict__ occMap, uint16_t* __restrict__ geo0,
                                                            uint16_t* __restrict__ geo1, uint32_t* __restrict__ error ) {
  const uint32_t tile  = blockIdx.x;
  const PlaceDev p     = place[tilePatch[tile]];
  const int      local = int( tile ) - p.tileBase;
  const int      u = ( local % p.sizeU0 ) * 16 + int( threadIdx.x & 15 ), v = ( local / p.sizeU0 ) * 16 + int( threadIdx.x >> 4 );
  if ( u >= p.sizeU || v >= p.sizeV ) return;
  const size_t  q = size_t( p.depthOff ) + size_t( v ) * p.sizeU + u;
  const int16_t d = depth0[q];
  if ( d == 32767 ) return;
  int x, y;
  toCanvas( p, u, v, x, y );
  if ( x >= W || y >= H ) {  // the reference exit(180)s (PCCPatch.cpp:236-243); reported as an error code here
    *error = 1;
    return;
  }
  const size_t c = size_t( y ) * W + x;
  occMap[c]      = 1;
  geo0[c]        = uint16_t( d );
  geo1[c]        = uint16_t( depth1[q] );
}

// occupancy video: one cell per p x p pixels, 1 if any pixel is occupied
__global__ __launch_bounds__( 256 ) void occVideoKernel( const uint8_t* __restrict__ occMap, int W, int Wv, int Hv, int p,
                                                          uint8_t* __restrict__ occVideo ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= Wv * Hv ) return;
  const int xv = i % Wv, yv = i / Wv;
  uint32_t  any = 0;
  for ( int j = 0; j < p; ++j )
    for ( int k = 0; k < p; ++k ) any |= occMap[size_t( yv * p + j ) * W + xv * p + k];
  occVideo[i] = any ? 1 : 0;
}

// blockToPatch[block] = 1 + list position of the LAST patch (packing order) whose box covers the block,
// provided the block has any occupancy-video pixel set; 0 otherwise
__global__ __launch_bounds__( 256 ) void blockToPatchKernel( const PlaceDev* __restrict__ place, int P,
                                                              const uint8_t* __restrict__ occVideo, int Wb, int Hb,
                                                              int Wv, int p, uint32_t* __restrict__ blockToPatch ) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if ( b >= Wb * Hb ) return;
  const int bx = b % Wb, by = b / Wb;
  const int cells = 16 / p;
  bool      any = false;
  for ( int j = 0; j < cells && !any; ++j )
    for ( int i = 0; i < cells; ++i ) any |= occVideo[size_t( by * cells + j ) * Wv + bx * cells + i] != 0;
  uint32_t owner = 0;
  if ( any ) {
    for ( int k = 0; k < P; ++k ) {
      const PlaceDev q = place[k];
      const int      w = q.orient == 0 ? q.sizeU0 : q.sizeV0, h = q.orient == 0 ? q.sizeV0 : q.sizeU0;
      if ( bx >= q.u0 && bx < q.u0 + w && by >= q.v0 && by < q.v0 + h ) owner = uint32_t( k + 1 );
    }
  }
  blockToPatch[b] = owner;
}

// block dilation of partially filled blocks; empty blocks are flagged for the carry passes
__global__ __launch_bounds__( 256 ) void dilateBlockKernel( const uint8_t* __restrict__ occMap, int W, int H,
                                                             uint16_t* __restrict__ geo /* 2 maps */,
                                                             uint8_t* __restrict__ emptyBlock ) {
  __shared__ uint32_t lvl[18][18];  // level with a one-pixel apron of "outside the block" (0xFFFFFFFF)
  __shared__ uint32_t val[18][18];
  __shared__ int      counters[2];
  const int Wb = W / 16;
  const int bx = blockIdx.x % Wb, by = blockIdx.x / Wb;
  const int i = threadIdx.x & 15, j = threadIdx.x >> 4;
  uint16_t* img = geo + size_t( blockIdx.y ) * W * H;
  const size_t c = size_t( by * 16 + j ) * W + bx * 16 + i;
  for ( int t = threadIdx.x; t < 18 * 18; t += 256 ) ( &lvl[0][0] )[t] = 0xFFFFFFFFu;
  if ( threadIdx.x == 0 ) counters[0] = 0;
  __syncthreads();
  uint32_t myLevel = occMap[c] ? 1u : 0u;
  uint32_t myVal   = img[c];
  lvl[j + 1][i + 1] = myLevel;
  val[j + 1][i + 1] = myVal;
  if ( myLevel ) atomicAdd( &counters[0], 1 );
  __syncthreads();
  int filled = counters[0];
  if ( filled == 0 ) {
    if ( threadIdx.x == 0 && blockIdx.y == 0 ) emptyBlock[blockIdx.x] = 1;
    return;
  }
  if ( threadIdx.x == 0 && blockIdx.y == 0 ) emptyBlock[blockIdx.x] = 0;
  for ( uint32_t it = 1; filled < 256; ++it ) {
    uint32_t sum = 0, cnt = 0;
    if ( myLevel == 0 ) {
      if ( lvl[j][i + 1] == it ) { sum += val[j][i + 1]; ++cnt; }
      if ( lvl[j + 1][i] == it ) { sum += val[j + 1][i]; ++cnt; }
      if ( lvl[j + 1][i + 2] == it ) { sum += val[j + 1][i + 2]; ++cnt; }
      if ( lvl[j + 2][i + 1] == it ) { sum += val[j + 2][i + 1]; ++cnt; }
    }
    __syncthreads();
    if ( threadIdx.x == 0 ) counters[1] = 0;
    __syncthreads();
    if ( cnt ) {
      myLevel           = it + 1;
      myVal             = ( sum + cnt / 2 ) / cnt;
      lvl[j + 1][i + 1] = myLevel;
      val[j + 1][i + 1] = myVal;
      atomicAdd( &counters[1], 1 );
    }
    __syncthreads();
    filled += counters[1];
    __syncthreads();
  }
  img[c] = uint16_t( myVal );
}

// first block column, top-down: an empty block repeats the bottom row of the block above
__global__ void carryFirstColumnKernel( const uint8_t* __restrict__ emptyBlock, int W, int H, uint16_t* __restrict__ geo,
                                        uint8_t* __restrict__ resolved ) {
  const int x = threadIdx.x;  // 0..15
  uint16_t* img = geo + size_t( blockIdx.x ) * W * H;
  const int Wb = W / 16, Hb = H / 16;
  uint16_t  carry = 0;
  for ( int by = 0; by < Hb; ++by ) {
    if ( emptyBlock[size_t( by ) * Wb] ) {
      if ( by > 0 )
        for ( int j = 0; j < 16; ++j ) img[size_t( by * 16 + j ) * W + x] = carry;
      // by == 0: block (0,0) keeps its zeros; carry below is its (zero) bottom row
      if ( by == 0 ) carry = img[size_t( 15 ) * W + x];
    } else {
      carry = img[size_t( by * 16 + 15 ) * W + x];
    }
  }
  (void)resolved;
}

// remaining block columns, left to right: an empty block repeats, row by row, the last column of its left neighbour
__global__ __launch_bounds__( 64 ) void carryRowsKernel( const uint8_t* __restrict__ emptyBlock, int W, int H,
                                                          uint16_t* __restrict__ geo ) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if ( y >= H ) return;
  uint16_t* row = geo + size_t( blockIdx.y ) * W * H + size_t( y ) * W;
  const int Wb  = W / 16;
  const uint8_t* e = emptyBlock + size_t( y / 16 ) * Wb;
  uint16_t carry = row[15];
  for ( int bx = 1; bx < Wb; ++bx ) {
    if ( e[bx] ) {
#pragma unroll
      for ( int i = 0; i < 16; ++i ) row[bx * 16 + i] = carry;
    } else {
      carry = row[bx * 16 + 15];
    }
  }
}

__global__ __launch_bounds__( 256 ) void groupDilateKernel( const uint8_t* __restrict__ occVideo, int W, int H, int Wv,
                                                             int p, uint16_t* __restrict__ geo ) {
  const size_t c = size_t( blockIdx.x ) * blockDim.x + threadIdx.x;
  if ( c >= size_t( W ) * H ) return;
  const int x = int( c % W ), y = int( c / W );
  if ( occVideo[size_t( y / p ) * Wv + x / p ] ) return;
  const uint32_t a = geo[c], b = geo[size_t( W ) * H + c];
  const uint16_t avg = uint16_t( ( a + b + 1 ) >> 1 );
  geo[c]                   = avg;
  geo[size_t( W ) * H + c] = avg;
}

}  // namespace

// placement table + list of 16x16 patch blocks, both in packing order: what the raster and the reconstruction kernels index
int uploadPlacement( tmc2_frame* f ) {
  const int P = int( f->patches.size() );
  size_t    tiles = 0;
  for ( int k = 0; k < P; ++k ) {
    const tmc2_patch& t = f->patches[size_t( f->packOrder[size_t( k )] )];
    if ( t.patchOrientation != 0 && t.patchOrientation != 1 ) {
      setError( "patch orientation %d unsupported", t.patchOrientation );
      return TMC2_E_UNSUPPORTED;
    }
    if ( t.sizeU0 < 0 || t.sizeV0 < 0 || t.u0 < 0 || t.v0 < 0 ) {
      setError( "patch %d: negative block size or position", k );
      return TMC2_E_INVALID;
    }
    tiles += size_t( t.sizeU0 ) * size_t( t.sizeV0 );
  }
  // both tables in the context's page-locked staging: the copies are DMA straight from it and nobody waits for them here (the
  // caller synchronises the stream before the staging is refilled)
  const size_t placeBytes = ( size_t( P ) * sizeof( PlaceDev ) + 15 ) & ~size_t( 15 );
  uint8_t*     staging    = f->ctx->hostTables.get<uint8_t>( std::max<size_t>( placeBytes + tiles * 4, size_t( 1 ) << 20 ) );
  if ( !staging ) {
    setError( "uploadPlacement: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  PlaceDev* place     = reinterpret_cast<PlaceDev*>( staging );
  uint32_t* tilePatch = reinterpret_cast<uint32_t*>( staging + placeBytes );
  size_t    at        = 0;
  for ( int k = 0; k < P; ++k ) {
    const tmc2_patch& t = f->patches[size_t( f->packOrder[size_t( k )] )];
    PlaceDev&         d = place[size_t( k )];
    d.u0 = t.u0, d.v0 = t.v0, d.orient = t.patchOrientation;
    d.sizeU = t.sizeU, d.sizeV = t.sizeV, d.sizeU0 = t.sizeU0, d.sizeV0 = t.sizeV0;
    d.tileBase = int32_t( at );
    d.depthOff = t.depthOffset;
    d.u1 = t.u1, d.v1 = t.v1, d.d1 = t.d1;
    d.axN = t.normalAxis, d.axT = t.tangentAxis, d.axB = t.bitangentAxis, d.mode = t.projectionMode;
    d.pad = 0;
    for ( size_t b = size_t( t.sizeU0 ) * size_t( t.sizeV0 ); b > 0; --b ) tilePatch[at++] = uint32_t( k );
  }
  hipStream_t s = f->ctx->stream;
  TMC2_TRY( f->d_place.alloc( size_t( std::max( P, 1 ) ) ) );
  TMC2_TRY( f->d_tilePatch.alloc( std::max<size_t>( tiles, 1 ) ) );
  if ( P ) {
    TMC2_HIP( hipMemcpyAsync( f->d_place.p, place, size_t( P ) * sizeof( PlaceDev ), hipMemcpyHostToDevice, s ) );
    if ( tiles ) TMC2_HIP( hipMemcpyAsync( f->d_tilePatch.p, tilePatch, tiles * 4, hipMemcpyHostToDevice, s ) );
  }
  f->tileCount = uint32_t( tiles );
  return TMC2_OK;
}

// decoder side: a frame that has no source cloud, only what the bitstream carries -- patch records in list order, the
// decoded occupancy video and the two decoded geometry maps.  blockToPatch is derived here (S13).
int createDecoderFrame( tmc2_ctx* ctx, const tmc2_patch* patches, int count, int W, int H, int occPrecision, const uint8_t* occVideo,
                        const uint16_t* geometry, tmc2_frame** out ) {
  if ( W <= 0 || H <= 0 || W % 16 || H % 16 || W > kMaxCanvasDim || H > kMaxCanvasDim || occPrecision < 1 || 16 % occPrecision ) {
    setError( "decoder_frame_create: unsupported geometry %dx%d (at most %d a side), precision %d", W, H, kMaxCanvasDim, occPrecision );
    return TMC2_E_UNSUPPORTED;
  }
  std::unique_ptr<tmc2_frame> f( new tmc2_frame() );
  f->ticket.bind( ctx );
  f->ctx = ctx;
  f->n   = 0;
  f->patches.assign( patches, patches + count );
  f->packOrder.resize( size_t( count ) );
  for ( int k = 0; k < count; ++k ) {
    f->packOrder[size_t( k )] = k;
    const tmc2_patch& t = f->patches[size_t( k )];
    // records come from a bitstream: everything the kernels index with is checked here (the axes address a 3-vector)
    const int axes = ( 1 << t.normalAxis ) | ( 1 << t.tangentAxis ) | ( 1 << t.bitangentAxis );
    if ( t.normalAxis < 0 || t.normalAxis > 2 || t.tangentAxis < 0 || t.tangentAxis > 2 || t.bitangentAxis < 0 ||
         t.bitangentAxis > 2 || axes != 7 || t.projectionMode < 0 || t.projectionMode > 1 || t.patchOrientation < 0 ||
         t.patchOrientation > 1 || t.sizeU0 <= 0 || t.sizeV0 <= 0 ) {
      setError( "decoder_frame_create: patch %d: axes (%d, %d, %d) / projection mode %d / orientation %d / block size %dx%d invalid",
                k, t.normalAxis, t.tangentAxis, t.bitangentAxis, t.projectionMode, t.patchOrientation, t.sizeU0, t.sizeV0 );
      return TMC2_E_INVALID;
    }
    const int64_t bw = t.patchOrientation == 0 ? t.sizeU0 : t.sizeV0, bh = t.patchOrientation == 0 ? t.sizeV0 : t.sizeU0;
    if ( t.u0 < 0 || t.v0 < 0 || ( int64_t( t.u0 ) + bw ) * 16 > W || ( int64_t( t.v0 ) + bh ) * 16 > H ) {
      setError( "decoder_frame_create: patch %d lies outside the %dx%d canvas", k, W, H );
      return TMC2_E_INVALID;
    }
  }
  f->packMatch.assign( size_t( count ), -1 );
  f->havePatches = f->havePacking = true;
  hipStream_t  s    = ctx->stream;
  const size_t area = size_t( W ) * H;
  const int    Wv = W / occPrecision, Hv = H / occPrecision, Wb = W / 16, Hb = H / 16;
  TMC2_TRY( uploadPlacement( f.get() ) );
  TMC2_TRY( f->d_occVideo.alloc( size_t( Wv ) * Hv ) );
  TMC2_TRY( f->d_blockToPatch.alloc( size_t( Wb ) * Hb ) );
  TMC2_TRY( f->d_geo.alloc( 2 * area ) );
  TMC2_TRY( f->d_occMap.alloc( area ) );
  TMC2_HIP( hipMemcpyAsync( f->d_occVideo.p, occVideo, size_t( Wv ) * Hv, hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( f->d_geo.p, geometry, 2 * area * sizeof( uint16_t ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemsetAsync( f->d_occMap.p, 0, area, s ) );  // (the precise encoder-side map does not exist on this side)
  const int sid = ctx->stageBegin( "block_to_patch" );
  hipLaunchKernelGGL( blockToPatchKernel, dim3( ( Wb * Hb + 255 ) / 256 ), dim3( 256 ), 0, s, f->d_place.p, count, f->d_occVideo.p, Wb,
                      Hb, Wv, occPrecision, f->d_blockToPatch.p );
  ctx->stageEnd( sid );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  f->canvasW = W, f->canvasH = H, f->occPrecision = occPrecision;
  f->haveGeometryImages = true;
  *out                  = f.release();
  return TMC2_OK;
}

int generateGeometryImages( tmc2_frame* f, int W, int H, int occRes, int occPrecision ) {
  if ( !f->havePacking ) {
    setError( "generateGeometryImages: frame not packed" );
    return TMC2_E_STATE;
  }
  if ( occRes != 16 || W <= 0 || H <= 0 || W % 16 || H % 16 || W > kMaxCanvasDim || H > kMaxCanvasDim || occPrecision < 1 ||
       16 % occPrecision ) {
    setError( "generateGeometryImages: unsupported geometry %dx%d (at most %d a side), occupancyResolution %d, precision %d", W, H,
              kMaxCanvasDim, occRes, occPrecision );
    return TMC2_E_UNSUPPORTED;
  }
  // new canvases: whatever was derived from the old ones is stale
  f->haveGeometryImages = false;
  invalidateReconstruction( f );
  tmc2_ctx*   ctx = f->ctx;
  hipStream_t s   = ctx->stream;
  const int   P   = int( f->patches.size() );
  const size_t area = size_t( W ) * H;
  const int    Wv = W / occPrecision, Hv = H / occPrecision, Wb = W / 16, Hb = H / 16;
  TMC2_TRY( uploadPlacement( f ) );
  DevBuf<PlaceDev>& d_place     = f->d_place;
  DevBuf<uint32_t>& d_tilePatch = f->d_tilePatch;
  // (the "patch outside the canvas" flag is a page-locked word of the context: a kernel that finds one stores there, the host reads it
  //  after the synchronisation below -- no copy)
  volatile uint32_t* h_err = ctx->answerLine( tmc2_ctx::kAnswerGeoError );
  *h_err                   = 0;
  DevBuf<uint8_t>  d_empty;
  TMC2_TRY( d_empty.alloc( size_t( Wb ) * Hb ) );
  TMC2_TRY( f->d_occMap.alloc( area ) );
  TMC2_TRY( f->d_occVideo.alloc( size_t( Wv ) * Hv ) );
  TMC2_TRY( f->d_blockToPatch.alloc( size_t( Wb ) * Hb ) );
  TMC2_TRY( f->d_geo.alloc( 2 * area ) );
  const int sid = ctx->stageBegin( "geometry_images" );
  TMC2_TRY( fillRegions( ctx, {{f->d_occMap.p, area, 0}, {f->d_geo.p, 2 * area * sizeof( uint16_t ), 0}} ) );
  const dim3 blk( 256 );
  if ( f->tileCount )
    hipLaunchKernelGGL( rasterTileKernel, dim3( f->tileCount ), blk, 0, s, d_place.p, d_tilePatch.p,
                        f->d_depth0.p, f->d_depth1.p, W, H, f->d_occMap.p, f->d_geo.p, f->d_geo.p + area, const_cast<uint32_t*>( h_err ) );
  hipLaunchKernelGGL( occVideoKernel, dim3( ( Wv * Hv + 255 ) / 256 ), blk, 0, s, f->d_occMap.p, W, Wv, Hv, occPrecision,
                      f->d_occVideo.p );
  hipLaunchKernelGGL( blockToPatchKernel, dim3( ( Wb * Hb + 255 ) / 256 ), blk, 0, s, d_place.p, P, f->d_occVideo.p, Wb,
                      Hb, Wv, occPrecision, f->d_blockToPatch.p );
  hipLaunchKernelGGL( dilateBlockKernel, dim3( Wb * Hb, 2 ), blk, 0, s, f->d_occMap.p, W, H, f->d_geo.p, d_empty.p );
  hipLaunchKernelGGL( carryFirstColumnKernel, dim3( 2 ), dim3( 16 ), 0, s, d_empty.p, W, H, f->d_geo.p, (uint8_t*)nullptr );
  hipLaunchKernelGGL( carryRowsKernel, dim3( ( H + 63 ) / 64, 2 ), dim3( 64 ), 0, s, d_empty.p, W, H, f->d_geo.p );
  hipLaunchKernelGGL( groupDilateKernel, dim3( uint32_t( ( area + 255 ) / 256 ) ), blk, 0, s, f->d_occVideo.p, W, H, Wv,
                      occPrecision, f->d_geo.p );
  ctx->stageEnd( sid );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  const uint32_t err = *h_err;
  if ( err ) {
    setError( "generateGeometryImages: a patch falls outside the %dx%d canvas (the reference exits with code 180)", W, H );
    return TMC2_E_INVALID;
  }
  f->canvasW            = W;
  f->canvasH            = H;
  f->occPrecision       = occPrecision;
  f->haveGeometryImages = true;
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

int tmc2_encoder_generate_geometry_images( tmc2_frame* f, int width, int height, int occupancyPrecision ) {
  if ( !f ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  return tmc2::generateGeometryImages( f, width, height, 16, occupancyPrecision );
}

int tmc2_decoder_frame_create( tmc2_ctx* ctx, const tmc2_patch* patches, int count, int width, int height, int occupancyPrecision,
                               const uint8_t* occVideo, const uint16_t* geometry, tmc2_frame** out ) {
  if ( !ctx || !out || count < 0 || ( count && !patches ) || !occVideo || !geometry ) return TMC2_E_INVALID;
  *out = nullptr;
  tmc2::ApiScope scope( ctx );
  return tmc2::createDecoderFrame( ctx, patches, count, width, height, occupancyPrecision, occVideo, geometry, out );
}

int tmc2_frame_set_decoded_geometry( tmc2_frame* f, const uint8_t* occVideo, const uint16_t* geometry ) {
  if ( !f || !f->haveGeometryImages ) {
    tmc2::setError( "set_decoded_geometry: canvases not generated" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  hipStream_t    s    = f->ctx->stream;
  const size_t   area = size_t( f->canvasW ) * f->canvasH;
  if ( occVideo )
    TMC2_HIP( hipMemcpyAsync( f->d_occVideo.p, occVideo, area / ( size_t( f->occPrecision ) * f->occPrecision ),
                              hipMemcpyHostToDevice, s ) );
  if ( geometry ) TMC2_HIP( hipMemcpyAsync( f->d_geo.p, geometry, 2 * area * sizeof( uint16_t ), hipMemcpyHostToDevice, s ) );
  if ( occVideo ) {  // block ownership derives from the DECODED occupancy video (PCCEncoder.cpp:168, PCCCodec.cpp:1736-1775)
    const int Wb = f->canvasW / 16, Hb = f->canvasH / 16;
    hipLaunchKernelGGL( tmc2::blockToPatchKernel, dim3( ( Wb * Hb + 255 ) / 256 ), dim3( 256 ), 0, s, f->d_place.p,
                        int( f->patches.size() ), f->d_occVideo.p, Wb, Hb, f->canvasW / f->occPrecision, f->occPrecision,
                        f->d_blockToPatch.p );
  }
  tmc2::invalidateReconstruction( f );  // whatever was reconstructed from the previous canvases is stale
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

int tmc2_frame_device_images( tmc2_frame* f, void** occupancy, void** occVideo, void** blockToPatch, void** geometry ) {
  if ( !f || !f->haveGeometryImages ) {
    tmc2::setError( "device_images: not generated" );
    return TMC2_E_STATE;
  }
  if ( occupancy ) *occupancy = f->d_occMap.p;
  if ( occVideo ) *occVideo = f->d_occVideo.p;
  if ( blockToPatch ) *blockToPatch = f->d_blockToPatch.p;
  if ( geometry ) *geometry = f->d_geo.p;
  return TMC2_OK;
}

int tmc2_frame_get_geometry_images( tmc2_frame* f, uint8_t* occupancy, uint8_t* occVideo, uint32_t* blockToPatch,
                                    uint16_t* geometryD0, uint16_t* geometryD1 ) {
  if ( !f || !f->haveGeometryImages ) {
    tmc2::setError( "get_geometry_images: not generated" );
    return TMC2_E_STATE;
  }
  tmc2::ApiScope scope( f->ctx );
  hipStream_t  s    = f->ctx->stream;
  const size_t area = size_t( f->canvasW ) * f->canvasH;
  const size_t av   = area / ( size_t( f->occPrecision ) * f->occPrecision );
  if ( occupancy ) TMC2_HIP( hipMemcpyAsync( occupancy, f->d_occMap.p, area, hipMemcpyDeviceToHost, s ) );
  if ( occVideo ) TMC2_HIP( hipMemcpyAsync( occVideo, f->d_occVideo.p, av, hipMemcpyDeviceToHost, s ) );
  if ( blockToPatch ) TMC2_HIP( hipMemcpyAsync( blockToPatch, f->d_blockToPatch.p, area / 256 * 4, hipMemcpyDeviceToHost, s ) );
  if ( geometryD0 ) TMC2_HIP( hipMemcpyAsync( geometryD0, f->d_geo.p, area * 2, hipMemcpyDeviceToHost, s ) );
  if ( geometryD1 ) TMC2_HIP( hipMemcpyAsync( geometryD1, f->d_geo.p + area, area * 2, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}
}
