// weight_normal.hip -- projection-footprint axis weights (S0) on gfx950.
//
// Replaces PCCEncoder::calculateWeightNormal (reference: source/lib/PccLibEncoder/source/PCCEncoder.cpp:3569-3626,
// enhancedPP branch): rasterise the cloud onto the three axis-aligned planes, count the lit pixels,
// rank the three counts and map them to weights in [minWeightEPP, 1].
// Device: bit-plane rasterisation with atomicOr (3 x (2^bits)^2 bits = 1.5 MiB at vox10) + popcount
// reduction; host: the 3-element ranking and the two divisions (scalar, order-sensitive fp64).
#include <algorithm>

#include "internal.h"

namespace tmc2 {
namespace {

__global__ __launch_bounds__( 256 ) void footprintKernel( const Pt* __restrict__ pts, uint32_t n, int M,
                                                           uint32_t* __restrict__ bits ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const Pt       p  = pts[i];
  const int      p0 = min( M - 1, max( 0, int( p.x ) ) );
  const int      p1 = min( M - 1, max( 0, int( p.y ) ) );
  const int      p2 = min( M - 1, max( 0, int( p.z ) ) );
  const uint32_t plane = uint32_t( M ) * uint32_t( M );
  const uint32_t a = uint32_t( p2 ) * M + p1;              // seen along x
  const uint32_t b = uint32_t( p0 ) * M + p2 + plane;      // along y
  const uint32_t c = uint32_t( p1 ) * M + p0 + 2 * plane;  // along z
  atomicOr( &bits[a >> 5], 1u << ( a & 31 ) );
  atomicOr( &bits[b >> 5], 1u << ( b & 31 ) );
  atomicOr( &bits[c >> 5], 1u << ( c & 31 ) );
}

__global__ __launch_bounds__( 256 ) void popcountKernel( const uint32_t* __restrict__ bits, uint32_t wordsPerPlane,
                                                          uint32_t* __restrict__ counts ) {
  const uint32_t plane = blockIdx.y;
  uint32_t       acc   = 0;
  for ( uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < wordsPerPlane; w += gridDim.x * blockDim.x )
    acc += __popc( bits[plane * wordsPerPlane + w] );
  for ( int off = 32; off > 0; off >>= 1 ) acc += __shfl_down( acc, off, 64 );
  if ( ( threadIdx.x & 63 ) == 0 && acc ) atomicAdd( &counts[plane], acc );
}

}  // namespace

int weightNormal( tmc2_frame* f, int bits, double minWeightEPP, double w[3] ) {
  if ( bits < 3 || bits > 13 ) {  // (the bit planes are addressed by whole 32-bit words)
    setError( "weightNormal: geometryBitDepth3D=%d out of range", bits );
    return TMC2_E_INVALID;
  }
  const int       M             = 1 << bits;
  const uint32_t  wordsPerPlane = uint32_t( ( size_t( M ) * M ) >> 5 );
  DevBuf<uint32_t> d_bits;
  TMC2_TRY( d_bits.alloc( size_t( wordsPerPlane ) * 3 + 4 ) );
  uint32_t*   d_counts = d_bits.p + size_t( wordsPerPlane ) * 3;
  hipStream_t s        = f->ctx->stream;
  const int   sid      = f->ctx->stageBegin( "weight_normal" );
  TMC2_HIP( hipMemsetAsync( d_bits.p, 0, d_bits.bytes(), s ) );
  hipLaunchKernelGGL( footprintKernel, dim3( uint32_t( ( f->n + 255 ) / 256 ) ), dim3( 256 ), 0, s, f->d_pts.p,
                      uint32_t( f->n ), M, d_bits.p );
  hipLaunchKernelGGL( popcountKernel, dim3( 128, 3 ), dim3( 256 ), 0, s, d_bits.p, wordsPerPlane, d_counts );
  f->ctx->stageEnd( sid );
  uint32_t counts[3];
  TMC2_HIP( hipMemcpyAsync( counts, d_counts, sizeof( counts ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  // rank ascending by count, stable (the reference's std::sort on 3 elements is an insertion sort)
  int order[3] = {0, 1, 2};
  std::stable_sort( order, order + 3, [&]( int a, int b ) { return counts[a] < counts[b]; } );
  const double big = double( counts[order[2]] );
  const double r0 = double( counts[order[0]] ) / big, r1 = double( counts[order[1]] ) / big;
  if ( r0 >= minWeightEPP ) {
    w[order[0]] = r0;
    w[order[1]] = r1;
    w[order[2]] = 1.0;
  } else {
    w[order[0]] = minWeightEPP;
    w[order[2]] = 1.0;
    w[order[1]] = minWeightEPP + ( r1 - r0 ) / ( 1.0 - r0 ) * ( 1 - minWeightEPP );
  }
  return TMC2_OK;
}

}  // namespace tmc2
