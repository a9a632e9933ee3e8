// scan.hip -- exclusive prefix sum of uint32 on gfx950 (two-level: 2048-element tiles, then one block
// walks the tile totals).  Used for stream compaction / CSR offsets throughout the path.
#include "internal.h"

namespace tmc2 {
namespace {
constexpr int kTile = 2048;  // 256 threads x 8

__device__ __forceinline__ uint32_t waveInclusive( uint32_t v, int lane ) {
#pragma unroll
  for ( int off = 1; off < 64; off <<= 1 ) {
    const uint32_t t = __shfl_up( v, off, 64 );
    if ( lane >= off ) v += t;
  }
  return v;
}

// per tile: local exclusive scan written to out, tile total to sums[tile]
__global__ __launch_bounds__( 256 ) void scanTiles( const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     uint32_t* __restrict__ sums, uint32_t n ) {
  __shared__ uint32_t waveSum[4];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t      base = blockIdx.x * kTile + threadIdx.x * 8;
  uint32_t            v[8], run = 0;
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    v[k] = ( base + k < n ) ? in[base + k] : 0u;
    run += v[k];
  }
  const uint32_t inc = waveInclusive( run, lane );
  if ( lane == 63 ) waveSum[wave] = inc;
  __syncthreads();
  uint32_t offset = inc - run;
  for ( int w = 0; w < wave; ++w ) offset += waveSum[w];
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    if ( base + k < n ) out[base + k] = offset;
    offset += v[k];
  }
  if ( threadIdx.x == 255 ) sums[blockIdx.x] = offset;
}

// one block: exclusive scan of the tile totals in place, grand total to *total
__global__ __launch_bounds__( 256 ) void scanSums( uint32_t* __restrict__ sums, uint32_t tiles,
                                                    uint32_t* __restrict__ total ) {
  __shared__ uint32_t waveSum[4];
  __shared__ uint32_t carry;
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ( threadIdx.x == 0 ) carry = 0;
  __syncthreads();
  for ( uint32_t base = 0; base < tiles; base += 256 ) {
    const uint32_t i   = base + threadIdx.x;
    const uint32_t v   = i < tiles ? sums[i] : 0u;
    const uint32_t inc = waveInclusive( v, lane );
    if ( lane == 63 ) waveSum[wave] = inc;
    __syncthreads();
    uint32_t offset = carry + inc - v;
    for ( int w = 0; w < wave; ++w ) offset += waveSum[w];
    if ( i < tiles ) sums[i] = offset;
    __syncthreads();
    if ( threadIdx.x == 255 ) carry = offset + v;
    __syncthreads();
  }
  if ( threadIdx.x == 0 && total ) *total = carry;
}

__global__ __launch_bounds__( 256 ) void addOffsets( uint32_t* __restrict__ out, const uint32_t* __restrict__ sums,
                                                      uint32_t n ) {
  const uint32_t base = blockIdx.x * kTile + threadIdx.x * 8;
  const uint32_t off  = sums[blockIdx.x];
#pragma unroll
  for ( int k = 0; k < 8; ++k )
    if ( base + k < n ) out[base + k] += off;
}
}  // namespace

int exclusiveScanU32( tmc2_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, uint32_t* d_total ) {
  if ( n == 0 ) {
    if ( d_total ) TMC2_HIP( hipMemsetAsync( d_total, 0, 4, ctx->stream ) );
    return TMC2_OK;
  }
  const uint32_t tiles = uint32_t( ( n + kTile - 1 ) / kTile );
  TMC2_TRY( ctx->scratchU32.alloc( tiles + 8 ) );
  hipLaunchKernelGGL( scanTiles, dim3( tiles ), dim3( 256 ), 0, ctx->stream, d_in, d_out, ctx->scratchU32.p,
                      uint32_t( n ) );
  hipLaunchKernelGGL( scanSums, dim3( 1 ), dim3( 256 ), 0, ctx->stream, ctx->scratchU32.p, tiles, d_total );
  hipLaunchKernelGGL( addOffsets, dim3( tiles ), dim3( 256 ), 0, ctx->stream, d_out, ctx->scratchU32.p, uint32_t( n ) );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2
