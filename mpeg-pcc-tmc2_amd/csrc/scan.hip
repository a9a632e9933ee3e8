// scan.hip -- exclusive prefix sum of uint32 on gfx950 (one launch: 2048-element tiles with decoupled look-back), used for
// stream compaction / CSR offsets throughout the path; and a several-buffers-at-once fill.
#include <algorithm>

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

__device__ __forceinline__ uint32_t waveSum( uint32_t v ) {
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) v += __shfl_xor( v, off, 64 );
  return v;
}

// One launch: every tile scans its 2048 elements, publishes its total, and finds the sum of everything before it by looking
// back over the totals / running sums its predecessors have published (decoupled look-back, one wavefront reads 64
// predecessors at a time).  A state word is {epoch : 30, flag : 2, value : 32}, written and read whole at agent scope (the
// tiles run on different XCDs); the epoch makes last call's words read as "not yet published", so nothing is cleared between
// calls.  Tiles are handed out by a ticket counter (never reset: the host passes where this call's tickets start), so a tile's
// predecessors are always already running.
constexpr unsigned long long kAggregate = 1ull << 32, kInclusive = 2ull << 32;
__global__ __launch_bounds__( 256 ) void scanLookBackKernel( const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              uint32_t n, unsigned long long* __restrict__ state,
                                                              uint32_t epoch, uint32_t ticketBase, uint32_t tiles,
                                                              uint32_t* __restrict__ total, ScanAnswer answer ) {
  __shared__ uint32_t waveTotal[4];
  __shared__ uint32_t sTile, sPrefix;
  if ( threadIdx.x == 0 ) sTile = atomicAdd( reinterpret_cast<uint32_t*>( state ), 1u ) - ticketBase;
  __syncthreads();
  const uint32_t tile = sTile;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t base = tile * kTile + threadIdx.x * 8;
  uint32_t       v[8], run = 0;
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    v[k] = ( base + k < n ) ? in[base + k] : 0u;
    run += v[k];
  }
  const uint32_t inc = waveInclusive( run, lane );
  if ( lane == 63 ) waveTotal[wave] = inc;
  __syncthreads();
  uint32_t offset = inc - run;
  for ( int w = 0; w < wave; ++w ) offset += waveTotal[w];
  if ( wave == 0 ) {
    const uint32_t           aggregate = waveTotal[0] + waveTotal[1] + waveTotal[2] + waveTotal[3];
    const unsigned long long tag       = static_cast<unsigned long long>( epoch ) << 34;
    unsigned long long*      st        = state + 1;
    if ( lane == 0 )
      __hip_atomic_store( &st[tile], tag | ( tile == 0 ? kInclusive : kAggregate ) | aggregate, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT );
    uint32_t prefix = 0;
    if ( tile > 0 ) {
      long long j = static_cast<long long>( tile ) - 1 - lane;
      for ( ;; ) {
        unsigned long long w = tag | kInclusive;  // before the first tile: a running sum of 0
        if ( j >= 0 ) {
          do { w = __hip_atomic_load( &st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); } while ( ( w >> 34 ) != epoch );
        }
        const unsigned long long running = __ballot( ( ( w >> 32 ) & 3u ) == 2u );
        const int                first   = running ? __ffsll( static_cast<long long>( running ) ) - 1 : 64;
        prefix += waveSum( lane <= first ? static_cast<uint32_t>( w ) : 0u );
        if ( running ) break;
        j -= 64;
      }
      if ( lane == 0 )
        __hip_atomic_store( &st[tile], tag | kInclusive | ( prefix + aggregate ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    }
    if ( lane == 0 ) {
      sPrefix = prefix;
      if ( tile + 1 == tiles ) {
        if ( total ) *total = prefix + aggregate;
        if ( answer.host ) {  // the host's copy, and the words that ride along (written by earlier launches of the stream)
          answer.host[0] = prefix + aggregate;
          for ( int k = 0; k < answer.carryWords; ++k ) answer.host[1 + k] = answer.carry[k];
        }
      }
    }
  }
  __syncthreads();
  offset += sPrefix;
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    if ( base + k < n ) out[base + k] = offset;
    offset += v[k];
  }
}

// ---- several regions set to a byte value each, one launch
constexpr int    kFillMax   = 12;
constexpr size_t kFillBlock = 16384;  // bytes per block: 256 threads x 4 x 16
struct FillArgs {
  unsigned long long base[kFillMax], bytes[kFillMax];
  uint32_t           firstBlock[kFillMax + 1], word[kFillMax];
  int                count;
};
__global__ __launch_bounds__( 256 ) void fillRegionsKernel( const FillArgs a ) {
  int r = 0;
  while ( r + 1 < a.count && blockIdx.x >= a.firstBlock[r + 1] ) ++r;
  uint8_t* const p     = reinterpret_cast<uint8_t*>( a.base[r] );
  const size_t   head  = reinterpret_cast<size_t>( p ) & 15, end = head + a.bytes[r];  // in bytes from the 16-byte line p lies in
  uint8_t* const line0 = p - head;
  const uint32_t w     = a.word[r];
  const size_t   first = size_t( blockIdx.x - a.firstBlock[r] ) * kFillBlock;
#pragma unroll
  for ( int k = 0; k < 4; ++k ) {
    const size_t lo = first + ( size_t( k ) * 256 + threadIdx.x ) * 16;
    if ( lo >= end ) break;
    if ( lo >= head && lo + 16 <= end ) {
      *reinterpret_cast<uint4*>( line0 + lo ) = make_uint4( w, w, w, w );
    } else {
      const size_t b = lo > head ? lo : head, e = lo + 16 < end ? lo + 16 : end;
      for ( size_t i = b; i < e; ++i ) line0[i] = uint8_t( w );
    }
  }
}
}  // namespace

int exclusiveScanU32( tmc2_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n, uint32_t* d_total, ScanAnswer answer ) {
  if ( n == 0 ) {
    if ( d_total ) TMC2_HIP( hipMemsetAsync( d_total, 0, 4, ctx->stream ) );
    if ( answer.host ) {
      setError( "exclusiveScanU32: an answer line needs at least one element" );
      return TMC2_E_INVALID;
    }
    return TMC2_OK;
  }
  if ( n > 0xFFFFFFFFull - kTile ) {
    setError( "exclusiveScanU32: %zu elements", n );
    return TMC2_E_INVALID;
  }
  const uint32_t tiles = uint32_t( ( n + kTile - 1 ) / kTile );
  ctx->scanEpoch       = ( ctx->scanEpoch + 1 ) & 0x3FFFFFFFu;
  if ( ctx->scanState.count < size_t( tiles ) + 1 || ctx->scanEpoch == 0 ) {  // (re)allocated or the epochs wrapped: start clean
    TMC2_TRY( ctx->scanState.alloc( std::max<size_t>( size_t( tiles ) + 1, 4096 ) ) );
    TMC2_HIP( hipMemsetAsync( ctx->scanState.p, 0, ctx->scanState.bytes(), ctx->stream ) );
    ctx->scanEpoch   = 1;
    ctx->scanTickets = 0;
  }
  hipLaunchKernelGGL( scanLookBackKernel, dim3( tiles ), dim3( 256 ), 0, ctx->stream, d_in, d_out, uint32_t( n ),
                      ctx->scanState.p, ctx->scanEpoch, ctx->scanTickets, tiles, d_total, answer );
  // (one scan state per context: every scan of a context is queued on ctx->stream, in order)
  const hipError_t launched = hipGetLastError();
  if ( launched != hipSuccess ) {
    // the launch was refused: the device ticket counter did not move.  Start the next scan from a clean state instead of
    // letting the host-side ticket base run ahead of it (tiles would index out of range, or wait for predecessors forever).
    ctx->scanState.release();
    setError( "exclusiveScanU32: %s", hipGetErrorString( launched ) );
    return TMC2_E_HIP;
  }
  ctx->scanTickets += tiles;
  return TMC2_OK;
}

int fillRegions( tmc2_ctx* ctx, std::initializer_list<FillRegion> regions ) {
  FillArgs a{};
  uint32_t blocks = 0;
  for ( const FillRegion& r : regions ) {
    if ( r.bytes == 0 ) continue;
    if ( a.count == kFillMax ) {
      setError( "fillRegions: more than %d regions", kFillMax );
      return TMC2_E_INVALID;
    }
    a.base[a.count]       = reinterpret_cast<unsigned long long>( r.p );
    a.bytes[a.count]      = r.bytes;
    a.word[a.count]       = r.value * 0x01010101u;
    a.firstBlock[a.count] = blocks;
    blocks += uint32_t( ( ( reinterpret_cast<size_t>( r.p ) & 15 ) + r.bytes + kFillBlock - 1 ) / kFillBlock );
    ++a.count;
  }
  if ( a.count == 0 ) return TMC2_OK;
  a.firstBlock[a.count] = blocks;
  hipLaunchKernelGGL( fillRegionsKernel, dim3( blocks ), dim3( 256 ), 0, ctx->stream, a );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2

