// metrics.hip -- D1 / D2 / colour distortion (S23) on gfx950.
//
// Replaces PCCMetrics::compute for one frame (reference: source/lib/PccLibMetrics/source/PCCMetrics.cpp:324-375),
// QualityMetrics::compute (:73-229) and operator+ (:289-322), with PCCPointSet3::removeDuplicate
// (PccLibCommon/source/PCCPointSet.cpp:169-220), copyNormals (:2282-2320) and scaleNormals (:2322-2380).
//
// Every neighbour query of the metric asks for "all points at the minimum distance" (k grows 5, 10, .. 30 until the k-th result
// is farther than the first; PCCMetrics.cpp:91-96, PCCPointSet.cpp:2340-2346, 2362-2368).  A search for k results returns the
// first k entries of the search for any K > k (the k-d tree visiting order does not depend on the bound: DESIGN.md section 2), so
// one exact search serves all of them: K = 16 first -- the group then has fewer than 16 members in practice -- and, where some
// query's 16 results are all equidistant, the whole metric again with K = 32, of which the first 30 count: exactly the
// reference's last attempt.
// Everything runs on the device: the lexicographic de-duplication (an own stable LSD radix sort of re-based (x, y, z) keys, then
// run heads, a prefix sum and one thread per distinct position that averages the colours of its run), the tree builds, the
// four query batches (exact nanoflann-order k-NN kernel), the per-recon-point ordered normal accumulation, the per-point
// distortion terms and the final sums.  D1 is a sum of integers (exact in fp64, order-free): a parallel 64-bit reduction.  D2
// and the colour errors are fp64 sums whose value depends on the order: lanes walk the terms in the reference's order (one
// dependent add per point each) while the rest of the workgroup streams the next chunk into LDS.  Only the 3 x 8 results cross
// PCIe on the way back.  The clouds come either from the host (tmc2_metrics_compute) or straight from a frame's resident
// arrays (tmc2_metrics_compute_frame: no upload at all).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>

#include "internal.h"
#include "ordered_sum.h"

namespace tmc2 {
namespace {

// a cloud as its owner keeps it: positions as int16 triples `xyzStride` int16 apart, colours as byte triples `rgbStride` bytes
// apart (host arrays: 3 / 3; a frame's Pt / rgb4 arrays: 4 / 4), both on the device
struct CloudView {
  const int16_t* xyz;
  const uint8_t* rgb;
  uint32_t       n;
  int            xyzStride, rgbStride;
};

// a cloud + its tree on the device
struct DevCloud {
  size_t           n = 0;
  KdTreeHost       tree;
  DevBuf<Pt>       pts, ptsTree;
  DevBuf<uint32_t> perm;
  DevBuf<KdNode>   nodes;
  DevBuf<uint8_t>  rgb4;
  DevBuf<double>   nrm;
  TreeDev          dev() const {
    TreeDev t;
    t.ptsTree = ptsTree.p, t.perm = perm.p, t.nodes = nodes.p;
    for ( int d = 0; d < 3; ++d ) t.lo[d] = tree.lo[d], t.hi[d] = tree.hi[d];
    t.depth = tree.depth, t.n = n;
    return t;
  }
  int buildTree( tmc2_ctx* ctx ) { return buildKdTreeDevice( ctx, pts.p, n, ptsTree, perm, nodes, tree.lo, tree.hi, tree.depth ); }
  // this cloud's tree as queried with the points of `q` (whose own tree gives their bounding box: the packed LDS-stack
  // traversal of the k-NN kernel needs every query coordinate within [-4096, 12287])
  TreeDev devFor( const DevCloud& q ) const {
    TreeDev t        = dev();
    t.queriesBounded = true;
    for ( int d = 0; d < 3; ++d ) t.queriesBounded = t.queriesBounded && q.tree.lo[d] >= -4096 && q.tree.hi[d] <= 12287;
    return t;
  }
};

// ---- PCCPointSet3::removeDuplicate on the device ---------------------------------------------------------------------------
// (x, y, z) order = ascending key with the coordinates re-based to the cloud's bounding box (x in the top bits); the sort is
// stable, so the first element of a run of equal keys is the duplicate with the smallest input index -- the one the reference
// keeps the position of.
struct KeyBase {
  int      lo[3];
  uint32_t bitsY, bitsZ, bits;  // widths of the y and z fields, total key width
};
// (round 6: a stride loop over at most kBoundsBlocks workgroups, folded per workgroup in LDS, six atomics per WORKGROUP.  Rounds 2-5
//  had one wavefront per 64 points report to the six words behind a look-before-you-atomic -- but the 8 192 wavefronts that are
//  resident when the kernel starts all look at the untouched box and all report: ~ 8 000 atomics per word, 11 ns each, 0.26 ms
//  for a reduction over a million points)
constexpr uint32_t kBoundsBlocks = 1024;
__global__ __launch_bounds__( 256 ) void boundsKernel( CloudView c, int* __restrict__ box /* [6] min, max */ ) {
  __shared__ int sMin[4][3], sMax[4][3];
  int            mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[3] = {int( 0x80000000 ), int( 0x80000000 ), int( 0x80000000 )};
  for ( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += gridDim.x * blockDim.x ) {
#pragma unroll
    for ( int d = 0; d < 3; ++d ) {
      const int v = c.xyz[size_t( i ) * c.xyzStride + d];
      mn[d] = min( mn[d], v ), mx[d] = max( mx[d], v );
    }
  }
#pragma unroll
  for ( int d = 0; d < 3; ++d ) {
#pragma unroll
    for ( int off = 32; off > 0; off >>= 1 ) mn[d] = min( mn[d], __shfl_xor( mn[d], off, 64 ) ), mx[d] = max( mx[d], __shfl_xor( mx[d], off, 64 ) );
  }
  if ( ( threadIdx.x & 63 ) == 0 ) {
#pragma unroll
    for ( int d = 0; d < 3; ++d ) sMin[threadIdx.x >> 6][d] = mn[d], sMax[threadIdx.x >> 6][d] = mx[d];
  }
  __syncthreads();
  if ( threadIdx.x < 3 ) {
    const int d = threadIdx.x;
    const int lo = min( min( sMin[0][d], sMin[1][d] ), min( sMin[2][d], sMin[3][d] ) ), hi = max( max( sMax[0][d], sMax[1][d] ), max( sMax[2][d], sMax[3][d] ) );
    if ( lo < loadStaleOk( &box[d] ) ) atomicMin( &box[d], lo );
    if ( hi > loadStaleOk( &box[3 + d] ) ) atomicMax( &box[3 + d], hi );
  }
}
__global__ __launch_bounds__( 256 ) void positionKeysKernel( CloudView c, KeyBase kb, uint64_t* __restrict__ key, uint32_t* __restrict__ index ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= c.n ) return;
  const int16_t* q = c.xyz + size_t( i ) * c.xyzStride;
  const uint64_t x = uint64_t( int( q[0] ) - kb.lo[0] ), y = uint64_t( int( q[1] ) - kb.lo[1] ), z = uint64_t( int( q[2] ) - kb.lo[2] );
  key[i]   = ( x << ( kb.bitsY + kb.bitsZ ) ) | ( y << kb.bitsZ ) | z;
  index[i] = i;
}

// Stable LSD radix sort, 8 bits per pass, tiles of kSortTile keys per workgroup: per-tile digit counts (radixCountKernel), one
// prefix sum over counts[digit][tile] (digit-major: the exclusive sum IS the digit's base plus the tiles before), then the
// scatter: a tile is ranked in sub-tiles of 256 keys -- within a wavefront the lanes holding the same digit find each other
// with eight ballots, the waves' counts are prefixed through LDS, running per-digit offsets carry over the sub-tiles.
constexpr int kSortTile = 2048;
__global__ __launch_bounds__( 256 ) void radixCountKernel( const uint64_t* __restrict__ key, uint32_t n, int shift, uint32_t tiles,
                                                            uint32_t* __restrict__ counts ) {
  __shared__ uint32_t bins[256];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for ( uint32_t i = base + threadIdx.x; i < min( n, base + uint32_t( kSortTile ) ); i += 256 ) atomicAdd( &bins[( key[i] >> shift ) & 0xFF], 1u );
  __syncthreads();
  counts[size_t( threadIdx.x ) * tiles + blockIdx.x] = bins[threadIdx.x];
}
__global__ __launch_bounds__( 256 ) void radixScatterKernel( const uint64_t* __restrict__ keyIn, const uint32_t* __restrict__ idxIn,
                                                              uint32_t n, int shift, uint32_t tiles,
                                                              const uint32_t* __restrict__ bases, uint64_t* __restrict__ keyOut,
                                                              uint32_t* __restrict__ idxOut ) {
  __shared__ uint32_t running[256], waveCount[4][256];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  running[threadIdx.x] = bases[size_t( threadIdx.x ) * tiles + blockIdx.x];
  const uint32_t base = blockIdx.x * kSortTile, end = min( n, base + uint32_t( kSortTile ) );
  for ( uint32_t sub = base; sub < end; sub += 256 ) {
#pragma unroll
    for ( int w = 0; w < 4; ++w ) waveCount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i     = sub + threadIdx.x;
    const bool     valid = i < end;
    const uint64_t k     = valid ? keyIn[i] : 0;
    const uint32_t d     = uint32_t( k >> shift ) & 0xFF;
    unsigned long long peers = __ballot( valid );
#pragma unroll
    for ( int bit = 0; bit < 8; ++bit ) {
      const unsigned long long m = __ballot( ( d >> bit ) & 1u );
      peers &= ( ( d >> bit ) & 1u ) ? m : ~m;
    }
    const uint32_t rankInWave = uint32_t( __popcll( peers & ( ( 1ull << lane ) - 1ull ) ) );
    if ( valid && rankInWave == 0 ) waveCount[wave][d] = uint32_t( __popcll( peers ) );
    __syncthreads();
    if ( valid ) {
      uint32_t at = running[d] + rankInWave;
      for ( int w = 0; w < wave; ++w ) at += waveCount[w][d];
      keyOut[at] = k;
      idxOut[at] = idxIn[i];
    }
    __syncthreads();
    running[threadIdx.x] += waveCount[0][threadIdx.x] + waveCount[1][threadIdx.x] + waveCount[2][threadIdx.x] + waveCount[3][threadIdx.x];
    __syncthreads();
  }
}
// keys / payload of `a` sorted; the result is in (keyA, idxA) or (keyB, idxB): returns which through *inA
int radixSortPairs( tmc2_ctx* ctx, uint64_t* keyA, uint32_t* idxA, uint64_t* keyB, uint32_t* idxB, uint32_t n, uint32_t bits, bool* inA ) {
  hipStream_t      s     = ctx->stream;
  const uint32_t   tiles = ( n + kSortTile - 1 ) / kSortTile;
  DevBuf<uint32_t> d_counts;
  TMC2_TRY( d_counts.alloc( size_t( 256 ) * tiles ) );
  bool fromA = true;
  for ( uint32_t shift = 0; shift < bits; shift += 8 ) {
    uint64_t* kin  = fromA ? keyA : keyB;
    uint32_t* iin  = fromA ? idxA : idxB;
    uint64_t* kout = fromA ? keyB : keyA;
    uint32_t* iout = fromA ? idxB : idxA;
    hipLaunchKernelGGL( radixCountKernel, dim3( tiles ), dim3( 256 ), 0, s, kin, n, int( shift ), tiles, d_counts.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_counts.p, d_counts.p, size_t( 256 ) * tiles, nullptr ) );
    hipLaunchKernelGGL( radixScatterKernel, dim3( tiles ), dim3( 256 ), 0, s, kin, iin, n, int( shift ), tiles, d_counts.p, kout, iout );
    fromA = !fromA;
  }
  TMC2_HIP( hipGetLastError() );
  *inA = fromA;
  return TMC2_OK;
}

__global__ __launch_bounds__( 256 ) void runHeadKernel( const uint64_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ head ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) head[i] = ( i == 0 || key[i] != key[i - 1] ) ? 1u : 0u;
}
// one thread per run: position of its first element, colour = integer mean over the run (removeDuplicate :188-206)
__global__ __launch_bounds__( 256 ) void emitDistinctKernel( const uint64_t* __restrict__ key, const uint32_t* __restrict__ index,
                                                              const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank,
                                                              CloudView c, Pt* __restrict__ pts, uint8_t* __restrict__ rgb4,
                                                              uint32_t* __restrict__ first ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= c.n || !head[i] ) return;
  const uint64_t k = key[i];
  uint32_t       r = 0, g = 0, b = 0, cnt = 0;
  for ( uint32_t j = i; j < c.n && key[j] == k; ++j ) {
    const uint8_t* col = c.rgb + size_t( index[j] ) * c.rgbStride;
    r += col[0], g += col[1], b += col[2];
    ++cnt;
  }
  const uint32_t u = rank[i];
  const int16_t* q = c.xyz + size_t( index[i] ) * c.xyzStride;
  pts[u]           = Pt{q[0], q[1], q[2], 0};
  reinterpret_cast<uchar4*>( rgb4 )[u] = make_uchar4( (unsigned char)( r / cnt ), (unsigned char)( g / cnt ), (unsigned char)( b / cnt ), 0 );
  first[u]                             = index[i];
}
__global__ __launch_bounds__( 256 ) void rawPointsKernel( CloudView c, Pt* __restrict__ pts ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= c.n ) return;
  const int16_t* q = c.xyz + size_t( i ) * c.xyzStride;
  pts[i]           = Pt{q[0], q[1], q[2], 0};
}
__global__ __launch_bounds__( 256 ) void gatherPointsKernel( const Pt* __restrict__ pts, const uint32_t* __restrict__ which, uint32_t n,
                                                              Pt* __restrict__ out ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) out[i] = pts[which[i]];
}
__global__ __launch_bounds__( 256 ) void gatherNormalsKernel( const double* __restrict__ nrm, const uint32_t* __restrict__ first,
                                                               uint32_t n, double* __restrict__ out ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const size_t o = 3 * size_t( first[i] );
  out[3 * size_t( i )] = nrm[o], out[3 * size_t( i ) + 1] = nrm[o + 1], out[3 * size_t( i ) + 2] = nrm[o + 2];
}

// the cloud as its owner keeps it -> the distinct positions in (x, y, z) order with averaged colours; first[u] = input index of
// the duplicate whose position / normal the reference keeps
int removeDuplicatesDevice( tmc2_ctx* ctx, const CloudView& c, DevCloud& out, DevBuf<uint32_t>& d_first ) {
  hipStream_t      s = ctx->stream;
  const uint32_t   n = c.n;
  DevBuf<uint64_t> d_keyA, d_keyB;
  DevBuf<uint32_t> d_idxA, d_idxB, d_head, d_rank, d_small;
  TMC2_TRY( d_keyA.alloc( n ) );
  TMC2_TRY( d_keyB.alloc( n ) );
  TMC2_TRY( d_idxA.alloc( n ) );
  TMC2_TRY( d_idxB.alloc( n ) );
  TMC2_TRY( d_head.alloc( n ) );
  TMC2_TRY( d_rank.alloc( n ) );
  TMC2_TRY( d_small.alloc( 8 ) );  // [0 .. 5] bounding box, [6] distinct positions
  const dim3 blk( 256 ), grd( ( n + 255 ) / 256 );
  int        box[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, int( 0x80000000 ), int( 0x80000000 ), int( 0x80000000 )};
  TMC2_HIP( hipMemcpyAsync( d_small.p, box, sizeof( box ), hipMemcpyHostToDevice, s ) );
  hipLaunchKernelGGL( boundsKernel, dim3( std::min( grd.x, kBoundsBlocks ) ), blk, 0, s, c, reinterpret_cast<int*>( d_small.p ) );
  TMC2_HIP( hipMemcpyAsync( box, d_small.p, sizeof( box ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  KeyBase  kb{};
  uint32_t width[3];
  for ( int d = 0; d < 3; ++d ) {
    kb.lo[d] = box[d];
    width[d] = 1;
    while ( ( uint64_t( 1 ) << width[d] ) <= uint64_t( box[3 + d] - box[d] ) ) ++width[d];
  }
  kb.bitsY = width[1], kb.bitsZ = width[2], kb.bits = width[0] + width[1] + width[2];
  hipLaunchKernelGGL( positionKeysKernel, grd, blk, 0, s, c, kb, d_keyA.p, d_idxA.p );
  bool inA = true;
  TMC2_TRY( radixSortPairs( ctx, d_keyA.p, d_idxA.p, d_keyB.p, d_idxB.p, n, kb.bits, &inA ) );
  const uint64_t* key = inA ? d_keyA.p : d_keyB.p;
  const uint32_t* idx = inA ? d_idxA.p : d_idxB.p;
  hipLaunchKernelGGL( runHeadKernel, grd, blk, 0, s, key, n, d_head.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_head.p, d_rank.p, n, d_small.p + 6 ) );
  uint32_t distinct = 0;
  TMC2_HIP( hipMemcpyAsync( &distinct, d_small.p + 6, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  out.n = distinct;
  TMC2_TRY( out.pts.alloc( distinct ) );
  TMC2_TRY( out.rgb4.alloc( 4 * size_t( distinct ) ) );
  TMC2_TRY( d_first.alloc( distinct ) );
  hipLaunchKernelGGL( emitDistinctKernel, grd, blk, 0, s, key, idx, d_head.p, d_rank.p, c, out.pts.p, out.rgb4.p, d_first.p );
  TMC2_HIP( hipGetLastError() );
  TMC2_HIP( hipStreamSynchronize( s ) );  // (the temporaries go back to the pool)
  return TMC2_OK;
}

// size of the minimum-distance group of one K-NN row (leading entries equal to the first distance), at most cap: 16 in the
// first attempt (a full group raises the flag that sends the whole metric to the second one), 30 of K = 32 in the second
__device__ __forceinline__ int groupSize( const uint32_t* dist, int cap ) {
  int g = 1;
  while ( g < cap && dist[g] == dist[0] ) ++g;
  return g;
}
constexpr int kMaxGroup = 30;  // num_results_max of the reference
__host__ __device__ __forceinline__ int groupCap( int K ) { return K == 16 ? 16 : kMaxGroup; }

// scaleNormals, pass 1: every source point votes for its nearest reconstructed points
__global__ __launch_bounds__( 256 ) void votesCountKernel( const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist, int K,
                                                            uint32_t n, uint32_t* __restrict__ count, uint32_t* __restrict__ error ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const int g = groupSize( dist + size_t( i ) * K, groupCap( K ) );
  if ( K == 16 && g == K ) *error = 1;
  for ( int j = 0; j < g; ++j ) atomicAdd( &count[idx[size_t( i ) * K + j]], 1u );
}
__global__ __launch_bounds__( 256 ) void votesFillKernel( const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist, int K,
                                                           uint32_t n, const uint32_t* __restrict__ offset,
                                                           uint32_t* __restrict__ cursor, uint32_t* __restrict__ voters ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const int g = groupSize( dist + size_t( i ) * K, groupCap( K ) );
  for ( int j = 0; j < g; ++j ) {
    const uint32_t r             = idx[size_t( i ) * K + j];
    voters[offset[r] + atomicAdd( &cursor[r], 1u )] = i;
  }
}
// pass 2: per reconstructed point, add the voters' normals in increasing source index, divide by the count
__global__ __launch_bounds__( 256 ) void votesReduceKernel( const uint32_t* __restrict__ count, const uint32_t* __restrict__ offset,
                                                             const uint32_t* __restrict__ voters,
                                                             const double* __restrict__ srcNormals, uint32_t m,
                                                             double* __restrict__ recNormals ) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if ( r >= m ) return;
  const uint32_t  c = count[r];
  const uint32_t* v = voters + offset[r];
  double          x = 0.0, y = 0.0, z = 0.0;
  uint32_t        last = 0;
  for ( uint32_t k = 0; k < c; ++k ) {
    uint32_t best = 0xFFFFFFFFu;
    for ( uint32_t j = 0; j < c; ++j )
      if ( ( k == 0 || v[j] > last ) && v[j] < best ) best = v[j];
    last = best;
    x += srcNormals[3 * size_t( best )];
    y += srcNormals[3 * size_t( best ) + 1];
    z += srcNormals[3 * size_t( best ) + 2];
  }
  if ( c ) {
    x = __ddiv_rn( x, double( c ) );
    y = __ddiv_rn( y, double( c ) );
    z = __ddiv_rn( z, double( c ) );
  }
  recNormals[3 * size_t( r )] = x, recNormals[3 * size_t( r ) + 1] = y, recNormals[3 * size_t( r ) + 2] = z;
}
// reconstructed points nobody voted for: mean normal of their own nearest source points, in RESULT order
__global__ __launch_bounds__( 256 ) void orphanNormalsKernel( const uint32_t* __restrict__ orphan, uint32_t nOrphan,
                                                               const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist, int K,
                                                               const double* __restrict__ srcNormals,
                                                               double* __restrict__ recNormals, uint32_t* __restrict__ error ) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if ( o >= nOrphan ) return;
  const int g = groupSize( dist + size_t( o ) * K, groupCap( K ) );
  if ( K == 16 && g == K ) *error = 1;
  double x = 0.0, y = 0.0, z = 0.0;
  for ( int j = 0; j < g; ++j ) {
    const size_t s = idx[size_t( o ) * K + j];
    x += srcNormals[3 * s], y += srcNormals[3 * s + 1], z += srcNormals[3 * s + 2];
  }
  const size_t r = orphan[o];
  recNormals[3 * r] = __ddiv_rn( x, double( g ) ), recNormals[3 * r + 1] = __ddiv_rn( y, double( g ) ),
                 recNormals[3 * r + 2] = __ddiv_rn( z, double( g ) );
}

__device__ __forceinline__ void yuv709( const uchar4 c, float& y, float& u, float& v ) {
  y = float( __ddiv_rn( 0.2126 * double( c.x ) + 0.7152 * double( c.y ) + 0.0722 * double( c.z ), 255.0 ) );
  u = float( __ddiv_rn( -0.1146 * double( c.x ) - 0.3854 * double( c.y ) + 0.5000 * double( c.z ), 255.0 ) + 0.5000 );
  v = float( __ddiv_rn( 0.5000 * double( c.x ) - 0.4542 * double( c.y ) - 0.0458 * double( c.z ), 255.0 ) + 0.5000 );
}

// per point of A: D1 term (min squared distance), D2 term, three colour terms against its nearest group in B
__global__ __launch_bounds__( 256 ) void distortionTermsKernel( const Pt* __restrict__ ptsA, const uint8_t* __restrict__ rgbA,
                                                                 const Pt* __restrict__ ptsB, const uint8_t* __restrict__ rgbB,
                                                                 const double* __restrict__ nrmB,
                                                                 const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist,
                                                                 int K, uint32_t nA, double* __restrict__ terms /* [nA][5] */,
                                                                 uint32_t* __restrict__ error ) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if ( a >= nA ) return;
  const uint32_t* d = dist + size_t( a ) * K;
  const int       g = groupSize( d, groupCap( K ) );
  if ( K == 16 && g == K ) *error = 1;
  uint32_t same[kMaxGroup];
  for ( int j = 0; j < g; ++j ) same[j] = idx[size_t( a ) * K + j];
  for ( int j = 1; j < g; ++j ) {  // ascending index
    const uint32_t v = same[j];
    int            k = j - 1;
    while ( k >= 0 && same[k] > v ) {
      same[k + 1] = same[k];
      --k;
    }
    same[k + 1] = v;
  }
  const Pt pa  = ptsA[a];
  double   c2p = 0.0;
  unsigned r = 0, gg = 0, b = 0;
  for ( int j = 0; j < g; ++j ) {
    const uint32_t ib = same[j];
    if ( nrmB ) {
      const Pt     pb = ptsB[ib];
      const double e0 = double( int( pa.x ) - int( pb.x ) ), e1 = double( int( pa.y ) - int( pb.y ) ), e2 = double( int( pa.z ) - int( pb.z ) );
      const double dp = e0 * nrmB[3 * size_t( ib )] + e1 * nrmB[3 * size_t( ib ) + 1] + e2 * nrmB[3 * size_t( ib ) + 2];
      c2p += dp * dp;
    }
    const uchar4 cb = reinterpret_cast<const uchar4*>( rgbB )[ib];
    r += cb.x, gg += cb.y, b += cb.z;
  }
  if ( nrmB ) c2p = __ddiv_rn( c2p, double( g ) );
  const uchar4 avg = make_uchar4( (unsigned char)round( __ddiv_rn( double( r ), double( g ) ) ),
                                  (unsigned char)round( __ddiv_rn( double( gg ), double( g ) ) ),
                                  (unsigned char)round( __ddiv_rn( double( b ), double( g ) ) ), 0 );
  float ya, ua, va, yb, ub, vb;
  yuv709( reinterpret_cast<const uchar4*>( rgbA )[a], ya, ua, va );
  yuv709( avg, yb, ub, vb );
  const float dy = ya - yb, du = ua - ub, dv = va - vb;
  double*     t  = terms + size_t( a ) * 5;
  t[0]           = double( d[0] );
  t[1]           = c2p;
  t[2]           = double( float( dy * dy ) );
  t[3]           = double( float( du * du ) );
  t[4]           = double( float( dv * dv ) );
}

// ---- the sums over the points ---------------------------------------------------------------------------------------------
// terms[a][0] (squared distances: integers) are summed as 64-bit integers by everybody; terms[a][1 .. 4] (D2 and the three
// colour errors) in the reference's order, a = 0, 1, 2, ... (`sse += dist`, PCCMetrics.cpp:73-229): eight ordered fp64 sums
// for the two directions of the metric.  Rounds 2-5 added them term by term: one dependent add per term and chain, 4.4 ms for
// 0.95 M points and 15.8 ms for 3 M on ONE workgroup -- the longest kernel of every trace.  Round 6 (ordered_sum.h): while a sum
// stays in one binade an add is INTEGER arithmetic on its mantissa that needs nothing of the sum so far but its parity, so a
// block of kOsumBlock consecutive terms is reduced in parallel to a two-entry step, and the chain over the blocks is one 64-bit
// integer add (to the double's bit pattern) per block; the binade a block's step is computed for is guessed from an
// approximate prefix sum and CHECKED on the exact values when the blocks are chained -- the dozen blocks per sum that straddle
// a power of two (and the first one) are added term by term, as before.  Four launches:
//   osumBlockKernel<false>   per block and chain an approximate sum (any order), the D1 integers
//   osumGuessKernel          per chain a prefix sum over the blocks' approximate sums -> the binade each block is expected in
//   osumBlockKernel<true>    per block and chain the step of the block for that binade (in-order tree of compositions)
//   osumChainKernel          per chain (one wavefront each) the walk over the blocks; out[0 .. 4] = the five sums of direction
//                            A as doubles, out[5 .. 9] = of direction B
// Option METRICS_SUMS=sequential: no block is trusted (every one is added term by term) -- the cross-check of the form, and
// the test of its fallback.
constexpr int kOsumBlock = 1024;  // terms per block: 256 threads x 4 consecutive terms
struct OsumBufs {                 // chain = direction * 4 + column; entry [chain * nbMax + block]
  double*             approx;
  unsigned long long* d1;  // [direction * nbMax + block]
  int*                expo;
  osum::Step*         step;
  uint32_t            nbA, nbB, nbMax;
};

template <bool kSteps>
__global__ __launch_bounds__( 256 ) void osumBlockKernel( const double* __restrict__ termsA, uint32_t nA,
                                                           const double* __restrict__ termsB, uint32_t nB, OsumBufs o ) {
  const int      dir  = blockIdx.x >= o.nbA;
  const uint32_t k    = dir ? blockIdx.x - o.nbA : blockIdx.x;
  const double*  term = dir ? termsB : termsA;
  const uint32_t n    = dir ? nB : nA;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bits[4][4];  // [term of the thread][column]
  unsigned long long d1 = 0;
#pragma unroll
  for ( int j = 0; j < 4; ++j ) {
    const uint32_t a = k * uint32_t( kOsumBlock ) + 4u * threadIdx.x + uint32_t( j );
    if ( a < n ) {
      const double* t = term + 5 * size_t( a );
      if ( !kSteps ) d1 += (unsigned long long)t[0];
#pragma unroll
      for ( int c = 0; c < 4; ++c ) bits[j][c] = (unsigned long long)__double_as_longlong( t[1 + c] );
    } else {
#pragma unroll
      for ( int c = 0; c < 4; ++c ) bits[j][c] = 0;
    }
  }
  if ( !kSteps ) {
    __shared__ double             sSum[4][4];
    __shared__ unsigned long long sD1[4];
    double v[4];
#pragma unroll
    for ( int c = 0; c < 4; ++c ) {
      bool bad = false;
      v[c]     = 0.0;
#pragma unroll
      for ( int j = 0; j < 4; ++j ) {
        const unsigned long long b = bits[j][c];
        bad |= ( ( b >> 63 ) && ( b << 1 ) ) || ( ( b >> 52 ) & 0x7FF ) == 0x7FF;  // negative or not finite: nothing the form covers
        v[c] += __longlong_as_double( (long long)b );
      }
      if ( bad ) v[c] = __longlong_as_double( 0x7FF8000000000000ll );  // (a NaN: this block and the guesses behind it fall back)
#pragma unroll
      for ( int off = 32; off > 0; off >>= 1 ) v[c] += __shfl_xor( v[c], off, 64 );
    }
#pragma unroll
    for ( int off = 32; off > 0; off >>= 1 ) d1 += __shfl_xor( d1, off, 64 );
    if ( lane == 0 ) {
#pragma unroll
      for ( int c = 0; c < 4; ++c ) sSum[wave][c] = v[c];
      sD1[wave] = d1;
    }
    __syncthreads();
    if ( threadIdx.x < 4 )
      o.approx[( size_t( dir ) * 4 + threadIdx.x ) * o.nbMax + k] =
          ( sSum[0][threadIdx.x] + sSum[1][threadIdx.x] ) + ( sSum[2][threadIdx.x] + sSum[3][threadIdx.x] );
    if ( threadIdx.x == 4 ) o.d1[size_t( dir ) * o.nbMax + k] = sD1[0] + sD1[1] + sD1[2] + sD1[3];
  } else {
    __shared__ osum::Step sStep[4][4];
    __shared__ int        sBad[4];
    if ( threadIdx.x < 4 ) sBad[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for ( int c = 0; c < 4; ++c ) {
      const int E = o.expo[( size_t( dir ) * 4 + c ) * o.nbMax + k];  // (uniform over the workgroup)
      if ( E < 1 ) continue;
      osum::Step s{0ull, 0ull};
      bool       ok = true;
#pragma unroll
      for ( int j = 0; j < 4; ++j ) {
        osum::Step e;
        ok &= osum::stepOf( bits[j][c], E, e );
        s = osum::then( s, e );
      }
      if ( !ok ) sBad[c] = 1;
      // in order: lane L takes [L, L + off) then [L + off, L + 2 off)
#pragma unroll
      for ( int off = 1; off < 64; off <<= 1 ) {
        osum::Step p;
        p.d0 = __shfl_down( s.d0, off, 64 );
        p.d1 = __shfl_down( s.d1, off, 64 );
        if ( ( lane & ( 2 * off - 1 ) ) == 0 ) s = osum::then( s, p );
      }
      if ( lane == 0 ) sStep[wave][c] = s;
    }
    __syncthreads();
    if ( threadIdx.x < 4 ) {
      const int    c  = threadIdx.x;
      const size_t at = ( size_t( dir ) * 4 + c ) * o.nbMax + k;
      if ( o.expo[at] >= 1 ) {
        if ( sBad[c] )
          o.expo[at] = osum::kExpUnsafe;
        else
          o.step[at] = osum::then( osum::then( sStep[0][c], sStep[1][c] ), osum::then( sStep[2][c], sStep[3][c] ) );
      }
    }
  }
}

// one wavefront per chain: prefix sums of the blocks' approximate sums, 64 blocks at a time
__global__ __launch_bounds__( 64 ) void osumGuessKernel( OsumBufs o, int trustNothing ) {
  const uint32_t chain = blockIdx.x, nb = ( chain >> 2 ) ? o.nbB : o.nbA;
  const int      lane  = threadIdx.x;
  double         carry = 0.0;
  for ( uint32_t k0 = 0; k0 < nb; k0 += 64 ) {
    const uint32_t k = k0 + uint32_t( lane );
    const double   v = k < nb ? o.approx[size_t( chain ) * o.nbMax + k] : 0.0;
    double         incl = v;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const double t = __shfl_up( incl, off, 64 );
      if ( lane >= off ) incl += t;
    }
    double excl = __shfl_up( incl, 1, 64 );
    if ( lane == 0 ) excl = 0.0;
    const double lo = carry + excl, hi = carry + incl;  // (approximate on purpose: the chain checks the guess on the exact values)
    if ( k < nb ) o.expo[size_t( chain ) * o.nbMax + k] = trustNothing ? osum::kExpUnsafe : ( v == 0.0 ? osum::kExpIdentity : osum::guessExponent( lo, hi ) );
    carry += __shfl( incl, 63, 64 );
  }
}

// one wavefront per chain, every lane on the same walk (what is sequential costs a wavefront the same on one lane or on all)
__global__ __launch_bounds__( 64 ) void osumChainKernel( const double* __restrict__ termsA, uint32_t nA,
                                                          const double* __restrict__ termsB, uint32_t nB, OsumBufs o,
                                                          double* __restrict__ out, uint32_t* __restrict__ fallbacks ) {
  __shared__ double buf[kOsumBlock];
  const uint32_t    chain = blockIdx.x, dir = chain >> 2, col = chain & 3;
  const uint32_t    nb = dir ? o.nbB : o.nbA, n = dir ? nB : nA;
  const double*     term = ( dir ? termsB : termsA ) + 1 + col;
  const int         lane = threadIdx.x;
  unsigned long long sum = 0;  // the bit pattern of the sum so far
  uint32_t           fb  = 0;
  for ( uint32_t k0 = 0; k0 < nb; k0 += 64 ) {
    const uint32_t k   = k0 + uint32_t( lane );
    const size_t   at  = size_t( chain ) * o.nbMax + k;
    const int      myE = k < nb ? o.expo[at] : osum::kExpIdentity;
    osum::Step     mine{0ull, 0ull};
    if ( k < nb && myE >= 1 ) mine = o.step[at];
    // (fully unrolled: lane j's record through v_readlane with a constant lane -- scalar registers, no LDS crossbar; blocks past
    //  the end read as identities)
#pragma unroll
    for ( int j = 0; j < 64; ++j ) {
      const int  E = __builtin_amdgcn_readlane( myE, j );
      osum::Step st;
      st.d0 = ( (unsigned long long)(uint32_t)__builtin_amdgcn_readlane( int( mine.d0 >> 32 ), j ) << 32 ) | (uint32_t)__builtin_amdgcn_readlane( int( mine.d0 ), j );
      st.d1 = ( (unsigned long long)(uint32_t)__builtin_amdgcn_readlane( int( mine.d1 >> 32 ), j ) << 32 ) | (uint32_t)__builtin_amdgcn_readlane( int( mine.d1 ), j );
      if ( osum::apply( sum, st, E ) ) continue;
      // the block as the reference adds it
      ++fb;
      const uint32_t first = ( k0 + uint32_t( j ) ) * uint32_t( kOsumBlock ), terms = min( uint32_t( kOsumBlock ), n - first );
      __syncthreads();
      for ( uint32_t e = uint32_t( lane ); e < terms; e += 64 ) buf[e] = term[5 * size_t( first + e )];
      __syncthreads();
      double   acc = __longlong_as_double( (long long)sum );
      uint32_t e   = 0;
      for ( ; e + 8 <= terms; e += 8 ) {  // the loads ahead of the (dependent) adds
        const double2 a = *reinterpret_cast<const double2*>( buf + e ), b = *reinterpret_cast<const double2*>( buf + e + 2 );
        const double2 c = *reinterpret_cast<const double2*>( buf + e + 4 ), d = *reinterpret_cast<const double2*>( buf + e + 6 );
        acc += a.x, acc += a.y, acc += b.x, acc += b.y, acc += c.x, acc += c.y, acc += d.x, acc += d.y;
      }
      for ( ; e < terms; ++e ) acc += buf[e];
      sum = (unsigned long long)__double_as_longlong( acc );
    }
  }
  if ( lane == 0 ) {
    out[5 * dir + 1 + col] = __longlong_as_double( (long long)sum );
    if ( fallbacks ) fallbacks[chain] = fb;
  }
  if ( col == 0 ) {  // D1: integers
    unsigned long long d1 = 0;
    for ( uint32_t k = uint32_t( lane ); k < nb; k += 64 ) d1 += o.d1[size_t( dir ) * o.nbMax + k];
#pragma unroll
    for ( int off = 32; off > 0; off >>= 1 ) d1 += __shfl_xor( d1, off, 64 );
    if ( lane == 0 ) out[5 * dir] = double( d1 );
  }
}

// the eight ordered sums + the two integer ones of termsA[nA][5] / termsB[nB][5] -> out[10] (device)
int orderedSums( tmc2_ctx* ctx, const double* termsA, uint32_t nA, const double* termsB, uint32_t nB, double* out ) {
  hipStream_t s = ctx->stream;
  OsumBufs    o{};
  o.nbA = ( nA + kOsumBlock - 1 ) / kOsumBlock, o.nbB = ( nB + kOsumBlock - 1 ) / kOsumBlock, o.nbMax = std::max( 1u, std::max( o.nbA, o.nbB ) );
  DevBuf<double>             d_approx;
  DevBuf<unsigned long long> d_d1, d_step;
  DevBuf<uint32_t>           d_expo;
  TMC2_TRY( d_approx.alloc( 8 * size_t( o.nbMax ) ) );
  TMC2_TRY( d_d1.alloc( 2 * size_t( o.nbMax ) ) );
  TMC2_TRY( d_step.alloc( 16 * size_t( o.nbMax ) ) );
  TMC2_TRY( d_expo.alloc( 8 * size_t( o.nbMax ) ) );
  o.approx = d_approx.p, o.d1 = d_d1.p, o.expo = reinterpret_cast<int*>( d_expo.p ), o.step = reinterpret_cast<osum::Step*>( d_step.p );
  const char* form         = ctxOption( ctx, "METRICS_SUMS" );
  const int   trustNothing = form && form[0] == 's';
  if ( o.nbA + o.nbB ) hipLaunchKernelGGL( osumBlockKernel<false>, dim3( o.nbA + o.nbB ), dim3( 256 ), 0, s, termsA, nA, termsB, nB, o );
  hipLaunchKernelGGL( osumGuessKernel, dim3( 8 ), dim3( 64 ), 0, s, o, trustNothing );
  if ( ( o.nbA + o.nbB ) && !trustNothing ) hipLaunchKernelGGL( osumBlockKernel<true>, dim3( o.nbA + o.nbB ), dim3( 256 ), 0, s, termsA, nA, termsB, nB, o );
  DevBuf<uint32_t> d_fb;  // (test hook: the blocks added term by term, per chain)
  if ( ctxOption( ctx, "METRICS_SUMS_DEBUG" ) ) TMC2_TRY( d_fb.alloc( 8 ) );
  uint32_t* d_fallbacks = d_fb.p;
  hipLaunchKernelGGL( osumChainKernel, dim3( 8 ), dim3( 64 ), 0, s, termsA, nA, termsB, nB, o, out, d_fallbacks );
  if ( d_fallbacks ) {
    uint32_t fb[8];
    TMC2_HIP( hipMemcpyAsync( fb, d_fallbacks, sizeof( fb ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    fprintf( stderr, "ordered sums: %u + %u blocks, added term by term per chain: %u %u %u %u | %u %u %u %u\n", o.nbA, o.nbB, fb[0], fb[1], fb[2],
             fb[3], fb[4], fb[5], fb[6], fb[7] );
  }
  return TMC2_OK;
}

double psnr( double dist, double p, double factor ) { return 10 * std::log10( ( factor * p * p ) / dist ); }

// The metric's searches against a DE-DUPLICATED cloud (round 6: in two launches, knn.hip launchKnnSplit): most points of one cloud
// are points of the other, and in a tree without duplicate positions the group "all points at the minimum distance" of such a
// query is that one point.  Option KNN_SPLIT=0: one launch, as before.
int metricsKnn( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int K, uint32_t* d_idx, uint32_t* d_dist ) {
  const char* splitEnv = ctxOption( ctx, "KNN_SPLIT" );
  if ( splitEnv && splitEnv[0] == '0' ) return launchKnnTree( ctx, tree, d_queries, nq, K, d_idx, d_dist, "metrics_knn" );
  DevBuf<uint32_t> d_easy;
  TMC2_TRY( d_easy.alloc( std::max<uint64_t>( nq, 1 ) ) );
  return launchKnnSplit( ctx, tree, d_queries, nq, K, d_easy.p, d_idx, d_dist, "metrics_knn", true );
}

// per-point terms of one direction (A's points against their nearest neighbours in B)
int qualityTerms( tmc2_ctx* ctx, const DevCloud& A, const DevCloud& B, bool withNormals, int K, DevBuf<double>& d_terms,
                  uint32_t* d_error ) {
  hipStream_t      s  = ctx->stream;
  const uint32_t   nA = uint32_t( A.n );
  DevBuf<uint32_t> d_idx, d_dist;
  TMC2_TRY( d_idx.alloc( size_t( nA ) * K ) );
  TMC2_TRY( d_dist.alloc( size_t( nA ) * K ) );
  TMC2_TRY( d_terms.alloc( size_t( nA ) * 5 ) );
  TMC2_TRY( metricsKnn( ctx, B.devFor( A ), A.pts.p, nA, K, d_idx.p, d_dist.p ) );
  const int sid = ctx->stageBegin( "metrics_terms" );
  hipLaunchKernelGGL( distortionTermsKernel, dim3( ( nA + 255 ) / 256 ), dim3( 256 ), 0, s, A.pts.p, A.rgb4.p, B.pts.p, B.rgb4.p,
                      withNormals ? B.nrm.p : (const double*)nullptr, d_idx.p, d_dist.p, K, nA, d_terms.p, d_error );
  ctx->stageEnd( sid );
  return TMC2_OK;
}

void qualityFromSums( const double* sse, double num, bool withNormals, double resolution, double* out ) {
  out[0] = sse[0] / num;
  out[1] = psnr( out[0], resolution, 3 );
  out[2] = withNormals ? sse[1] / num : 0.0;
  out[3] = withNormals ? psnr( out[2], resolution, 3 ) : 0.0;
  for ( int i = 0; i < 3; ++i ) out[4 + i] = sse[2 + i] / num;
  out[7] = psnr( out[4], 1.0, 1.0 );
}

struct StageScope {  // closes a stage on every way out
  tmc2_ctx* ctx;
  int       id;
  ~StageScope() { ctx->stageEnd( id ); }
};

// PCCMetrics::compute for one frame from clouds resident on the device.  d_srcNormals: fp64[src.n][3] or null.
// *overflow = true: some query's K results were all equidistant (K = 16 only): the caller repeats with K = 32.
int metricsDevice( tmc2_ctx* ctx, const CloudView& src, const CloudView& rec, const double* d_srcNormals, double resolution, int K,
                   double* out, int64_t* counts, bool* overflow ) {
  hipStream_t s = ctx->stream;
  *overflow     = false;
  DevCloud         dS, dR, dN;  // de-duplicated source, de-duplicated reconstruction, normal cloud (original source order)
  DevBuf<uint32_t> d_firstS, d_firstR;
  const bool       withNormals = d_srcNormals != nullptr;
  const dim3       blk( 256 );
  {
    StageScope prep{ctx, ctx->stageBegin( "metrics_prepare" )};
    TMC2_TRY( removeDuplicatesDevice( ctx, src, dS, d_firstS ) );
    TMC2_TRY( removeDuplicatesDevice( ctx, rec, dR, d_firstR ) );
    if ( counts ) counts[0] = int64_t( dS.n ), counts[1] = int64_t( dR.n );
    if ( dS.n < size_t( K ) || dR.n < size_t( K ) ) {
      setError( "metrics_compute: clouds smaller than %d points unsupported", K );
      return TMC2_E_UNSUPPORTED;
    }
    if ( withNormals ) {
      if ( dS.n != src.n ) {
        setError( "metrics_compute: the source has duplicate positions; normals cannot be attached (the reference exits)" );
        return TMC2_E_INVALID;
      }
      // copyNormals: the de-duplicated source is the lexicographic sort of the input
      TMC2_TRY( dS.nrm.alloc( 3 * size_t( src.n ) ) );
      hipLaunchKernelGGL( gatherNormalsKernel, dim3( ( src.n + 255 ) / 256 ), blk, 0, s, d_srcNormals, d_firstS.p, src.n, dS.nrm.p );
    }
    TMC2_TRY( dS.buildTree( ctx ) );
    TMC2_TRY( dR.buildTree( ctx ) );
  }
  DevBuf<uint32_t> d_error;
  TMC2_TRY( d_error.alloc( 1 ) );
  TMC2_HIP( hipMemsetAsync( d_error.p, 0, 4, s ) );
  if ( withNormals ) {
    // scaleNormals
    const uint32_t mR = uint32_t( dR.n ), nS = src.n;
    dN.n = nS;  // the normal cloud: the source in its original order (no duplicates: checked above)
    TMC2_TRY( dN.pts.alloc( nS ) );
    hipLaunchKernelGGL( rawPointsKernel, dim3( ( nS + 255 ) / 256 ), blk, 0, s, src, dN.pts.p );
    DevBuf<uint32_t> d_idx, d_dist, d_count, d_offset, d_cursor, d_voters, d_total;
    TMC2_TRY( d_idx.alloc( size_t( nS ) * K ) );
    TMC2_TRY( d_dist.alloc( size_t( nS ) * K ) );
    TMC2_TRY( d_count.alloc( mR ) );
    TMC2_TRY( d_offset.alloc( mR ) );
    TMC2_TRY( d_cursor.alloc( mR ) );
    TMC2_TRY( d_total.alloc( 1 ) );
    TMC2_TRY( dR.nrm.alloc( 3 * size_t( mR ) ) );
    TMC2_TRY( metricsKnn( ctx, dR.devFor( dS ), dN.pts.p, nS, K, d_idx.p, d_dist.p ) );  // (dN = the points of dS)
    TMC2_TRY( fillRegions( ctx, {{d_count.p, size_t( mR ) * 4, 0}, {d_cursor.p, size_t( mR ) * 4, 0}} ) );
    hipLaunchKernelGGL( votesCountKernel, dim3( ( nS + 255 ) / 256 ), blk, 0, s, d_idx.p, d_dist.p, K, nS, d_count.p, d_error.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_offset.p, mR, d_total.p ) );
    uint32_t total = 0;
    TMC2_HIP( hipMemcpyAsync( &total, d_total.p, 4, hipMemcpyDeviceToHost, s ) );
    std::vector<uint32_t> h_count( mR );
    TMC2_HIP( hipMemcpyAsync( h_count.data(), d_count.p, size_t( mR ) * 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    TMC2_TRY( d_voters.alloc( std::max( total, 1u ) ) );
    hipLaunchKernelGGL( votesFillKernel, dim3( ( nS + 255 ) / 256 ), blk, 0, s, d_idx.p, d_dist.p, K, nS, d_offset.p, d_cursor.p,
                        d_voters.p );
    hipLaunchKernelGGL( votesReduceKernel, dim3( ( mR + 255 ) / 256 ), blk, 0, s, d_count.p, d_offset.p, d_voters.p, d_srcNormals, mR,
                        dR.nrm.p );
    std::vector<uint32_t> orphans;
    for ( uint32_t r = 0; r < mR; ++r )
      if ( h_count[r] == 0 ) orphans.push_back( r );
    if ( !orphans.empty() ) {
      // these query the tree of the NORMAL cloud (original source order)
      TMC2_TRY( dN.buildTree( ctx ) );
      const uint32_t   nO = uint32_t( orphans.size() );
      DevBuf<Pt>       d_q;
      DevBuf<uint32_t> d_orph, d_oi, d_od;
      TMC2_TRY( d_q.alloc( nO ) );
      TMC2_TRY( d_orph.alloc( nO ) );
      TMC2_TRY( d_oi.alloc( size_t( nO ) * K ) );
      TMC2_TRY( d_od.alloc( size_t( nO ) * K ) );
      TMC2_HIP( hipMemcpyAsync( d_orph.p, orphans.data(), size_t( nO ) * 4, hipMemcpyHostToDevice, s ) );
      hipLaunchKernelGGL( gatherPointsKernel, dim3( ( nO + 255 ) / 256 ), blk, 0, s, dR.pts.p, d_orph.p, nO, d_q.p );
      TMC2_TRY( launchKnnTree( ctx, dN.devFor( dR ), d_q.p, nO, K, d_oi.p, d_od.p, "metrics_knn" ) );
      hipLaunchKernelGGL( orphanNormalsKernel, dim3( ( nO + 255 ) / 256 ), blk, 0, s, d_orph.p, nO, d_oi.p, d_od.p, K, d_srcNormals,
                          dR.nrm.p, d_error.p );
      TMC2_HIP( hipStreamSynchronize( s ) );
    }
  }
  {
    DevBuf<double> d_termsS, d_termsR, d_sums;
    TMC2_TRY( d_sums.alloc( 16 ) );
    TMC2_TRY( qualityTerms( ctx, dS, dR, withNormals, K, d_termsS, d_error.p ) );
    TMC2_TRY( qualityTerms( ctx, dR, dS, withNormals, K, d_termsR, d_error.p ) );
    const int    sid = ctx->stageBegin( "metrics_sums" );
    TMC2_TRY( orderedSums( ctx, d_termsS.p, uint32_t( dS.n ), d_termsR.p, uint32_t( dR.n ), d_sums.p ) );
    ctx->stageEnd( sid );
    double   sse[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t err     = 0;
    TMC2_HIP( hipMemcpyAsync( sse, d_sums.p, sizeof( sse ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipMemcpyAsync( &err, d_error.p, 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    TMC2_HIP( hipGetLastError() );
    if ( err ) {
      *overflow = true;
      return TMC2_OK;
    }
    qualityFromSums( sse, double( dS.n ), withNormals, resolution, out );
    qualityFromSums( sse + 5, double( dR.n ), withNormals, resolution, out + 8 );
  }
  for ( int i = 0; i < 8; ++i ) {
    const bool isPsnr = ( i == 1 || i == 3 || i == 7 );
    out[16 + i]       = isPsnr ? std::min( out[i], out[8 + i] ) : std::max( out[i], out[8 + i] );
  }
  return TMC2_OK;
}

// K = 16, and once more with K = 32 (the reference's 30) where a minimum-distance group filled all 16 slots
int metricsBothWidths( tmc2_ctx* ctx, const CloudView& src, const CloudView& rec, const double* d_srcNormals, double resolution,
                       double* out, int64_t* counts ) {
  bool overflow = false;
  // (test hook TMC2_METRICS_K=32: the wide search straight away)
  const char* kEnv = ctxOption( ctx, "METRICS_K" );
  if ( !( kEnv && atoi( kEnv ) == 32 ) ) {
    TMC2_TRY( metricsDevice( ctx, src, rec, d_srcNormals, resolution, 16, out, counts, &overflow ) );
    if ( !overflow ) return TMC2_OK;
  }
  ctx->stageAddHostMs( "metrics_wide_search", 1.0 );  // (a count: how often the 30-neighbour search was needed)
  return metricsDevice( ctx, src, rec, d_srcNormals, resolution, 32, out, counts, &overflow );
}

}  // namespace
}  // namespace tmc2

namespace tmc2 {
namespace {
// the source cloud (and its normals) as the caller holds them on the host, to the device
struct UploadedSource {
  DevBuf<int16_t> xyz;
  DevBuf<uint8_t> rgb;
  DevBuf<double>  nrm;
  int put( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n, const double* srcNormals, CloudView& view ) {
    hipStream_t s = ctx->stream;
    StageScope  up{ctx, ctx->stageBegin( "metrics_upload" )};
    TMC2_TRY( xyz.alloc( 3 * size_t( n ) ) );
    TMC2_TRY( rgb.alloc( 3 * size_t( n ) ) );
    TMC2_HIP( hipMemcpyAsync( xyz.p, srcXyz, 6 * size_t( n ), hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipMemcpyAsync( rgb.p, srcRgb, 3 * size_t( n ), hipMemcpyHostToDevice, s ) );
    if ( srcNormals ) {
      TMC2_TRY( nrm.alloc( 3 * size_t( n ) ) );
      TMC2_HIP( hipMemcpyAsync( nrm.p, srcNormals, 3 * size_t( n ) * sizeof( double ), hipMemcpyHostToDevice, s ) );
    }
    view = CloudView{xyz.p, rgb.p, uint32_t( n ), 3, 3};
    return TMC2_OK;
  }
};
// the reconstruction a frame holds: which 0: of the attribute-image step, 1: the finished cloud of the post-reconstruction tail
int residentReconstruction( tmc2_frame* f, int which, CloudView& rec ) {
  if ( !f->haveReconstruction || f->reconCount == 0 || f->reconCount > 0x7FFFFFFFull ) {
    setError( "metrics: the frame holds no reconstruction" );
    return TMC2_E_STATE;
  }
  rec = CloudView{reinterpret_cast<const int16_t*>( f->d_recon.p ), f->d_reconRgb.p, uint32_t( f->reconCount ), 4, 4};
  if ( which == 0 ) {
    if ( !f->haveAttributeImages ) {
      setError( "metrics: the reconstruction has no colours yet (tmc2_encoder_generate_attribute_images)" );
      return TMC2_E_STATE;
    }
  } else if ( which == 1 ) {
    if ( !f->haveRgbPost ) {
      setError( "metrics: the post-reconstruction tail has not run (tmc2_codec_convert_yuv16_to_rgb8 is its last step)" );
      return TMC2_E_STATE;
    }
    if ( f->haveSmoothed ) rec.xyz = reinterpret_cast<const int16_t*>( f->d_reconSmoothed.p );
    rec.rgb = f->d_rgbPost.p;
  } else {
    setError( "metrics: which = %d (0: the reconstruction of the attribute-image step, 1: the finished cloud)", which );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}
}  // namespace
}  // namespace tmc2

extern "C" int tmc2_metrics_compute( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n,
                                     const int16_t* recXyz, const uint8_t* recRgb, uint64_t m, const double* srcNormals,
                                     double resolution, double* out, int64_t* counts ) {
  using namespace tmc2;
  if ( !ctx || !srcXyz || !srcRgb || !recXyz || !recRgb || !out || n == 0 || m == 0 || n > 0x7FFFFFFFull || m > 0x7FFFFFFFull ) {
    setError( "metrics_compute: invalid argument (null pointer, empty cloud or more than 2^31 - 1 points)" );
    return TMC2_E_INVALID;
  }
  ApiScope       scope( ctx );
  UploadedSource src, rec;  // (9 bytes per point each way; fp64 normals: 24 per source point)
  CloudView      vs{}, vr{};
  TMC2_TRY( src.put( ctx, srcXyz, srcRgb, n, srcNormals, vs ) );
  TMC2_TRY( rec.put( ctx, recXyz, recRgb, m, nullptr, vr ) );
  return metricsBothWidths( ctx, vs, vr, srcNormals ? src.nrm.p : nullptr, resolution, out, counts );
}

extern "C" int tmc2_metrics_ordered_sums( tmc2_ctx* ctx, const double* termsA, uint64_t nA, const double* termsB, uint64_t nB, double* out ) {
  using namespace tmc2;
  if ( !ctx || !out || ( nA && !termsA ) || ( nB && !termsB ) || nA > 0x7FFFFFFFull || nB > 0x7FFFFFFFull ) {
    setError( "metrics_ordered_sums: invalid argument (null pointer or more than 2^31 - 1 terms)" );
    return TMC2_E_INVALID;
  }
  ApiScope       scope( ctx );
  hipStream_t    s = ctx->stream;
  DevBuf<double> d_a, d_b, d_out;
  TMC2_TRY( d_a.alloc( std::max<size_t>( 5 * size_t( nA ), 1 ) ) );
  TMC2_TRY( d_b.alloc( std::max<size_t>( 5 * size_t( nB ), 1 ) ) );
  TMC2_TRY( d_out.alloc( 16 ) );
  if ( nA ) TMC2_HIP( hipMemcpyAsync( d_a.p, termsA, 5 * size_t( nA ) * sizeof( double ), hipMemcpyHostToDevice, s ) );
  if ( nB ) TMC2_HIP( hipMemcpyAsync( d_b.p, termsB, 5 * size_t( nB ) * sizeof( double ), hipMemcpyHostToDevice, s ) );
  const int sid = ctx->stageBegin( "metrics_sums" );
  TMC2_TRY( orderedSums( ctx, d_a.p, uint32_t( nA ), d_b.p, uint32_t( nB ), d_out.p ) );
  ctx->stageEnd( sid );
  TMC2_HIP( hipMemcpyAsync( out, d_out.p, 10 * sizeof( double ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

extern "C" int tmc2_metrics_compute_frame( tmc2_frame* f, int which, int useNormals, double resolution, double* out, int64_t* counts ) {
  using namespace tmc2;
  if ( !f || !out ) {
    setError( "metrics_compute_frame: invalid argument" );
    return TMC2_E_INVALID;
  }
  ApiScope scope( f->ctx );
  if ( f->n == 0 || f->d_rgb.count == 0 || f->n > 0x7FFFFFFFull ) {
    setError( "metrics_compute_frame: the frame has no source cloud with colours (a decoder-side frame: tmc2_metrics_compute_frame_source)" );
    return TMC2_E_STATE;
  }
  if ( useNormals && !f->haveNormals ) {
    setError( "metrics_compute_frame: no normals on the frame (tmc2_normals_compute / tmc2_frame_set_normals)" );
    return TMC2_E_STATE;
  }
  CloudView src{reinterpret_cast<const int16_t*>( f->d_pts.p ), f->d_rgb.p, uint32_t( f->n ), 4, 4}, rec{};
  TMC2_TRY( residentReconstruction( f, which, rec ) );
  return metricsBothWidths( f->ctx, src, rec, useNormals ? f->d_normals.p : nullptr, resolution, out, counts );
}

extern "C" int tmc2_metrics_compute_frame_source( tmc2_frame* f, int which, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n,
                                                  const double* srcNormals, double resolution, double* out, int64_t* counts ) {
  using namespace tmc2;
  if ( !f || !out || !srcXyz || !srcRgb || n == 0 || n > 0x7FFFFFFFull ) {
    setError( "metrics_compute_frame_source: invalid argument" );
    return TMC2_E_INVALID;
  }
  ApiScope       scope( f->ctx );
  CloudView      rec{}, vs{};
  UploadedSource src;
  TMC2_TRY( residentReconstruction( f, which, rec ) );
  TMC2_TRY( src.put( f->ctx, srcXyz, srcRgb, n, srcNormals, vs ) );
  return metricsBothWidths( f->ctx, vs, rec, srcNormals ? src.nrm.p : nullptr, resolution, out, counts );
}
