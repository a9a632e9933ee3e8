// metrics.hip -- D1 / D2 / colour distortion (S23) on gfx950.
//
// Replaces PCCMetrics::compute for one frame (reference: source/lib/PccLibMetrics/source/PCCMetrics.cpp:324-375),
// QualityMetrics::compute (:73-229) and operator+ (:289-322), with PCCPointSet3::removeDuplicate
// (PccLibCommon/source/PCCPointSet.cpp:169-220), copyNormals (:2282-2320) and scaleNormals (:2322-2380).
//
// Every neighbour query of the metric asks for "all points at the minimum distance" (k grows 5,10,..30 until the
// k-th result is farther than the first).  That set is canonical as long as it has fewer than k members, and its
// RESULT order (needed only where the reference sums fp64 normals in result order) is the k-d tree visiting order,
// independent of k -- so one exact k=16 search per query serves all of them; a group that fills all 16 slots is
// reported as unsupported instead of being approximated.
// Everything runs on the device: the lexicographic de-duplication (a stable radix sort of (x, y, z) keys -- hipCUB's device
// radix sort is the one library primitive used -- then run heads, a prefix sum and one thread per distinct position that
// averages the colours of its run), the tree builds, the four query batches (exact nanoflann-order k-NN kernel), the
// per-recon-point ordered normal accumulation, the per-point distortion terms and the final sums.  D1 is a sum of
// integers (exact in fp64, order-free): a parallel 64-bit reduction.  D2 and the colour errors are fp64 sums whose value
// depends on the order: four lanes walk the terms in the reference's order (one dependent add per point each) while the
// rest of the workgroup streams the next chunk into LDS.  Only the 3 x 8 results cross PCIe on the way back.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>

#include <hipcub/hipcub.hpp>

#include "internal.h"

namespace tmc2 {
namespace {

constexpr int K = 16;

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
// (x, y, z) order = ascending 48-bit key (coordinates biased to unsigned); the sort is stable, so the first element of a
// run of equal keys is the duplicate with the smallest input index -- the one the reference keeps the position of.
__global__ __launch_bounds__( 256 ) void positionKeysKernel( const int16_t* __restrict__ xyz, uint32_t n, uint64_t* __restrict__ key,
                                                              uint32_t* __restrict__ index ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint64_t x = uint16_t( int( xyz[3 * size_t( i )] ) + 32768 ), y = uint16_t( int( xyz[3 * size_t( i ) + 1] ) + 32768 ),
                 z = uint16_t( int( xyz[3 * size_t( i ) + 2] ) + 32768 );
  key[i]   = ( x << 32 ) | ( y << 16 ) | z;
  index[i] = i;
}
__global__ __launch_bounds__( 256 ) void runHeadKernel( const uint64_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ head ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) head[i] = ( i == 0 || key[i] != key[i - 1] ) ? 1u : 0u;
}
// one thread per run: position of its first element, colour = integer mean over the run (removeDuplicate :188-206)
__global__ __launch_bounds__( 256 ) void emitDistinctKernel( const uint64_t* __restrict__ key, const uint32_t* __restrict__ index,
                                                              const uint32_t* __restrict__ head, const uint32_t* __restrict__ rank,
                                                              const uint8_t* __restrict__ rgb, uint32_t n, Pt* __restrict__ pts,
                                                              uint8_t* __restrict__ rgb4, uint32_t* __restrict__ first ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n || !head[i] ) return;
  const uint64_t k = key[i];
  uint32_t       r = 0, g = 0, b = 0, c = 0;
  for ( uint32_t j = i; j < n && key[j] == k; ++j ) {
    const size_t o = 3 * size_t( index[j] );
    r += rgb[o], g += rgb[o + 1], b += rgb[o + 2];
    ++c;
  }
  const uint32_t u = rank[i];
  pts[u]           = Pt{int16_t( int( ( k >> 32 ) & 0xFFFF ) - 32768 ), int16_t( int( ( k >> 16 ) & 0xFFFF ) - 32768 ),
              int16_t( int( k & 0xFFFF ) - 32768 ), 0};
  reinterpret_cast<uchar4*>( rgb4 )[u] = make_uchar4( (unsigned char)( r / c ), (unsigned char)( g / c ), (unsigned char)( b / c ), 0 );
  first[u]                             = index[i];
}
__global__ __launch_bounds__( 256 ) void rawPointsKernel( const int16_t* __restrict__ xyz, uint32_t n, Pt* __restrict__ pts ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) pts[i] = Pt{xyz[3 * size_t( i )], xyz[3 * size_t( i ) + 1], xyz[3 * size_t( i ) + 2], 0};
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

// d_xyz / d_rgb: the cloud as the caller gave it (int16[n][3], uint8[n][3]) -> the distinct positions in (x, y, z) order with
// averaged colours; first[u] = input index of the duplicate whose position / normal the reference keeps
int removeDuplicatesDevice( tmc2_ctx* ctx, const int16_t* d_xyz, const uint8_t* d_rgb, uint32_t n, DevCloud& out,
                            DevBuf<uint32_t>& d_first ) {
  hipStream_t      s = ctx->stream;
  DevBuf<uint64_t> d_keyIn, d_keyOut;
  DevBuf<uint32_t> d_idxIn, d_idxOut, d_head, d_rank, d_total;
  DevBuf<uint8_t>  d_tmp;
  TMC2_TRY( d_keyIn.alloc( n ) );
  TMC2_TRY( d_keyOut.alloc( n ) );
  TMC2_TRY( d_idxIn.alloc( n ) );
  TMC2_TRY( d_idxOut.alloc( n ) );
  TMC2_TRY( d_head.alloc( n ) );
  TMC2_TRY( d_rank.alloc( n ) );
  TMC2_TRY( d_total.alloc( 1 ) );
  const dim3 blk( 256 ), grd( ( n + 255 ) / 256 );
  hipLaunchKernelGGL( positionKeysKernel, grd, blk, 0, s, d_xyz, n, d_keyIn.p, d_idxIn.p );
  size_t tmpBytes = 0;
  TMC2_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, tmpBytes, d_keyIn.p, d_keyOut.p, d_idxIn.p, d_idxOut.p, int( n ), 0, 48, s ) );
  TMC2_TRY( d_tmp.alloc( tmpBytes + 16 ) );
  TMC2_HIP( hipcub::DeviceRadixSort::SortPairs( d_tmp.p, tmpBytes, d_keyIn.p, d_keyOut.p, d_idxIn.p, d_idxOut.p, int( n ), 0, 48, s ) );
  hipLaunchKernelGGL( runHeadKernel, grd, blk, 0, s, d_keyOut.p, n, d_head.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_head.p, d_rank.p, n, d_total.p ) );
  uint32_t distinct = 0;
  TMC2_HIP( hipMemcpyAsync( &distinct, d_total.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  out.n = distinct;
  TMC2_TRY( out.pts.alloc( distinct ) );
  TMC2_TRY( out.rgb4.alloc( 4 * size_t( distinct ) ) );
  TMC2_TRY( d_first.alloc( distinct ) );
  hipLaunchKernelGGL( emitDistinctKernel, grd, blk, 0, s, d_keyOut.p, d_idxOut.p, d_head.p, d_rank.p, d_rgb, n, out.pts.p,
                      out.rgb4.p, d_first.p );
  TMC2_HIP( hipGetLastError() );
  TMC2_HIP( hipStreamSynchronize( s ) );  // (the temporaries go back to the pool)
  return TMC2_OK;
}

// size of the minimum-distance group of one 16-NN row (leading entries equal to the first distance)
__device__ __forceinline__ int groupSize( const uint32_t* dist ) {
  int g = 1;
  while ( g < K && dist[g] == dist[0] ) ++g;
  return g;
}

// scaleNormals, pass 1: every source point votes for its nearest reconstructed points
__global__ __launch_bounds__( 256 ) void votesCountKernel( const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist,
                                                            uint32_t n, uint32_t* __restrict__ count, uint32_t* __restrict__ error ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const int g = groupSize( dist + size_t( i ) * K );
  if ( g == K ) *error = 1;
  for ( int j = 0; j < g; ++j ) atomicAdd( &count[idx[size_t( i ) * K + j]], 1u );
}
__global__ __launch_bounds__( 256 ) void votesFillKernel( const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist,
                                                           uint32_t n, const uint32_t* __restrict__ offset,
                                                           uint32_t* __restrict__ cursor, uint32_t* __restrict__ voters ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const int g = groupSize( dist + size_t( i ) * K );
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
                                                               const uint32_t* __restrict__ idx, const uint32_t* __restrict__ dist,
                                                               const double* __restrict__ srcNormals,
                                                               double* __restrict__ recNormals, uint32_t* __restrict__ error ) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if ( o >= nOrphan ) return;
  const int g = groupSize( dist + size_t( o ) * K );
  if ( g == K ) *error = 1;
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
                                                                 uint32_t nA, double* __restrict__ terms /* [nA][5] */,
                                                                 uint32_t* __restrict__ error ) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if ( a >= nA ) return;
  const uint32_t* d = dist + size_t( a ) * K;
  const int       g = groupSize( d );
  if ( g == K ) *error = 1;
  uint32_t same[K];
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
// colour errors) in the reference's order, a = 0, 1, 2, ...  Both directions of the metric in one launch: eight ordered sums,
// one per lane 0 .. 7 of the first wave (a dependent fp64 add costs a wave the same whether one lane or eight take part),
// column-major in LDS so that a lane fetches two consecutive terms per load, chunk by chunk while the other waves fetch the
// next chunk.  out[0 .. 4] = the five sums of direction A as doubles, out[5 .. 9] = of direction B.
constexpr int kSumChunk = 1024, kSumStride = kSumChunk + 2;  // (columns 16 bytes apart in the LDS banks: the eight lanes read side by side)
__global__ __launch_bounds__( 1024 ) void orderedSumsKernel( const double* __restrict__ termsA, uint32_t nA,
                                                              const double* __restrict__ termsB, uint32_t nB,
                                                              double* __restrict__ out ) {
  extern __shared__ double buf[];  // [2 slots][8 columns][kSumStride]
  __shared__ unsigned long long d1Total[2];
  if ( threadIdx.x < 2 ) d1Total[threadIdx.x] = 0;
  const uint32_t     chunks = ( max( nA, nB ) + kSumChunk - 1 ) / kSumChunk;
  double             acc    = 0.0;  // (lanes 0 .. 7: one ordered sum each)
  unsigned long long d1A = 0, d1B = 0;
  auto               fetch = [&]( uint32_t c, int slot ) {
    const uint32_t a   = c * kSumChunk + threadIdx.x;
    double*        col = buf + size_t( slot ) * 8 * kSumStride + threadIdx.x;
    if ( a < nA ) {
      const double* t = termsA + 5 * size_t( a );
      d1A += (unsigned long long)t[0];
#pragma unroll
      for ( int k = 0; k < 4; ++k ) col[k * kSumStride] = t[1 + k];
    }
    if ( a < nB ) {
      const double* t = termsB + 5 * size_t( a );
      d1B += (unsigned long long)t[0];
#pragma unroll
      for ( int k = 0; k < 4; ++k ) col[( 4 + k ) * kSumStride] = t[1 + k];
    }
  };
  if ( chunks ) fetch( 0, 0 );
  __syncthreads();
  for ( uint32_t c = 0; c < chunks; ++c ) {
    const int slot = int( c & 1 );
    if ( threadIdx.x >= 64 ) {
      if ( c + 1 < chunks ) fetch( c + 1, slot ^ 1 );
    } else if ( threadIdx.x < 8 ) {
      const uint32_t n     = threadIdx.x < 4 ? nA : nB, first = c * kSumChunk;
      const uint32_t cnt   = first < n ? min( uint32_t( kSumChunk ), n - first ) : 0u;
      const double*  col   = buf + ( size_t( slot ) * 8 + threadIdx.x ) * kSumStride;
      const double2* pairs = reinterpret_cast<const double2*>( col );
      uint32_t       j     = 0;
      for ( ; j + 8 <= cnt; j += 8 ) {  // the loads ahead of the (dependent) adds
        const double2 a = pairs[j / 2], b = pairs[j / 2 + 1], d = pairs[j / 2 + 2], e = pairs[j / 2 + 3];
        acc += a.x, acc += a.y, acc += b.x, acc += b.y, acc += d.x, acc += d.y, acc += e.x, acc += e.y;
      }
      for ( ; j < cnt; ++j ) acc += col[j];
    }
    __syncthreads();
    if ( threadIdx.x < 64 && c + 1 < chunks ) fetch( c + 1, slot ^ 1 );  // (the first wave's share of the next chunk)
    __syncthreads();
  }
  atomicAdd( &d1Total[0], d1A );
  atomicAdd( &d1Total[1], d1B );
  __syncthreads();
  if ( threadIdx.x < 2 ) out[5 * threadIdx.x] = double( d1Total[threadIdx.x] );
  if ( threadIdx.x < 8 ) out[5 * ( threadIdx.x / 4 ) + 1 + ( threadIdx.x & 3 )] = acc;
}

double psnr( double dist, double p, double factor ) { return 10 * std::log10( ( factor * p * p ) / dist ); }

// per-point terms of one direction (A's points against their nearest neighbours in B)
int qualityTerms( tmc2_ctx* ctx, const DevCloud& A, const DevCloud& B, bool withNormals, DevBuf<double>& d_terms, uint32_t* d_error ) {
  hipStream_t      s  = ctx->stream;
  const uint32_t   nA = uint32_t( A.n );
  DevBuf<uint32_t> d_idx, d_dist;
  TMC2_TRY( d_idx.alloc( size_t( nA ) * K ) );
  TMC2_TRY( d_dist.alloc( size_t( nA ) * K ) );
  TMC2_TRY( d_terms.alloc( size_t( nA ) * 5 ) );
  TMC2_TRY( launchKnnTree( ctx, B.devFor( A ), A.pts.p, nA, K, d_idx.p, d_dist.p, "metrics_knn16" ) );
  const int sid = ctx->stageBegin( "metrics_terms" );
  hipLaunchKernelGGL( distortionTermsKernel, dim3( ( nA + 255 ) / 256 ), dim3( 256 ), 0, s, A.pts.p, A.rgb4.p, B.pts.p, B.rgb4.p,
                      withNormals ? B.nrm.p : (const double*)nullptr, d_idx.p, d_dist.p, nA, d_terms.p, d_error );
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

}  // namespace
}  // namespace tmc2

extern "C" int tmc2_metrics_compute( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n,
                                     const int16_t* recXyz, const uint8_t* recRgb, uint64_t m, const double* srcNormals,
                                     double resolution, double* out, int64_t* counts ) {
  using namespace tmc2;
  if ( !ctx || !srcXyz || !srcRgb || !recXyz || !recRgb || !out || n == 0 || m == 0 ) {
    setError( "metrics_compute: invalid argument" );
    return TMC2_E_INVALID;
  }
  ApiScope    scope( ctx );
  hipStream_t s  = ctx->stream;
  const int   sidPrep = ctx->stageBegin( "metrics_prepare" );
  // the clouds as given, to the device (9 bytes per point each way; fp64 normals: 24 per source point)
  DevBuf<int16_t> d_srcXyz, d_recXyz;
  DevBuf<uint8_t> d_srcRgb, d_recRgb;
  DevBuf<double>  d_srcNrm;
  TMC2_TRY( d_srcXyz.alloc( 3 * size_t( n ) ) );
  TMC2_TRY( d_srcRgb.alloc( 3 * size_t( n ) ) );
  TMC2_TRY( d_recXyz.alloc( 3 * size_t( m ) ) );
  TMC2_TRY( d_recRgb.alloc( 3 * size_t( m ) ) );
  TMC2_HIP( hipMemcpyAsync( d_srcXyz.p, srcXyz, 6 * size_t( n ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( d_srcRgb.p, srcRgb, 3 * size_t( n ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( d_recXyz.p, recXyz, 6 * size_t( m ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipMemcpyAsync( d_recRgb.p, recRgb, 3 * size_t( m ), hipMemcpyHostToDevice, s ) );
  const bool withNormals = srcNormals != nullptr;
  if ( withNormals ) {
    TMC2_TRY( d_srcNrm.alloc( 3 * size_t( n ) ) );
    TMC2_HIP( hipMemcpyAsync( d_srcNrm.p, srcNormals, 3 * size_t( n ) * sizeof( double ), hipMemcpyHostToDevice, s ) );
  }
  DevCloud         dS, dR, dN;  // de-duplicated source, de-duplicated reconstruction, normal cloud (original source order)
  DevBuf<uint32_t> d_firstS, d_firstR;
  TMC2_TRY( removeDuplicatesDevice( ctx, d_srcXyz.p, d_srcRgb.p, uint32_t( n ), dS, d_firstS ) );
  TMC2_TRY( removeDuplicatesDevice( ctx, d_recXyz.p, d_recRgb.p, uint32_t( m ), dR, d_firstR ) );
  if ( counts ) counts[0] = int64_t( dS.n ), counts[1] = int64_t( dR.n );
  if ( dS.n < size_t( K ) || dR.n < size_t( K ) ) {
    setError( "metrics_compute: clouds smaller than %d points unsupported", K );
    return TMC2_E_UNSUPPORTED;
  }
  const dim3 blk( 256 );
  if ( withNormals ) {
    if ( dS.n != n ) {
      setError( "metrics_compute: the source has duplicate positions; normals cannot be attached (the reference exits)" );
      return TMC2_E_INVALID;
    }
    // copyNormals: the de-duplicated source is the lexicographic sort of the input
    TMC2_TRY( dS.nrm.alloc( 3 * size_t( n ) ) );
    hipLaunchKernelGGL( gatherNormalsKernel, dim3( uint32_t( ( n + 255 ) / 256 ) ), blk, 0, s, d_srcNrm.p, d_firstS.p, uint32_t( n ),
                        dS.nrm.p );
  }
  TMC2_TRY( dS.buildTree( ctx ) );
  TMC2_TRY( dR.buildTree( ctx ) );
  ctx->stageEnd( sidPrep );
  DevBuf<uint32_t> d_error;
  TMC2_TRY( d_error.alloc( 1 ) );
  TMC2_HIP( hipMemsetAsync( d_error.p, 0, 4, s ) );
  if ( withNormals ) {
    // scaleNormals
    const uint32_t mR = uint32_t( dR.n ), nS = uint32_t( n );
    dN.n = n;  // the normal cloud: the source in its original order (no duplicates: checked above)
    TMC2_TRY( dN.pts.alloc( n ) );
    hipLaunchKernelGGL( rawPointsKernel, dim3( uint32_t( ( n + 255 ) / 256 ) ), blk, 0, s, d_srcXyz.p, uint32_t( n ), dN.pts.p );
    const double* d_nrmN = d_srcNrm.p;
    DevBuf<uint32_t> d_idx, d_dist, d_count, d_offset, d_cursor, d_voters, d_total;
    TMC2_TRY( d_idx.alloc( size_t( nS ) * K ) );
    TMC2_TRY( d_dist.alloc( size_t( nS ) * K ) );
    TMC2_TRY( d_count.alloc( mR ) );
    TMC2_TRY( d_offset.alloc( mR ) );
    TMC2_TRY( d_cursor.alloc( mR ) );
    TMC2_TRY( d_total.alloc( 1 ) );
    TMC2_TRY( dR.nrm.alloc( 3 * size_t( mR ) ) );
    TMC2_TRY( launchKnnTree( ctx, dR.devFor( dS ), dN.pts.p, nS, K, d_idx.p, d_dist.p, "metrics_knn16" ) );  // (dN = the points of dS)
    TMC2_HIP( hipMemsetAsync( d_count.p, 0, size_t( mR ) * 4, s ) );
    TMC2_HIP( hipMemsetAsync( d_cursor.p, 0, size_t( mR ) * 4, s ) );
    hipLaunchKernelGGL( votesCountKernel, dim3( ( nS + 255 ) / 256 ), blk, 0, s, d_idx.p, d_dist.p, nS, d_count.p, d_error.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_offset.p, mR, d_total.p ) );
    uint32_t total = 0;
    TMC2_HIP( hipMemcpyAsync( &total, d_total.p, 4, hipMemcpyDeviceToHost, s ) );
    std::vector<uint32_t> h_count( mR );
    TMC2_HIP( hipMemcpyAsync( h_count.data(), d_count.p, size_t( mR ) * 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    TMC2_TRY( d_voters.alloc( std::max( total, 1u ) ) );
    hipLaunchKernelGGL( votesFillKernel, dim3( ( nS + 255 ) / 256 ), blk, 0, s, d_idx.p, d_dist.p, nS, d_offset.p, d_cursor.p,
                        d_voters.p );
    hipLaunchKernelGGL( votesReduceKernel, dim3( ( mR + 255 ) / 256 ), blk, 0, s, d_count.p, d_offset.p, d_voters.p, d_nrmN, mR,
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
      TMC2_TRY( launchKnnTree( ctx, dN.devFor( dR ), d_q.p, nO, K, d_oi.p, d_od.p, "metrics_knn16" ) );
      hipLaunchKernelGGL( orphanNormalsKernel, dim3( ( nO + 255 ) / 256 ), blk, 0, s, d_orph.p, nO, d_oi.p, d_od.p, d_nrmN,
                          dR.nrm.p, d_error.p );
      TMC2_HIP( hipStreamSynchronize( s ) );
    }
  }
  {
    DevBuf<double> d_termsS, d_termsR, d_sums;
    TMC2_TRY( d_sums.alloc( 16 ) );
    TMC2_TRY( qualityTerms( ctx, dS, dR, withNormals, d_termsS, d_error.p ) );
    TMC2_TRY( qualityTerms( ctx, dR, dS, withNormals, d_termsR, d_error.p ) );
    const int    sid = ctx->stageBegin( "metrics_sums" );
    const size_t lds = size_t( 2 ) * 8 * kSumStride * sizeof( double );
    TMC2_TRY( allowLargeLds( reinterpret_cast<const void*>( orderedSumsKernel ), lds, ctx->device ) );
    hipLaunchKernelGGL( orderedSumsKernel, dim3( 1 ), dim3( 1024 ), lds, s, d_termsS.p, uint32_t( dS.n ), d_termsR.p, uint32_t( dR.n ),
                        d_sums.p );
    ctx->stageEnd( sid );
    double sse[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    TMC2_HIP( hipMemcpyAsync( sse, d_sums.p, sizeof( sse ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    qualityFromSums( sse, double( dS.n ), withNormals, resolution, out );
    qualityFromSums( sse + 5, double( dR.n ), withNormals, resolution, out + 8 );
  }
  for ( int i = 0; i < 8; ++i ) {
    const bool isPsnr = ( i == 1 || i == 3 || i == 7 );
    out[16 + i]       = isPsnr ? std::min( out[i], out[8 + i] ) : std::max( out[i], out[8 + i] );
  }
  uint32_t err = 0;
  TMC2_HIP( hipMemcpyAsync( &err, d_error.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_HIP( hipGetLastError() );
  if ( err ) {
    setError( "metrics_compute: a query has 16 or more equidistant nearest neighbours (the reference extends its search "
              "to 30; not reproduced)" );
    return TMC2_E_UNSUPPORTED;
  }
  return TMC2_OK;
}
