// orient_contract.hip -- device side of the normal orientation (S3): contraction of the orientation graph.
//
// Part of the replacement of PCCNormalsGenerator3::orientNormals, SPANNING_TREE branch (reference:
// source/lib/PccLibEncoder/source/PCCNormalsGenerator.cpp:198-242, 521-548).  The growth itself is sequential and runs
// on the host (orient_host.cpp, which also states why the contraction is exact); what it walks need not be the 0.8 M
// points and 13 M edges of the k-NN graph:
//   * points joined by MUTUAL strong edges (both list each other, |n_u . n_v| >= tau) are absorbed together, and if
//     their strong edges agree with one relative sign assignment only the sign of the whole cluster is left to decide.
//     Clusters + relative signs = a union-find with PARITY over the mutual strong edges: lock-free, one 32-bit word
//     per point (parent << 1 | parity-to-parent), hooking by compare-and-swap under the root of smaller hashed
//     priority, path halving that composes parities -- every word ever stored states a true relation, so stale reads
//     are harmless.  A second pass checks every mutual strong edge against the parities (a disagreement = a cluster
//     whose orientation would depend on the order of the growth: the caller falls back to the point-level walk);
//   * what is left for the host are the CROSS edges (ends in different clusters): counted per source cluster, prefix
//     summed, scattered -- about 4 % of the edges and ~18 K clusters at longdress size with tau = 0.98, 16 MB over
//     PCIe instead of 180 MB, and a 14 ms walk instead of 180 ms.
#include "internal.h"

namespace tmc2 {
namespace {

__device__ __forceinline__ uint32_t ufPriority( uint32_t x ) { return x * 2654435761u; }  // odd multiplier: a bijection

// The kernels below that read a point's own k-NN row (16 ids = 64 bytes) and dot-product row (16 doubles = 128 bytes) give the
// point 16 lanes, lane j holding edge j: a wavefront reads four rows back to back (fully coalesced) instead of 64 rows
// at a stride, which thrashed L1 / L2 (round-1 counters: 6 - 13 x the algorithmic bytes reached HBM).  The per-point
// results are ballots over the 16 lanes.
__device__ __forceinline__ uint32_t ballot16( bool pred, int lane ) {  // the 16 lanes of this point, as bits 0 .. 15
  return uint32_t( __ballot( pred ) >> ( lane & 48 ) ) & 0xFFFFu;
}

// bit j: knn[u][j] is a mutual strong neighbour of u (mutual bits: ensureMutualMask, shared with S7)
template <int K>
__global__ __launch_bounds__( 256 ) void strongMutualMaskKernel( const uint16_t* __restrict__ mutual, const double* __restrict__ edgeDot,
                                                                  uint32_t n, double tau, uint16_t* __restrict__ mask ) {
  static_assert( K == 16, "16 lanes per point" );
  const uint32_t u = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  const int      j = threadIdx.x & 15, lane = threadIdx.x & 63;
  const bool     in = u < n;
  const bool     strong = in && ( ( mutual[u] >> j ) & 1u ) && fabs( edgeDot[size_t( u ) * K + j] ) >= tau;
  const uint32_t out    = ballot16( strong, lane );
  if ( in && j == 0 ) mask[u] = uint16_t( out );
}

// Initial forest without a single atomic: every point hooks itself under the mutual strong neighbour of smallest hashed
// priority, if that is smaller than its own (priorities strictly decrease along parent links: no cycles; the word
// states a true relation).  Most of the union work is done before the first compare-and-swap, and the paths the
// union pass walks end at local priority minima a few steps away.
template <int K>
__global__ __launch_bounds__( 256 ) void initWordsKernel( const uint32_t* __restrict__ knn, const double* __restrict__ edgeDot,
                                                           const uint16_t* __restrict__ mask, uint32_t n,
                                                           uint32_t* __restrict__ word, uint32_t* __restrict__ count ) {
  static_assert( K == 16, "16 lanes per point" );
  const uint32_t i = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  const int      j = threadIdx.x & 15;
  if ( i > n ) return;  // (uniform over the 16 lanes of a point)
  if ( i == n ) {
    if ( j == 0 ) count[n] = 0;
    return;
  }
  // candidate of this lane: neighbour j if it is a mutual strong one, else the point itself (first minimum wins below, as
  // the sequential scan over the set bits in ascending j did: strict "<" kept the earliest)
  const bool     cand = ( mask[i] >> j ) & 1u;
  const uint32_t v    = cand ? knn[size_t( i ) * K + j] : i;
  uint32_t       prio = cand ? ufPriority( v ) : 0xFFFFFFFFu;
  uint32_t       best = v, parity = cand && edgeDot[size_t( i ) * K + j] < 0.0 ? 1u : 0u;
  uint32_t       slot = uint32_t( j );
#pragma unroll
  for ( int off = 8; off > 0; off >>= 1 ) {  // minimum of (priority, j) over the 16 lanes
    const uint32_t op = __shfl_xor( prio, off, 64 ), ob = __shfl_xor( best, off, 64 ), oq = __shfl_xor( parity, off, 64 ),
                   os = __shfl_xor( slot, off, 64 );
    if ( op < prio || ( op == prio && os < slot ) ) prio = op, best = ob, parity = oq, slot = os;
  }
  if ( j == 0 ) {
    count[i] = 0;
    if ( prio >= ufPriority( i ) ) best = i;  // (no mutual strong neighbour of smaller priority: its own root)
    word[i] = ( best << 1 ) | ( best == i ? 0u : parity );
  }
}

// root of x and the parity of x relative to it; halves the path on the way
// (the climb reads through the XCD's L2 -- workgroup-scope loads, a view possibly behind the other XCDs': a link read there is
// still a link of the forest with its parity -- and only "is this a root" goes to the coherent level, climbing on if it is not)
__device__ __forceinline__ uint32_t parityFind( uint32_t* word, uint32_t x, uint32_t& parity, bool agent ) {
  uint32_t acc = 0;
  for ( ;; ) {
    const uint32_t w = loadStaleOk( &word[x], agent );
    const uint32_t p = w >> 1;
    if ( p == x ) break;
    const uint32_t wp = loadStaleOk( &word[p], agent );
    const uint32_t gp = wp >> 1;
    if ( gp != p ) __hip_atomic_store( &word[x], ( gp << 1 ) | ( ( w ^ wp ) & 1u ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    acc ^= w & 1u;
    x = p;
  }
  for ( ;; ) {
    const uint32_t w = __hip_atomic_load( &word[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    const uint32_t p = w >> 1;
    if ( p == x ) break;
    acc ^= w & 1u;
    x = p;
  }
  parity = acc;
  return x;
}

// "Are a and b in one cluster already?" from this CU's possibly stale view, without a store or an atomic (the same walk as
// ufSameSetStale in patches.hip: the end of larger priority climbs; true is final because every word ever stored links two
// members of one cluster, false only sends the caller to the coherent loop).  Whether the parities along the two paths agree
// with the edge is not this walk's business: verifyCountKernel checks every mutual strong edge on the settled forest.
__device__ __forceinline__ bool paritySameSetStale( const uint32_t* word, uint32_t a, uint32_t b, bool agent ) {
  uint32_t pa = ufPriority( a ), pb = ufPriority( b );
  for ( ;; ) {
    if ( a == b ) return true;
    if ( pa < pb ) {
      const uint32_t t = a;
      a                = b;
      b                = t;
      const uint32_t q = pa;
      pa               = pb;
      pb               = q;
    }
    const uint32_t up = loadStaleOk( &word[a], agent ) >> 1;
    if ( up == a ) return false;
    a  = up;
    pa = ufPriority( a );
  }
}

template <int K>
__global__ __launch_bounds__( 256 ) void parityUnionKernel( const uint32_t* __restrict__ knn, const double* __restrict__ edgeDot,
                                                             const uint16_t* __restrict__ mask, uint32_t n,
                                                             uint32_t* __restrict__ word, int precheck, bool agent ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u >= n ) return;
  uint32_t m = mask[u];
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = knn[size_t( u ) * K + j];
    if ( v > u ) continue;  // every mutual edge is seen from both ends: the larger one acts
    if ( precheck && paritySameSetStale( word, u, v, agent ) ) continue;
    const uint32_t s = edgeDot[size_t( u ) * K + j] < 0.0 ? 1u : 0u;  // 1: the two normals must get opposite signs
    for ( ;; ) {
      uint32_t pa, pb;
      uint32_t a = parityFind( word, u, pa, agent ), b = parityFind( word, v, pb, agent );
      if ( a == b ) break;  // (whether the parities agree with s is checked afterwards, on the settled forest)
      if ( ufPriority( a ) < ufPriority( b ) ) {
        const uint32_t t = a;
        a                = b;
        b                = t;
      }
      // hook a under b: sign(a) sign(b) = sign(u) sign(v) (-1)^(pa ^ pb) = (-1)^(s ^ pa ^ pb)
      if ( atomicCAS( &word[a], a << 1, ( b << 1 ) | ( pa ^ pb ^ s ) ) == ( a << 1 ) ) break;
    }
  }
}

// Debug invariants of the settled forest (TMC2_UF_CHECK=1): links fall in priority; both ends of every mutual strong edge
// have one root (agent-scope climbs only).  bad[0] = broken links, bad[1] = split edges.
template <int K>
__global__ __launch_bounds__( 256 ) void parityCheckKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ mask,
                                                             uint32_t n, uint32_t* word, uint32_t* __restrict__ bad ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u >= n ) return;
  auto rootOf = [&]( uint32_t x ) {
    for ( uint32_t hops = 0; hops <= n; ++hops ) {
      const uint32_t q = __hip_atomic_load( &word[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) >> 1;
      if ( q == x ) return x;
      if ( q >= n || ufPriority( q ) >= ufPriority( x ) ) return 0xFFFFFFFFu;
      x = q;
    }
    return 0xFFFFFFFFu;
  };
  const uint32_t ru = rootOf( u );
  if ( ru == 0xFFFFFFFFu ) {
    atomicAdd( &bad[0], 1u );
    return;
  }
  uint32_t m = mask[u];
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = knn[size_t( u ) * K + j];
    if ( v > u ) continue;
    if ( rootOf( v ) != ru ) atomicAdd( &bad[1], 1u );
  }
}

__global__ __launch_bounds__( 256 ) void flattenKernel( uint32_t n, uint32_t* __restrict__ word, uint32_t* __restrict__ root,
                                                         uint8_t* __restrict__ parity, bool agent ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u >= n ) return;
  uint32_t       p;
  const uint32_t r = parityFind( word, u, p, agent );
  root[u]          = r;
  parity[u]        = uint8_t( p );
}

// every mutual strong edge against the settled parities; cross edges counted per source cluster
template <int K>
__global__ __launch_bounds__( 256 ) void verifyCountKernel( const uint32_t* __restrict__ knn, const double* __restrict__ edgeDot,
                                                             const uint16_t* __restrict__ mask, const uint32_t* __restrict__ root,
                                                             const uint8_t* __restrict__ parity, uint32_t n,
                                                             uint32_t* __restrict__ count, uint16_t* __restrict__ crossMask,
                                                             uint32_t* __restrict__ bad ) {
  static_assert( K == 16, "16 lanes per point" );
  const uint32_t u = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  const int      j = threadIdx.x & 15, lane = threadIdx.x & 63;
  const bool     in = u < n;
  bool           isCross = false, wrong = false;
  uint32_t       ru = 0;
  if ( in ) {
    ru                 = root[u];
    const uint32_t v   = knn[size_t( u ) * K + j];
    isCross            = root[v] != ru;
    if ( ( mask[u] >> j ) & 1u ) {
      const uint32_t s = edgeDot[size_t( u ) * K + j] < 0.0 ? 1u : 0u;
      wrong            = ( uint32_t( parity[u] ) ^ parity[v] ) != s;
    }
  }
  const uint32_t cross = ballot16( isCross, lane );  // bit j: edge j leaves the cluster (kept for the scatter pass)
  if ( __ballot( wrong ) && lane == 0 ) *bad = 1u;
  if ( in && j == 0 ) crossMask[u] = uint16_t( cross );
  // A smooth body part is ONE cluster of 100 K points: tens of thousands of reports on one counter, ~ 10 ns each, were most of
  // this kernel.  The 16 points of a workgroup are neighbours in index order and mostly share their cluster: the first of
  // them with a given root reports for all.
  __shared__ uint32_t sRoot[16], sCross[16];
  const int p = threadIdx.x >> 4;
  if ( j == 0 ) sRoot[p] = in ? ru : 0xFFFFFFFFu, sCross[p] = in ? uint32_t( __popc( cross ) ) : 0u;
  __syncthreads();
  if ( in && j == 0 ) {
    bool     first = true;
    uint32_t total = 0;
    for ( int q = 0; q < 16; ++q ) {
      if ( sRoot[q] != ru ) continue;
      if ( q < p ) first = false;
      total += sCross[q];
    }
    if ( first && total ) atomicAdd( &count[ru], total );
  }
}

template <int K>
__global__ __launch_bounds__( 256 ) void scatterCrossKernel( const uint32_t* __restrict__ knn, const double* __restrict__ edgeDot,
                                                              const uint32_t* __restrict__ root, const uint32_t* __restrict__ off,
                                                              const uint16_t* __restrict__ crossMask, uint32_t n,
                                                              uint32_t* __restrict__ cursor, OrientCrossEdge* __restrict__ edges ) {
  static_assert( K == 16, "16 lanes per point" );
  const uint32_t u = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  const int      j = threadIdx.x & 15, p = threadIdx.x >> 4;
  const bool     in = u < n;
  const uint32_t m  = in ? crossMask[u] : 0u;
  // one reservation per (workgroup, cluster), as the counts were reported (verifyCountKernel): the first point of the
  // workgroup with a given root reserves for all, the others take their share in index order
  __shared__ uint32_t sRoot[16], sCross[16], sBase[16];
  const uint32_t ru = in ? root[u] : 0xFFFFFFFFu;
  if ( j == 0 ) sRoot[p] = ru, sCross[p] = uint32_t( __popc( m ) );
  __syncthreads();
  if ( j == 0 && m ) {
    bool     first = true;
    uint32_t total = 0;
    for ( int q = 0; q < 16; ++q ) {
      if ( sRoot[q] != ru || !sCross[q] ) continue;
      if ( q < p ) first = false;
      total += sCross[q];
    }
    if ( first ) sBase[p] = off[ru] + atomicAdd( &cursor[ru], total );
  }
  __syncthreads();
  if ( !m ) return;  // (uniform over the 16 lanes of a point)
  uint32_t at = 0;
  if ( j == 0 ) {
    int lead = p;
    for ( int q = 0; q < p; ++q ) {
      if ( sRoot[q] != ru || !sCross[q] ) continue;
      if ( lead == p ) lead = q;  // the first point of this root that has cross edges made the reservation
      at += sCross[q];
    }
    at += sBase[lead];
  }
  at = __shfl( at, ( threadIdx.x & 63 ) & 48, 64 );
  if ( ( m >> j ) & 1u )  // (edges of a point in ascending j, as the sequential scan over the set bits wrote them)
    edges[at + __popc( m & ( ( 1u << j ) - 1u ) )] = OrientCrossEdge{u, knn[size_t( u ) * K + j], edgeDot[size_t( u ) * K + j]};
}

__global__ __launch_bounds__( 256 ) void clusterSignsKernel( const uint32_t* __restrict__ root, const uint8_t* __restrict__ parity,
                                                              const int8_t* __restrict__ clusterSign, uint32_t n,
                                                              int8_t* __restrict__ sign ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u < n ) sign[u] = int8_t( parity[u] ? -clusterSign[root[u]] : clusterSign[root[u]] );
}

// inputs of the reference's seed rule for the listed seeds: their k-NN rows and the normals of (0) the seed, (1) the
// point before it in index order, (2 + t) its t-th neighbour
template <int K>
__global__ __launch_bounds__( 256 ) void seedTableKernel( const uint32_t* __restrict__ seeds, uint32_t count,
                                                           const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                           uint32_t* __restrict__ rows, double* __restrict__ out ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= count * ( K + 2 ) ) return;
  const uint32_t s = t / ( K + 2 ), j = t % ( K + 2 );
  const uint32_t i = seeds[s];
  const uint32_t v = j == 0 ? i : ( j == 1 ? ( i ? i - 1 : 0 ) : knn[size_t( i ) * K + ( j - 2 )] );
  if ( j >= 2 ) rows[size_t( s ) * K + ( j - 2 )] = v;
  for ( int c = 0; c < 3; ++c ) out[3 * size_t( t ) + c] = normals[3 * size_t( v ) + c];
}

}  // namespace

// rows: [count][16], normals: [count][18][3] (host vectors), for the seeds the walk has listed
int gatherSeedTables( tmc2_frame* f, const std::vector<uint32_t>& seeds, std::vector<uint32_t>& rows, std::vector<double>& normals ) {
  const uint32_t count = uint32_t( seeds.size() );
  rows.assign( size_t( count ) * 16, 0 );
  normals.assign( size_t( count ) * 18 * 3, 0.0 );
  if ( !count ) return TMC2_OK;
  hipStream_t      s = f->ctx->stream;
  DevBuf<uint32_t> d_seeds, d_rows;
  DevBuf<double>   d_out;
  TMC2_TRY( d_seeds.alloc( count ) );
  TMC2_TRY( d_rows.alloc( size_t( count ) * 16 ) );
  TMC2_TRY( d_out.alloc( size_t( count ) * 18 * 3 ) );
  TMC2_HIP( hipMemcpyAsync( d_seeds.p, seeds.data(), size_t( count ) * 4, hipMemcpyHostToDevice, s ) );
  hipLaunchKernelGGL( seedTableKernel<16>, dim3( ( count * 18 + 255 ) / 256 ), dim3( 256 ), 0, s, d_seeds.p, count, f->d_knn.p,
                      f->d_normals.p, d_rows.p, d_out.p );
  TMC2_HIP( hipMemcpyAsync( rows.data(), d_rows.p, rows.size() * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( normals.data(), d_out.p, normals.size() * 8, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

// Contracts the orientation graph of frame f on the device (k = 16) and brings it to the host (page-locked staging of the
// context).  ok = false: some cluster's strong edges disagree.  d_root / d_parity stay valid for launchClusterSigns.
int contractOrientationDevice( tmc2_frame* f, const double* d_edgeDot, double tau, DevBuf<uint32_t>& d_root,
                               DevBuf<uint8_t>& d_parity, OrientContraction& g, bool& ok ) {
  tmc2_ctx*      ctx = f->ctx;
  hipStream_t    s   = ctx->stream;
  const uint32_t n   = uint32_t( f->n );
  ok                 = false;
  if ( f->k != 16 ) return TMC2_OK;  // not instantiated: the caller walks the points
  DevBuf<uint32_t> d_word, d_count, d_off, d_cursor, d_small;
  DevBuf<uint16_t> d_mask;
  TMC2_TRY( d_word.alloc( n ) );
  TMC2_TRY( d_count.alloc( size_t( n ) + 1 ) );
  TMC2_TRY( d_off.alloc( size_t( n ) + 1 ) );
  TMC2_TRY( d_cursor.alloc( n ) );
  TMC2_TRY( d_small.alloc( 4 ) );  // [0] bad flag, [1] total cross edges
  TMC2_TRY( d_mask.alloc( n ) );
  TMC2_TRY( d_root.alloc( n ) );
  TMC2_TRY( d_parity.alloc( n ) );
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  TMC2_TRY( fillRegions( ctx, {{d_small.p, 16, 0}, {d_cursor.p, size_t( n ) * 4, 0}} ) );
  TMC2_TRY( ensureMutualMask( f ) );
  const dim3 grdN16( ( n + 15 ) / 16 ), grdN16p( ( n + 16 ) / 16 );  // 16 lanes per point (... and one more "point" for count[n])
  hipLaunchKernelGGL( strongMutualMaskKernel<16>, grdN16, blk, 0, s, f->d_mutual.p, d_edgeDot, n, tau, d_mask.p );
  hipLaunchKernelGGL( initWordsKernel<16>, grdN16p, blk, 0, s, f->d_knn.p, d_edgeDot, d_mask.p, n, d_word.p, d_count.p );
  hipLaunchKernelGGL( parityUnionKernel<16>, grdN, blk, 0, s, f->d_knn.p, d_edgeDot, d_mask.p, n, d_word.p, unionPrecheck(), unionAgentScope() );
  if ( unionCheck() ) {  // debug invariants (soak tests): costs a round trip
    uint32_t bad[2] = {0, 0};
    hipLaunchKernelGGL( parityCheckKernel<16>, grdN, blk, 0, s, f->d_knn.p, d_mask.p, n, d_word.p, d_small.p + 2 );
    TMC2_HIP( hipMemcpyAsync( bad, d_small.p + 2, 8, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    if ( bad[0] | bad[1] ) {
      setError( "orientNormals: union-find invariant broken (%u bad links, %u split edges)", bad[0], bad[1] );
      return TMC2_E_HIP;
    }
  }
  hipLaunchKernelGGL( flattenKernel, grdN, blk, 0, s, n, d_word.p, d_root.p, d_parity.p, unionAgentScope() );
  DevBuf<uint16_t> d_crossMask;
  TMC2_TRY( d_crossMask.alloc( n ) );
  hipLaunchKernelGGL( verifyCountKernel<16>, grdN16, blk, 0, s, f->d_knn.p, d_edgeDot, d_mask.p, d_root.p, d_parity.p, n,
                      d_count.p, d_crossMask.p, d_small.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_off.p, size_t( n ) + 1, d_small.p + 1 ) );
  uint32_t head[2] = {0, 0};
  TMC2_HIP( hipMemcpyAsync( head, d_small.p, 8, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  if ( head[0] ) return TMC2_OK;  // inconsistent cluster
  const uint32_t   E = head[1];
  DevBuf<OrientCrossEdge> d_edges;
  TMC2_TRY( d_edges.alloc( std::max<uint32_t>( E, 1u ) ) );
  hipLaunchKernelGGL( scatterCrossKernel<16>, grdN16, blk, 0, s, f->d_knn.p, d_edgeDot, d_root.p, d_off.p, d_crossMask.p, n,
                      d_cursor.p, d_edges.p );
  uint32_t*        h_root   = ctx->hostA.get<uint32_t>( 2 * size_t( n ) + 2 );  // root | off
  uint8_t*         h_parity = ctx->hostC.get<uint8_t>( 2 * size_t( n ) );       // parity | (cluster signs, see the caller)
  OrientCrossEdge* h_edges  = ctx->hostE.get<OrientCrossEdge>( std::max<uint32_t>( E, 1u ) );
  if ( !h_root || !h_parity || !h_edges ) {
    setError( "orientNormals: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  uint32_t* h_off = h_root + n;
  TMC2_HIP( hipMemcpyAsync( h_root, d_root.p, size_t( n ) * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( h_off, d_off.p, ( size_t( n ) + 1 ) * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( h_parity, d_parity.p, n, hipMemcpyDeviceToHost, s ) );
  if ( E ) TMC2_HIP( hipMemcpyAsync( h_edges, d_edges.p, size_t( E ) * sizeof( OrientCrossEdge ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  g.root = h_root, g.off = h_off, g.parity = h_parity, g.edges = h_edges;
  ok     = true;
  return TMC2_OK;
}

// sign[v] = clusterSign[root[v]] * (-1)^parity[v]
int launchClusterSigns( tmc2_frame* f, const uint32_t* d_root, const uint8_t* d_parity, const int8_t* d_clusterSign,
                        int8_t* d_sign ) {
  const uint32_t n = uint32_t( f->n );
  hipLaunchKernelGGL( clusterSignsKernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, f->ctx->stream, d_root, d_parity,
                      d_clusterSign, n, d_sign );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2
