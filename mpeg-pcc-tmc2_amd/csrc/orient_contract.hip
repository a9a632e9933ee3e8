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

// n_u . n_v with the operand order and rounding of the reference's dot product (PCCMath.h operator*; -ffp-contract=off): the ONE
// place the contraction computes it.  Rounds 2-5 kept the 16 N doubles in HBM (edgeDotKernel: 107 MB written, read again by four
// kernels); since round 6 a pass gets the classes it needs as bits (strongAll / negAll below) and the few edges whose VALUE matters
// -- the cross edges, ~ 4 % -- are recomputed from the two normals where they are used (same expression: same bits).
__device__ __forceinline__ double edgeDotOf( const double* __restrict__ normals, uint32_t u, uint32_t v ) {
  const double* a = normals + 3 * size_t( u );
  const double* b = normals + 3 * size_t( v );
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// Which 16-point group a workgroup takes next.  The passes over the edges walk the points in TREE order (perm: tree position ->
// point; neighbours in space are neighbours in the tree) and XCD x works through the x-th eighth of the groups (block b runs on
// XCD b % 8 -- observed, not promised: only speed depends on it; the mapping of knnKernel): the rows, words and cluster records a
// group's neighbours need are the ones its own XCD's L2 has just fetched, instead of every L2 seeing every region of the cloud.
// perm == nullptr (an adjacency that came from the caller: no tree): index order, same eighths.
struct GroupWalk {
  uint32_t g, end, step;
  __device__ __forceinline__ GroupWalk( uint32_t groups, bool chunked ) {  // (chunked: gridDim.x is a multiple of 8)
    if ( !chunked ) {
      g = blockIdx.x, end = groups, step = gridDim.x;
      return;
    }
    const uint32_t x = blockIdx.x & 7u, perXcd = gridDim.x >> 3;
    const uint32_t begin = uint32_t( ( uint64_t( groups ) * x ) >> 3 );
    end  = uint32_t( ( uint64_t( groups ) * ( x + 1u ) ) >> 3 );
    g    = begin + ( blockIdx.x >> 3 );
    step = perXcd;
  }
};
// ... and which point a lane of a one-point-per-lane pass takes (n: none)
__device__ __forceinline__ uint32_t pointOfLane( const uint32_t* __restrict__ perm, bool chunked, uint32_t n ) {
  uint32_t block = blockIdx.x;
  if ( chunked ) block = ( blockIdx.x & 7u ) * ( gridDim.x >> 3 ) + ( blockIdx.x >> 3 );
  const uint32_t at = block * blockDim.x + threadIdx.x;
  return at < n ? ( perm ? perm[at] : at ) : n;
}

// Initial forest without a single atomic: every point hooks itself under the mutual strong neighbour of smallest hashed
// priority, if that is smaller than its own (priorities strictly decrease along parent links: no cycles; the word
// states a true relation).  Most of the union work is done before the first compare-and-swap, and the paths the
// union pass walks end at local priority minima a few steps away.
// Also the classes of the point's 16 edges as bits: strongAll (|n_u . n_v| >= tau), negAll (n_u . n_v < 0), mask = strongAll &
// mutual (mutual bits: ensureMutualMask, shared with S7).
template <int K>
__global__ __launch_bounds__( 256 ) void initWordsKernel( const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                           const uint16_t* __restrict__ mutual, const uint32_t* __restrict__ perm,
                                                           bool chunked, double tau, uint32_t n, uint16_t* __restrict__ mask,
                                                           uint16_t* __restrict__ strongAll, uint16_t* __restrict__ negAll,
                                                           uint32_t* __restrict__ word, uint32_t* __restrict__ count ) {
  static_assert( K == 16, "16 lanes per point" );
  const int j = threadIdx.x & 15, lane = threadIdx.x & 63;
  if ( blockIdx.x == 0 && threadIdx.x == 0 ) count[n] = 0;
  for ( GroupWalk w( ( n + 15 ) / 16, chunked ); w.g < w.end; w.g += w.step ) {  // (uniform over the 16 lanes of a point)
  const uint32_t at = w.g * 16 + ( threadIdx.x >> 4 );
  if ( at >= n ) continue;
  const uint32_t i  = perm ? perm[at] : at;
  const uint32_t vj = knn[size_t( i ) * K + j];
  const double   dj = edgeDotOf( normals, i, vj );
  const bool     strong = fabs( dj ) >= tau;
  const bool     cand   = ( ( mutual[i] >> j ) & 1u ) && strong;
  {
    const uint32_t bits = ballot16( cand, lane ), sb = ballot16( strong, lane ), nb = ballot16( dj < 0.0, lane );
    if ( j == 0 ) mask[i] = uint16_t( bits ), strongAll[i] = uint16_t( sb ), negAll[i] = uint16_t( nb );
  }
  // candidate of this lane: neighbour j if it is a mutual strong one, else the point itself (first minimum wins below, as
  // the sequential scan over the set bits in ascending j did: strict "<" kept the earliest)
  const uint32_t v    = cand ? vj : i;
  uint32_t       prio = cand ? ufPriority( v ) : 0xFFFFFFFFu;
  uint32_t       best = v, parity = cand && dj < 0.0 ? 1u : 0u;
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
__global__ __launch_bounds__( 256 ) void parityUnionKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ negAll,
                                                             const uint16_t* __restrict__ mask, const uint32_t* __restrict__ perm,
                                                             bool chunked, uint32_t n, uint32_t* __restrict__ word, int precheck,
                                                             bool agent ) {
  // (one point per lane; chunked: XCD x takes the x-th eighth of the blocks -- the words a lane's climbs touch belong to points
  //  around its own, which the same L2 serves)
  const uint32_t u = pointOfLane( perm, chunked, n );
  if ( u >= n ) return;
  uint32_t       m = mask[u];
  const uint32_t neg = negAll[u];
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = knn[size_t( u ) * K + j];
    if ( v > u ) continue;  // every mutual edge is seen from both ends: the larger one acts
    if ( precheck && paritySameSetStale( word, u, v, agent ) ) continue;
    const uint32_t s = ( neg >> j ) & 1u;  // 1: the two normals must get opposite signs
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

__global__ __launch_bounds__( 256 ) void flattenKernel( uint32_t n, const uint32_t* __restrict__ perm, bool chunked,
                                                         uint32_t* __restrict__ word, uint32_t* __restrict__ root,
                                                         uint8_t* __restrict__ parity, uint32_t* __restrict__ minIdx, bool agent ) {
  const uint32_t u = pointOfLane( perm, chunked, n );  // (as parityUnionKernel)
  if ( u >= n ) return;
  uint32_t       p;
  const uint32_t r = parityFind( word, u, p, agent );
  root[u]          = r;
  parity[u]        = uint8_t( p );
  // first member of the cluster (the running minimum only falls: look before the atomic -- the early points of a big cluster
  // settle it, the other hundred thousand skip)
  if ( minIdx && u < loadStaleOk( &minIdx[r] ) ) atomicMin( &minIdx[r], u );
}

// ---- compact form of the contracted graph ---------------------------------------------------------------------------------
// What the host walk needs is small: the clusters (~ 18 K at longdress size), numbered in the order of their first member (the
// order the walk opens components in), and per ordered pair of clusters (c, c2) ONE light cross edge -- the best one in the
// reference's edge order (|n_u . n_v|, start, end): a cluster's cross edges are all offered the moment it is absorbed, and of
// the offers into one target cluster only the best can ever be accepted (edges that tie in |n_u . n_v| all go: the walk's heap
// orders them by start and end) -- plus one strong (one-way) cross edge per implied relative sign (they absorb the target at once; a pair that carries both signs is what the walk reports as inconsistent).
// 540 K cross edges become ~ 130 K, and every per-point array of the walk (3.3 MB each, randomly accessed) a per-cluster
// one that stays in the host's L1 / L2.  Pairs are found in an open-addressing hash table in HBM (key = the two cluster
// ids; a probe sequence that runs too long raises the overflow flag and the frame takes the edge list as it is).
struct PairTable {
  unsigned long long *key, *bestW;  // key: (c + 1) << 32 | c2; bestW: the largest |d| (as bits) among the pair's light edges
  uint32_t*           strongFirst;  // [2 * slot + s]: the first strong edge (u * 16 + j) of implied relative sign s -- the one that is kept
  uint32_t            mask;         // capacity - 1 (a power of two)
};
constexpr int kPairProbes = 128;

__device__ __forceinline__ uint32_t pairHash( uint32_t c, uint32_t c2 ) {
  unsigned long long h = ( (unsigned long long)c << 32 | c2 ) * 0x9E3779B97F4A7C15ull;
  return uint32_t( h >> 29 );
}
// slot of the pair (c, c2), claimed if new; 0xFFFFFFFF: the table is too full here
__device__ __forceinline__ uint32_t pairSlot( const PairTable& t, uint32_t c, uint32_t c2, bool insert ) {
  const unsigned long long key = ( (unsigned long long)( c + 1u ) << 32 ) | c2;
  uint32_t                 s   = pairHash( c, c2 ) & t.mask;
  for ( int probe = 0; probe < kPairProbes; ++probe ) {
    unsigned long long k = __hip_atomic_load( &t.key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if ( k == 0 && insert ) {
      const unsigned long long prev = atomicCAS( &t.key[s], 0ull, key );
      k                             = prev == 0 ? key : prev;
    }
    if ( k == key ) return s;
    if ( k == 0 ) return 0xFFFFFFFFu;  // (lookup of a pair that was never inserted: cannot happen after a clean insert pass)
    s = ( s + 1 ) & t.mask;
  }
  return 0xFFFFFFFFu;
}

// first member of every cluster (flattenKernel took the minima): the clusters are numbered in that order
__global__ __launch_bounds__( 256 ) void clusterFlagKernel( const uint32_t* __restrict__ root, const uint32_t* __restrict__ minIdx, uint32_t n,
                                                             uint32_t* __restrict__ flag ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u < n ) flag[u] = minIdx[root[u]] == u ? 1u : 0u;
}
// pass 1 over the edges (16 lanes per point, lane j = edge j): the cluster ids (rank of the cluster's first member), the
// mutual strong edges against the settled parities, the cross edges into the pair table -- light ones raise the pair's best |d|
template <int K>
__global__ __launch_bounds__( 256 ) void pairInsertKernel( const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                            const uint16_t* __restrict__ strongAll, const uint16_t* __restrict__ negAll,
                                                            const uint32_t* __restrict__ perm, bool chunked, const uint32_t* __restrict__ root,
                                                            const uint32_t* __restrict__ minIdx, const uint32_t* __restrict__ rank,
                                                            const uint8_t* __restrict__ parity, uint32_t n, PairTable t,
                                                            uint32_t* __restrict__ cid, uint16_t* __restrict__ crossMask,
                                                            uint32_t* __restrict__ flags /* [0] bad, [2] overflow */ ) {
  static_assert( K == 16, "16 lanes per point" );
  const int j = threadIdx.x & 15, lane = threadIdx.x & 63;
  for ( GroupWalk w( ( n + 15 ) / 16, chunked ); w.g < w.end; w.g += w.step ) {
  const uint32_t at = w.g * 16 + ( threadIdx.x >> 4 );
  const bool     in = at < n;
  const uint32_t u  = in ? ( perm ? perm[at] : at ) : 0u;
  bool           isCross = false, wrong = false, full = false;
  if ( in ) {
    const uint32_t cu = rank[minIdx[root[u]]], v = knn[size_t( u ) * K + j], cv = rank[minIdx[root[v]]];
    const bool     strong = ( strongAll[u] >> j ) & 1u;
    const uint32_t neg    = ( negAll[u] >> j ) & 1u;
    if ( j == 0 ) cid[u] = cu;
    isCross = cv != cu;
    // every strong edge with both ends in one cluster -- the mutual ones the cluster was built from AND the one-way ones that
    // happen to fall inside it -- must agree with the parities: the growth may take any of them first (orient_host.cpp,
    // contractOnHost)
    if ( !isCross && strong ) wrong = ( uint32_t( parity[u] ) ^ parity[v] ) != neg;
    if ( isCross ) {
      const uint32_t s = pairSlot( t, cu, cv, true );
      if ( s == 0xFFFFFFFFu )
        full = true;
      else if ( !strong ) {
        const unsigned long long wgt = (unsigned long long)__double_as_longlong( fabs( edgeDotOf( normals, u, v ) ) );
        if ( wgt > __hip_atomic_load( &t.bestW[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) atomicMax( &t.bestW[s], wgt );
      } else {
        // a strong one-way edge: one per implied relative sign is enough, and it is the FIRST in (start, slot) order -- the one
        // the host-only reduction keeps (compactOnHost), whatever the scheduling
        uint32_t* first = &t.strongFirst[2 * size_t( s ) + ( neg ^ ( uint32_t( parity[u] ) ^ parity[v] ) )];
        const uint32_t id = u * 16u + uint32_t( j );
        if ( id < __hip_atomic_load( first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) atomicMin( first, id );
      }
    }
  }
  const uint32_t cross = ballot16( isCross, lane );
  if ( __ballot( wrong ) && lane == 0 ) flags[0] = 1u;
  if ( __ballot( full ) && lane == 0 ) flags[2] = 1u;
  if ( in && j == 0 ) crossMask[u] = uint16_t( cross );
  }
}
// ---- passes 2 and 3: which cross edges the walk gets, and the edges themselves, grouped by source cluster --------------------
// Both walk the points 256 per workgroup, 64 per wavefront: every lane fetches ONE point's mask (a coalesced load) and the
// wavefront then takes only the groups of four consecutive points that hold a set bit, sixteen lanes per point -- most points
// have no cross edge at all.  What the passes cost in rounds 3-5 (127 + 111 us) was neither that walk nor its barriers but the
// global atomics on the counters of a few BIG clusters (the body of a figure is one cluster with tens of thousands of cross
// edges): thousands of adds to one word, ~ 11 ns each, one after the other (tools/gpu/r6/call26.sh: pairSelectKernel without
// its adds takes 67 us instead of 147).  A workgroup's 256 consecutive points belong to a handful of clusters: their counts
// are folded in an LDS table first (key = cluster, open addressing, 512 slots for at most 256 keys) and every (workgroup,
// cluster) pair costs ONE global atomic.
constexpr uint32_t kFoldSlots = 512, kFoldEmpty = 0xFFFFFFFFu;
// (one lane) adds `add` to cluster cu's entry of the workgroup's table; returns the slot, and in `old` what the entry held before
__device__ __forceinline__ uint32_t foldAdd( uint32_t* key, uint32_t* val, uint32_t cu, uint32_t add, uint32_t& old ) {
  uint32_t slot = ( cu * 2654435761u ) >> 23;  // (9 bits)
  for ( ;; ) {
    const uint32_t k = atomicCAS( &key[slot], kFoldEmpty, cu );
    if ( k == kFoldEmpty || k == cu ) break;
    slot = ( slot + 1u ) & ( kFoldSlots - 1u );  // (at most 256 distinct keys per round: a free slot is always found)
  }
  old = atomicAdd( &val[slot], add );
  return slot;
}

// pass 2: bit j of keepMask[u] = the walk gets edge j of point u; count[c] = the edges cluster c keeps.  dedupe = false: every
// cross edge (the table overflowed).
template <int K>
__global__ __launch_bounds__( 256 ) void pairSelectKernel( const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                            const uint16_t* __restrict__ strongAll, const uint16_t* __restrict__ negAll,
                                                            const uint32_t* __restrict__ perm, const uint16_t* __restrict__ crossMask,
                                                            const uint32_t* __restrict__ cid, const uint8_t* __restrict__ parity,
                                                            uint32_t n, PairTable t, int dedupe, uint16_t* __restrict__ keepMask,
                                                            uint32_t* __restrict__ count ) {
  static_assert( K == 16, "16 lanes per point" );
  __shared__ uint32_t fKey[kFoldSlots], fVal[kFoldSlots];
  const int j = threadIdx.x & 15, lane = threadIdx.x & 63, p = lane >> 4;
  for ( uint32_t s = threadIdx.x; s < kFoldSlots; s += blockDim.x ) fKey[s] = kFoldEmpty, fVal[s] = 0;
  __syncthreads();
  for ( uint32_t wgBase = blockIdx.x * 256u; wgBase < n; wgBase += gridDim.x * 256u ) {  // (uniform over the workgroup: barriers inside)
  const uint32_t mineAt = wgBase + threadIdx.x;
  const uint32_t mineU  = mineAt < n ? ( perm ? perm[mineAt] : mineAt ) : 0u;
  const uint32_t mineCm = mineAt < n ? crossMask[mineU] : 0u;
  if ( mineAt < n && !mineCm ) keepMask[mineU] = 0;
  unsigned long long todo = __ballot( mineCm != 0u );
  while ( todo ) {  // (uniform over the wavefront)
    const int k = ( __ffsll( (long long)todo ) - 1 ) >> 2;
    todo &= ~( 0xFull << ( 4 * k ) );
    const uint32_t u  = __shfl( mineU, 4 * k + p, 64 );
    const uint32_t cm = __shfl( mineCm, 4 * k + p, 64 );
    const bool     in = cm != 0u;  // (a point without cross edges has its zero already)
    bool           keep = false;
    uint32_t       cu = 0xFFFFFFFFu;
    if ( in ) {
      cu   = cid[u];
      keep = ( cm >> j ) & 1u;
      if ( keep && dedupe ) {
        const uint32_t v = knn[size_t( u ) * K + j];
        const uint32_t sl = pairSlot( t, cu, cid[v], false );
        if ( sl == 0xFFFFFFFFu ) {
          keep = true;  // (unreachable after a clean insert pass; harmless: an extra edge)
        } else if ( ( strongAll[u] >> j ) & 1u ) {
          keep = t.strongFirst[2 * size_t( sl ) + ( ( ( negAll[u] >> j ) & 1u ) ^ ( uint32_t( parity[u] ) ^ parity[v] ) )] == u * 16u + uint32_t( j );
        } else {
          keep = (unsigned long long)__double_as_longlong( fabs( edgeDotOf( normals, u, v ) ) ) == t.bestW[sl];
        }
      }
    }
    const uint32_t kept = ballot16( keep, lane );
    if ( in && j == 0 ) {
      keepMask[u] = uint16_t( kept );
      uint32_t old;
      if ( kept ) (void)foldAdd( fKey, fVal, cu, uint32_t( __popc( kept ) ), old );
    }
  }
  __syncthreads();
  for ( uint32_t s = threadIdx.x; s < kFoldSlots; s += blockDim.x ) {  // this round's clusters: one global add each, and the table is empty again
    const uint32_t c = fKey[s], v = fVal[s];
    if ( c != kFoldEmpty ) {
      if ( v ) atomicAdd( &count[c], v );
      fKey[s] = kFoldEmpty, fVal[s] = 0;
    }
  }
  __syncthreads();
  }
}

// pass 3: the kept edges, per source cluster, with the target cluster and the two ends' parities folded into the dot product; and
// the cluster records (where a cluster's edges start, its seed point and parity: one copy to the host instead of three; the
// sentinel record by the last point).  Queued before the host knows the number of clusters / kept edges: both come from the
// device, writes stay inside the buffers -- a frame that needs more is repeated with exact sizes.
template <int K>
__global__ __launch_bounds__( 256 ) void scatterCompactKernel( const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                                const uint32_t* __restrict__ perm,
                                                                const uint32_t* __restrict__ cid, const uint8_t* __restrict__ parity,
                                                                const uint32_t* __restrict__ off, const uint16_t* __restrict__ keepMask,
                                                                const uint32_t* __restrict__ root, const uint32_t* __restrict__ minIdx,
                                                                uint32_t n, const uint32_t* __restrict__ clustersPtr, uint32_t edgeCap,
                                                                uint32_t recCap, uint32_t* __restrict__ cursor,
                                                                OrientCompactEdge* __restrict__ edges, OrientClusterRec* __restrict__ rec ) {
  static_assert( K == 16, "16 lanes per point" );
  __shared__ uint32_t fKey[kFoldSlots], fVal[kFoldSlots];  // (fVal: a cluster's edges in this round; after the reservation: where they start)
  const int      j = threadIdx.x & 15, lane = threadIdx.x & 63, p = lane >> 4;
  const uint32_t clusters = *clustersPtr;
  for ( uint32_t s = threadIdx.x; s < kFoldSlots; s += blockDim.x ) fKey[s] = kFoldEmpty, fVal[s] = 0;
  __syncthreads();
  for ( uint32_t wgBase = blockIdx.x * 256u; wgBase < n; wgBase += gridDim.x * 256u ) {  // (uniform over the workgroup: barriers inside)
  const uint32_t mineAt = wgBase + threadIdx.x;
  const uint32_t mineU  = mineAt < n ? ( perm ? perm[mineAt] : mineAt ) : 0u;
  const uint32_t mineM  = mineAt < n ? keepMask[mineU] : 0u;
  uint32_t       slot = 0, within = 0;  // this lane's point: its cluster's entry, and its edges' place among the round's edges of the cluster
  if ( mineAt < n ) {
    const uint32_t c = ( mineM || minIdx[root[mineU]] == mineU ) ? cid[mineU] : 0u;
    if ( minIdx[root[mineU]] == mineU && c < recCap ) rec[c] = OrientClusterRec{off[c], mineU, parity[mineU]};
    if ( mineU == n - 1 && clusters < recCap ) rec[clusters] = OrientClusterRec{off[clusters], 0u, 0u};
    if ( mineM ) slot = foldAdd( fKey, fVal, c, uint32_t( __popc( mineM ) ), within );
  }
  __syncthreads();
  for ( uint32_t s = threadIdx.x; s < kFoldSlots; s += blockDim.x ) {  // one reservation per cluster of the round
    const uint32_t c = fKey[s];
    if ( c != kFoldEmpty ) fVal[s] = off[c] + atomicAdd( &cursor[c], fVal[s] );
  }
  __syncthreads();
  const uint32_t minePos = mineM ? fVal[slot] + within : 0u;
  unsigned long long todo = __ballot( mineM != 0u );
  while ( todo ) {  // (uniform over the wavefront)
    const int k = ( __ffsll( (long long)todo ) - 1 ) >> 2;
    todo &= ~( 0xFull << ( 4 * k ) );
    const uint32_t u    = __shfl( mineU, 4 * k + p, 64 );
    const uint32_t m    = __shfl( mineM, 4 * k + p, 64 );
    const uint32_t pos0 = __shfl( minePos, 4 * k + p, 64 );
    if ( ( m >> j ) & 1u ) {
      const uint32_t v = knn[size_t( u ) * K + j];
      const double   d = edgeDotOf( normals, u, v );
      const uint32_t pos = pos0 + __popc( m & ( ( 1u << j ) - 1u ) );
      if ( pos < edgeCap ) edges[pos] = OrientCompactEdge{u, v, cid[v], 0u, ( ( parity[u] ^ parity[v] ) & 1 ) ? -d : d};
    }
  }
  __syncthreads();
  for ( uint32_t s = threadIdx.x; s < kFoldSlots; s += blockDim.x ) fKey[s] = kFoldEmpty, fVal[s] = 0;
  __syncthreads();
  }
}

__global__ __launch_bounds__( 256 ) void compactSignsKernel( const uint32_t* __restrict__ cid, const uint8_t* __restrict__ parity,
                                                              const int8_t* __restrict__ clusterSign, uint32_t n, int8_t* __restrict__ sign ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u < n ) sign[u] = int8_t( parity[u] ? -clusterSign[cid[u]] : clusterSign[cid[u]] );
}

// inputs of the reference's seed rule for the listed seeds, compact form: their k-NN rows' cluster ids and parities
// ([count][K + 1]: entry 0 = the point before the seed in index order, 1 + t = neighbour t) and the normals of (0) the seed,
// (1) the point before it, (2 + t) its t-th neighbour
template <int K>
__global__ __launch_bounds__( 256 ) void seedCompactKernel( const uint32_t* __restrict__ seeds, uint32_t count,
                                                             const uint32_t* __restrict__ knn, const double* __restrict__ normals,
                                                             const uint32_t* __restrict__ cid, const uint8_t* __restrict__ parity,
                                                             uint32_t* __restrict__ who /* [count][K + 1][3]: point, cluster, parity */,
                                                             double* __restrict__ out ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= count * ( K + 2 ) ) return;
  const uint32_t s = t / ( K + 2 ), j = t % ( K + 2 );
  const uint32_t i = seeds[s];
  const uint32_t v = j == 0 ? i : ( j == 1 ? ( i ? i - 1 : 0 ) : knn[size_t( i ) * K + ( j - 2 )] );
  if ( j >= 1 ) {
    uint32_t* w = who + ( size_t( s ) * ( K + 1 ) + ( j - 1 ) ) * 3;
    w[0] = v, w[1] = cid[v], w[2] = parity[v];
  }
  for ( int c = 0; c < 3; ++c ) out[3 * size_t( t ) + c] = normals[3 * size_t( v ) + c];
}

}  // namespace

// who: [count][17][3] (point, cluster, parity of: the point before the seed, then its 16 neighbours), normals: [count][18][3]
// (host vectors), for the seeds the walk has listed
int gatherSeedTables( tmc2_frame* f, const uint32_t* d_cid, const uint8_t* d_parity, const std::vector<uint32_t>& seeds,
                      std::vector<uint32_t>& who, std::vector<double>& normals ) {
  const uint32_t count = uint32_t( seeds.size() );
  who.assign( size_t( count ) * 17 * 3, 0 );
  normals.assign( size_t( count ) * 18 * 3, 0.0 );
  if ( !count ) return TMC2_OK;
  hipStream_t      s = f->ctx->stream;
  DevBuf<uint32_t> d_seeds, d_who;
  DevBuf<double>   d_out;
  TMC2_TRY( d_seeds.alloc( count ) );
  TMC2_TRY( d_who.alloc( who.size() ) );
  TMC2_TRY( d_out.alloc( normals.size() ) );
  TMC2_HIP( hipMemcpyAsync( d_seeds.p, seeds.data(), size_t( count ) * 4, hipMemcpyHostToDevice, s ) );
  hipLaunchKernelGGL( seedCompactKernel<16>, dim3( ( count * 18 + 255 ) / 256 ), dim3( 256 ), 0, s, d_seeds.p, count, f->d_knn.p,
                      f->d_normals.p, d_cid, d_parity, d_who.p, d_out.p );
  TMC2_HIP( hipMemcpyAsync( who.data(), d_who.p, who.size() * 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( normals.data(), d_out.p, normals.size() * 8, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

// Contracts the orientation graph of frame f on the device (k = 16) and brings its COMPACT form to the host (page-locked
// staging of the context): clusters numbered by first member, one light cross edge per ordered pair of clusters (+ the strong
// one-way ones).  ok = false: some cluster's strong edges disagree.  d_cid / d_parity stay valid for launchClusterSigns.
int contractOrientationDevice( tmc2_frame* f, double tau, DevBuf<uint32_t>& d_cid, DevBuf<uint8_t>& d_parity, OrientCompact& g, bool& ok ) {
  tmc2_ctx*      ctx = f->ctx;
  hipStream_t    s   = ctx->stream;
  const uint32_t n   = uint32_t( f->n );
  ok                 = false;
  if ( f->k != 16 ) return TMC2_OK;  // not instantiated: the caller walks the points
  DevBuf<uint32_t> d_word, d_count, d_off, d_cursor, d_small, d_root, d_minIdx, d_flag, d_rank, d_strongFirst;
  DevBuf<uint16_t> d_mask, d_strongAll, d_negAll, d_crossMask, d_keepMask;
  DevBuf<unsigned long long> d_pairs;
  TMC2_TRY( d_word.alloc( n ) );
  TMC2_TRY( d_small.alloc( 8 ) );  // [0] bad flag, [1] kept cross edges, [2] pair table overflow, [3] clusters; [4], [5]: debug check
  TMC2_TRY( d_mask.alloc( n ) );
  TMC2_TRY( d_strongAll.alloc( n ) );
  TMC2_TRY( d_negAll.alloc( n ) );
  TMC2_TRY( d_root.alloc( n ) );
  TMC2_TRY( d_minIdx.alloc( n ) );
  TMC2_TRY( d_flag.alloc( n ) );
  TMC2_TRY( d_rank.alloc( n ) );
  TMC2_TRY( d_cid.alloc( n ) );
  TMC2_TRY( d_parity.alloc( n ) );
  TMC2_TRY( d_crossMask.alloc( n ) );
  TMC2_TRY( d_keepMask.alloc( n ) );
  // (test hook TMC2_ORIENT_PAIRS: log2 of the pair table's capacity; small values force the overflow path)
  const char*    pairsEnv = ctxOption( f->ctx, "ORIENT_PAIRS" );
  const uint32_t pairCap  = 1u << ( pairsEnv ? std::min( 24, std::max( 4, atoi( pairsEnv ) ) ) : 20 );
  TMC2_TRY( d_pairs.alloc( 2 * size_t( pairCap ) ) );
  TMC2_TRY( d_strongFirst.alloc( 2 * size_t( pairCap ) ) );
  if ( n >= ( 1u << 28 ) ) return TMC2_OK;  // (edge ids are u * 16 + j in 32 bits; the caller walks the points)
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  TMC2_TRY( fillRegions( ctx, {{d_small.p, 32, 0},
                               {d_minIdx.p, size_t( n ) * 4, 0xFF},
                               {d_pairs.p, 2 * size_t( pairCap ) * 8, 0},
                               {d_strongFirst.p, 2 * size_t( pairCap ) * 4, 0xFF}} ) );
  TMC2_TRY( ensureMutualMask( f ) );
  // 16 lanes per point, groups of 16 points in a stride loop; the grids are multiples of 8 (GroupWalk: XCD x takes the x-th eighth)
  const dim3 grdN16( ( cappedBlocks( ctx, ( size_t( n ) + 15 ) / 16 ) + 7u ) & ~7u ), grdN8( ( grdN.x + 7u ) & ~7u );
  // option ORIENT_ORDER: "input" = index order, blocks as they come (rounds 2-5); "chunk" = index order, XCD x on the x-th eighth of
  // the blocks; "tree" = tree order (perm), same eighths.  Unset: what was fastest pass by pass with the GPU to itself
  // (profiles/r06_pass_order.txt) -- the union-find passes in eighths (their climbs touch the words of points around their own:
  // parityUnionKernel 349 -> 272 us), the pair-table passes as the blocks come (in eighths the inserts of a pair of clusters
  // meet in time: pairInsertKernel 95 -> 106 us); tree order costs every pass its coalesced own-row reads and wins nothing on
  // clouds that arrive in scan order.
  const char*     orderOpt = ctxOption( ctx, "ORIENT_ORDER" );
  const bool      chunked  = !( orderOpt && orderOpt[0] == 'i' );
  const bool      chunkedPairs = orderOpt && orderOpt[0] != 'i';
  const uint32_t* perm     = orderOpt && orderOpt[0] == 't' && f->haveTree && f->d_perm.p && f->d_perm.count >= n ? f->d_perm.p : nullptr;
  const double*   normals  = f->d_normals.p;
  TMC2_TRY( d_count.alloc( size_t( n ) + 1 ) );  // (initWordsKernel zeroes it; later: kept edges per cluster, C + 1 used)
  hipLaunchKernelGGL( initWordsKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, f->d_mutual.p, perm, chunked, tau, n, d_mask.p, d_strongAll.p,
                      d_negAll.p, d_word.p, d_count.p );
  hipLaunchKernelGGL( parityUnionKernel<16>, grdN8, blk, 0, s, f->d_knn.p, d_negAll.p, d_mask.p, perm, chunked, n, d_word.p, unionPrecheck( f->ctx ), unionAgentScope( f->ctx ) );
  if ( unionCheck( f->ctx ) ) {  // debug invariants (soak tests): costs a round trip
    uint32_t bad[2] = {0, 0};
    hipLaunchKernelGGL( parityCheckKernel<16>, grdN, blk, 0, s, f->d_knn.p, d_mask.p, n, d_word.p, d_small.p + 4 );
    TMC2_HIP( hipMemcpyAsync( bad, d_small.p + 4, 8, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    if ( bad[0] | bad[1] ) {
      setError( "orientNormals: union-find invariant broken (%u bad links, %u split edges)", bad[0], bad[1] );
      return TMC2_E_HIP;
    }
  }
  hipLaunchKernelGGL( flattenKernel, grdN8, blk, 0, s, n, perm, chunked, d_word.p, d_root.p, d_parity.p, d_minIdx.p, unionAgentScope( f->ctx ) );
  hipLaunchKernelGGL( clusterFlagKernel, grdN, blk, 0, s, d_root.p, d_minIdx.p, n, d_flag.p );
  TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_rank.p, n, d_small.p + 3 ) );
  TMC2_TRY( d_off.alloc( size_t( n ) + 1 ) );  // (per-cluster arrays sized for the worst case, n clusters: the used part is known
  TMC2_TRY( d_cursor.alloc( n ) );            //  only after the round trip below)
  PairTable t{d_pairs.p, d_pairs.p + pairCap, d_strongFirst.p, pairCap - 1};
  hipLaunchKernelGGL( pairInsertKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, d_strongAll.p, d_negAll.p, perm, chunkedPairs, d_root.p, d_minIdx.p,
                      d_rank.p, d_parity.p, n, t, d_cid.p, d_crossMask.p, d_small.p );
  // First attempt without knowing the sizes: room for kSpecEdges kept edges and kSpecClusters clusters (three times what a
  // longdress frame needs), everything queued back to back and ONE round trip -- counters, cluster records and edges come
  // back together.  A frame that needs more room (or whose pair table overflowed) is repeated with exact sizes, two round trips.
  // (test hook TMC2_ORIENT_SPEC = "<edges>,<clusters>": shrinks the speculative room so that small clouds take the repeat)
  uint32_t kSpecEdges = 384 * 1024, kSpecClusters = 64 * 1024;
  if ( const char* spec = ctxOption( f->ctx, "ORIENT_SPEC" ) ) {
    unsigned e = 0, c = 0;
    if ( sscanf( spec, "%u,%u", &e, &c ) == 2 ) kSpecEdges = std::max( 1u, std::min( kSpecEdges, e ) ), kSpecClusters = std::max( 2u, std::min( kSpecClusters, c ) );
  }
  DevBuf<OrientCompactEdge> d_edges;
  DevBuf<OrientClusterRec>  d_rec;
  TMC2_TRY( d_edges.alloc( kSpecEdges ) );
  TMC2_TRY( d_rec.alloc( kSpecClusters ) );
  uint32_t*          h_head  = ctx->hostC.get<uint32_t>( 4 + ( size_t( n ) + 4 ) / 4 + 4 );  // counters | (cluster signs, see the caller)
  OrientClusterRec*  h_rec   = ctx->hostA.get<OrientClusterRec>( kSpecClusters );
  OrientCompactEdge* h_edges = ctx->hostE.get<OrientCompactEdge>( kSpecEdges );
  if ( !h_head || !h_rec || !h_edges ) {
    setError( "orientNormals: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  TMC2_TRY( fillRegions( ctx, {{d_count.p, ( size_t( n ) + 1 ) * 4, 0}, {d_cursor.p, size_t( n ) * 4, 0}} ) );
  hipLaunchKernelGGL( pairSelectKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, d_strongAll.p, d_negAll.p, perm, d_crossMask.p, d_cid.p,
                      d_parity.p, n, t, 1, d_keepMask.p, d_count.p );
  // (the four counters in the context's page-locked line, stored by this scan's last tile: [0] kept cross edges -- its total --,
  //  [1] bad flag, [2] the same total again, [3] pair table overflow, [4] clusters: all settled by earlier launches -- no copy)
  volatile uint32_t* headLine = ctx->answerLine( tmc2_ctx::kAnswerOrientHead );
  TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_off.p, size_t( n ) + 1, d_small.p + 1, ScanAnswer{headLine, d_small.p, 4} ) );
  hipLaunchKernelGGL( scatterCompactKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, perm, d_cid.p, d_parity.p, d_off.p, d_keepMask.p, d_root.p,
                      d_minIdx.p, n, d_small.p + 3, kSpecEdges, kSpecClusters, d_cursor.p, d_edges.p, d_rec.p );
  // (what a frame typically needs, plus a margin, comes along right away; the rest -- if any -- after the counters are known)
  const uint32_t kFirstEdges = std::min( 192u * 1024, kSpecEdges ), kFirstClusters = std::min( 32u * 1024, kSpecClusters );
  TMC2_HIP( hipMemcpyAsync( h_rec, d_rec.p, size_t( kFirstClusters ) * sizeof( OrientClusterRec ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( h_edges, d_edges.p, size_t( kFirstEdges ) * sizeof( OrientCompactEdge ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  uint32_t head[4] = {headLine[1], headLine[0], headLine[3], headLine[4]};  // (d_small[1] is being written by the scan whose last tile carries the line: its total is [0])
  if ( head[0] ) return TMC2_OK;  // inconsistent cluster
  uint32_t E = head[1], C = head[3];
  if ( !head[2] && E <= kSpecEdges && C + 1 <= kSpecClusters ) {
    bool more = false;
    if ( C + 1 > kFirstClusters ) {
      TMC2_HIP( hipMemcpyAsync( h_rec + kFirstClusters, d_rec.p + kFirstClusters, size_t( C + 1 - kFirstClusters ) * sizeof( OrientClusterRec ),
                                hipMemcpyDeviceToHost, s ) );
      more = true;
    }
    if ( E > kFirstEdges ) {
      TMC2_HIP( hipMemcpyAsync( h_edges + kFirstEdges, d_edges.p + kFirstEdges, size_t( E - kFirstEdges ) * sizeof( OrientCompactEdge ),
                                hipMemcpyDeviceToHost, s ) );
      more = true;
    }
    if ( more ) TMC2_HIP( hipStreamSynchronize( s ) );
  } else {
    if ( head[2] ) {
      // the pair table overflowed: every cross edge goes (selection and counts again, without the table)
      ctx->stageAddHostMs( "orient_pair_table_overflow", 0.0 );  // (counts the frames that took every cross edge)
      TMC2_TRY( fillRegions( ctx, {{d_count.p, ( size_t( n ) + 1 ) * 4, 0}, {d_cursor.p, size_t( n ) * 4, 0}} ) );
      hipLaunchKernelGGL( pairSelectKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, d_strongAll.p, d_negAll.p, perm, d_crossMask.p, d_cid.p,
                          d_parity.p, n, t, 0, d_keepMask.p, d_count.p );
      TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_off.p, size_t( n ) + 1, d_small.p + 1 ) );
      TMC2_HIP( hipMemcpyAsync( head, d_small.p, 16, hipMemcpyDeviceToHost, s ) );
      TMC2_HIP( hipStreamSynchronize( s ) );
      E = head[1], C = head[3];
    } else {
      // only the room was short: the selection, its counts and offsets stand as they are (they never depended on the room);
      // the cursors start again and the scatter runs with exact sizes
      ctx->stageAddHostMs( "orient_exact_size_repeat", 0.0 );
      TMC2_TRY( fillRegions( ctx, {{d_cursor.p, size_t( n ) * 4, 0}} ) );
    }
    TMC2_TRY( d_edges.alloc( std::max<uint32_t>( E, 1u ) ) );
    TMC2_TRY( d_rec.alloc( size_t( C ) + 1 ) );
    hipLaunchKernelGGL( scatterCompactKernel<16>, grdN16, blk, 0, s, f->d_knn.p, normals, perm, d_cid.p, d_parity.p, d_off.p, d_keepMask.p, d_root.p,
                        d_minIdx.p, n, d_small.p + 3, std::max<uint32_t>( E, 1u ), C + 1, d_cursor.p, d_edges.p, d_rec.p );
    h_rec   = ctx->hostA.get<OrientClusterRec>( size_t( C ) + 1 );
    h_edges = ctx->hostE.get<OrientCompactEdge>( std::max<uint32_t>( E, 1u ) );
    if ( !h_rec || !h_edges ) {
      setError( "orientNormals: hipHostMalloc failed" );
      return TMC2_E_HIP;
    }
    TMC2_HIP( hipMemcpyAsync( h_rec, d_rec.p, ( size_t( C ) + 1 ) * sizeof( OrientClusterRec ), hipMemcpyDeviceToHost, s ) );
    if ( E ) TMC2_HIP( hipMemcpyAsync( h_edges, d_edges.p, size_t( E ) * sizeof( OrientCompactEdge ), hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
  }
  if ( ctxOption( f->ctx, "ORIENT_SPEC" ) ) ctx->stageAddHostMs( "orient_compact_edges", double( E ) );  // (test hook: the size of the compact graph)
  g.clusters = C, g.rec = h_rec, g.edges = h_edges;
  ok         = true;
  return TMC2_OK;
}

// sign[v] = clusterSign[cluster of v] * (-1)^parity[v]
int launchClusterSigns( tmc2_frame* f, const uint32_t* d_cid, const uint8_t* d_parity, const int8_t* d_clusterSign, int8_t* d_sign ) {
  const uint32_t n = uint32_t( f->n );
  hipLaunchKernelGGL( compactSignsKernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, f->ctx->stream, d_cid, d_parity, d_clusterSign, n, d_sign );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2
