// patches.hip -- connected-component patch extraction (S7), per-patch depth maps (S8) and the raw-point
// update (S9) on gfx950.
//
// Replaces PCCPatchSegmenter3::segmentPatches (reference: source/lib/PccLibEncoder/source/
// PCCPatchSegmenter.cpp:542-1320, CTC branch: no EOM / patch expansion / partitioning / gradient
// separation) together with resampledPointcloud (:362-470) and the per-round k-d tree + 1-NN (:1291-1298).
//
// S7  The reference floods components sequentially (LIFO over the DIRECTED 16-NN lists, same plane,
//     seeds in increasing index among raw points farther than 9 from the resampled cloud).  The set a
//     seed absorbs is exactly { v : s is the SMALLEST eligible seed that reaches v }, so
//     the components are the fixpoint of label[v] = min(label[v], label[u]) over edges u->v; patch
//     order = increasing label.  Push-style atomicMin sweeps; several sweeps per host round-trip.
// S8  All per-patch quantities are min/max/count reductions (inputs have no duplicate positions), so
//     they are atomics: bbox, then a 64-bit atomicMin/Max of (depth << 32 | point id) per pixel gives D0
//     and its source point in one pass.  Depth maps are processed as 16x16 tiles = one 256-lane
//     workgroup per occupancy block: block peak by wave reduction, outlier filter, D1 seeding, block
//     occupancy by ballot, conversion to patch-local depth.
// S9  Only thresholds of the distance to the resampled cloud matter (> 9 seeds, > 1 stays raw), so the
//     per-round k-d tree is replaced by a dense 3-D occupancy bitmap of the resampled voxels in HBM
//     (2^30 bits = 128 MiB at vox10) probed in increasing-distance order.
#include <algorithm>
#include <cmath>

#include "internal.h"

namespace tmc2 {
namespace {

constexpr uint32_t kNoLabel = 0xFFFFFFFFu;
constexpr uint32_t kFar     = 0xFFFFFFFFu;  // "distance to the resampled cloud unknown / beyond the probe radius"

struct PatchDev {
  int32_t u1, v1, d1;
  int32_t sizeU, sizeV, sizeU0, sizeV0;
  int32_t axN, axT, axB, mode;
  int32_t blockBase;  // first tile of this patch in the round's tile list
  int64_t depthOff;   // into the frame's depth pools
  int64_t occOff;     // into the frame's occupancy pool
};

__device__ __forceinline__ int coordOf( const Pt p, int axis ) { return axis == 0 ? p.x : ( axis == 1 ? p.y : p.z ); }

// Wave-aggregated reductions keyed by a small integer (patch / component id).  Neighbouring points almost always
// share the key, so instead of 64 atomics on one address per wavefront the lanes holding the same key are reduced
// with cross-lane shuffles and ONE lane issues the atomics.  Loops once per distinct key present in the wave.
// All 64 lanes must call these (key < 0 = nothing to contribute).
// A body-sized patch still receives one report per wave (thousands per address): look before the atomic -- the running
// value is monotone, so a stale read can only cause a redundant atomic, never a missed one.
__device__ __forceinline__ void lazyAtomicMin( int32_t* a, int v ) {
  if ( v < loadStaleOk( a ) ) atomicMin( a, v );
}
__device__ __forceinline__ void lazyAtomicMax( int32_t* a, int v ) {
  if ( v > loadStaleOk( a ) ) atomicMax( a, v );
}
__device__ __forceinline__ int waveMinMasked( int v, bool mine ) {
  v = mine ? v : 0x7FFFFFFF;
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) v = min( v, __shfl_xor( v, off, 64 ) );
  return v;
}
__device__ __forceinline__ int waveMaxMasked( int v, bool mine ) {
  v = mine ? v : int( 0x80000000 );
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) v = max( v, __shfl_xor( v, off, 64 ) );
  return v;
}

// ---- S7 ---------------------------------------------------------------------------------------------
// Connected components of the reference = "smallest eligible seed that reaches v" over the DIRECTED k-NN graph
// restricted to raw points of one plane.  Pushing labels along edges needs as many dependent steps as
// the graph is deep (hundreds of hops on a body-sized patch).  Instead:
//   1. mutual edges (u in knn[v] and v in knn[u]) are bidirectional, so everything they connect is reached by
//      exactly the same seeds: a lock-free union-find over the mutual edges (hooking with atomicCAS under the root of
//      smaller HASHED priority -- hooking by index would grow chains as long as the scan order of the cloud -- and
//      path halving) collapses each such group in O(log) dependent steps;
//   2. lab[root] = smallest ELIGIBLE member (atomicMin);
//   3. the remaining one-way edges connect groups; lab[] is relaxed along them until nothing changes -- on the
//      condensed graph that is a handful of sweeps, each touching only the one-way edges.
// The result is the unique fixpoint, so it does not depend on scheduling.

// bit j of mutual[u] = knn[u][j] lists u in its own row.  Depends only on the adjacency: once per call.
// Every row is read by the point itself and by its (up to) sixteen in-neighbours: 64 N bytes if each row reached HBM once, sixteen
// times that if none stayed cached.  Rounds 1-5 dealt the blocks round-robin over the eight XCDs: every L2 saw every region of the
// cloud (counters: 325 MB for 55 MB of contract bytes).  Round 6: XCD x works through the x-th eighth of the blocks (the mapping of
// knnKernel) -- the rows a workgroup needs are the rows its neighbours on the same L2 have just fetched (85 MB); perm != nullptr:
// the points in TREE order instead of input order (no faster on clouds that arrive in scan order: option MUTUAL_ORDER=tree).
template <int K>
__global__ __launch_bounds__( 256 ) void ccMutualMaskKernel( const uint32_t* __restrict__ knn, const uint32_t* __restrict__ perm, bool chunked,
                                                              uint32_t n, uint16_t* __restrict__ mutual ) {
  uint32_t block = blockIdx.x;
  if ( chunked ) {  // (grid: a multiple of 8 blocks; block b runs on XCD b % 8 -- observed, not promised: only speed depends on it)
    const uint32_t perXcd = gridDim.x >> 3;
    block                 = ( blockIdx.x & 7u ) * perXcd + ( blockIdx.x >> 3 );
  }
  const uint32_t at = block * blockDim.x + threadIdx.x;
  if ( at >= n ) return;
  const uint32_t u = perm ? perm[at] : at;
  uint32_t     nb[K];
  const uint4* row = reinterpret_cast<const uint4*>( knn + size_t( u ) * K );
#pragma unroll
  for ( int j = 0; j < K / 4; ++j ) {
    const uint4 r = row[j];
    nb[4 * j] = r.x, nb[4 * j + 1] = r.y, nb[4 * j + 2] = r.z, nb[4 * j + 3] = r.w;
  }
  uint32_t m = 0;
#pragma unroll
  for ( int j = 0; j < K; ++j ) {
    const uint32_t v = nb[j];
    if ( v == u ) continue;
    const uint4* rv  = reinterpret_cast<const uint4*>( knn + size_t( v ) * K );
    bool         hit = false;
#pragma unroll
    for ( int t = 0; t < K / 4; ++t ) {
      const uint4 r = rv[t];
      hit |= ( r.x == u ) | ( r.y == u ) | ( r.z == u ) | ( r.w == u );
    }
    m |= hit ? ( 1u << j ) : 0u;
  }
  mutual[u] = uint16_t( m );
}

__device__ __forceinline__ uint32_t ufPriority( uint32_t x ) { return x * 2654435761u; }  // odd multiplier: a bijection

// The point of this lane (n: none).  chunked: XCD x works through the x-th eighth of the blocks (block b runs on XCD b % 8 -- observed,
// not promised: only speed depends on it; the grid is a multiple of 8 blocks); perm: the points in tree order instead of input order.
__device__ __forceinline__ uint32_t pointOfLane( const uint32_t* __restrict__ perm, bool chunked, uint32_t n ) {
  uint32_t block = blockIdx.x;
  if ( chunked ) block = ( blockIdx.x & 7u ) * ( gridDim.x >> 3 ) + ( blockIdx.x >> 3 );
  const uint32_t at = block * blockDim.x + threadIdx.x;
  return at < n ? ( perm ? perm[at] : at ) : n;
}

// Initial forest without atomics: every raw point hooks itself under the eligible mutual neighbour of smallest hashed
// priority, if smaller than its own (priorities strictly decrease along parent links: acyclic).  Most unions are done
// before the first compare-and-swap, and the paths the union pass walks end at local priority minima.
template <int K>
__global__ __launch_bounds__( 256 ) void ccInitKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ mutual,
                                                        const uint8_t* __restrict__ partition, const uint8_t* __restrict__ raw,
                                                        const uint32_t* __restrict__ perm, bool chunked, uint32_t n,
                                                        uint32_t* __restrict__ parent, uint32_t* __restrict__ lab,
                                                        uint32_t* __restrict__ ccCount ) {
  const uint32_t i = pointOfLane( perm, chunked, n );
  if ( i >= n ) return;
  lab[i]        = kNoLabel;
  ccCount[i]    = 0;
  uint32_t best = i;
  if ( raw[i] ) {
    uint32_t      bestPrio = ufPriority( i ), m = mutual[i];
    const uint8_t pi       = partition[i];
    while ( m ) {
      const int j = __ffs( int( m ) ) - 1;
      m &= m - 1;
      const uint32_t v = knn[size_t( i ) * K + j];
      if ( ufPriority( v ) < bestPrio && raw[v] && partition[v] == pi ) {
        best     = v;
        bestPrio = ufPriority( v );
      }
    }
  }
  parent[i] = best;
}

// a parent always has a smaller priority than its child, so the forest stays acyclic and a stale read during find
// is still an ancestor-or-self of the truth
// The climb reads through the XCD's L2 (workgroup-scope loads: the view of this XCD, possibly behind the other seven -- still
// ancestors); only the last step, "is this really a root", goes to the coherent level and climbs on from there if it is not.
// An agent-scope load per hop is a trip past the L2 for every link of every path.
__device__ __forceinline__ uint32_t ufFind( uint32_t* parent, uint32_t x, bool agent ) {
  uint32_t p = loadStaleOk( &parent[x], agent );
  while ( p != x ) {
    const uint32_t g = loadStaleOk( &parent[p], agent );
    if ( g != p ) __hip_atomic_store( &parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );  // path halving
    x = p;
    p = g;
  }
  for ( ;; ) {
    const uint32_t q = __hip_atomic_load( &parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if ( q == x ) return x;
    x = q;
  }
}

// "Are a and b in one set already?" answered from this CU's possibly stale view, without a store or an atomic: the two
// climbs advance the end of LARGER priority (priorities fall strictly along every link ever written, so the end of smaller
// priority cannot lie below the other one's path) and meet at a common ancestor if the view has one.  true is final -- every
// word ever stored in parent[] links two members of one set and sets only grow; false only means "not known here", and the
// caller goes on to the coherent find / compare-and-swap loop.  Most mutual edges join points that ccInitKernel (or an
// earlier union) has put into one tree already: they end here, a few L1 / L2 hits each.
__device__ __forceinline__ bool ufSameSetStale( const uint32_t* parent, uint32_t a, uint32_t b, bool agent ) {
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
    const uint32_t up = loadStaleOk( &parent[a], agent );
    if ( up == a ) return false;  // a root of larger priority than b: nothing above it in this view
    a  = up;
    pa = ufPriority( a );
  }
}

template <int K>
__global__ __launch_bounds__( 256 ) void ccUnionKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ mutual,
                                                         const uint8_t* __restrict__ partition,
                                                         const uint8_t* __restrict__ raw, const uint32_t* __restrict__ perm,
                                                         bool chunked, uint32_t n, uint32_t* __restrict__ parent, int precheck,
                                                         bool agent ) {
  const uint32_t u = pointOfLane( perm, chunked, n );
  if ( u >= n || !raw[u] ) return;
  uint32_t m = mutual[u];
  if ( !m ) return;
  const uint8_t   pu  = partition[u];
  const uint32_t* row = knn + size_t( u ) * K;
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = row[j];
    if ( v > u || !raw[v] || partition[v] != pu ) continue;  // every mutual edge is seen from both ends: larger one acts
    if ( precheck && ufSameSetStale( parent, u, v, agent ) ) continue;
    uint32_t a = u, b = v;
    while ( true ) {
      a = ufFind( parent, a, agent );
      b = ufFind( parent, b, agent );
      if ( a == b ) break;
      if ( ufPriority( a ) < ufPriority( b ) ) {
        const uint32_t t = a;
        a                = b;
        b                = t;
      }
      if ( atomicCAS( &parent[a], a, b ) == a ) break;  // hook the root of larger priority under the other
    }
  }
}

// Debug invariants of the settled forest (TMC2_UF_CHECK=1, the soak tests): every link goes to a raw point of the same plane
// and of smaller priority; the two ends of every eligible mutual edge have one root.  Climbs at agent scope only (the
// coherent truth, no stale view involved).  bad[0] = broken links, bad[1] = edges whose ends ended up in different sets.
template <int K>
__global__ __launch_bounds__( 256 ) void ccCheckKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ mutual,
                                                         const uint8_t* __restrict__ partition, const uint8_t* __restrict__ raw,
                                                         uint32_t n, uint32_t* parent, uint32_t* __restrict__ bad ) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if ( u >= n || !raw[u] ) return;
  auto rootOf = [&]( uint32_t x ) {
    for ( uint32_t hops = 0; hops <= n; ++hops ) {
      const uint32_t q = __hip_atomic_load( &parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( q == x ) return x;
      if ( q >= n || ufPriority( q ) >= ufPriority( x ) ) return 0xFFFFFFFFu;
      x = q;
    }
    return 0xFFFFFFFFu;
  };
  const uint32_t p = __hip_atomic_load( &parent[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  if ( p != u && ( p >= n || !raw[p] || partition[p] != partition[u] || ufPriority( p ) >= ufPriority( u ) ) ) atomicAdd( &bad[0], 1u );
  const uint32_t ru = rootOf( u );
  if ( ru == 0xFFFFFFFFu ) {
    atomicAdd( &bad[0], 1u );
    return;
  }
  uint32_t m = mutual[u];
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = knn[size_t( u ) * K + j];
    if ( v > u || !raw[v] || partition[v] != partition[u] ) continue;
    if ( rootOf( v ) != ru ) atomicAdd( &bad[1], 1u );
  }
}

__global__ __launch_bounds__( 256 ) void ccFlattenSeedKernel( const uint8_t* __restrict__ raw, const uint32_t* __restrict__ dist,
                                                               uint32_t thrDetection, uint32_t n,
                                                               uint32_t* __restrict__ parent, uint32_t* __restrict__ root,
                                                               uint32_t* __restrict__ lab, bool agent ) {
  const uint32_t u    = chunkedIndex();
  const int      lane = threadIdx.x & 63;
  uint32_t       r    = kNoLabel;
  bool           seed = false;
  if ( u < n && raw[u] ) {
    // The flat view goes to its OWN array.  Writing r into parent[u] raced with the path-halving stores of the finds that
    // pass through u at the same moment: a halving store issued from a stale view (parent[u] = some ancestor) could land after
    // this one and leave parent[u] short of the root -- and the kernels below read the array as flat.  Rare while the union
    // pass had compressed nearly every path, frequent once most edges end in the store-free pre-check.
    r       = ufFind( parent, u, agent );
    root[u] = r;
    seed    = dist[u] > thrDetection;
  }
  // a body-sized group has hundreds of thousands of members: one atomic per wave and group, and only if it can lower
  unsigned long long todo = __ballot( seed );
  while ( todo ) {
    const int                leader = __ffsll( (long long)todo ) - 1;
    const uint32_t           key    = __shfl( r, leader, 64 );
    const unsigned long long same   = __ballot( seed && r == key );
    // lanes are in index order, so the leader (lowest lane of its group) holds the group's smallest index in the wave
    if ( lane == leader && loadStaleOk( &lab[key], agent ) > u )
      atomicMin( &lab[key], u );
    todo &= ~same;
  }
}

// one relaxation sweep over the one-way edges between groups (parent = the flat roots ccFlattenSeedKernel wrote)
template <int K>
__global__ __launch_bounds__( 256 ) void ccRelaxKernel( const uint32_t* __restrict__ knn, const uint16_t* __restrict__ mutual,
                                                         const uint8_t* __restrict__ partition,
                                                         const uint8_t* __restrict__ raw, const uint32_t* __restrict__ parent,
                                                         const uint32_t* __restrict__ perm, bool chunked, uint32_t n,
                                                         uint32_t* __restrict__ lab, uint32_t* __restrict__ changed, uint32_t token,
                                                         bool agent ) {
  const uint32_t u = pointOfLane( perm, chunked, n );
  if ( u >= n || !raw[u] ) return;
  uint32_t m = ~uint32_t( mutual[u] ) & ( ( 1u << K ) - 1u );
  if ( !m ) return;
  const uint32_t ru = parent[u];
  const uint32_t lu = loadStaleOk( &lab[ru], agent );
  if ( lu == kNoLabel ) return;
  const uint8_t   pu  = partition[u];
  const uint32_t* row = knn + size_t( u ) * K;
  bool            any = false;
  while ( m ) {
    const int j = __ffs( int( m ) ) - 1;
    m &= m - 1;
    const uint32_t v = row[j];
    if ( v == u || !raw[v] || partition[v] != pu ) continue;
    const uint32_t rv = parent[v];
    if ( rv != ru && loadStaleOk( &lab[rv], agent ) > lu &&
         atomicMin( &lab[rv], lu ) > lu )
      any = true;
  }
  if ( any ) *changed = token;  // (which sweep changed something last: nothing to clear between sweeps)
}

constexpr uint32_t kRawCounters = 16;  // this round's count of points still raw: sixteen words 128 bytes apart, one add per workgroup
// start of a round's per-patch accumulators: bounding box {min 3 x INT_MAX, max 3 x 0 -- the reference starts its max at 0},
// minimum (u, v), the two resampling counters; and this round's count of points still raw
__global__ __launch_bounds__( 256 ) void patchBoundsInitKernel( uint32_t P, int32_t* __restrict__ bbox, int32_t* __restrict__ minUv,
                                                                 int32_t* __restrict__ patchStat, uint32_t* __restrict__ rawCount ) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if ( p < kRawCounters ) rawCount[p * 32] = 0;  // (the first workgroup always exists: P >= 1)
  if ( p >= P ) return;
#pragma unroll
  for ( int d = 0; d < 3; ++d ) {
    bbox[6 * p + d]     = 0x7FFFFFFF;
    bbox[6 * p + 3 + d] = 0;
  }
  minUv[2 * p] = minUv[2 * p + 1] = 0x7F7F7F7F;
  patchStat[2 * p] = patchStat[2 * p + 1] = 0;
}

// label of every point still raw (the label of its group), and the size of every component (one atomic per run of equal
// labels in a wavefront)
// (round 6: the sizes are only ever compared with minCount -- ccSeedFlagKernel, ccAssignKernel -- so a component that has reached it
//  stops counting: the body of a figure is ONE component of 10^5 points, and 13 000 wavefronts adding to its one word queued up for
//  ~ 11 ns each; the look is a possibly stale view of a counter that only grows -- a stale read costs an add, never a wrong answer)
__global__ __launch_bounds__( 256 ) void ccLabelCountKernel( const uint8_t* __restrict__ raw, const uint32_t* __restrict__ parent,
                                                              const uint32_t* __restrict__ lab, uint32_t n, uint32_t minCount,
                                                              uint32_t* __restrict__ label, uint32_t* __restrict__ ccCount ) {
  const uint32_t i    = chunkedIndex();
  const int      lane = threadIdx.x & 63;
  const uint32_t l    = ( i < n && raw[i] ) ? lab[parent[i]] : kNoLabel;
  if ( i < n ) label[i] = l;
  unsigned long long todo = __ballot( l != kNoLabel );
  while ( todo ) {
    const int                leader = __ffsll( (long long)todo ) - 1;
    const uint32_t           key    = __shfl( l, leader, 64 );
    const unsigned long long same   = __ballot( l == key );
    if ( lane == leader && loadStaleOk( &ccCount[key] ) < minCount ) atomicAdd( &ccCount[key], uint32_t( __popcll( same ) ) );
    todo &= ~same;
  }
}

__global__ __launch_bounds__( 256 ) void ccSeedFlagKernel( const uint32_t* __restrict__ label,
                                                            const uint32_t* __restrict__ ccCount, uint32_t minCount,
                                                            uint32_t n, uint32_t* __restrict__ flag ) {
  const uint32_t i = chunkedIndex();
  if ( i < n ) flag[i] = ( label[i] == i && ccCount[i] >= minCount ) ? 1u : 0u;
}

// per point: patch (this round's rank of its component) or -1; per patch: plane of the seed
__global__ __launch_bounds__( 256 ) void ccAssignKernel( const uint32_t* __restrict__ label,
                                                          const uint32_t* __restrict__ ccCount,
                                                          const uint32_t* __restrict__ rank,
                                                          const uint8_t* __restrict__ partition, uint32_t minCount,
                                                          uint32_t n, int32_t* __restrict__ pointPatch,
                                                          int32_t* __restrict__ patchView ) {
  const uint32_t i = chunkedIndex();
  if ( i >= n ) return;
  const uint32_t l = label[i];
  int32_t        p = -1;
  if ( l != kNoLabel && ccCount[l] >= minCount ) {
    p = int32_t( rank[l] );
    if ( l == i ) patchView[p] = partition[i];
  }
  pointPatch[i] = p;
}

// ---- S8 reductions ------------------------------------------------------------------------------------
// stats[p] = { minU, minV } (splitting window anchor)
// Per-patch minima / maxima over the points of a workgroup, in LDS first: a small open-addressing table keyed by patch (LDS
// atomics cost next to nothing and meet no other workgroup), then one look-before-you-atomic per table entry and field.
// Points come in input order -- a wave's 64 points belong to up to dozens of patches -- so reducing patch by patch inside the
// wave (a masked butterfly per field and distinct patch) was 0.13 - 0.3 ms per launch; per-point global atomics would meet on
// the handful of words of a big patch.  A point whose patch finds no slot (more than kAggSlots distinct patches in one
// workgroup) reports to global memory itself.
constexpr int kAggSlots = 128;
template <int FIELDS>
struct PatchAgg {
  int32_t key[kAggSlots];
  int32_t val[kAggSlots][FIELDS];
};
// MINS: the first MINS fields are minima, the others maxima
template <int FIELDS, int MINS>
__device__ __forceinline__ void aggInit( PatchAgg<FIELDS>& t ) {
  for ( int i = threadIdx.x; i < kAggSlots; i += blockDim.x ) {
    t.key[i] = -1;
#pragma unroll
    for ( int k = 0; k < FIELDS; ++k ) t.val[i][k] = k < MINS ? 0x7FFFFFFF : int32_t( 0x80000000 );
  }
  __syncthreads();
}
template <int FIELDS, int MINS>
__device__ __forceinline__ void aggAdd( PatchAgg<FIELDS>& t, int32_t patch, const int ( &v )[FIELDS], int32_t* __restrict__ out ) {
  int slot = int( ( uint32_t( patch ) * 2654435761u ) >> 25 );  // (kAggSlots = 2^7)
  for ( int tries = 0; tries < kAggSlots; ++tries, slot = ( slot + 1 ) & ( kAggSlots - 1 ) ) {
    const int32_t k = atomicCAS( &t.key[slot], -1, patch );
    if ( k == -1 || k == patch ) {
#pragma unroll
      for ( int f = 0; f < FIELDS; ++f ) {
        if ( f < MINS )
          atomicMin( &t.val[slot][f], v[f] );
        else
          atomicMax( &t.val[slot][f], v[f] );
      }
      return;
    }
  }
#pragma unroll
  for ( int f = 0; f < FIELDS; ++f ) {  // no slot left
    if ( f < MINS )
      lazyAtomicMin( &out[FIELDS * patch + f], v[f] );
    else
      lazyAtomicMax( &out[FIELDS * patch + f], v[f] );
  }
}
template <int FIELDS, int MINS>
__device__ __forceinline__ void aggFlush( PatchAgg<FIELDS>& t, int32_t* __restrict__ out ) {
  __syncthreads();
  for ( int i = threadIdx.x; i < kAggSlots * FIELDS; i += blockDim.x ) {
    const int     slot = i / FIELDS, f = i % FIELDS;
    const int32_t k    = t.key[slot];
    if ( k < 0 ) continue;
    if ( f < MINS )
      lazyAtomicMin( &out[FIELDS * k + f], t.val[slot][f] );
    else
      lazyAtomicMax( &out[FIELDS * k + f], t.val[slot][f] );
  }
}

__global__ __launch_bounds__( 256 ) void patchMinUvKernel( const Pt* __restrict__ pts, const int32_t* __restrict__ pointPatch,
                                                            const int32_t* __restrict__ patchView, uint32_t n,
                                                            int32_t* __restrict__ minUv ) {
  __shared__ PatchAgg<2> agg;
  aggInit<2, 2>( agg );
  const uint32_t i = chunkedIndex();
  const int32_t  p = i < n ? pointPatch[i] : -1;
  if ( p >= 0 ) {
    const int view = patchView[p] % 3;
    const int axT = view == 0 ? 2 : ( view == 1 ? 2 : 0 ), axB = view == 0 ? 1 : ( view == 1 ? 0 : 1 );
    const Pt  q    = pts[i];
    const int uv[2] = {coordOf( q, axT ), coordOf( q, axB )};
    aggAdd<2, 2>( agg, p, uv, minUv );
  }
  aggFlush<2, 2>( agg, minUv );
}

__global__ __launch_bounds__( 256 ) void patchTrimBboxKernel( const Pt* __restrict__ pts, const int32_t* __restrict__ patchView,
                                                               const int32_t* __restrict__ minUv, int splitting,
                                                               int maxPatchSize, uint32_t n,
                                                               int32_t* __restrict__ pointPatch, int32_t* __restrict__ bbox ) {
  __shared__ PatchAgg<6> agg;
  aggInit<6, 3>( agg );
  const uint32_t i = chunkedIndex();
  int32_t        p = i < n ? pointPatch[i] : -1;
  if ( p >= 0 ) {
    const Pt q = pts[i];
    if ( splitting ) {
      const int view = patchView[p] % 3;
      const int axT = view == 0 ? 2 : ( view == 1 ? 2 : 0 ), axB = view == 0 ? 1 : ( view == 1 ? 0 : 1 );
      if ( !( coordOf( q, axT ) - minUv[2 * p] < maxPatchSize && coordOf( q, axB ) - minUv[2 * p + 1] < maxPatchSize ) ) {
        pointPatch[i] = -1;  // trimmed: stays raw for a later round
        p             = -1;
      }
    }
    if ( p >= 0 ) {
      const int box[6] = {q.x, q.y, q.z, q.x, q.y, q.z};
      aggAdd<6, 3>( agg, p, box, bbox );
    }
  }
  aggFlush<6, 3>( agg, bbox );
}

// D0 candidates: 64-bit (depth << 32 | point) min (mode 0) / max (mode 1) per pixel
__global__ __launch_bounds__( 256 ) void patchDepth0Kernel( const Pt* __restrict__ pts, const int32_t* __restrict__ pointPatch,
                                                             const PatchDev* __restrict__ patches, uint32_t n,
                                                             int64_t roundDepthBase,
                                                             unsigned long long* __restrict__ map64 ) {
  const uint32_t i = chunkedIndex();
  if ( i >= n ) return;
  const int32_t p = pointPatch[i];
  if ( p < 0 ) return;
  const PatchDev pd = patches[p];
  const Pt       q  = pts[i];
  const int      d = coordOf( q, pd.axN ), u = coordOf( q, pd.axT ) - pd.u1, v = coordOf( q, pd.axB ) - pd.v1;
  const size_t   px  = size_t( pd.depthOff - roundDepthBase ) + size_t( v ) * pd.sizeU + u;
  const unsigned long long val = ( (unsigned long long)uint32_t( d + 1 ) << 32 ) | i;  // +1: 0 stays a sentinel
  if ( pd.mode == 0 )
    atomicMin( &map64[px], val );
  else
    atomicMax( &map64[px], val );
}

// per tile: reset the 64-bit D0 candidates to the projection mode's sentinel
__global__ __launch_bounds__( 256 ) void patchInitTileKernel( const PatchDev* __restrict__ patches,
                                                               const uint32_t* __restrict__ tilePatch,
                                                               int64_t roundDepthBase, int occRes,
                                                               unsigned long long* __restrict__ map64 ) {
  const uint32_t tile  = blockIdx.x;
  const PatchDev pd    = patches[tilePatch[tile]];
  const int      local = int( tile ) - pd.blockBase;
  const int      u = ( local % pd.sizeU0 ) * occRes + int( threadIdx.x ) % occRes;
  const int      v = ( local / pd.sizeU0 ) * occRes + int( threadIdx.x ) / occRes;
  if ( u < pd.sizeU && v < pd.sizeV )
    map64[size_t( pd.depthOff - roundDepthBase ) + size_t( v ) * pd.sizeU + u] = pd.mode == 0 ? ~0ull : 0ull;
}

// one 16x16 tile (occupancy block) per workgroup: block peak, outlier filter, D0/D1 seed, source point id
__global__ __launch_bounds__( 256 ) void patchFilterTileKernel( const PatchDev* __restrict__ patches,
                                                                 const uint32_t* __restrict__ tilePatch,
                                                                 const unsigned long long* __restrict__ map64,
                                                                 int64_t roundDepthBase, int occRes, int surfaceThickness,
                                                                 int maxAllowedDepth, int32_t* __restrict__ d0tmp,
                                                                 int32_t* __restrict__ d1tmp, uint32_t* __restrict__ d0src ) {
  __shared__ int  wavePeak[4];
  const uint32_t  tile = blockIdx.x;
  const uint32_t  p    = tilePatch[tile];
  const PatchDev  pd   = patches[p];
  const int       local = int( tile ) - pd.blockBase;
  const int       bu = local % pd.sizeU0, bv = local / pd.sizeU0;
  const int       u = bu * occRes + int( threadIdx.x ) % occRes, v = bv * occRes + int( threadIdx.x ) / occRes;
  const bool      inside = u < pd.sizeU && v < pd.sizeV;
  const size_t    px     = size_t( pd.depthOff - roundDepthBase ) + size_t( v ) * pd.sizeU + u;
  const int       INF    = 32767;
  int             depth  = INF;
  uint32_t        src    = 0xFFFFFFFFu;
  if ( inside ) {
    const unsigned long long m = map64[px];
    const bool               empty = pd.mode == 0 ? ( m == ~0ull ) : ( m == 0ull );
    if ( !empty ) {
      depth = int( m >> 32 ) - 1;
      src   = uint32_t( m );
    }
  }
  // block peak: min depth (mode 0) or max depth (mode 1) over valid pixels
  int pk = depth == INF ? ( pd.mode == 0 ? INF : 0 ) : depth;
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) {
    const int o = __shfl_xor( pk, off, 64 );
    pk          = pd.mode == 0 ? min( pk, o ) : max( pk, o );
  }
  if ( ( threadIdx.x & 63 ) == 0 ) wavePeak[threadIdx.x >> 6] = pk;
  __syncthreads();
  pk = wavePeak[0];
#pragma unroll
  for ( int w = 1; w < 4; ++w ) pk = pd.mode == 0 ? min( pk, wavePeak[w] ) : max( pk, wavePeak[w] );
  if ( !inside ) return;
  if ( depth != INF ) {
    const int     dir = 1 - 2 * pd.mode;
    const int16_t a   = int16_t( abs( depth - pk ) );
    const int16_t b   = int16_t( int16_t( surfaceThickness ) + dir * depth );
    const int16_t c   = int16_t( dir * pd.d1 + int16_t( maxAllowedDepth ) );
    if ( a > 32 || b > c ) {
      depth = INF;
      src   = 0xFFFFFFFFu;
    }
  }
  d0tmp[px] = depth;
  d1tmp[px] = depth;
  d0src[px] = src;
}

// D1: farthest same-pixel depth within surfaceThickness of D0, colour-similar to the D0 point
__global__ __launch_bounds__( 256 ) void patchDepth1Kernel( const Pt* __restrict__ pts, const uint8_t* __restrict__ rgb4,
                                                             const int32_t* __restrict__ pointPatch,
                                                             const PatchDev* __restrict__ patches, uint32_t n,
                                                             int64_t roundDepthBase, int surfaceThickness,
                                                             const int32_t* __restrict__ d0tmp,
                                                             const uint32_t* __restrict__ d0src, int32_t* __restrict__ d1tmp ) {
  const uint32_t i = chunkedIndex();
  if ( i >= n ) return;
  const int32_t p = pointPatch[i];
  if ( p < 0 ) return;
  const PatchDev pd = patches[p];
  const Pt       q  = pts[i];
  const int      d = coordOf( q, pd.axN ), u = coordOf( q, pd.axT ) - pd.u1, v = coordOf( q, pd.axB ) - pd.v1;
  const size_t   px = size_t( pd.depthOff - roundDepthBase ) + size_t( v ) * pd.sizeU + u;
  const int      z0 = d0tmp[px];
  if ( !( z0 < 32767 ) ) return;
  const int     dir   = 1 - 2 * pd.mode;
  const int16_t delta = int16_t( dir * ( d - z0 ) );
  if ( !( delta <= int16_t( surfaceThickness ) && delta >= 0 ) ) return;
  const uchar4 ci = reinterpret_cast<const uchar4*>( rgb4 )[i];
  const uchar4 c0 = reinterpret_cast<const uchar4*>( rgb4 )[d0src[px]];
  if ( !( abs( int( c0.x ) - int( ci.x ) ) < 128 && abs( int( c0.y ) - int( ci.y ) ) < 128 &&
          abs( int( c0.z ) - int( ci.z ) ) < 128 ) )
    return;
  if ( pd.mode == 0 )
    atomicMax( &d1tmp[px], d );
  else
    atomicMin( &d1tmp[px], d );
}

// per tile: block occupancy, patch-local depths into the frame pools, resampled voxels into the bitmap,
// per-patch sizeD (max local depth) and d0Count
__global__ __launch_bounds__( 256 ) void patchResampleTileKernel( const PatchDev* __restrict__ patches,
                                                                   const uint32_t* __restrict__ tilePatch,
                                                                   int64_t roundDepthBase, int occRes,
                                                                   const int32_t* __restrict__ d0tmp,
                                                                   const int32_t* __restrict__ d1tmp, int bitmapBits,
                                                                   uint32_t* __restrict__ bitmap,
                                                                   int16_t* __restrict__ depth0, int16_t* __restrict__ depth1,
                                                                   uint8_t* __restrict__ occupancy,
                                                                   int32_t* __restrict__ patchStat /* [p][2] sizeD, d0Count */ ) {
  __shared__ int  anyValid[4], tileMax[4];
  const uint32_t  tile = blockIdx.x;
  const uint32_t  p    = tilePatch[tile];
  const PatchDev  pd   = patches[p];
  const int       local = int( tile ) - pd.blockBase;
  const int       bu = local % pd.sizeU0, bv = local / pd.sizeU0;
  const int       u = bu * occRes + int( threadIdx.x ) % occRes, v = bv * occRes + int( threadIdx.x ) / occRes;
  const bool      inside = u < pd.sizeU && v < pd.sizeV;
  bool            valid  = false;
  int             l0 = 0, l1 = 0;
  if ( inside ) {
    const size_t px = size_t( pd.depthOff - roundDepthBase ) + size_t( v ) * pd.sizeU + u;
    const int    z0 = d0tmp[px], z1 = d1tmp[px];
    int16_t      o0 = 32767, o1 = 32767;
    if ( z0 < 32767 ) {
      valid         = true;
      const int dir = 1 - 2 * pd.mode;
      l0            = dir * ( z0 - pd.d1 );
      l1            = dir * ( z1 - pd.d1 );
      o0            = int16_t( l0 );
      o1            = int16_t( l1 );
      // resampled points (D0 and D1) -> bitmap
      int c[3];
      c[pd.axT] = u + pd.u1;
      c[pd.axB] = v + pd.v1;
      c[pd.axN] = z0;
      size_t bit = size_t( c[0] ) + ( size_t( c[1] ) << bitmapBits ) + ( size_t( c[2] ) << ( 2 * bitmapBits ) );
      atomicOr( &bitmap[bit >> 5], 1u << ( bit & 31 ) );
      if ( z1 != z0 ) {
        c[pd.axN] = z1;
        bit       = size_t( c[0] ) + ( size_t( c[1] ) << bitmapBits ) + ( size_t( c[2] ) << ( 2 * bitmapBits ) );
        atomicOr( &bitmap[bit >> 5], 1u << ( bit & 31 ) );
      }
    }
    depth0[size_t( pd.depthOff ) + size_t( v ) * pd.sizeU + u] = o0;
    depth1[size_t( pd.depthOff ) + size_t( v ) * pd.sizeU + u] = o1;
  }
  // reductions over the tile
  const unsigned long long m  = __ballot( valid );
  int                      mx = valid ? max( l0, l1 ) : 0;
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) mx = max( mx, __shfl_xor( mx, off, 64 ) );
  // (one report per TILE -- rounds 1-5: one per wavefront -- and the maximum only if it can raise the patch's: a big patch is
  //  thousands of tiles adding to the same two words, ~ 11 ns each, one after the other; the look is a possibly stale view of a
  //  word that only grows)
  if ( ( threadIdx.x & 63 ) == 0 ) {
    anyValid[threadIdx.x >> 6] = __popcll( m );
    tileMax[threadIdx.x >> 6]  = mx;
  }
  __syncthreads();
  if ( threadIdx.x == 0 ) {
    const int count = anyValid[0] + anyValid[1] + anyValid[2] + anyValid[3];
    occupancy[size_t( pd.occOff ) + size_t( bv ) * pd.sizeU0 + bu] = count ? 1 : 0;
    if ( count ) {
      const int tmx = max( max( tileMax[0], tileMax[1] ), max( tileMax[2], tileMax[3] ) );
      if ( tmx > loadStaleOk( &patchStat[2 * p] ) ) atomicMax( &patchStat[2 * p], tmx );
      atomicAdd( &patchStat[2 * p + 1], count );
    }
  }
}

// ---- S9 -------------------------------------------------------------------------------------------------
// dist2 of every input point to the resampled cloud, exact up to the probe radius, kFar beyond it
__global__ __launch_bounds__( 256 ) void rawDistanceKernel( const Pt* __restrict__ pts, uint32_t n,
                                                             const uint32_t* __restrict__ bitmap, int bitmapBits,
                                                             const int* __restrict__ offsets, int nOffsets,
                                                             uint32_t thrSelection, uint32_t* __restrict__ dist,
                                                             uint8_t* __restrict__ raw, uint32_t* __restrict__ rawCount ) {
  const uint32_t i = chunkedIndex();
  bool           isRaw = false;
  if ( i < n ) {
    const Pt  q    = pts[i];
    const int size = 1 << bitmapBits;
    uint32_t  best = kFar;
    for ( int o = 0; o < nOffsets; ++o ) {
      const int packed = offsets[o];
      const int dx = ( packed & 0xFF ) - 128, dy = ( ( packed >> 8 ) & 0xFF ) - 128, dz = ( ( packed >> 16 ) & 0xFF ) - 128;
      const int x = q.x + dx, y = q.y + dy, z = q.z + dz;
      if ( x < 0 || y < 0 || z < 0 || x >= size || y >= size || z >= size ) continue;
      const size_t bit = size_t( x ) + ( size_t( y ) << bitmapBits ) + ( size_t( z ) << ( 2 * bitmapBits ) );
      if ( bitmap[bit >> 5] & ( 1u << ( bit & 31 ) ) ) {
        best = uint32_t( packed >> 24 );  // offsets are sorted by d2 and carry it in the top byte
        break;
      }
    }
    dist[i] = best;
    isRaw   = best > thrSelection;
    raw[i]  = isRaw ? 1 : 0;
  }
  // (rounds 1-5: one add per wavefront to ONE word -- 13 000 of them, ~ 11 ns each, in a queue; now folded per workgroup in LDS
  //  and spread over kRawCounters words)
  __shared__ uint32_t wgRaw;
  if ( threadIdx.x == 0 ) wgRaw = 0;
  __syncthreads();
  const unsigned long long m = __ballot( isRaw );
  if ( ( threadIdx.x & 63 ) == 0 && m ) atomicAdd( &wgRaw, uint32_t( __popcll( m ) ) );
  __syncthreads();
  if ( threadIdx.x == 0 && wgRaw ) atomicAdd( &rawCount[( blockIdx.x % kRawCounters ) * 32], wgRaw );
}

}  // namespace

// bit j of mutual[i] = knn[i][j] lists i in its own row.  Depends only on the adjacency: once per frame.
int ensureMutualMask( tmc2_frame* f ) {
  if ( f->haveMutual ) return TMC2_OK;
  const uint32_t n = uint32_t( f->n );
  TMC2_TRY( f->d_mutual.alloc( n ) );
  const int kt = f->ctx->stageBegin( "k:ccMutualMask" );
  // option MUTUAL_ORDER (this pass and S7's union / relaxation passes): "input" = index order, blocks as they come (rounds 1-5);
  // "chunk" = index order, XCD x on the x-th eighth of the blocks; "tree" = tree order, same eighths
  const char*     order   = ctxOption( f->ctx, "MUTUAL_ORDER" );
  const bool      chunked = !( order && order[0] == 'i' );
  const uint32_t* perm    = order && order[0] == 't' && f->haveTree && f->d_perm.p && f->d_perm.count >= n ? f->d_perm.p : nullptr;
  const uint32_t  blocks  = ( n + 255 ) / 256;
  hipLaunchKernelGGL( ccMutualMaskKernel<16>, dim3( chunked ? ( ( blocks + 7 ) & ~7u ) : blocks ), dim3( 256 ), 0, f->ctx->stream, f->d_knn.p,
                      perm, chunked, n, f->d_mutual.p );
  f->ctx->stageEnd( kt );
  TMC2_HIP( hipGetLastError() );
  f->haveMutual = true;
  return TMC2_OK;
}

int segmentPatches( tmc2_frame* f, const tmc2_segmenter_params* sp ) {
  if ( !f->haveKnn || !f->havePartition ) {
    setError( "segmentPatches: adjacency / partition missing" );
    return TMC2_E_STATE;
  }
  if ( f->k != sp->maxNNCountPatchSegmentation || f->k != 16 ) {
    setError( "segmentPatches: maxNNCountPatchSegmentation=%d must equal the resident adjacency k=%d (16)",
              sp->maxNNCountPatchSegmentation, f->k );
    return TMC2_E_UNSUPPORTED;
  }
  if ( f->d_rgb.count == 0 ) {
    setError( "segmentPatches: the frame has no colours (needed by the D1 colour-similarity test)" );
    return TMC2_E_STATE;
  }
  if ( sp->occupancyResolution != 16 ) {
    setError( "segmentPatches: occupancyResolution=%d unsupported (tiles are 16x16)", sp->occupancyResolution );
    return TMC2_E_UNSUPPORTED;
  }
  tmc2_ctx*      ctx = f->ctx;
  hipStream_t    s   = ctx->stream;
  const uint32_t n   = uint32_t( f->n );
  const int      occRes = sp->occupancyResolution;
  const uint32_t thrDet = uint32_t( std::floor( sp->maxAllowedDist2RawPointsDetection ) );
  const uint32_t thrSel = uint32_t( std::floor( sp->maxAllowedDist2RawPointsSelection ) );
  const int      probeR2 = int( std::max( thrDet, thrSel ) );
  if ( probeR2 > 27 ) {
    setError( "segmentPatches: raw-point distance thresholds above 27 unsupported" );
    return TMC2_E_UNSUPPORTED;
  }
  // probe offsets sorted by d2 (d2 in the top byte)
  std::vector<int> offsets;
  {
    int R = 0;
    while ( R * R <= probeR2 ) ++R;
    std::vector<std::pair<int, int>> tmp;
    for ( int dz = -R; dz <= R; ++dz )
      for ( int dy = -R; dy <= R; ++dy )
        for ( int dx = -R; dx <= R; ++dx ) {
          const int d2 = dx * dx + dy * dy + dz * dz;
          if ( d2 <= probeR2 ) tmp.emplace_back( d2, ( dx + 128 ) | ( ( dy + 128 ) << 8 ) | ( ( dz + 128 ) << 16 ) );
        }
    std::sort( tmp.begin(), tmp.end() );
    for ( auto& t : tmp ) offsets.push_back( t.second | ( t.first << 24 ) );
  }
  int bitmapBits = 1;
  while ( ( 1 << bitmapBits ) <= int( f->geoMax ) ) ++bitmapBits;
  const size_t bitmapWords = ( size_t( 1 ) << ( 3 * bitmapBits ) ) >> 5;
  TMC2_TRY( ctx->voxelBitmap.alloc( bitmapWords ) );

  DevBuf<uint32_t> d_label, d_ccCount, d_flag, d_rank, d_dist, d_small, d_tilePatch, d_d0src, d_parent, d_root, d_lab;
  DevBuf<uint8_t>  d_raw;
  DevBuf<int32_t>  d_pointPatch, d_minUv, d_bbox, d_patchStat, d_d0tmp, d_d1tmp;
  DevBuf<PatchDev> d_patches;
  DevBuf<unsigned long long> d_map64;
  TMC2_TRY( d_label.alloc( n ) );
  TMC2_TRY( d_parent.alloc( n ) );
  TMC2_TRY( d_root.alloc( n ) );
  TMC2_TRY( d_lab.alloc( n ) );
  TMC2_TRY( d_ccCount.alloc( n ) );
  TMC2_TRY( d_flag.alloc( n ) );
  TMC2_TRY( d_rank.alloc( n ) );
  TMC2_TRY( d_dist.alloc( n ) );
  TMC2_TRY( d_raw.alloc( n ) );
  TMC2_TRY( d_pointPatch.alloc( n ) );
  TMC2_TRY( d_small.alloc( 16 ) );
  const int* d_offsets = ctx->constTable( ( uint64_t( 0x5339 ) << 32 ) | uint64_t( probeR2 ), offsets );  // (S9's probe offsets: a function of the thresholds)
  if ( !d_offsets ) return TMC2_E_HIP;
  TMC2_TRY( fillRegions( ctx, {{ctx->voxelBitmap.p, bitmapWords * 4, 0},
                               {d_raw.p, n, 1},
                               {d_dist.p, size_t( n ) * 4, 0xFF},
                               {d_small.p, 64, 0}} ) );

  f->patches.clear();
  f->depthCount = 0;
  f->occCount   = 0;
  // (option POINT_CHUNK=0: the one-point-per-lane passes of S7-S9 with the blocks as they come, rounds 1-5; default: XCD x takes the
  //  x-th eighth of the blocks -- chunkedIndex)
  const char* pcOpt = ctxOption( ctx, "POINT_CHUNK" );
  const dim3  blk( 256 ), grdN( chunkedGrid( ( n + 255 ) / 256, !( pcOpt && pcOpt[0] == '0' ) ) );
  uint32_t    rawCount = n, relaxToken = 0;
  int        rounds   = 0;
  TMC2_TRY( ensureMutualMask( f ) );  // usually there already: the orientation (S3) needs the same bits
  DevBuf<uint16_t>& d_mutual   = f->d_mutual;
  const bool        agentScope = unionAgentScope( f->ctx );
  // the union / relaxation passes: option MUTUAL_ORDER as in ensureMutualMask (here the default is "chunk")
  const char*     ccOrder = ctxOption( ctx, "MUTUAL_ORDER" );
  const bool      chunked = !( ccOrder && ccOrder[0] == 'i' );
  const uint32_t* perm    = ccOrder && ccOrder[0] == 't' && f->haveTree && f->d_perm.p && f->d_perm.count >= n ? f->d_perm.p : nullptr;
  const dim3      grdT( chunked ? ( ( grdN.x + 7u ) & ~7u ) : grdN.x );
  while ( rawCount > 0 ) {
    // ---- S7 -----------------------------------------------------------------------------------------
    int sid = ctx->stageBegin( "patches_cc" );
    hipLaunchKernelGGL( ccInitKernel<16>, grdT, blk, 0, s, f->d_knn.p, d_mutual.p, f->d_partition.p, d_raw.p, perm, chunked, n, d_parent.p,
                        d_lab.p, d_ccCount.p );
    {
      const int kt = ctx->stageBegin( "k:ccUnion" );
      hipLaunchKernelGGL( ccUnionKernel<16>, grdT, blk, 0, s, f->d_knn.p, d_mutual.p, f->d_partition.p, d_raw.p, perm, chunked, n,
                          d_parent.p, unionPrecheck( f->ctx ), agentScope );
      ctx->stageEnd( kt );
      if ( unionCheck( f->ctx ) ) {  // debug invariants (soak tests): costs a round trip
        uint32_t bad[2] = {0, 0};
        TMC2_HIP( hipMemsetAsync( d_small.p + 8, 0, 8, s ) );
        hipLaunchKernelGGL( ccCheckKernel<16>, grdN, blk, 0, s, f->d_knn.p, d_mutual.p, f->d_partition.p, d_raw.p, n,
                            d_parent.p, d_small.p + 8 );
        TMC2_HIP( hipMemcpyAsync( bad, d_small.p + 8, 8, hipMemcpyDeviceToHost, s ) );
        TMC2_HIP( hipStreamSynchronize( s ) );
        if ( bad[0] | bad[1] ) {
          setError( "segmentPatches: union-find invariant broken in round %d (%u bad links, %u split edges)", rounds, bad[0], bad[1] );
          return TMC2_E_HIP;
        }
      }
      hipLaunchKernelGGL( ccFlattenSeedKernel, grdN, blk, 0, s, d_raw.p, d_dist.p, thrDet, n, d_parent.p, d_root.p, d_lab.p,
                          agentScope );
    }
    // a few sweeps, then -- speculatively -- the labelling and the seed count, and ONE round trip for both answers: "did the
    // last sweep of the batch still change a label" (then sweep on and label again) and the number of patches
    uint32_t P = 0;
    for ( int guard = 0; guard < 1 << 20; ++guard ) {
      const int kt = ctx->stageBegin( "k:ccRelax" );
      for ( int b = 0; b < 3; ++b )
        hipLaunchKernelGGL( ccRelaxKernel<16>, grdT, blk, 0, s, f->d_knn.p, d_mutual.p, f->d_partition.p, d_raw.p,
                            d_root.p, perm, chunked, n, d_lab.p, d_small.p, ++relaxToken, agentScope );
      ctx->stageEnd( kt );
      hipLaunchKernelGGL( ccLabelCountKernel, grdN, blk, 0, s, d_raw.p, d_root.p, d_lab.p, n,
                          uint32_t( sp->minPointCountPerCCPatchSegmentation ), d_label.p, d_ccCount.p );
      hipLaunchKernelGGL( ccSeedFlagKernel, grdN, blk, 0, s, d_label.p, d_ccCount.p,
                          uint32_t( sp->minPointCountPerCCPatchSegmentation ), n, d_flag.p );
      // (both answers in the context's page-locked line, stored by the scan's last tile: [0] the number of patches, [1] the token of
      //  the last sweep that changed a label -- no copy)
      volatile uint32_t* answer = ctx->answerLine( tmc2_ctx::kAnswerPatchRound );
      TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_rank.p, n, d_small.p + 1, ScanAnswer{answer, d_small.p, 1} ) );
      TMC2_HIP( hipStreamSynchronize( s ) );
      P = answer[0];
      if ( answer[1] != relaxToken ) break;
      // (ccCount is an accumulation: start it over before labelling again)
      TMC2_HIP( hipMemsetAsync( d_ccCount.p, 0, size_t( n ) * 4, s ) );
    }
    ctx->stageEnd( sid );
    if ( P == 0 ) break;
    // ---- S8 -----------------------------------------------------------------------------------------
    sid = ctx->stageBegin( "patches_build" );
    TMC2_TRY( d_minUv.alloc( 2 * size_t( P ) ) );
    TMC2_TRY( d_bbox.alloc( 7 * size_t( P ) ) );           // boxes, then the views: one copy to the host
    const size_t statWords = 2 * size_t( P ) + size_t( kRawCounters ) * 32;  // counters, then the round's counts of points still raw: one copy
    TMC2_TRY( d_patchStat.alloc( statWords ) );
    int32_t*  d_view     = d_bbox.p + 6 * size_t( P );
    uint32_t* d_rawCount = reinterpret_cast<uint32_t*>( d_patchStat.p + 2 * size_t( P ) );
    TMC2_TRY( d_patches.alloc( P ) );
    hipLaunchKernelGGL( ccAssignKernel, grdN, blk, 0, s, d_label.p, d_ccCount.p, d_rank.p, f->d_partition.p,
                        uint32_t( sp->minPointCountPerCCPatchSegmentation ), n, d_pointPatch.p, d_view );
    hipLaunchKernelGGL( patchBoundsInitKernel, dim3( ( P + 255 ) / 256 ), blk, 0, s, P, d_bbox.p, d_minUv.p, d_patchStat.p,
                        d_rawCount );
    if ( sp->enablePatchSplitting )
      hipLaunchKernelGGL( patchMinUvKernel, grdN, blk, 0, s, f->d_pts.p, d_pointPatch.p, d_view, n, d_minUv.p );
    hipLaunchKernelGGL( patchTrimBboxKernel, grdN, blk, 0, s, f->d_pts.p, d_view, d_minUv.p,
                        sp->enablePatchSplitting, sp->maxPatchSize, n, d_pointPatch.p, d_bbox.p );
    // (boxes + views, and further down the counters, land in the context's page-locked staging: plain DMA, no staging copy behind it)
    int32_t* h_records = ctx->hostRecords.get<int32_t>( std::max<size_t>( 7 * size_t( P ) + statWords, size_t( 1 ) << 16 ) );
    if ( !h_records ) {
      setError( "segmentPatches: hipHostMalloc failed" );
      return TMC2_E_HIP;
    }
    const int32_t* h_bbox = h_records;
    TMC2_HIP( hipMemcpyAsync( h_records, d_bbox.p, 7 * size_t( P ) * 4, hipMemcpyDeviceToHost, s ) );
    const int32_t* h_view = h_bbox + 6 * size_t( P );
    TMC2_HIP( hipStreamSynchronize( s ) );
    // patch geometry on the host (P is a few hundred): axes, sizes, depth origin, pool offsets, tile list
    static const int       AX[3][3] = {{0, 2, 1}, {1, 2, 0}, {2, 0, 1}};
    std::vector<PatchDev>  h_pdLocal( P );
    PatchDev*              h_pd           = h_pdLocal.data();
    size_t                 tilesSoFar     = 0;
    const size_t           patchBase      = f->patches.size();
    const int64_t          roundDepthBase = f->depthCount;
    for ( uint32_t p = 0; p < P; ++p ) {
      tmc2_patch T{};
      T.index          = int32_t( patchBase + p );
      T.viewId         = h_view[p];
      const int* ax    = AX[T.viewId % 3];
      T.normalAxis     = ax[0];
      T.tangentAxis    = ax[1];
      T.bitangentAxis  = ax[2];
      T.projectionMode = T.viewId / 3;
      const int32_t* bb = &h_bbox[6 * p];
      T.u1    = bb[ax[1]];
      T.v1    = bb[ax[2]];
      T.sizeU = 1 + bb[3 + ax[1]] - bb[ax[1]];
      T.sizeV = 1 + bb[3 + ax[2]] - bb[ax[2]];
      const int L = sp->minLevel;
      T.d1        = T.projectionMode == 0 ? ( bb[ax[0]] / L ) * L
                                          : int( std::ceil( double( bb[3 + ax[0]] ) / double( L ) ) ) * L;
      T.sizeU0    = ( T.sizeU - 1 ) / occRes + 1;
      T.sizeV0    = ( T.sizeV - 1 ) / occRes + 1;
      T.size2DXInPixel = T.sizeU;
      T.size2DYInPixel = T.sizeV;
      if ( sp->quantizerSizeX )
        T.size2DXInPixel = int( std::ceil( double( T.sizeU ) / double( sp->quantizerSizeX ) ) * sp->quantizerSizeX );
      if ( sp->quantizerSizeY )
        T.size2DYInPixel = int( std::ceil( double( T.sizeV ) / double( sp->quantizerSizeY ) ) * sp->quantizerSizeY );
      T.depthOffset = f->depthCount;
      T.occOffset   = f->occCount;
      f->depthCount += int64_t( T.sizeU ) * T.sizeV;
      f->occCount += int64_t( T.sizeU0 ) * T.sizeV0;
      PatchDev& D = h_pd[p];
      D.u1 = T.u1, D.v1 = T.v1, D.d1 = T.d1;
      D.sizeU = T.sizeU, D.sizeV = T.sizeV, D.sizeU0 = T.sizeU0, D.sizeV0 = T.sizeV0;
      D.axN = ax[0], D.axT = ax[1], D.axB = ax[2], D.mode = T.projectionMode;
      D.blockBase = int32_t( tilesSoFar );
      D.depthOff  = T.depthOffset;
      D.occOff    = T.occOffset;
      tilesSoFar += size_t( T.sizeU0 ) * T.sizeV0;
      f->patches.push_back( T );
    }
    const size_t roundArea = size_t( f->depthCount - roundDepthBase );
    const uint32_t tiles   = uint32_t( tilesSoFar );
    // the round's two tables in the context's page-locked staging (patch records | tile -> patch): the copies below are DMA from it;
    // the round's last synchronisation (the counters) comes before the next round refills it
    const size_t pdBytes = ( size_t( P ) * sizeof( PatchDev ) + 15 ) & ~size_t( 15 );
    uint8_t*     h_tables = ctx->hostTables.get<uint8_t>( std::max<size_t>( pdBytes + size_t( tiles ) * 4, size_t( 1 ) << 20 ) );
    if ( !h_tables ) {
      setError( "segmentPatches: hipHostMalloc failed" );
      return TMC2_E_HIP;
    }
    memcpy( h_tables, h_pd, size_t( P ) * sizeof( PatchDev ) );
    uint32_t* h_tilePatch = reinterpret_cast<uint32_t*>( h_tables + pdBytes );
    for ( uint32_t p = 0, at = 0; p < P; ++p )
      for ( uint32_t b = uint32_t( h_pd[p].sizeU0 ) * uint32_t( h_pd[p].sizeV0 ); b > 0; --b ) h_tilePatch[at++] = p;
    TMC2_TRY( f->growPools() );
    TMC2_TRY( d_map64.alloc( roundArea ) );
    TMC2_TRY( d_d0tmp.alloc( roundArea ) );
    TMC2_TRY( d_d1tmp.alloc( roundArea ) );
    TMC2_TRY( d_d0src.alloc( roundArea ) );
    TMC2_TRY( d_tilePatch.alloc( tiles ) );
    TMC2_HIP( hipMemcpyAsync( d_patches.p, h_tables, size_t( P ) * sizeof( PatchDev ), hipMemcpyHostToDevice, s ) );
    TMC2_HIP( hipMemcpyAsync( d_tilePatch.p, h_tilePatch, size_t( tiles ) * 4, hipMemcpyHostToDevice, s ) );
    hipLaunchKernelGGL( patchInitTileKernel, dim3( tiles ), blk, 0, s, d_patches.p, d_tilePatch.p, roundDepthBase, occRes,
                        d_map64.p );
    hipLaunchKernelGGL( patchDepth0Kernel, grdN, blk, 0, s, f->d_pts.p, d_pointPatch.p, d_patches.p, n, roundDepthBase,
                        d_map64.p );
    hipLaunchKernelGGL( patchFilterTileKernel, dim3( tiles ), blk, 0, s, d_patches.p, d_tilePatch.p, d_map64.p,
                        roundDepthBase, occRes, sp->surfaceThickness, sp->maxAllowedDepth, d_d0tmp.p, d_d1tmp.p,
                        d_d0src.p );
    if ( sp->surfaceThickness > 0 )
      hipLaunchKernelGGL( patchDepth1Kernel, grdN, blk, 0, s, f->d_pts.p, f->d_rgb.p, d_pointPatch.p, d_patches.p, n,
                          roundDepthBase, sp->surfaceThickness, d_d0tmp.p, d_d0src.p, d_d1tmp.p );
    hipLaunchKernelGGL( patchResampleTileKernel, dim3( tiles ), blk, 0, s, d_patches.p, d_tilePatch.p, roundDepthBase,
                        occRes, d_d0tmp.p, d_d1tmp.p, bitmapBits, ctx->voxelBitmap.p, f->d_depth0.p, f->d_depth1.p,
                        f->d_occupancy.p, d_patchStat.p );
    int32_t* h_stat = h_records + 7 * size_t( P );  // (copied after the raw-point update below: its count rides along)
    // ---- S9 -----------------------------------------------------------------------------------------
    hipLaunchKernelGGL( rawDistanceKernel, grdN, blk, 0, s, f->d_pts.p, n, ctx->voxelBitmap.p, bitmapBits, d_offsets,
                        int( offsets.size() ), thrSel, d_dist.p, d_raw.p, d_rawCount );
    TMC2_HIP( hipMemcpyAsync( h_stat, d_patchStat.p, statWords * 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    rawCount = 0;
    for ( uint32_t c = 0; c < kRawCounters; ++c ) rawCount += uint32_t( h_stat[2 * size_t( P ) + size_t( c ) * 32] );
    ctx->stageEnd( sid );
    for ( uint32_t p = 0; p < P; ++p ) {
      tmc2_patch& T = f->patches[patchBase + p];
      const int   sizeD = h_stat[2 * p];
      T.sizeDPixel      = sizeD;
      const int bits    = std::min( sp->geometryBitDepth3D, sp->geometryBitDepth2D );
      const int L       = sp->minLevel;
      const int sd      = std::min( ( 1 << bits ) - 1, sizeD );
      const int bitsD   = bits - int( std::log2( L ) );
      int       q       = sd == 0 ? 0 : ( ( sd - 1 ) / L + 1 );
      q                 = std::min( q, ( 1 << bitsD ) - 1 );
      T.sizeD           = q == 0 ? 0 : ( q * L - 1 );
      T.d0Count         = h_stat[2 * p + 1];
      T.eomAndD1Count   = 0;
    }
    ++rounds;
  }
  TMC2_HIP( hipGetLastError() );
  f->rounds      = rounds;
  f->havePatches = true;
  return TMC2_OK;
}

}  // namespace tmc2
