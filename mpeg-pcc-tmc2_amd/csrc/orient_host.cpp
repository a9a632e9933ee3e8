// orient_host.cpp -- normal orientation (S3), host side, exact.
//
// Replaces PCCNormalsGenerator3::orientNormals, SPANNING_TREE branch, and addNeighbors
// (reference: source/lib/PccLibEncoder/source/PCCNormalsGenerator.cpp:198-242, 521-548).
//
// The reference grows a tree from point 0 over the DIRECTED 16-NN graph: a std::priority_queue holds
// every edge (|n_u . n_v|, u, v) pushed from visited u to not-yet-visited v, the largest edge (ties:
// larger start, then larger end -- PCCNormalsGenerator.h:64-71) is popped, and if its end is still
// unvisited it is flipped to agree with its start and expanded.  That is inherently sequential (the
// graph is directed, so the result is not an MST and no Boruvka-style shortcut is exact), hence it
// stays on the host, fed by the device-computed k-NN lists and normals.
//
// Formulation used here (exactly equivalent, ~4x less heap work than the reference): only the BEST
// pending in-edge of each unvisited vertex can ever be accepted (all worse edges into the same vertex
// are popped after it has been visited and are then ignored), so we keep one key (w, start) per
// unvisited vertex in an indexed max-heap and raise it when a better in-edge appears.  The heap's
// maximum is the same edge the reference's queue would accept next; stale entries never exist.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>

#include "internal.h"

namespace tmc2 {

namespace {

// per-vertex state: the slot of the vertex's best pending in-edge in the heap, or one of two markers.  Four bytes per
// vertex keep the randomly accessed part of the working set at n * 4 bytes (3.3 MB at longdress size): several walks
// can share one last-level cache, which is what bounds the number of frames the host can orient at once.
using VState = int32_t;
constexpr int32_t kAbsent = -1, kVisited = -2;  // no pending in-edge yet / already oriented

// heap entries carry their own key, so sifting compares contiguous records instead of chasing vertex ids.
// Field order matters: read as two 64-bit words, (|d| bits, s:v) is the reference's edge order (weight, start, end)
// as one 128-bit unsigned number -- |d| >= 0, so its IEEE bit pattern orders like the value -- and the comparison
// compiles to branch-free code (the outcomes are coin flips; mispredictions used to dominate the sifts).
struct HeapEntry {
  double   d;
  uint32_t v, s;
};
static_assert( sizeof( HeapEntry ) == 16, "HeapEntry layout" );

__attribute__( ( always_inline ) ) inline unsigned __int128 edgeKey( const HeapEntry& e ) {
  uint64_t w[2];
  memcpy( w, &e, 16 );
  return ( static_cast<unsigned __int128>( w[0] & 0x7FFFFFFFFFFFFFFFull ) << 64 ) | w[1];
}

// 4-ary max-heap: half the levels of a binary heap, and the four children of a slot share one cache line
struct VertexHeap {
  std::vector<HeapEntry> heap;
  VState*                st;
  const uint32_t*        owner;  // nullptr: one pending entry per vertex; else per owner[vertex] (cluster walk)

  VertexHeap( size_t n, VState* state, const uint32_t* ownerOf = nullptr ) : st( state ), owner( ownerOf ) {
    for ( size_t i = 0; i < n; ++i ) st[i] = kAbsent;
    heap.reserve( n / 4 + 16 );
  }
  size_t slotOf( uint32_t v ) const { return owner ? owner[v] : v; }
  // strict "a has a smaller key than b" in the reference's edge order (weight, start, end)
  static bool less( const HeapEntry& a, const HeapEntry& b ) { return edgeKey( a ) < edgeKey( b ); }
  void siftUp( size_t i ) {
    const HeapEntry e = heap[i];
    while ( i > 0 ) {
      const size_t p = ( i - 1 ) >> 2;
      if ( !less( heap[p], e ) ) break;
      heap[i]           = heap[p];
      st[slotOf( heap[i].v )] = int32_t( i );
      i                 = p;
    }
    heap[i]     = e;
    st[slotOf( e.v )] = int32_t( i );
  }
  void siftDown( size_t i ) {
    const size_t    n = heap.size();
    const HeapEntry e = heap[i];
    for ( ;; ) {
      const size_t c0 = 4 * i + 1;
      if ( c0 >= n ) break;
      size_t       c  = c0;
      const size_t ce = c0 + 4 < n ? c0 + 4 : n;
      unsigned __int128 best = edgeKey( heap[c0] );
      for ( size_t t = c0 + 1; t < ce; ++t ) {
        const unsigned __int128 kt = edgeKey( heap[t] );
        const bool              gt = best < kt;
        c                          = gt ? t : c;
        best                       = gt ? kt : best;
      }
      if ( !less( e, heap[c] ) ) break;
      heap[i]           = heap[c];
      st[slotOf( heap[i].v )] = int32_t( i );
      i                 = c;
    }
    heap[i]     = e;
    st[slotOf( e.v )] = int32_t( i );
  }
  // offer in-edge (signed dot d as it stands now, start) to unvisited vertex v whose state is `pos`; the key of a
  // pending vertex lives in its heap entry (the heap is small and cache resident)
  void offer( uint32_t v, int32_t pos, double d, uint32_t start ) {
    if ( pos == kAbsent ) {
      heap.push_back( HeapEntry{d, v, start} );
      siftUp( heap.size() - 1 );
    } else {
      HeapEntry&      h = heap[size_t( pos )];
      const HeapEntry c{d, v, start};  // same end vertex in the per-vertex walk; any vertex of the cluster otherwise
      if ( less( h, c ) ) {
        h = c;
        siftUp( size_t( pos ) );
      }
    }
  }
  // delete the pending entry in slot i (its vertex has been reached another way)
  void remove( size_t i ) {
    const HeapEntry last = heap.back();
    heap.pop_back();
    if ( i == heap.size() ) return;
    const bool up = less( heap[i], last );
    heap[i]       = last;
    st[slotOf( last.v )] = int32_t( i );
    if ( up )
      siftUp( i );
    else
      siftDown( i );
  }
  HeapEntry popMax() {
    const HeapEntry top  = heap[0];
    const HeapEntry last = heap.back();
    heap.pop_back();
    if ( !heap.empty() ) {
      heap[0] = last;
      siftDown( 0 );
    }
    st[slotOf( top.v )] = kVisited;
    return top;
  }
};

inline double dot( const double* a, const double* b ) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

}  // namespace

// Spanning-tree growth proper.  edgeDot[u][j] = n_u . n_knn[u][j] on the ORIGINAL normals (fp64, evaluated as dot()
// above -- on the device path it is computed by edgeDotKernel).  Negating a normal negates its dot products exactly,
// so the walk never touches the normals: the weight of an edge is |edgeDot|, and with sign[u] = +-1 the orientation
// already given to u, the reference's test "n_u(now) . n_v < 0" is sign[u] * edgeDot < 0.  Normals are read only at
// the (rare) seeds of new components.  Output: sign[i] = -1 where the reference would have negated normal i by the end
// of the growth (the global majority flip is left to the caller).
//
// Strong edges.  The growth always takes the heaviest pending edge, so once it stands on a vertex it exhausts every
// edge of weight >= tau it can reach before it takes any lighter one: between two light edges it absorbs exactly the
// set R reachable from the entry vertex over strong (>= tau) directed edges, and no strong edge leads from anything
// visited earlier into R (it would have been taken before the light edge).  WHICH strong edges become tree edges
// depends on their order, but if every strong edge with both ends in R agrees with one sign assignment (the signed
// graph on R is balanced -- the normal case: neighbouring normals that are nearly parallel up to sign), every
// admissible tree yields that same assignment.  So R is absorbed by a plain breadth-first sweep: no heap traffic for
// strong edges, light edges are offered as usual (their order of arrival is immaterial: the heap keeps the best
// in-edge per vertex under an order-independent rule, and none of them can be taken before R is complete).  Every
// strong edge inside R is checked against the assignment; a single disagreement means the order would have
// mattered, and the caller repeats the whole growth the plain way (tau = infinity).
// scratch: n * 8 bytes or nullptr.
// The walk is bound by the latency of its scattered reads (the 16 state records of a row's neighbours, the rows of the
// next vertex), so those are issued together / ahead of use.
static bool growSigns( const int16_t* xyz, size_t n, const uint32_t* knn, int k, const double* normals,
                       const double* edgeDot, int8_t* sign, void* scratch, double tau ) {
  VState*    st    = reinterpret_cast<VState*>( scratch );
  uint32_t*  phase = reinterpret_cast<uint32_t*>( scratch ) + n;  // which absorption reached the vertex
  VertexHeap heap( n, st );
  for ( size_t i = 0; i < n; ++i ) sign[i] = 1, phase[i] = 0;
  constexpr int         kMaxK = 64;
  std::vector<uint32_t> queue;
  queue.reserve( 1 << 16 );
  uint32_t epoch = 0;
  auto prefetchRows = [&]( uint32_t v ) {
    const char* r = reinterpret_cast<const char*>( knn + size_t( v ) * k );
    const char* d = reinterpret_cast<const char*>( edgeDot + size_t( v ) * k );
    // rows are read exactly once: non-temporal, so that they do not evict the state records from the shared L3
    for ( int o = 0; o < k * 4; o += 64 ) __builtin_prefetch( r + o, 0, 0 );
    for ( int o = 0; o < k * 8; o += 64 ) __builtin_prefetch( d + o, 0, 0 );
  };
  // v0 has just been visited and signed: absorb everything strongly reachable from it.  false = inconsistent
  auto absorb = [&]( uint32_t v0 ) -> bool {
    ++epoch;
    phase[v0] = epoch;
    queue.clear();
    queue.push_back( v0 );
    for ( size_t head = 0; head < queue.size(); ++head ) {
      const uint32_t cur = queue[head];
      if ( head + 1 < queue.size() ) prefetchRows( queue[head + 1] );
      const double    sc  = double( sign[cur] );
      const uint32_t* row = knn + size_t( cur ) * k;
      const double*   dr  = edgeDot + size_t( cur ) * k;
      int32_t         pos[kMaxK];
      for ( int j = 0; j < k; ++j ) pos[j] = st[row[j]];  // independent loads: the misses overlap
      for ( int j = 0; j < k; ++j ) {
        const uint32_t v      = row[j];
        const double   d      = sc * dr[j];
        const bool     strong = std::fabs( d ) >= tau;
        const int32_t  pv     = st[v];  // an earlier edge of this row may have changed it (cached by now)
        if ( pv == kVisited ) {
          if ( strong && phase[v] == epoch && ( d < 0.0 ) != ( sign[v] < 0 ) && v != cur ) return false;
        } else if ( strong ) {
          if ( pv >= 0 ) heap.remove( size_t( pv ) );
          st[v]    = kVisited;
          sign[v]  = d < 0.0 ? -1 : 1;
          phase[v] = epoch;
          queue.push_back( v );
        } else {
          heap.offer( v, pv, d, cur );
        }
      }
      (void)pos;
    }
    return true;
  };
  for ( size_t seed = 0; seed < n; ++seed ) {
    if ( st[seed] == kVisited ) continue;
    // a seed is never in the heap: the heap is empty whenever a new seed is picked
    st[seed] = kVisited;
    // reference direction of the seed: sum of its already oriented neighbours (row order), else the previous point's
    // normal as it stands, else the view point
    double          acc[3]   = {0.0, 0.0, 0.0};
    size_t          accCount = 0;
    const uint32_t* row      = knn + seed * k;
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = row[j];
      if ( st[v] == kVisited && v != seed ) {
        const double sv = double( sign[v] );
        acc[0] += sv * normals[3 * size_t( v )];
        acc[1] += sv * normals[3 * size_t( v ) + 1];
        acc[2] += sv * normals[3 * size_t( v ) + 2];
        ++accCount;
      }
    }
    if ( accCount == 0 ) {
      if ( seed != 0 ) {
        const double sp = double( sign[seed - 1] );
        acc[0] = sp * normals[3 * ( seed - 1 )];
        acc[1] = sp * normals[3 * ( seed - 1 ) + 1];
        acc[2] = sp * normals[3 * ( seed - 1 ) + 2];
      } else {
        acc[0] = 0.0 - xyz[0];
        acc[1] = 0.0 - xyz[1];
        acc[2] = 0.0 - xyz[2];
      }
    }
    if ( dot( normals + 3 * seed, acc ) < 0.0 ) sign[seed] = -1;
    if ( !absorb( uint32_t( seed ) ) ) return false;
    while ( !heap.heap.empty() ) {
      const HeapEntry e = heap.popMax();
      // the new top is the likeliest next pop: have its rows on the way while this vertex is expanded
      if ( !heap.heap.empty() ) prefetchRows( heap.heap[0].v );
      if ( e.d < 0.0 ) sign[e.v] = -1;
      if ( !absorb( e.v ) ) return false;
    }
  }
  return true;
}

// ---- the same growth on the CONTRACTED graph ----------------------------------------------------------------------
// Vertices joined by MUTUAL strong edges (u lists v, v lists u, |n_u . n_v| >= tau) reach each other over strong
// edges, so the growth absorbs such a cluster as a whole the moment it touches it (see above); if the cluster's strong
// edges agree with one relative sign assignment (parity to a root -- established and verified by the contraction,
// device: orient_contract.hip, host: contractOnHost below), only the cluster's sign S is left to decide:
// sign[v] = S[root[v]] * (-1)^parity[v].  What remains of the graph are the CROSS edges (both ends in different
// clusters), a few per cent of all edges on a smooth surface, grouped by source cluster.  The walk below is
// growSigns() with clusters for vertices: strong cross edges (one-way ones) are absorbed breadth-first and checked,
// light ones go through the heap, whose key is still the reference's (weight, start vertex, end vertex) -- one pending
// entry per unvisited cluster: the best edge into any of its vertices is the one the per-vertex heap would pop first.
// Seeds.  The reference orients the first point of every connected component by a rule that reads the normals of its
// already oriented neighbours (orientSeedSign below).  Everything else in the component is oriented RELATIVE to its
// seed, so the walk gives every seed +1 provisionally, numbers the components in the order it opens them
// (component[c], per cluster) and lists the seeds; resolveSeedSigns() then applies the rule seed by seed -- "already
// oriented" at the time seed k was picked means exactly "in a component < k" -- and a cluster's final sign is its
// provisional one times its component's.  The rule's inputs (one k-NN row and <= k + 2 normals per seed) can thus be
// fetched in one piece after the walk.  Returns false if a strong cross edge disagrees.
bool orientContractedSigns( size_t n, const OrientContraction& g, double tau, int8_t* clusterSign, uint32_t* component,
                            std::vector<uint32_t>& seeds, void* scratch ) {
  VState*    st    = reinterpret_cast<VState*>( scratch );       // by cluster (root vertex id)
  uint32_t*  phase = reinterpret_cast<uint32_t*>( scratch ) + n;
  VertexHeap heap( n, st, g.root );
  for ( size_t i = 0; i < n; ++i ) clusterSign[i] = 1, phase[i] = 0;
  std::vector<uint32_t> queue;
  queue.reserve( 1 << 12 );
  uint32_t epoch = 0;
  auto absorb = [&]( uint32_t c0 ) -> bool {
    ++epoch;
    phase[c0] = epoch;
    queue.clear();
    queue.push_back( c0 );
    for ( size_t head = 0; head < queue.size(); ++head ) {
      const uint32_t c  = queue[head];
      const double   sc = double( clusterSign[c] );
      for ( uint32_t e = g.off[c]; e < g.off[c + 1]; ++e ) {
        const OrientCrossEdge& x  = g.edges[e];
        const uint32_t         c2 = g.root[x.v];
        const double           d  = ( ( g.parity[x.u] ^ g.parity[x.v] ) & 1 ) ? -( sc * x.d ) : sc * x.d;  // = sign[u] n_u.n_v (-1)^parity[v]
        const bool             strong = std::fabs( x.d ) >= tau;
        const int32_t          pv     = st[c2];
        if ( pv == kVisited ) {
          if ( strong && phase[c2] == epoch && ( d < 0.0 ) != ( clusterSign[c2] < 0 ) ) return false;
        } else if ( strong ) {
          if ( pv >= 0 ) heap.remove( size_t( pv ) );
          st[c2]          = kVisited;
          clusterSign[c2] = d < 0.0 ? -1 : 1;
          phase[c2]       = epoch;
          queue.push_back( c2 );
        } else {
          heap.offer( x.v, pv, d, x.u );
        }
      }
    }
    return true;
  };
  seeds.clear();
  for ( size_t seed = 0; seed < n; ++seed ) {
    const uint32_t c = g.root[seed];
    if ( st[c] == kVisited ) continue;
    seeds.push_back( uint32_t( seed ) );
    st[c]          = kVisited;
    clusterSign[c] = int8_t( ( g.parity[seed] & 1 ) ? -1 : 1 );  // the seed itself: +1 for now
    if ( !absorb( c ) ) return false;
    while ( !heap.heap.empty() ) {
      const HeapEntry e  = heap.popMax();
      const uint32_t  c2 = g.root[e.v];
      clusterSign[c2]    = e.d < 0.0 ? -1 : 1;
      if ( !absorb( c2 ) ) return false;
    }
    // everything visited since the last seed belongs to this component: stamp it (phase[] holds absorption epochs,
    // the epochs of one component are consecutive)
  }
  // component of a cluster = number of seeds opened before (or at) its absorption: epochs are handed out in order, so
  // a second pass over the seeds' epochs is enough
  {
    std::vector<uint32_t> firstEpoch( seeds.size() );
    for ( size_t k = 0; k < seeds.size(); ++k ) firstEpoch[k] = phase[g.root[seeds[k]]];
    for ( size_t i = 0; i < n; ++i ) {
      if ( g.root[i] != i ) continue;
      const uint32_t ep = phase[i];
      component[i]      = uint32_t( std::upper_bound( firstEpoch.begin(), firstEpoch.end(), ep ) - firstEpoch.begin() ) - 1u;
    }
  }
  return true;
}

// Seed rule, after the walk: compSign[k] for every component, and the clusters' final signs.
// rowOf( k ) / normalOf( k, j ): the k-NN row of seed k and the normals of (j = 0) the seed, (j = 1) the point before it
// in index order, (j = 2 + t) its t-th neighbour.
void resolveSeedSigns( size_t n, const OrientContraction& g, int kNN, const std::vector<uint32_t>& seeds,
                       const uint32_t* component, const std::function<const uint32_t*( size_t )>& rowOf,
                       const std::function<const double*( size_t, int )>& normalOf, const int16_t* xyz0,
                       int8_t* clusterSign ) {
  std::vector<int8_t> compSign( seeds.size(), 1 );
  for ( size_t k = 0; k < seeds.size(); ++k ) {
    const uint32_t  i   = seeds[k];
    const uint32_t* row = rowOf( k );
    // sign of v as it stood when seed k was picked: 0 if v had not been oriented yet
    const auto signThen = [&]( uint32_t v ) -> int {
      const uint32_t c = g.root[v];
      if ( component[c] >= k ) return 0;
      const int sc = int( clusterSign[c] ) * int( compSign[component[c]] );
      return ( g.parity[v] & 1 ) ? -sc : sc;
    };
    double acc[3]   = {0.0, 0.0, 0.0};
    size_t accCount = 0;
    for ( int j = 0; j < kNN; ++j ) {
      const uint32_t v  = row[j];
      const int      sv = v != i ? signThen( v ) : 0;
      if ( sv != 0 ) {
        const double* nv = normalOf( k, 2 + j );
        acc[0] += double( sv ) * nv[0];
        acc[1] += double( sv ) * nv[1];
        acc[2] += double( sv ) * nv[2];
        ++accCount;
      }
    }
    if ( accCount == 0 ) {
      if ( i != 0 ) {
        const int     sp = signThen( i - 1 );  // i is the smallest index not oriented yet: i - 1 has been
        const double* np = normalOf( k, 1 );
        acc[0] = double( sp == 0 ? 1 : sp ) * np[0];
        acc[1] = double( sp == 0 ? 1 : sp ) * np[1];
        acc[2] = double( sp == 0 ? 1 : sp ) * np[2];
      } else {
        acc[0] = 0.0 - xyz0[0];
        acc[1] = 0.0 - xyz0[1];
        acc[2] = 0.0 - xyz0[2];
      }
    }
    compSign[k] = dot( normalOf( k, 0 ), acc ) < 0.0 ? -1 : 1;
  }
  for ( size_t i = 0; i < n; ++i )
    if ( g.root[i] == i && compSign[component[i]] < 0 ) clusterSign[i] = int8_t( -clusterSign[i] );
}

// ---- the walk on the COMPACT contracted graph (device contraction) -----------------------------------------------------------
// orientContractedSigns() with clusters as the only vertices: every array is per cluster (18 K entries instead of 0.84 M: the
// whole working set sits in the host's L1 / L2), a cluster's edge list holds one light edge per target cluster (the only one of
// its offers into that cluster that can be accepted) and the strong one-way edges, with the target cluster and the ends'
// parities already folded in.  Clusters are numbered by first member, so "the smallest point not oriented yet" -- the next seed
// -- is the first member of the first cluster not visited yet.
namespace {
struct ClusterHeap {  // indexed 4-ary max-heap, one pending entry per cluster; key = the reference's (|d|, start, end)
  struct Entry {
    double   d;
    uint32_t v, s, c, pad;
  };
  std::vector<Entry>   heap;
  std::vector<int32_t> st;  // slot of the cluster's entry, kAbsent, kVisited
  explicit ClusterHeap( size_t clusters ) : st( clusters, kAbsent ) { heap.reserve( clusters / 2 + 16 ); }
  static unsigned __int128 key( const Entry& e ) {
    uint64_t w[2];
    memcpy( w, &e, 16 );
    return ( static_cast<unsigned __int128>( w[0] & 0x7FFFFFFFFFFFFFFFull ) << 64 ) | w[1];
  }
  void place( size_t i, const Entry& e ) {
    heap[i]   = e;
    st[e.c]   = int32_t( i );
  }
  void siftUp( size_t i ) {
    const Entry e = heap[i];
    while ( i > 0 ) {
      const size_t p = ( i - 1 ) >> 2;
      if ( !( key( heap[p] ) < key( e ) ) ) break;
      place( i, heap[p] );
      i = p;
    }
    place( i, e );
  }
  void siftDown( size_t i ) {
    const size_t n = heap.size();
    const Entry  e = heap[i];
    for ( ;; ) {
      const size_t c0 = 4 * i + 1;
      if ( c0 >= n ) break;
      size_t c = c0;
      for ( size_t t = c0 + 1; t < std::min( c0 + 4, n ); ++t )
        if ( key( heap[c] ) < key( heap[t] ) ) c = t;
      if ( !( key( e ) < key( heap[c] ) ) ) break;
      place( i, heap[c] );
      i = c;
    }
    place( i, e );
  }
  void offer( uint32_t c, double d, uint32_t v, uint32_t start ) {
    const Entry   cand{d, v, start, c, 0};
    const int32_t pos = st[c];
    if ( pos == kAbsent ) {
      heap.push_back( cand );
      siftUp( heap.size() - 1 );
    } else if ( key( heap[size_t( pos )] ) < key( cand ) ) {
      heap[size_t( pos )] = cand;
      siftUp( size_t( pos ) );
    }
  }
  void remove( size_t i ) {
    const Entry last = heap.back();
    heap.pop_back();
    if ( i == heap.size() ) return;
    const bool up = key( heap[i] ) < key( last );
    heap[i]       = last;
    st[last.c]    = int32_t( i );
    if ( up )
      siftUp( i );
    else
      siftDown( i );
  }
  Entry popMax() {
    const Entry top  = heap[0];
    const Entry last = heap.back();
    heap.pop_back();
    if ( !heap.empty() ) {
      heap[0] = last;
      siftDown( 0 );
    }
    st[top.c] = kVisited;
    return top;
  }
};
}  // namespace

// clusterSign[c]: the sign of cluster c's root-relative frame (as orientContractedSigns); component[c]: the component it
// belongs to; seeds / seedClusters: the first point and the cluster of every component, in the order they were opened
bool orientCompactSigns( const OrientCompact& g, double tau, int8_t* clusterSign, uint32_t* component, std::vector<uint32_t>& seeds,
                         std::vector<uint32_t>& seedClusters ) {
  const uint32_t        C = g.clusters;
  ClusterHeap           heap( C );
  std::vector<uint32_t> phase( C, 0 ), queue;
  for ( uint32_t c = 0; c < C; ++c ) clusterSign[c] = 1;
  queue.reserve( 1 << 10 );
  uint32_t epoch  = 0;
  auto     absorb = [&]( uint32_t c0 ) -> bool {
    ++epoch;
    phase[c0] = epoch;
    queue.clear();
    queue.push_back( c0 );
    for ( size_t head = 0; head < queue.size(); ++head ) {
      const uint32_t c  = queue[head];
      const double   sc = double( clusterSign[c] );
      for ( uint32_t e = g.rec[c].off; e < g.rec[c + 1].off; ++e ) {
        const OrientCompactEdge& x      = g.edges[e];
        const uint32_t           c2     = x.c2;
        const double             d      = sc * x.d;  // = sign[u] n_u.n_v (-1)^parity[v]
        const bool               strong = std::fabs( x.d ) >= tau;
        const int32_t            pv     = heap.st[c2];
        if ( pv == kVisited ) {
          if ( strong && phase[c2] == epoch && ( d < 0.0 ) != ( clusterSign[c2] < 0 ) ) return false;
        } else if ( strong ) {
          if ( pv >= 0 ) heap.remove( size_t( pv ) );
          heap.st[c2]     = kVisited;
          clusterSign[c2] = d < 0.0 ? -1 : 1;
          phase[c2]       = epoch;
          queue.push_back( c2 );
        } else {
          heap.offer( c2, d, x.v, x.u );
        }
      }
    }
    return true;
  };
  seeds.clear();
  seedClusters.clear();
  std::vector<uint32_t> firstEpoch;
  for ( uint32_t c = 0; c < C; ++c ) {  // (clusters in the order of their first members)
    if ( heap.st[c] == kVisited ) continue;
    seeds.push_back( g.rec[c].seedPoint );
    seedClusters.push_back( c );
    heap.st[c]     = kVisited;
    clusterSign[c] = int8_t( ( g.rec[c].seedParity & 1 ) ? -1 : 1 );  // the seed itself: +1 for now
    firstEpoch.push_back( epoch + 1 );
    if ( !absorb( c ) ) return false;
    while ( !heap.heap.empty() ) {
      const ClusterHeap::Entry e = heap.popMax();
      clusterSign[e.c]           = e.d < 0.0 ? -1 : 1;
      if ( !absorb( e.c ) ) return false;
    }
  }
  for ( uint32_t c = 0; c < C; ++c )
    component[c] = uint32_t( std::upper_bound( firstEpoch.begin(), firstEpoch.end(), phase[c] ) - firstEpoch.begin() ) - 1u;
  return true;
}

// Seed rule after the compact walk.  who: [seeds][kNN + 1][3] = (point, cluster, parity) of the point before the seed in index
// order and of its kNN neighbours; normals: [seeds][kNN + 2][3] = seed, point before, neighbours (gatherSeedTables).
void resolveSeedSignsCompact( const OrientCompact& g, int kNN, const std::vector<uint32_t>& seeds, const std::vector<uint32_t>& seedClusters,
                              const uint32_t* component, const uint32_t* who, const double* normals, const int16_t* xyz0, int8_t* clusterSign ) {
  std::vector<int8_t> compSign( seeds.size(), 1 );
  for ( size_t k = 0; k < seeds.size(); ++k ) {
    const uint32_t  i   = seeds[k];
    const uint32_t* w   = who + k * size_t( kNN + 1 ) * 3;
    const double*   nrm = normals + k * size_t( kNN + 2 ) * 3;
    // sign of entry t of `who` as it stood when seed k was picked: 0 if that point had not been oriented yet
    const auto signThen = [&]( int t ) -> int {
      const uint32_t c = w[3 * t + 1];
      if ( component[c] >= k ) return 0;
      const int sc = int( clusterSign[c] ) * int( compSign[component[c]] );
      return ( w[3 * t + 2] & 1 ) ? -sc : sc;
    };
    double acc[3]   = {0.0, 0.0, 0.0};
    size_t accCount = 0;
    for ( int j = 0; j < kNN; ++j ) {
      const uint32_t v  = w[3 * ( 1 + j )];
      const int      sv = v != i ? signThen( 1 + j ) : 0;
      if ( sv != 0 ) {
        const double* nv = nrm + 3 * ( 2 + j );
        acc[0] += double( sv ) * nv[0];
        acc[1] += double( sv ) * nv[1];
        acc[2] += double( sv ) * nv[2];
        ++accCount;
      }
    }
    if ( accCount == 0 ) {
      if ( i != 0 ) {
        const int     sp = signThen( 0 );  // i is the smallest index not oriented yet: i - 1 has been
        const double* np = nrm + 3;
        acc[0] = double( sp == 0 ? 1 : sp ) * np[0];
        acc[1] = double( sp == 0 ? 1 : sp ) * np[1];
        acc[2] = double( sp == 0 ? 1 : sp ) * np[2];
      } else {
        acc[0] = 0.0 - xyz0[0];
        acc[1] = 0.0 - xyz0[1];
        acc[2] = 0.0 - xyz0[2];
      }
    }
    compSign[k] = dot( nrm, acc ) < 0.0 ? -1 : 1;
  }
  (void)seedClusters;
  for ( uint32_t c = 0; c < g.clusters; ++c )
    if ( compSign[component[c]] < 0 ) clusterSign[c] = int8_t( -clusterSign[c] );
}

// host-side contraction (the device path does the same in orient_contract.hip): union-find with parity over the
// mutual strong edges, consistency check, cross edges grouped by source cluster.  false = some cluster's strong edges
// disagree (the caller then grows the plain way / with a tighter threshold).
static bool contractOnHost( size_t n, const uint32_t* knn, int k, const double* edgeDot, double tau,
                            std::vector<uint32_t>& root, std::vector<uint8_t>& parity, std::vector<uint32_t>& off,
                            std::vector<OrientCrossEdge>& edges ) {
  std::vector<uint32_t> parent( n );
  std::vector<uint8_t>  par( n, 0 );  // parity to the parent
  for ( size_t i = 0; i < n; ++i ) parent[i] = uint32_t( i );
  auto find = [&]( uint32_t x, uint8_t& px ) -> uint32_t {  // with path compression
    uint32_t r = x;
    uint8_t  p = 0;
    while ( parent[r] != r ) p ^= par[r], r = parent[r];
    // second pass: point everything at the root with the accumulated parity
    uint32_t y = x;
    uint8_t  q = p;
    while ( parent[y] != y ) {
      const uint32_t next = parent[y];
      const uint8_t  step = par[y];
      parent[y]           = r;
      par[y]              = q;
      q ^= step;
      y = next;
    }
    px = p;
    return r;
  };
  for ( size_t u = 0; u < n; ++u )
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = knn[u * k + j];
      const double   d = edgeDot[u * k + j];
      if ( v >= u || std::fabs( d ) < tau ) continue;  // every mutual edge is seen from both ends: the larger acts
      bool mutual = false;
      for ( int t = 0; t < k; ++t ) mutual |= knn[size_t( v ) * k + t] == u;
      if ( !mutual ) continue;
      uint8_t        pu, pv;
      const uint32_t ru = find( uint32_t( u ), pu ), rv = find( v, pv );
      const uint8_t  s  = d < 0.0 ? 1 : 0;
      if ( ru == rv ) {
        if ( ( pu ^ pv ) != s ) return false;
      } else {
        parent[ru] = rv;
        par[ru]    = uint8_t( pu ^ pv ^ s );
      }
    }
  root.resize( n );
  parity.resize( n );
  for ( size_t i = 0; i < n; ++i ) root[i] = find( uint32_t( i ), parity[i] );
  // A strong edge that is NOT mutual may still have both ends in one cluster (they are joined by other, mutual, strong edges).
  // It is an edge of the strongly reachable set like any other: the growth may take it before the mutual ones (it does when it
  // is the heavier way in), so it too must agree with the cluster's parities -- or the order of the growth matters and the
  // caller has to grow the plain way.  (Round 4: frame 26 of the redandblack-like GOF has exactly one such edge in 12 M; the
  // reference follows it and orients two points against their cluster.)
  for ( size_t u = 0; u < n; ++u )
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = knn[u * k + j];
      const double   d = edgeDot[u * k + j];
      if ( std::fabs( d ) >= tau && root[v] == root[u] && ( uint8_t( parity[u] ^ parity[v] ) != ( d < 0.0 ? 1 : 0 ) ) ) return false;
    }
  off.assign( n + 1, 0 );
  for ( size_t u = 0; u < n; ++u )
    for ( int j = 0; j < k; ++j )
      if ( root[knn[u * k + j]] != root[u] ) ++off[root[u] + 1];
  for ( size_t i = 0; i < n; ++i ) off[i + 1] += off[i];
  edges.resize( off[n] );
  std::vector<uint32_t> cursor( off.begin(), off.end() - 1 );
  for ( size_t u = 0; u < n; ++u )
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = knn[u * k + j];
      if ( root[v] != root[u] ) edges[cursor[root[u]]++] = OrientCrossEdge{uint32_t( u ), v, edgeDot[u * k + j]};
    }
  return true;
}

// The contraction above in the compact form the device produces (orient_contract.hip: clusters numbered by first member, per
// ordered pair of clusters the light edges that carry the pair's largest |n_u . n_v| and one strong edge per implied relative
// sign): the host-only path walks the same structure as the device path -- and the CPU tier pins that walk and the reduction of
// the edge list to the reference's growth.
struct HostCompact {
  std::vector<OrientClusterRec>  rec;
  std::vector<OrientCompactEdge> edges;
  std::vector<uint32_t>          cid;
  OrientCompact view() const { return OrientCompact{uint32_t( rec.size() - 1 ), rec.data(), edges.data()}; }
};
static void compactOnHost( size_t n, const std::vector<uint32_t>& root, const std::vector<uint8_t>& parity, const std::vector<uint32_t>& off,
                           const std::vector<OrientCrossEdge>& edges, double tau, HostCompact& out ) {
  out.cid.assign( n, 0 );
  std::vector<uint32_t> idOfRoot( n, 0xFFFFFFFFu );
  uint32_t              C = 0;
  for ( size_t i = 0; i < n; ++i ) {  // clusters in the order of their first members
    uint32_t& id = idOfRoot[root[i]];
    if ( id == 0xFFFFFFFFu ) {
      id = C++;
      out.rec.push_back( OrientClusterRec{0u, uint32_t( i ), parity[i]} );
    }
    out.cid[i] = id;
  }
  out.rec.push_back( OrientClusterRec{0u, 0u, 0u} );
  std::vector<std::vector<OrientCompactEdge>> per( C );
  for ( size_t r = 0; r < n; ++r ) {
    if ( root[r] != r || off[r] == off[r + 1] ) continue;
    // per target cluster: best |d| among the light edges; the strong edges' sign classes seen
    std::vector<OrientCompactEdge> all;
    for ( uint32_t e = off[r]; e < off[r + 1]; ++e ) {
      const OrientCrossEdge& x = edges[e];
      all.push_back( OrientCompactEdge{x.u, x.v, out.cid[x.v], 0u, ( ( parity[x.u] ^ parity[x.v] ) & 1 ) ? -x.d : x.d} );
    }
    std::stable_sort( all.begin(), all.end(), []( const OrientCompactEdge& a, const OrientCompactEdge& b ) { return a.c2 < b.c2; } );
    std::vector<OrientCompactEdge>& keep = per[out.cid[r]];
    for ( size_t i = 0; i < all.size(); ) {
      size_t j    = i;
      double best = -1.0;
      while ( j < all.size() && all[j].c2 == all[i].c2 ) {
        if ( std::fabs( all[j].d ) < tau ) best = std::max( best, std::fabs( all[j].d ) );
        ++j;
      }
      bool strongSeen[2] = {false, false};
      for ( size_t t = i; t < j; ++t ) {
        const double w = std::fabs( all[t].d );
        if ( w >= tau ) {
          const int cls = all[t].d < 0.0 ? 1 : 0;
          if ( !strongSeen[cls] ) keep.push_back( all[t] );
          strongSeen[cls] = true;
        } else if ( w == best ) {
          keep.push_back( all[t] );
        }
      }
      i = j;
    }
  }
  for ( uint32_t c = 0; c < C; ++c ) {
    out.rec[c].off = uint32_t( out.edges.size() );
    out.edges.insert( out.edges.end(), per[c].begin(), per[c].end() );
  }
  out.rec[C].off = uint32_t( out.edges.size() );
}

// first strong-edge threshold: ~11 degrees (nearly every edge on a smooth surface is strong)
// The thresholds the CONTRACTED walk is tried with, in turn: any threshold at which every strong edge inside a cluster (and every
// strong cross edge of a strongly reachable set) agrees with one sign assignment gives the reference's orientation, so a frame
// whose graph is unbalanced at 0.98 -- one stray edge of weight 0.984 in 12 M is enough (round 4, redandblack-like frame 26) --
// is tried again a little tighter (a contraction + a cluster walk: milliseconds) before it falls back to the growth point by
// point (hundreds of milliseconds, and the GOF waits for it).
std::vector<double> orientTauLadder( const tmc2_ctx* ctx ) {
  const double        first = orientFirstTau( ctx );
  std::vector<double> l{first};
  for ( double t : {0.99, 0.995, 0.998} )
    if ( t > first && first <= 1.5 ) l.push_back( t );
  return l;
}

double orientFirstTau( const tmc2_ctx* ctx ) {
  const char* e = ctxOption( ctx, "ORIENT_TAU" );  // test hook; >= 2 goes straight to the plain growth
  return e ? atof( e ) : 0.98;
}

// returns the number of growths it took (1: the first attempt held; each disagreement costs one more).
// tryContraction: contract on the host first (the device path has done that -- or failed at it -- on the device)
int orientSpanningTreeSigns( const int16_t* xyz, size_t n, const uint32_t* knn, int k, const double* normals,
                             const double* edgeDot, int8_t* sign, void* scratch, bool tryContraction, const tmc2_ctx* ctx ) {
  if ( n == 0 ) return 0;
  std::vector<uint64_t> own;
  if ( !scratch ) {
    own.resize( n );
    scratch = own.data();
  }
  // thresholds tried in turn: the first (contracted, then point by point), then ~3.6 degrees, then none (the plain
  // growth, always exact by construction)
  const double first0 = orientFirstTau( ctx );
  int growths = 0;
  if ( tryContraction && first0 <= 1.5 && !ctxOption( ctx, "ORIENT_NO_CONTRACTION" ) )  // (test hook: point-level walk only)
  for ( const double first : orientTauLadder( ctx ) ) {
    // contracted walk: clusters of mutual strong edges first
    ++growths;
    std::vector<uint32_t>        root, off;
    std::vector<uint8_t>         parity;
    std::vector<OrientCrossEdge> edges;
    const auto tc0 = std::chrono::steady_clock::now();
    const bool okc = contractOnHost( n, knn, k, edgeDot, first, root, parity, off, edges );
    const auto tc1 = std::chrono::steady_clock::now();
    if ( ctxOption( ctx, "ORIENT_TIMING" ) ) {
      size_t clusters = 0;
      for ( size_t i = 0; okc && i < n; ++i ) clusters += root[i] == i;
      fprintf( stderr, "contraction %.1f ms ok=%d clusters %zu cross edges %zu of %zu\n",
               std::chrono::duration<double, std::milli>( tc1 - tc0 ).count(), int( okc ), clusters, edges.size(), n * size_t( k ) );
    }
    if ( okc ) {
      // (TMC2_ORIENT_HOST_WALK=points: the per-point-array walk over the full cross-edge list, kept as the cross-check)
      const char* walkEnv = ctxOption( ctx, "ORIENT_HOST_WALK" );
      const auto  tw0     = std::chrono::steady_clock::now();
      bool        okw;
      size_t      seedCount = 0;
      if ( walkEnv && walkEnv[0] == 'p' ) {
        const OrientContraction g{root.data(), parity.data(), off.data(), edges.data()};
        std::vector<int8_t>     clusterSign( n );
        std::vector<uint32_t>   component( n ), seeds;
        okw       = orientContractedSigns( n, g, first, clusterSign.data(), component.data(), seeds, scratch );
        seedCount = seeds.size();
        if ( okw ) {
          const auto rowOf    = [&]( size_t s ) { return knn + size_t( seeds[s] ) * k; };
          const auto normalOf = [&]( size_t s, int j ) {
            const uint32_t i = seeds[s];
            const uint32_t v = j == 0 ? i : ( j == 1 ? ( i ? i - 1 : 0 ) : knn[size_t( i ) * k + ( j - 2 )] );
            return normals + 3 * size_t( v );
          };
          resolveSeedSigns( n, g, k, seeds, component.data(), rowOf, normalOf, xyz, clusterSign.data() );
          for ( size_t i = 0; i < n; ++i ) sign[i] = int8_t( ( parity[i] & 1 ) ? -clusterSign[root[i]] : clusterSign[root[i]] );
        }
      } else {
        HostCompact hc;
        compactOnHost( n, root, parity, off, edges, first, hc );
        const OrientCompact   g = hc.view();
        std::vector<int8_t>   clusterSign( g.clusters + 1 );
        std::vector<uint32_t> component( g.clusters + 1 ), seeds, seedClusters;
        const auto tg0 = std::chrono::steady_clock::now();
        okw       = orientCompactSigns( g, first, clusterSign.data(), component.data(), seeds, seedClusters );
        if ( ctxOption( ctx, "ORIENT_TIMING" ) )
          fprintf( stderr, "  the growth over the compact graph alone: %.2f ms (%u clusters, %u edges)\n",
                   std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tg0 ).count(), g.clusters, g.rec[g.clusters].off );
        seedCount = seeds.size();
        if ( okw ) {
          // the seed rule's tables: (point, cluster, parity) of the point before the seed and of its neighbours; their normals
          std::vector<uint32_t> who( seeds.size() * size_t( k + 1 ) * 3 );
          std::vector<double>   nrm( seeds.size() * size_t( k + 2 ) * 3 );
          for ( size_t sd = 0; sd < seeds.size(); ++sd ) {
            const uint32_t i = seeds[sd];
            for ( int j = 0; j < k + 2; ++j ) {
              const uint32_t v = j == 0 ? i : ( j == 1 ? ( i ? i - 1 : 0 ) : knn[size_t( i ) * k + ( j - 2 )] );
              if ( j >= 1 ) {
                uint32_t* w = who.data() + ( sd * size_t( k + 1 ) + size_t( j - 1 ) ) * 3;
                w[0] = v, w[1] = hc.cid[v], w[2] = parity[v];
              }
              for ( int c = 0; c < 3; ++c ) nrm[( sd * size_t( k + 2 ) + size_t( j ) ) * 3 + c] = normals[3 * size_t( v ) + c];
            }
          }
          resolveSeedSignsCompact( g, k, seeds, seedClusters, component.data(), who.data(), nrm.data(), xyz, clusterSign.data() );
          for ( size_t i = 0; i < n; ++i ) sign[i] = int8_t( ( parity[i] & 1 ) ? -clusterSign[hc.cid[i]] : clusterSign[hc.cid[i]] );
        }
      }
      if ( ctxOption( ctx, "ORIENT_TIMING" ) )
        fprintf( stderr, "contracted walk %.1f ms ok=%d seeds %zu\n", std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tw0 ).count(), int( okw ), seedCount );
      if ( okw ) return growths;
    }
  }
  const double first = first0;
  for ( double tau : {first, 0.998} ) {
    if ( tau > 1.5 ) break;
    ++growths;
    if ( growSigns( xyz, n, knn, k, normals, edgeDot, sign, scratch, tau ) ) return growths;
  }
  growSigns( xyz, n, knn, k, normals, edgeDot, sign, scratch, 4.0 );  // no edge is "strong"
  return growths + 1;
}

// normals: [n][3] in/out (host), knn: [n][k] (host), xyz: [n][3]; scratch: n * 8 bytes or nullptr
void orientNormalsSpanningTree( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals, void* scratch ) {
  if ( n == 0 ) return;
  std::vector<double> edgeDotV( n * size_t( k ) );
  double*             edgeDot = edgeDotV.data();
  for ( size_t u = 0; u < n; ++u )
    for ( int j = 0; j < k; ++j ) edgeDot[u * k + j] = dot( normals + 3 * u, normals + 3 * size_t( knn[u * k + j] ) );
  std::vector<int8_t> sign( n );
  const auto          tt0 = std::chrono::steady_clock::now();
  const int growths = orientSpanningTreeSigns( xyz, n, knn, k, normals, edgeDot, sign.data(), scratch, true, nullptr );
  if ( ctxOption( nullptr, "ORIENT_TIMING" ) )  // test hook: time of the growth alone
    fprintf( stderr, "orient core %.1f ms (%d growth%s)\n",
             std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tt0 ).count(), growths, growths == 1 ? "" : "s" );
  size_t negCount = 0;
  for ( size_t i = 0; i < n; ++i ) {
    if ( sign[i] < 0 ) {
      normals[3 * i]     = -normals[3 * i];
      normals[3 * i + 1] = -normals[3 * i + 1];
      normals[3 * i + 2] = -normals[3 * i + 2];
    }
    const double toView[3] = {0.0 - xyz[3 * i], 0.0 - xyz[3 * i + 1], 0.0 - xyz[3 * i + 2]};
    if ( dot( normals + 3 * i, toView ) < 0.0 ) ++negCount;
  }
  if ( negCount > ( n + 1 ) / 2 )
    for ( size_t i = 0; i < 3 * n; ++i ) normals[i] = -normals[i];
}

int orientNormalsHost( tmc2_frame* f ) {
  if ( !f->haveKnn || !f->haveNormals ) {
    setError( "orientNormals: adjacency / normals not computed" );
    return TMC2_E_STATE;
  }
  const size_t n = f->n, edges = n * size_t( f->k );
  tmc2_ctx*    ctx = f->ctx;
  hipStream_t  s   = ctx->stream;
  DevBuf<double>   d_edgeDot;
  DevBuf<int8_t>   d_sign, d_clusterSign;
  DevBuf<uint32_t> d_negCount, d_root;
  DevBuf<uint8_t>  d_parity;
  TMC2_TRY( d_sign.alloc( n ) );
  TMC2_TRY( d_negCount.alloc( kOrientNegCountWords ) );
  if ( ctx->orientScratch.size() < 2 * n ) ctx->orientScratch.resize( 2 * n );
  // ---- fast path: contract on the device, walk the clusters on the host; a frame that is inconsistent at one threshold is
  // tried with the next (orientTauLadder) --------------------------------------------------------------------------------
  bool contracted = false;
  if ( orientFirstTau( ctx ) <= 1.5 && !ctxOption( ctx, "ORIENT_NO_CONTRACTION" ) )
  for ( const double tau : orientTauLadder( ctx ) ) {
    if ( tau != orientFirstTau( ctx ) ) ctx->stageAddHostMs( "orient_tau_retry", 0.0 );  // (counts the repeats with a tighter threshold)
    OrientCompact g{};
    const int     sid = ctx->stageBegin( "orient_contract" );
    TMC2_TRY( contractOrientationDevice( f, tau, d_root, d_parity, g, contracted ) );  // (d_root: cluster ids here)
    ctx->stageEnd( sid );
    if ( contracted ) {
      if ( f->beforeHostWalk ) {  // device work that overlaps the walk (once, whatever the number of thresholds tried)
        TMC2_TRY( f->beforeHostWalk() );
        f->beforeHostWalk = nullptr;
      }
      const uint32_t        C           = g.clusters;
      int8_t*               clusterSign = reinterpret_cast<int8_t*>( ctx->hostC.get<uint32_t>( 4 + ( n + 4 ) / 4 + 4 ) + 4 );  // (behind the counters)
      std::vector<uint32_t> seeds, seedClusters, component( C );
      bool                  ok;
      {
        HostGate   gate( ctx );
        const auto t0 = std::chrono::steady_clock::now();
        ok            = orientCompactSigns( g, tau, clusterSign, component.data(), seeds, seedClusters );
        const auto t1 = std::chrono::steady_clock::now();
        ctx->stageAddHostMs( "orient_normals_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
      }
      if ( ok ) {
        // the seed rule's inputs, gathered on the device for exactly the seeds the walk has listed (a few KB)
        std::vector<uint32_t> who;
        std::vector<double>   seedNormals;
        TMC2_TRY( gatherSeedTables( f, d_root.p, d_parity.p, seeds, who, seedNormals ) );
        resolveSeedSignsCompact( g, f->k, seeds, seedClusters, component.data(), who.data(), seedNormals.data(), f->h_xyz.data(), clusterSign );
        TMC2_TRY( d_clusterSign.alloc( std::max<uint32_t>( C, 1u ) ) );
        TMC2_HIP( hipMemcpyAsync( d_clusterSign.p, clusterSign, C, hipMemcpyHostToDevice, s ) );
        TMC2_TRY( launchClusterSigns( f, d_root.p, d_parity.p, d_clusterSign.p, d_sign.p ) );
        TMC2_TRY( launchApplyOrientation( f, d_sign.p, d_negCount.p ) );
        TMC2_HIP( hipStreamSynchronize( s ) );
        return TMC2_OK;
      }
    }
  }
  if ( orientFirstTau( ctx ) <= 1.5 && !ctxOption( ctx, "ORIENT_NO_CONTRACTION" ) )
    ctx->stageAddHostMs( "orient_normals_regrowth", 0.0 );  // counts the frames that needed the point-level walk

  // ---- point-level walk: rows, dot products and normals to the host ----------------------------------------------------
  // (the 16 N dot products exist only here: the contraction works on their classes as bits and recomputes the few values it needs)
  TMC2_TRY( d_edgeDot.alloc( edges ) );
  TMC2_TRY( launchEdgeDots( f, d_edgeDot.p ) );
  uint32_t* knn  = ctx->hostA.get<uint32_t>( edges );
  double*   nrm  = ctx->hostB.get<double>( n * 3 );
  double*   dots = ctx->hostE.get<double>( edges );
  int8_t*   sign = ctx->hostC.get<int8_t>( n );
  if ( !knn || !nrm || !dots || !sign ) {
    setError( "orientNormals: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  TMC2_HIP( hipMemcpyAsync( knn, f->d_knn.p, edges * sizeof( uint32_t ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( dots, d_edgeDot.p, edges * sizeof( double ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( nrm, f->d_normals.p, n * 3 * sizeof( double ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  std::chrono::steady_clock::time_point t0, t1;
  {
    HostGate gate( ctx );
    t0 = std::chrono::steady_clock::now();
    orientSpanningTreeSigns( f->h_xyz.data(), n, knn, f->k, nrm, dots, sign, ctx->orientScratch.data(), false, ctx );
    t1 = std::chrono::steady_clock::now();
  }
  ctx->stageAddHostMs( "orient_normals_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
  TMC2_HIP( hipMemcpyAsync( d_sign.p, sign, n, hipMemcpyHostToDevice, s ) );
  TMC2_TRY( launchApplyOrientation( f, d_sign.p, d_negCount.p ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

}  // namespace tmc2
