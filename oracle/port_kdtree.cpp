// oracle/port_kdtree.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// Restates the k-d tree the reference uses for every neighbour query on the hot path:
//   PCCKdTree (source/lib/PccLibCommon/source/PCCKdTree.cpp:40-79) =
//   nanoflann KDTreeSingleIndexAdaptor<L2_Simple, int16 coords, leaf size 10>
//   (dependencies/nanoflann/nanoflann.hpp: build 1041-1181, query 1186-1254, result sets 79-205).
// Parity matters down to the ORDER of equal-distance neighbours, which is a
// function of (a) the exact permutation the two-pass plane split leaves in the index array,
// (b) the node bounds divlow/divhigh after bottom-up tightening, (c) near-child-first traversal and
// (d) "insert after equal distances / reject when equal to the current worst" in the k-NN list.
// All arithmetic is exact: coordinates are < 2^12, squared distances < 2^26 (held in int64 here; the
// reference holds the same integers in float/double).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "oracle.h"

namespace {

struct Node {
  int32_t  dim;              // -1: leaf
  int32_t  divlow, divhigh;  // inner: upper bound of the left child's box / lower bound of the right child's box
  uint32_t a, b;             // leaf: [a,b) into perm;  inner: child node ids (near/far decided per query)
};

struct Box {
  int32_t lo[3], hi[3];
};

}  // namespace

struct orc_kdtree {
  std::vector<int16_t>  xyz;
  size_t                n = 0;
  std::vector<uint32_t> perm;
  std::vector<Node>     nodes;
  Box                   root;
  static const int      LEAF = 10;

  int32_t coord( uint32_t p, int d ) const { return xyz[3 * size_t( p ) + d]; }

  // nanoflann.hpp:1154-1181 -- two-pass in-place partition; the swap sequence is part of the contract.
  void planeSplit( uint32_t* ind, size_t count, int dim, int32_t cut, size_t& lim1, size_t& lim2 ) {
    size_t left = 0, right = count - 1;
    for ( ;; ) {
      while ( left <= right && coord( ind[left], dim ) < cut ) ++left;
      while ( right && left <= right && coord( ind[right], dim ) >= cut ) --right;
      if ( left > right || !right ) break;
      std::swap( ind[left], ind[right] );
      ++left;
      --right;
    }
    lim1  = left;
    right = count - 1;
    for ( ;; ) {
      while ( left <= right && coord( ind[left], dim ) <= cut ) ++left;
      while ( right && left <= right && coord( ind[right], dim ) > cut ) --right;
      if ( left > right || !right ) break;
      std::swap( ind[left], ind[right] );
      ++left;
      --right;
    }
    lim2 = left;
  }

  // nanoflann.hpp:1041-1142
  uint32_t divide( size_t left, size_t right, Box& box ) {
    const uint32_t id = uint32_t( nodes.size() );
    nodes.push_back( Node() );
    if ( right - left <= size_t( LEAF ) ) {
      for ( int d = 0; d < 3; ++d ) box.lo[d] = box.hi[d] = coord( perm[left], d );
      for ( size_t k = left + 1; k < right; ++k )
        for ( int d = 0; d < 3; ++d ) {
          const int32_t v = coord( perm[k], d );
          if ( box.lo[d] > v ) box.lo[d] = v;
          if ( box.hi[d] < v ) box.hi[d] = v;
        }
      nodes[id].dim = -1;
      nodes[id].a   = uint32_t( left );
      nodes[id].b   = uint32_t( right );
      return id;
    }
    uint32_t*    ind   = perm.data() + left;
    const size_t count = right - left;
    // choose the cut dimension: among dims whose box span is (numerically) the largest span, the one
    // whose points have the largest actual spread; first wins.
    int32_t maxSpan = box.hi[0] - box.lo[0];
    for ( int d = 1; d < 3; ++d ) maxSpan = std::max( maxSpan, box.hi[d] - box.lo[d] );
    int     cutDim    = 0;
    int32_t maxSpread = -1;
    for ( int d = 0; d < 3; ++d ) {
      const int32_t span = box.hi[d] - box.lo[d];
      if ( double( span ) > ( 1.0 - 0.00001 ) * double( maxSpan ) ) {
        int32_t mn = coord( ind[0], d ), mx = mn;
        for ( size_t k = 1; k < count; ++k ) {
          const int32_t v = coord( ind[k], d );
          mn              = std::min( mn, v );
          mx              = std::max( mx, v );
        }
        if ( mx - mn > maxSpread ) {
          cutDim    = d;
          maxSpread = mx - mn;
        }
      }
    }
    const int32_t mid = ( box.lo[cutDim] + box.hi[cutDim] ) / 2;  // integer midpoint of the BOX interval
    int32_t       mn = coord( ind[0], cutDim ), mx = mn;
    for ( size_t k = 1; k < count; ++k ) {
      const int32_t v = coord( ind[k], cutDim );
      mn              = std::min( mn, v );
      mx              = std::max( mx, v );
    }
    const int32_t cut = mid < mn ? mn : ( mid > mx ? mx : mid );
    size_t        lim1, lim2;
    planeSplit( ind, count, cutDim, cut, lim1, lim2 );
    size_t idx;
    if ( lim1 > count / 2 )
      idx = lim1;
    else if ( lim2 < count / 2 )
      idx = lim2;
    else
      idx = count / 2;

    Box lb         = box;
    lb.hi[cutDim]  = cut;
    const uint32_t c1 = divide( left, left + idx, lb );
    Box rb         = box;
    rb.lo[cutDim]  = cut;
    const uint32_t c2 = divide( left + idx, right, rb );
    nodes[id].dim     = cutDim;
    nodes[id].divlow  = lb.hi[cutDim];
    nodes[id].divhigh = rb.lo[cutDim];
    nodes[id].a       = c1;
    nodes[id].b       = c2;
    for ( int d = 0; d < 3; ++d ) {
      box.lo[d] = std::min( lb.lo[d], rb.lo[d] );
      box.hi[d] = std::max( lb.hi[d], rb.hi[d] );
    }
    return id;
  }

  void build( const int16_t* p, size_t count ) {
    n = count;
    xyz.assign( p, p + 3 * count );
    perm.resize( n );
    for ( size_t i = 0; i < n; ++i ) perm[i] = uint32_t( i );
    nodes.clear();
    nodes.reserve( n / 4 + 16 );
    if ( n == 0 ) return;
    for ( int d = 0; d < 3; ++d ) root.lo[d] = root.hi[d] = coord( 0, d );
    for ( size_t k = 1; k < n; ++k )
      for ( int d = 0; d < 3; ++d ) {
        const int32_t v = coord( uint32_t( k ), d );
        if ( v < root.lo[d] ) root.lo[d] = v;
        if ( v > root.hi[d] ) root.hi[d] = v;
      }
    Box b = root;
    divide( 0, n, b );
  }

  int64_t dist2( const int16_t* q, uint32_t p ) const {
    int64_t s = 0;
    for ( int d = 0; d < 3; ++d ) {
      const int64_t t = int64_t( q[d] ) - coord( p, d );
      s += t * t;
    }
    return s;
  }

  // ---- k-NN (nanoflann.hpp KNNResultSet 79-131, searchLevel 1207-1254) -------------------------
  struct Knn {
    int       k, count;
    int64_t*  d;
    uint32_t* i;
    int64_t   worst() const { return d[k - 1]; }
    void      add( int64_t dist, uint32_t index ) {
      int j;
      for ( j = count; j > 0; --j ) {
        if ( d[j - 1] > dist ) {
          if ( j < k ) {
            d[j] = d[j - 1];
            i[j] = i[j - 1];
          }
        } else
          break;
      }
      if ( j < k ) {
        d[j] = dist;
        i[j] = index;
      }
      if ( count < k ) ++count;
    }
  };

  void searchKnn( Knn& rs, const int16_t* q, uint32_t node, int64_t mind, int64_t* dd ) const {
    const Node& nd = nodes[node];
    if ( nd.dim < 0 ) {
      const int64_t worst = rs.worst();  // sampled once per leaf, as the reference does
      for ( uint32_t k = nd.a; k < nd.b; ++k ) {
        const uint32_t p  = perm[k];
        const int64_t  ds = dist2( q, p );
        if ( ds < worst ) rs.add( ds, p );
      }
      return;
    }
    const int     dim = nd.dim;
    const int64_t v = q[dim], d1 = v - nd.divlow, d2 = v - nd.divhigh;
    uint32_t      nearC, farC;
    int64_t       cut;
    if ( d1 + d2 < 0 ) {
      nearC = nd.a;
      farC  = nd.b;
      cut   = d2 * d2;
    } else {
      nearC = nd.b;
      farC  = nd.a;
      cut   = d1 * d1;
    }
    searchKnn( rs, q, nearC, mind, dd );
    const int64_t saved = dd[dim];
    mind                = mind + cut - saved;
    dd[dim]             = cut;
    if ( mind <= rs.worst() ) searchKnn( rs, q, farC, mind, dd );
    dd[dim] = saved;
  }

  void initialDists( const int16_t* q, int64_t* dd, int64_t& sum ) const {
    sum = 0;
    for ( int d = 0; d < 3; ++d ) {
      dd[d] = 0;
      if ( q[d] < root.lo[d] ) {
        dd[d] = int64_t( q[d] - root.lo[d] ) * ( q[d] - root.lo[d] );
        sum += dd[d];
      }
      if ( q[d] > root.hi[d] ) {
        dd[d] = int64_t( q[d] - root.hi[d] ) * ( q[d] - root.hi[d] );
        sum += dd[d];
      }
    }
  }

  int knn( const int16_t* q, int k, uint32_t* idx, int64_t* dist ) const {
    Knn rs = {k, 0, dist, idx};
    dist[k - 1] = INT64_MAX;
    int64_t dd[3], sum;
    initialDists( q, dd, sum );
    searchKnn( rs, q, 0, sum, dd );
    return rs.count;
  }

  // ---- radius search (nanoflann.hpp RadiusResultSet 143-190, radiusSearch 945-952) -------------
  void searchRadius( std::vector<std::pair<int64_t, uint32_t>>& out, int64_t r2, const int16_t* q, uint32_t node,
                     int64_t mind, int64_t* dd ) const {
    const Node& nd = nodes[node];
    if ( nd.dim < 0 ) {
      for ( uint32_t k = nd.a; k < nd.b; ++k ) {
        const int64_t ds = dist2( q, perm[k] );
        if ( ds < r2 ) out.emplace_back( ds, perm[k] );
      }
      return;
    }
    const int     dim = nd.dim;
    const int64_t v = q[dim], d1 = v - nd.divlow, d2 = v - nd.divhigh;
    uint32_t      nearC, farC;
    int64_t       cut;
    if ( d1 + d2 < 0 ) {
      nearC = nd.a;
      farC  = nd.b;
      cut   = d2 * d2;
    } else {
      nearC = nd.b;
      farC  = nd.a;
      cut   = d1 * d1;
    }
    searchRadius( out, r2, q, nearC, mind, dd );
    const int64_t saved = dd[dim];
    mind                = mind + cut - saved;
    dd[dim]             = cut;
    if ( mind <= r2 ) searchRadius( out, r2, q, farC, mind, dd );
    dd[dim] = saved;
  }
};

extern "C" {

orc_kdtree* orc_kdtree_build( const int16_t* xyz, size_t n ) {
  orc_kdtree* t = new orc_kdtree();
  t->build( xyz, n );
  return t;
}
void orc_kdtree_free( orc_kdtree* t ) { delete t; }

size_t orc_kdtree_node_count( const orc_kdtree* t ) { return t->nodes.size(); }
const uint32_t* orc_kdtree_perm( const orc_kdtree* t ) { return t->perm.data(); }

// k nearest neighbours of nq query points; idx[nq*k] (u32), dist[nq*k] (f64, may be NULL).
int orc_knn( const orc_kdtree* t, const int16_t* q, size_t nq, int k, uint32_t* idx, double* dist ) {
  if ( size_t( k ) > t->n || k < 1 || k > 64 ) return -1;
  int64_t  d[64];
  uint32_t id[64];
  for ( size_t i = 0; i < nq; ++i ) {
    t->knn( q + 3 * i, k, id, d );
    for ( int j = 0; j < k; ++j ) {
      idx[i * k + j] = id[j];
      if ( dist ) dist[i * k + j] = double( d[j] );
    }
  }
  return 0;
}

// all neighbours with squared distance < radius2, sorted by (distance, index), truncated to cap.
int orc_radius( const orc_kdtree* t, const int16_t* q, size_t nq, double radius2, int cap, int32_t* count,
                uint32_t* idx ) {
  std::vector<std::pair<int64_t, uint32_t>> out;
  // "dist < radius" on integer distances with a real radius: dist < ceil(radius)
  int64_t r2 = int64_t( radius2 );
  if ( double( r2 ) < radius2 ) ++r2;
  for ( size_t i = 0; i < nq; ++i ) {
    out.clear();
    int64_t dd[3], sum;
    t->initialDists( q + 3 * i, dd, sum );
    t->searchRadius( out, r2, q + 3 * i, 0, sum, dd );
    std::sort( out.begin(), out.end() );
    const size_t m = std::min( out.size(), size_t( cap ) );
    count[i]       = int32_t( m );
    for ( size_t j = 0; j < m; ++j ) idx[i * size_t( cap ) + j] = out[j].second;
  }
  return 0;
}

}  // extern "C"
