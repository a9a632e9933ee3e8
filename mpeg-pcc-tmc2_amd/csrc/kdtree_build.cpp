// kdtree_build.cpp -- host-side construction of the k-d tree that the device kernels traverse.
//
// Replaces PCCKdTree::init (reference: source/lib/PccLibCommon/source/PCCKdTree.cpp:56-59), i.e.
// nanoflann's KDTreeSingleIndexAdaptor::buildIndex / divideTree / middleSplit_ / planeSplit
// (dependencies/nanoflann/nanoflann.hpp:858-866, 1041-1181) for int16 3-D points and leaf size 10.
//
// The tree must be IDENTICAL to nanoflann's, not merely equivalent: voxelised clouds are full of
// distance ties and the reference's neighbour order/selection under ties follows the leaf visiting
// order and the in-leaf point order.  So this builder reproduces
//   * the split rule (box-midpoint clamped to the point range, on the widest-box / widest-spread dim),
//   * the permutation left behind by the two-pass "Hoare" partition (swap for swap),
//   * the balance fallback (count/2) and the bottom-up tightening that defines divlow/divhigh.
// Unlike the reference it is iterative (explicit frame stack), gathers the three per-dimension ranges
// of a node in ONE pass, emits a flat pre-order node array ready for upload, and carries the points
// along with the index permutation (8-byte records swapped together), so that every pass over a node
// streams through contiguous memory instead of gathering coordinates through the index array -- the
// build is memory-latency bound otherwise.
#include <algorithm>

#include "internal.h"

namespace tmc2 {

namespace {
struct Box3 {
  int32_t lo[3], hi[3];
};
struct Frame {
  uint32_t begin, end;   // range in perm
  uint32_t node;         // node id
  int32_t  parent;       // frame index of the parent (-1 root)
  uint8_t  side;         // 0: we are the parent's left child, 1: right child
  uint8_t  state;        // 0 fresh, 1 left child done, 2 right child done
  int16_t  cutDim;
  int32_t  cut;
  uint32_t mid;          // begin + idx
  int32_t  depth;
  Box3     box;          // in: loose box handed down; out: tight box handed up
  Box3     lb, rb;       // child boxes (tightened by the children)
};

// two-pass partition; returns lim1/lim2.
// nanoflann runs two Hoare sweeps (first with "< cut" on the left, then "<= cut" on the left of what remained).  A
// Hoare sweep ends with every left-class element in [0, nL) and has swapped, in order, the i-th misplaced element from
// the left with the i-th misplaced element from the right; elements already on their side never move.  Counting the
// class, listing the misplaced positions of both sides and swapping them pairwise therefore leaves EXACTLY the same
// permutation -- but as three branch-free streaming loops instead of two sweeps of coin-flip branches (the same
// closed form the device builder uses, kdtree_device.hip).
// one sweep over [0, count) whose left class is "value < c" and holds nL elements
inline void hoareSweep( uint32_t* ind, Pt* pts, uint32_t count, uint32_t nL, int dim, int32_t c, uint32_t* scratch ) {
  const int16_t* v         = &pts[0].x + dim;  // stride 4 int16
  uint32_t*      fromLeft  = scratch;           // misplaced (right-class) positions in [0, nL), ascending
  uint32_t*      fromRight = scratch + nL + 1;  // misplaced (left-class) positions in [nL, count), descending
  uint32_t       m = 0, m2 = 0;
  for ( uint32_t i = 0; i < nL; ++i ) {
    fromLeft[m] = i;
    m += uint32_t( int32_t( v[4 * size_t( i )] ) >= c );
  }
  for ( uint32_t j = count; j-- > nL; ) {
    fromRight[m2] = j;
    m2 += uint32_t( int32_t( v[4 * size_t( j )] ) < c );
  }
  for ( uint32_t t = 0; t < m; ++t ) {
    const uint32_t a = fromLeft[t], b = fromRight[t];
    const uint32_t x = ind[a];
    ind[a]           = ind[b];
    ind[b]           = x;
    const Pt q       = pts[a];
    pts[a]           = pts[b];
    pts[b]           = q;
  }
}
inline void partitionTwoPass( uint32_t* ind, Pt* pts, uint32_t count, int dim, int32_t cut, uint32_t& lim1, uint32_t& lim2,
                              uint32_t* scratch ) {
  const int16_t* v  = &pts[0].x + dim;
  uint32_t       lt = 0, le = 0;  // both class sizes in one pass: every "< cut" element is also "<= cut"
  for ( uint32_t i = 0; i < count; ++i ) {
    const int32_t x = v[4 * size_t( i )];
    lt += uint32_t( x < cut );
    le += uint32_t( x <= cut );
  }
  lim1 = lt;
  lim2 = le;
  hoareSweep( ind, pts, count, lt, dim, cut, scratch );
  hoareSweep( ind + lt, pts + lt, count - lt, le - lt, dim, cut + 1, scratch );
}
}  // namespace

void KdTreeHost::build( const int16_t* xyz, size_t n ) {
  perm.resize( n );
  ptsTree.resize( n );
  for ( size_t i = 0; i < n; ++i ) ptsTree[i] = Pt{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0};
  buildInPlace( ptsTree.data(), perm.data(), n );
}

// pts: the points in their original order on entry, in tree order on return; ind receives the permutation.
// (The device-facing callers pass page-locked staging here, so that the result is uploaded without another copy.)
void KdTreeHost::buildInPlace( Pt* ptsIO, uint32_t* ind, size_t n ) {
  for ( size_t i = 0; i < n; ++i ) ind[i] = uint32_t( i );
  nodes.clear();
  nodes.reserve( n / 4 + 16 );
  depth = 0;
  if ( n == 0 ) return;
  Box3 root;
  root.lo[0] = root.hi[0] = ptsIO[0].x, root.lo[1] = root.hi[1] = ptsIO[0].y, root.lo[2] = root.hi[2] = ptsIO[0].z;
  for ( size_t i = 1; i < n; ++i ) {
    const Pt p = ptsIO[i];
    root.lo[0] = std::min<int32_t>( root.lo[0], p.x ), root.hi[0] = std::max<int32_t>( root.hi[0], p.x );
    root.lo[1] = std::min<int32_t>( root.lo[1], p.y ), root.hi[1] = std::max<int32_t>( root.hi[1], p.y );
    root.lo[2] = std::min<int32_t>( root.lo[2], p.z ), root.hi[2] = std::max<int32_t>( root.hi[2], p.z );
  }
  for ( int d = 0; d < 3; ++d ) {
    lo[d] = root.lo[d];
    hi[d] = root.hi[d];
  }
  std::vector<uint32_t> scratch( n + 2 );  // the two position lists of a sweep: nL + 1 and count - nL + 1 entries... at most count + 2
  std::vector<Frame>    stack;
  stack.reserve( 128 );
  {
    Frame f{};
    f.begin  = 0;
    f.end    = uint32_t( n );
    f.parent = -1;
    f.state  = 0;
    f.depth  = 1;
    f.box    = root;
    stack.push_back( f );
  }
  auto finish = [&]( Frame& f ) {  // hand the tight box up and pop
    if ( f.parent >= 0 ) {
      Frame& p = stack[f.parent];
      ( f.side == 0 ? p.lb : p.rb ) = f.box;
    }
    stack.pop_back();
  };
  while ( !stack.empty() ) {
    const size_t fi = stack.size() - 1;
    Frame&       f  = stack[fi];
    if ( f.state == 0 ) {
      f.node = uint32_t( nodes.size() );
      nodes.push_back( KdNode{} );
      depth                = std::max( depth, f.depth );
      const uint32_t count = f.end - f.begin;
      uint32_t*      seg   = ind + f.begin;
      Pt*            pts   = ptsIO + f.begin;
      // one pass: actual range of the node's points in all three dims
      int32_t mn[3] = {pts[0].x, pts[0].y, pts[0].z}, mx[3] = {pts[0].x, pts[0].y, pts[0].z};
      for ( uint32_t i = 1; i < count; ++i ) {
        const Pt p = pts[i];
        mn[0] = std::min<int32_t>( mn[0], p.x ), mx[0] = std::max<int32_t>( mx[0], p.x );
        mn[1] = std::min<int32_t>( mn[1], p.y ), mx[1] = std::max<int32_t>( mx[1], p.y );
        mn[2] = std::min<int32_t>( mn[2], p.z ), mx[2] = std::max<int32_t>( mx[2], p.z );
      }
      if ( count <= 10 ) {
        KdNode& nd = nodes[f.node];
        nd.a       = int32_t( f.begin );
        nd.b       = int32_t( f.end );
        nd.dim     = -1;
        nd.divlow = nd.divhigh = 0;
        for ( int d = 0; d < 3; ++d ) {
          f.box.lo[d] = mn[d];
          f.box.hi[d] = mx[d];
        }
        finish( f );
        continue;
      }
      // cut dimension: among dims whose BOX span equals the widest box span (the reference's
      // "span > (1-1e-5)*max_span" in double, evaluated literally), the largest point spread; first wins.
      int32_t maxSpan = 0;
      for ( int d = 0; d < 3; ++d ) maxSpan = std::max( maxSpan, f.box.hi[d] - f.box.lo[d] );
      int     cutDim    = 0;
      int32_t bestSpread = -1;
      for ( int d = 0; d < 3; ++d ) {
        const int32_t span = f.box.hi[d] - f.box.lo[d];
        if ( double( span ) > ( 1.0 - 0.00001 ) * double( maxSpan ) ) {
          const int32_t spread = mx[d] - mn[d];
          if ( spread > bestSpread ) {
            bestSpread = spread;
            cutDim     = d;
          }
        }
      }
      const int32_t mid = ( f.box.lo[cutDim] + f.box.hi[cutDim] ) / 2;
      const int32_t cut = std::min( std::max( mid, mn[cutDim] ), mx[cutDim] );
      uint32_t      lim1, lim2;
      partitionTwoPass( seg, pts, count, cutDim, cut, lim1, lim2, scratch.data() );
      const uint32_t half = count / 2;
      const uint32_t idx  = lim1 > half ? lim1 : ( lim2 < half ? lim2 : half );
      f.cutDim            = int16_t( cutDim );
      f.cut               = cut;
      f.mid               = f.begin + idx;
      f.state             = 1;
      Frame c{};
      c.begin            = f.begin;
      c.end              = f.mid;
      c.parent           = int32_t( fi );
      c.side             = 0;
      c.state            = 0;
      c.depth            = f.depth + 1;
      c.box              = f.box;
      c.box.hi[cutDim]   = cut;
      stack.push_back( c );  // invalidates f
    } else if ( f.state == 1 ) {
      f.state = 2;
      Frame c{};
      c.begin            = f.mid;
      c.end              = f.end;
      c.parent           = int32_t( fi );
      c.side             = 1;
      c.state            = 0;
      c.depth            = f.depth + 1;
      c.box              = f.box;
      c.box.lo[f.cutDim] = f.cut;
      stack.push_back( c );
    } else {
      KdNode& nd = nodes[f.node];
      nd.a       = int32_t( f.node + 1 );
      // nd.b (right child id) is filled by the subtree-size sweep after the loop
      nd.dim     = f.cutDim;
      nd.divlow  = int16_t( f.lb.hi[f.cutDim] );
      nd.divhigh = int16_t( f.rb.lo[f.cutDim] );
      for ( int d = 0; d < 3; ++d ) {
        f.box.lo[d] = std::min( f.lb.lo[d], f.rb.lo[d] );
        f.box.hi[d] = std::max( f.lb.hi[d], f.rb.hi[d] );
      }
      finish( f );
    }
  }
  // right-child ids: in a pre-order layout the right child of an inner node is the node that follows
  // the last node of its left subtree.  One reverse sweep computes subtree sizes.
  std::vector<uint32_t> subtree( nodes.size(), 1 );
  for ( size_t i = nodes.size(); i-- > 0; ) {
    if ( nodes[i].dim >= 0 ) {
      const uint32_t l = uint32_t( i ) + 1;
      const uint32_t r = l + subtree[l];
      nodes[i].b       = int32_t( r );
      subtree[i]       = 1 + subtree[l] + subtree[r];
    }
  }
}

}  // namespace tmc2
