// knn.hip -- exact nanoflann-order k-NN on gfx950 (MI355X).
//
// Replaces the N x PCCKdTree::search(k) loops of the hot path
//   PCCNormalsGenerator3::computeNormals   (PccLibEncoder/source/PCCNormalsGenerator.cpp:158-185, k=16)
//   PCCNormalsGenerator3::addNeighbors     (:521-548, k=16, same lists)
//   PCCPatchSegmenter3::computeAdjacencyInfo (PCCPatchSegmenter.cpp:267-291, k=16, same lists)
//   PCCPointSet3::transferColors k=8 / k=1 searches (PccLibCommon/source/PCCPointSet.cpp:807-1124)
// i.e. nanoflann KDTreeSingleIndexAdaptor::findNeighbors / searchLevel / KNNResultSet::addPoint
// (dependencies/nanoflann/nanoflann.hpp:901-915, 1207-1254, 110-131).
//
// Design (MI355X-first, integer/HBM path -- no MFMA):
//   * one query per lane; queries are issued in TREE order, so the 64 lanes of a wavefront are spatial
//     neighbours, walk almost the same node sequence and hit the same leaves: node records (16 B,
//     one dwordx4 load) and leaf points (8 B each) are wave-uniform or near-uniform loads served by
//     L1/L2; the only HBM streams are the query points (8 B/pt) and the result rows (4k B/pt).
//   * the k-best list lives in VGPRs (fully unrolled insertion = the reference's "insert after equal
//     distances, reject when equal to the current worst" rule); the far-child stack lives in scratch.
//   * the recursion of searchLevel is unrolled into near-descend / far-stack form.  A far child is
//     visited iff mindist <= worst AT THE TIME THE NEAR SUBTREE HAS BEEN FINISHED -- exactly when it
//     is popped.  Because worst never grows, entries that already fail at push time can be dropped.
//   * the far-child stack is the memory hog: every query pushes ~one entry per tree level on its first descent.  As a
//     16-byte scratch record that is ~1 KB of private memory per lane, ~200 MB live across the chip: it streams
//     through L2 and evicts the (small, shared) tree, and the counters show 56x the algorithmic bytes.  So the
//     stack lives in LDS, packed to 8 bytes per entry: every per-dimension distance on the path is the SQUARE of an
//     integer offset to a split / box plane, so the entry keeps node id (22 bits) + three offsets (14 bits each).
//     A full-depth LDS stack (tree depth x 2 KiB per block) would cap occupancy at 3 waves per SIMD, and this kernel
//     lives on latency hiding.  But the stack is only deep during the first descent (unbounded list: one pending far
//     child per level); once the list is full it is purged of everything that already fails the bound and stays a
//     few entries deep.  So slots 0..kLdsTop-1 are in LDS (16 KiB per block, occupancy back at the register limit) and
//     the slots above spill to packed private memory, written and read once, while the wave's lanes are in step.
//     Callers that cannot bound their coordinates (|offset| < 16384) or their tree (< 4 M nodes, <= kLdsLevels levels)
//     get the scratch-stack kernel instead (same traversal, same results).
//   * all arithmetic is int32: coordinates < 2^12, squared distances < 2^26 (the reference holds the
//     same integers in float/double -- exact below 2^24 for <= 11-bit data, KDTreeVectorOfVectorsAdaptor.h:126).
#include <algorithm>

#include "internal.h"

namespace tmc2 {

namespace {

constexpr int      kMaxStack  = 64;
constexpr int      kLdsLevels = 40;  // deepest tree the packed-stack kernel takes
constexpr int      kLdsTop    = 8;   // stack slots kept in LDS (the top of the stack)
constexpr uint32_t kInf       = 0x7FFFFFFFu;

struct RootBox {
  int lo[3], hi[3];
};

template <int K>
__device__ __forceinline__ void knnInsert( uint32_t ( &bd )[K], uint32_t ( &bi )[K], uint32_t dist, uint32_t index ) {
  // precondition: dist < bd[K-1].  New entry goes after every entry with bd <= dist.  Branch-free, top slot first (a slot
  // reads the OLD value of the one below it): with the list sorted, the new bd[j] is the median of ( bd[j-1], dist, bd[j] ) --
  // bd[j] where it stays, dist where the entry lands, bd[j-1] above that -- and the index follows the same two comparisons.
  bool keep = bd[K - 1] <= dist;  // (false)
#pragma unroll
  for ( int j = K - 1; j > 0; --j ) {
    const bool keepBelow = bd[j - 1] <= dist;
    bi[j]                = keep ? bi[j] : ( keepBelow ? index : bi[j - 1] );
    bd[j]                = min( max( bd[j - 1], dist ), bd[j] );
    keep                 = keepBelow;
  }
  bi[0] = keep ? bi[0] : index;
  bd[0] = min( bd[0], dist );
}

// SELF = true : queries are the tree-order points themselves, row j is written to out[perm[j]]
// SELF = false: queries come from `queries` (any order), row j is written to out[j]
// LDS = true: far-child stack in LDS, 8-byte packed entries (see the header); false: 16-byte entries in scratch
// nqLive: non-null = the number of queries lives on the device (a compacted list: launchKnnSplit); rowMap: non-null = row j is
// written to out[rowMap[j]]
constexpr uint32_t kHardQuery = 0xFFFFFFFFu;
template <int K, bool SELF, bool LDS>
__global__ __launch_bounds__( 256 ) void knnKernel( const Pt* __restrict__ ptsTree, const uint32_t* __restrict__ perm,
                                                     const KdNode* __restrict__ nodes, RootBox root,
                                                     const Pt* __restrict__ queries, uint32_t nq,
                                                     uint32_t* __restrict__ outIdx, uint32_t* __restrict__ outDist,
                                                     int xcdAware, uint32_t nTree, int nodeBits,
                                                     const uint32_t* __restrict__ nqLive, const uint32_t* __restrict__ rowMap ) {
  if ( nqLive ) nq = min( nq, *nqLive );
  // packed far-child entry (LDS form): node id in the low nodeBits, three offsets of ( 64 - nodeBits ) / 3 bits above it --
  // 22 + 3 x 14 for trees of up to 2^21 points and queries within [-4096, 12287]; 25 + 3 x 13 for larger trees (the vox11
  // frames: 3 M points) whose queries lie inside [0, 8191] like the tree's box (offsets < 2^13)
  const int      offBits = ( 64 - nodeBits ) / 3, sh0 = nodeBits, sh1 = nodeBits + offBits, sh2 = nodeBits + 2 * offBits;
  const uint32_t nodeMask = ( 1u << nodeBits ) - 1u, offMask = ( 1u << offBits ) - 1u;
  (void)sh0, (void)sh1, (void)sh2, (void)nodeMask, (void)offMask;
  // Which 256 queries this workgroup takes.  Workgroups are handed to the eight XCDs round-robin (block b runs on XCD b % 8:
  // observed, not promised -- only speed depends on it), so consecutive blocks -- neighbours in tree order, walking the same
  // part of the tree -- land on eight different L2s, and every L2 ends up streaming the whole tree (9.4 MB at longdress size
  // against 4 MB of L2: 9.5 x the algorithmic bytes reached HBM).  With the mapping below XCD x works through the x-th eighth
  // of the queries: its L2 holds an eighth of the tree.  (The grid is a multiple of 8 blocks; blocks past the end leave.)
  uint32_t block = blockIdx.x;
  if ( xcdAware ) {
    // (a compacted list: the eighths are eighths of the LIVE blocks, the grid was sized for the worst case)
    const uint32_t perXcd = nqLive ? ( ( nq + 255u ) / 256u + 7u ) >> 3 : gridDim.x >> 3;
    if ( ( blockIdx.x >> 3 ) >= perXcd ) return;
    block = ( blockIdx.x & 7u ) * perXcd + ( blockIdx.x >> 3 );
  }
  const uint32_t j = block * blockDim.x + threadIdx.x;
  if ( j >= nq ) return;
  const Pt  qp = SELF ? ptsTree[j] : queries[j];
  const int qx = qp.x, qy = qp.y, qz = qp.z;

  uint32_t bd[K], bi[K];
#pragma unroll
  for ( int i = 0; i < K; ++i ) {
    bd[i] = kInf;
    bi[i] = 0;
  }
  // offset of the query to the root box, per dimension (nanoflann computeInitialDistances); the per-dimension
  // distances are the squares of these offsets all the way down
  int o0 = 0, o1 = 0, o2 = 0;
  if ( qx < root.lo[0] ) o0 = root.lo[0] - qx;
  if ( qx > root.hi[0] ) o0 = qx - root.hi[0];
  if ( qy < root.lo[1] ) o1 = root.lo[1] - qy;
  if ( qy > root.hi[1] ) o1 = qy - root.hi[1];
  if ( qz < root.lo[2] ) o2 = root.lo[2] - qz;
  if ( qz > root.hi[2] ) o2 = qz - root.hi[2];

  // An upper bound of the K-th distance BEFORE the traversal: K real points next to the query in tree order (the query's own
  // neighbours for SELF; the points around the leaf a stack-less descent ends in otherwise) -- their largest distance bounds
  // the K-th smallest distance of the whole cloud from above.  The traversal prunes with min( worst of the list, cap ): a
  // subtree farther than cap holds nothing that can end up in the list, and skipping it leaves the order in which the others
  // are visited as it was, so the list is the reference's, ties included.  What it buys: nanoflann's first descent runs with
  // an unbounded list and leaves one pending far child per tree level (~ 25 stack entries per query, most of them spilled to
  // private memory: ~ 90 MB written and read back per launch at longdress size, 6 x the algorithmic bytes); with the cap the
  // stack holds the handful of far children that are really near.
  uint32_t cap = 0;
  if ( nTree < uint32_t( K ) ) {
    cap = 0xFFFFFFFFu;  // fewer than K points in the tree: nothing to bound with, the list keeps its unfilled slots
  } else {
    uint32_t centre = j;
    if ( !SELF ) {
      uint32_t at = 0;
      KdNode   nd = nodes[0];
      while ( nd.dim >= 0 ) {
        const int v = nd.dim == 0 ? qx : ( nd.dim == 1 ? qy : qz );
        at          = ( ( v - nd.divlow ) + ( v - nd.divhigh ) ) < 0 ? uint32_t( nd.a ) : uint32_t( nd.b );
        nd          = nodes[at];
      }
      centre = uint32_t( nd.a + nd.b ) >> 1;
    }
    const uint32_t first = min( centre > uint32_t( K / 2 ) ? centre - uint32_t( K / 2 ) : 0u, nTree - uint32_t( K ) );
    for ( int i = 0; i < K; ++i ) {
      const Pt  c  = ptsTree[first + i];
      const int ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
      cap          = max( cap, uint32_t( ex * ex + ey * ey + ez * ez ) );
    }
  }
  __shared__ unsigned long long ldsStack[LDS ? kLdsTop * 256 : 1];  // [slot][thread], slots < kLdsTop
  unsigned long long            lowStack[LDS ? kLdsLevels : 1];       // slots >= kLdsTop (private memory)
  uint4                         scratchStack[LDS ? 1 : kMaxStack];
  bool     swept = false;  // LDS: the stack has been purged once (see below)
  int      sp    = 0;
  uint32_t node = 0;
  for ( ;; ) {
    KdNode nd = nodes[node];
    while ( nd.dim >= 0 ) {
      const int  v        = nd.dim == 0 ? qx : ( nd.dim == 1 ? qy : qz );
      const int  ocur     = nd.dim == 0 ? o0 : ( nd.dim == 1 ? o1 : o2 );
      const int  diff1    = v - nd.divlow;
      const int  diff2    = v - nd.divhigh;
      const bool leftNear = ( diff1 + diff2 ) < 0;
      const int  ofar     = leftNear ? abs( diff2 ) : abs( diff1 );
      const uint32_t nearC = leftNear ? uint32_t( nd.a ) : uint32_t( nd.b );
      const uint32_t farC  = leftNear ? uint32_t( nd.b ) : uint32_t( nd.a );
      const uint32_t farMin = uint32_t( o0 * o0 + o1 * o1 + o2 * o2 + ofar * ofar - ocur * ocur );
      if ( farMin <= min( bd[K - 1], cap ) && sp < ( LDS ? kLdsLevels : kMaxStack ) ) {
        const uint32_t f0 = uint32_t( nd.dim == 0 ? ofar : o0 ), f1 = uint32_t( nd.dim == 1 ? ofar : o1 ),
                       f2 = uint32_t( nd.dim == 2 ? ofar : o2 );
        if ( LDS ) {
          const unsigned long long e = (unsigned long long)farC | ( (unsigned long long)f0 << sh0 ) |
                                       ( (unsigned long long)f1 << sh1 ) | ( (unsigned long long)f2 << sh2 );
          if ( sp < kLdsTop )
            ldsStack[sp * 256 + threadIdx.x] = e;
          else
            lowStack[sp - kLdsTop] = e;
        } else {
          scratchStack[sp] = make_uint4( farC, f0, f1, f2 );
        }
        ++sp;
      }
      node = nearC;
      nd   = nodes[node];
    }
    // The leaf's points, two per 16-byte load (a pair starts at an even tree position; what lies outside [a, b) is skipped).
    // The list keeps TREE POSITIONS: the original indices are looked up once at the end, all lanes in step, instead of one
    // scattered 4-byte load per insertion.  (A candidate beyond cap cannot stay in the list and is never the reason another
    // one is rejected: it is skipped.)
    for ( int p = nd.a & ~1; p < nd.b; p += 2 ) {
      uint4 two;
      if ( uint32_t( p ) + 1u < nTree ) {
        two = *reinterpret_cast<const uint4*>( ptsTree + p );
      } else {
        const uint2 one = *reinterpret_cast<const uint2*>( ptsTree + p );
        two             = make_uint4( one.x, one.y, 0u, 0u );
      }
      if ( p >= nd.a ) {
        const int      ex = qx - int( int16_t( two.x & 0xFFFFu ) ), ey = qy - int( int16_t( two.x >> 16 ) ), ez = qz - int( int16_t( two.y & 0xFFFFu ) );
        const uint32_t dist = uint32_t( ex * ex + ey * ey + ez * ez );
        if ( dist < bd[K - 1] && dist <= cap ) knnInsert<K>( bd, bi, dist, uint32_t( p ) );
      }
      if ( p + 1 < nd.b ) {
        const int      ex = qx - int( int16_t( two.z & 0xFFFFu ) ), ey = qy - int( int16_t( two.z >> 16 ) ), ez = qz - int( int16_t( two.w & 0xFFFFu ) );
        const uint32_t dist = uint32_t( ex * ex + ey * ey + ez * ez );
        if ( dist < bd[K - 1] && dist <= cap ) knnInsert<K>( bd, bi, dist, uint32_t( p + 1 ) );
      }
    }
    // The first descent runs with an unbounded list and therefore leaves one pending far child per tree level.  As soon as
    // the list is full, nearly all of them (the far side of the coarse splits) fail the bound, and the bound only
    // tightens: purge them NOW, while the lanes of the wave are still in step (their private-memory slots are read
    // coalesced), instead of one by one at the very end, when every lane unwinds on its own (each read then costs whole
    // cache lines: 1.7 GB of HBM traffic per launch).  Same test, same order of the survivors: same traversal.
    if ( LDS && !swept && bd[K - 1] != kInf ) {
      swept = true;
      int w = 0;
      for ( int i = 0; i < sp; ++i ) {
        const unsigned long long e = i < kLdsTop ? ldsStack[i * 256 + threadIdx.x] : lowStack[i - kLdsTop];
        const uint32_t e0 = uint32_t( e >> sh0 ) & offMask, e1 = uint32_t( e >> sh1 ) & offMask, e2 = uint32_t( e >> sh2 ) & offMask;
        if ( e0 * e0 + e1 * e1 + e2 * e2 <= min( bd[K - 1], cap ) ) {
          if ( w < kLdsTop )
            ldsStack[w * 256 + threadIdx.x] = e;
          else
            lowStack[w - kLdsTop] = e;
          ++w;
        }
      }
      sp = w;
    }
    bool found = false;
    while ( sp > 0 ) {
      --sp;
      uint32_t en, e0, e1, e2;
      if ( LDS ) {
        const unsigned long long e = sp < kLdsTop ? ldsStack[sp * 256 + threadIdx.x] : lowStack[sp - kLdsTop];
        en = uint32_t( e ) & nodeMask, e0 = uint32_t( e >> sh0 ) & offMask, e1 = uint32_t( e >> sh1 ) & offMask, e2 = uint32_t( e >> sh2 ) & offMask;
      } else {
        const uint4 e = scratchStack[sp];
        en = e.x, e0 = e.y, e1 = e.z, e2 = e.w;
      }
      if ( e0 * e0 + e1 * e1 + e2 * e2 <= min( bd[K - 1], cap ) ) {
        node  = en;
        o0    = int( e0 );
        o1    = int( e1 );
        o2    = int( e2 );
        found = true;
        break;
      }
    }
    if ( !found ) break;
  }
  const size_t row = SELF ? size_t( perm[j] ) : ( rowMap ? size_t( rowMap[j] ) : size_t( j ) );
  uint32_t*    oi  = outIdx + row * K;
#pragma unroll
  for ( int i = 0; i < K; ++i ) oi[i] = perm[bi[i]];  // (the list is full: k <= n, and everything within cap was visited)
  if ( outDist ) {
    uint32_t* od = outDist + row * K;
#pragma unroll
    for ( int i = 0; i < K; ++i ) od[i] = bd[i];
  }
}

// nqLive / rowMap: see knnKernel
template <bool SELF>
int dispatch( const tmc2_ctx* ctx, hipStream_t s, const TreeDev& t, const Pt* q, uint64_t nq, int k, uint32_t* idx, uint32_t* dist,
              const uint32_t* nqLive = nullptr, const uint32_t* rowMap = nullptr ) {
  if ( t.depth > kMaxStack ) {
    setError( "k-d tree depth %d exceeds the traversal stack (%d)", t.depth, kMaxStack );
    return TMC2_E_UNSUPPORTED;
  }
  if ( uint64_t( k ) > t.n ) {
    setError( "k=%d larger than the cloud (%llu points)", k, (unsigned long long)t.n );
    return TMC2_E_INVALID;
  }
  RootBox rb;
  for ( int d = 0; d < 3; ++d ) {
    rb.lo[d] = t.lo[d];
    rb.hi[d] = t.hi[d];
  }
  const dim3 block( 256 );
  // (test hook TMC2_KNN_XCD=0: blocks in launch order)
  const char* xcdEnv   = ctxOption( ctx, "KNN_XCD" );
  const int   xcdAware = xcdEnv && xcdEnv[0] == '0' ? 0 : 1;
  const dim3  grid( xcdAware ? uint32_t( ( ( nq + 255 ) / 256 + 7 ) & ~uint64_t( 7 ) ) : uint32_t( ( nq + 255 ) / 256 ) );
  // the packed LDS stack needs: every offset < 2^14 (tree box and queries inside a 16383-wide window -- the caller
  // vouches for the queries with t.queriesBounded), node ids < 2^22, and at most kLdsLevels pending far children
  bool lds = t.depth <= kLdsLevels && ( ( t.queriesBounded && t.n <= ( uint64_t( 1 ) << 21 ) ) || ( ( SELF || t.queriesTight ) && t.n <= ( uint64_t( 1 ) << 23 ) ) );
  for ( int d = 0; d < 3; ++d ) lds = lds && t.lo[d] >= 0 && t.hi[d] <= 8191;
  const int nodeBits = t.n <= ( uint64_t( 1 ) << 21 ) ? 22 : 25;  // (node ids stay below 2 n + the id chunks of the build)
#define TMC2_LAUNCH_K( KK )                                                                                          \
  if ( lds ) {                                                                                                       \
    hipLaunchKernelGGL( ( knnKernel<KK, SELF, true> ), grid, block, 0, s, t.ptsTree, t.perm, t.nodes, rb, q,          \
                        uint32_t( nq ), idx, dist, xcdAware, uint32_t( t.n ), nodeBits, nqLive, rowMap );                             \
  } else {                                                                                                           \
    hipLaunchKernelGGL( ( knnKernel<KK, SELF, false> ), grid, block, 0, s, t.ptsTree, t.perm, t.nodes, rb, q,         \
                        uint32_t( nq ), idx, dist, xcdAware, uint32_t( t.n ), nodeBits, nqLive, rowMap );                             \
  }
  switch ( k ) {
    case 1: TMC2_LAUNCH_K( 1 ); break;
    case 4: TMC2_LAUNCH_K( 4 ); break;
    case 8: TMC2_LAUNCH_K( 8 ); break;
    case 16: TMC2_LAUNCH_K( 16 ); break;
    case 32: TMC2_LAUNCH_K( 32 ); break;  // (the metric's wide search: the reference's last attempt asks for 30)
    default: setError( "k=%d not instantiated (1, 4, 8, 16, 32)", k ); return TMC2_E_UNSUPPORTED;
  }
#undef TMC2_LAUNCH_K
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace

TreeDev frameTree( const tmc2_frame* f ) {
  TreeDev t;
  t.ptsTree = f->d_ptsTree.p;
  t.perm    = f->d_perm.p;
  t.nodes   = f->d_nodes.p;
  for ( int d = 0; d < 3; ++d ) {
    t.lo[d] = f->tree.lo[d];
    t.hi[d] = f->tree.hi[d];
  }
  t.depth = f->tree.depth;
  t.n     = f->n;
  // queries against a frame's tree are the frame's own points or its reconstruction (non-negative, < 2^13)
  t.queriesBounded = true;
  t.queriesTight   = true;  // (... and below 2^13: inside [0, 8191])
  return t;
}

int launchKnnSelf( tmc2_frame* f, int k ) {
  TMC2_TRY( f->d_knn.alloc( f->n * size_t( k ) ) );
  const int sid = f->ctx->stageBegin( "knn_self" );
  const int r   = dispatch<true>( f->ctx, f->ctx->stream, frameTree( f ), nullptr, f->n, k, f->d_knn.p, nullptr );
  f->ctx->stageEnd( sid );
  if ( r == TMC2_OK ) {
    f->k       = k;
    f->haveKnn    = true;
    f->haveMutual = false;
  }
  return r;
}

int launchKnnQueries( tmc2_frame* f, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx, uint32_t* d_dist,
                      bool queriesBounded ) {
  TreeDev t        = frameTree( f );
  t.queriesBounded = queriesBounded;
  t.queriesTight   = false;  // (a caller's own queries: anywhere in the bounded window)
  return launchKnnTree( f->ctx, t, d_queries, nq, k, d_idx, d_dist, "knn_query" );
}

int launchKnnTree( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx,
                   uint32_t* d_dist, const char* stage ) {
  const int sid = ctx->stageBegin( stage );
  const int r   = dispatch<false>( ctx, ctx->stream, tree, d_queries, nq, k, d_idx, d_dist );
  ctx->stageEnd( sid );
  return r;
}

namespace {
// Pass 1 of launchKnnSplit.  The reference's search descends to the leaf on the query's side of every split FIRST and scans it
// in tree order (nanoflann searchLevel); a point of that leaf that is IDENTICAL to the query is at distance 0, enters the result
// list at its head, and nothing can displace it from there (equal distances are inserted behind, a full list rejects them): the
// first identical point of the descent leaf IS the reference's first result.  easyIdx[j] = its original index (and 0 as its
// distance where the caller wants one), kHardQuery where the leaf has none -- such a query may still have an identical point
// elsewhere (points ON a split plane go to either side): it simply takes the ordinary search.
__global__ __launch_bounds__( 256 ) void easyQueryKernel( const Pt* __restrict__ ptsTree, const uint32_t* __restrict__ perm,
                                                           const KdNode* __restrict__ nodes, const Pt* __restrict__ queries, uint32_t nq,
                                                           uint32_t* __restrict__ easyIdx, uint32_t* __restrict__ easyDist,
                                                           uint32_t* __restrict__ rowIdx, uint32_t* __restrict__ rowDist, int rowK ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j >= nq ) return;
  const Pt  qp = queries[j];
  const int qx = qp.x, qy = qp.y, qz = qp.z;
  KdNode    nd = nodes[0];
  while ( nd.dim >= 0 ) {
    const int v = nd.dim == 0 ? qx : ( nd.dim == 1 ? qy : qz );
    nd          = nodes[( ( v - nd.divlow ) + ( v - nd.divhigh ) ) < 0 ? uint32_t( nd.a ) : uint32_t( nd.b )];
  }
  uint32_t found = kHardQuery;
  for ( int p = nd.b - 1; p >= nd.a; --p ) {  // (backwards: the FIRST identical point is what is left in `found`)
    const Pt c = ptsTree[p];
    if ( c.x == qp.x && c.y == qp.y && c.z == qp.z ) found = uint32_t( p );
  }
  easyIdx[j] = found == kHardQuery ? kHardQuery : perm[found];
  if ( easyDist && found != kHardQuery ) easyDist[j] = 0u;
  // a tree of UNIQUE positions (the metric's de-duplicated clouds): the query's results at distance 0 are this one point -- row
  // [ match, .. ] with distances [ 0, not 0, .. ] is all a consumer that asks for "every point at the minimum distance" reads
  if ( rowIdx && found != kHardQuery ) {
    rowIdx[size_t( j ) * rowK]      = perm[found];
    rowDist[size_t( j ) * rowK]     = 0u;
    rowDist[size_t( j ) * rowK + 1] = kInf;
  }
}
// flag[j] = 1 where the easy pass left query j to the second launch
__global__ __launch_bounds__( 256 ) void hardFlagKernel( const uint32_t* __restrict__ easyIdx, uint32_t nq, uint32_t* __restrict__ flag ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j < nq ) flag[j] = easyIdx[j] == kHardQuery ? 1u : 0u;
}
// the hard queries back to back, in their order, with the rows they came from
__global__ __launch_bounds__( 256 ) void gatherHardKernel( const uint32_t* __restrict__ easyIdx, const uint32_t* __restrict__ rank, const Pt* __restrict__ queries,
                                                            uint32_t nq, Pt* __restrict__ hardQueries, uint32_t* __restrict__ rowMap ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j >= nq || easyIdx[j] != kHardQuery ) return;
  const uint32_t at = rank[j];
  hardQueries[at]   = queries[j];
  rowMap[at]        = j;
}
}  // namespace

// The transfer's two searches in two launches each (round 6).  Most reconstructed points ARE source points (and the other way
// round): 84-96 % of the queries have an identical point in the tree -- but the one query in ten that has to look around for real
// kept every wavefront as long as itself.  Pass 1 (easyQueryKernel: one descent, one leaf) answers the queries whose descent leaf
// holds an identical point -- d_easy[j] = the reference's FIRST result, kHardQuery otherwise; pass 2 runs the ordinary k-NN kernel
// over the compacted rest, writing rows d_idx / d_dist [j][k] in place (the live count stays on the device: no round trip).  What
// a caller may do with an easy query depends on k: k = 1 -- d_easy IS the result (pass d_idx as d_easy: distance 0 is written
// too); k > 1 -- only where the consumer needs nothing but the first result of a search whose first distance is 0
// (transferColors' forward direction: `skipAvgIfIdenticalSourcePointPresent`, PCCPointSet.cpp:853-858): the rows of easy queries
// are NOT written.
// uniqueTreeRows (k > 1): the tree holds no position twice and the consumer reads a row only up to the first distance that differs
// from the first (PCCMetrics' "all points at the minimum distance", PCCMetrics.cpp:91-96): the rows of easy queries are written
// as [ match | 0, not 0 ].
int launchKnnSplit( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_easy, uint32_t* d_idx,
                    uint32_t* d_dist, const char* stage, bool uniqueTreeRows ) {
  hipStream_t s = ctx->stream;
  if ( nq == 0 ) return TMC2_OK;
  const int sid = ctx->stageBegin( stage );
  DevBuf<uint32_t> d_flag, d_rank, d_rowMap, d_count;
  DevBuf<Pt>       d_hard;
  TMC2_TRY( d_flag.alloc( nq ) );
  TMC2_TRY( d_rank.alloc( nq ) );
  TMC2_TRY( d_rowMap.alloc( nq ) );
  TMC2_TRY( d_hard.alloc( nq ) );
  TMC2_TRY( d_count.alloc( 1 ) );
  int r = TMC2_OK;
  if ( tree.n == 0 ) {
    setError( "k-NN: empty tree" );
    r = TMC2_E_INVALID;
  }
  if ( r == TMC2_OK ) {
    const dim3 blk( 256 ), grd( uint32_t( ( nq + 255 ) / 256 ) );
    // pass 1: with k = 1 the easy results go straight to their rows (distance 0 with them)
    const bool rows = uniqueTreeRows && k > 1;
    hipLaunchKernelGGL( easyQueryKernel, grd, blk, 0, s, tree.ptsTree, tree.perm, tree.nodes, d_queries, uint32_t( nq ), d_easy,
                        k == 1 ? d_dist : (uint32_t*)nullptr, rows ? d_idx : (uint32_t*)nullptr, rows ? d_dist : (uint32_t*)nullptr, k );
    hipLaunchKernelGGL( hardFlagKernel, grd, blk, 0, s, d_easy, uint32_t( nq ), d_flag.p );
    r = exclusiveScanU32( ctx, d_flag.p, d_rank.p, nq, d_count.p );
    if ( r == TMC2_OK ) {
      hipLaunchKernelGGL( gatherHardKernel, grd, blk, 0, s, d_easy, d_rank.p, d_queries, uint32_t( nq ), d_hard.p, d_rowMap.p );
      // pass 2: the grid is sized for the worst case, the live count is on the device (no round trip)
      r = dispatch<false>( ctx, s, tree, d_hard.p, nq, k, d_idx, d_dist, d_count.p, d_rowMap.p );
    }
  }
  ctx->stageEnd( sid );
  return r;
}

}  // namespace tmc2
