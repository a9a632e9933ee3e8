// kdtree_device.hip -- construction of the nanoflann-identical k-d tree ON the device (S1).
//
// Replaces PCCKdTree::init (reference: source/lib/PccLibCommon/source/PCCKdTree.cpp:56-59), i.e. nanoflann's
// buildIndex / divideTree / middleSplit_ / planeSplit (dependencies/nanoflann/nanoflann.hpp:858-866, 1041-1181)
// for int16 3-D points and leaf size 10.  kdtree_build.cpp is the same algorithm on the host (kept for the host-only
// entry point and as the cross-check of this one); see its header for why the tree has to be IDENTICAL, permutation
// included.
//
// The reference builds depth-first and partitions each node with a sequential two-pass Hoare sweep.  Here the tree is
// built LEVEL by level, all nodes of a level at once, with every per-node step turned into a data-parallel pass over
// the points (tree-order arrays, segment id per point):
//   * ranges  : segmented wave reduction + atomicMin/Max per node (one pass gives the ranges of all three dimensions);
//   * split   : per node -- widest loose-box dimension / largest spread, midpoint clamped to the range (as the host);
//   * partition: what the Hoare sweep leaves behind is closed-form.  With L = "value < cut", nL = #L: the sweep ends with
//     every L in [0,nL) and swaps, in order, the i-th misplaced non-L from the left with the i-th misplaced L from the
//     right; elements already on their side never move.  A prefix sum of the class flag gives every misplaced element
//     its rank, the right-hand ones publish their position by rank, the left-hand ones swap with it.  The second sweep
//     ("value <= cut" on [nL, count)) is the same on the sub-range.  lim1 / lim2 fall out of the same prefix sums;
//   * children: node ids and next-level slots from atomic counters (ids are arbitrary: the traversal follows explicit
//     child ids; only the ROOT must be node 0), loose boxes handed down, and each child reports its tight range to
//     the parent's divlow / divhigh when it is measured at the next level.
// Cost: ~12 short launches per level over <= n points, depth ~ 2 log2(n/10) levels; no host work except one 8-byte
// read-back per level once termination becomes possible.
#include <algorithm>

#include "internal.h"

namespace tmc2 {
namespace {

constexpr uint32_t kNone     = 0xFFFFFFFFu;
constexpr int      kScanTile = 2048;  // 256 threads x 8
constexpr int      kLeafMax  = 10;

struct BuildSeg {
  uint32_t begin, end;    // range in tree order
  uint32_t node, parent;  // own node id; parent's node id (kNone for the root)
  int32_t  mn[3], mx[3];  // tight range of the points (atomics)
  int16_t  lo[3], hi[3];  // loose box handed down by the parent
  uint8_t  side, pdim;    // which child of the parent we are, and the parent's cut dimension
  uint8_t  split, cutDim;
  int32_t  cut;
  uint32_t lim1, m1, r1b, r1m;  // first sweep : #L, #swaps, prefix at begin, prefix at begin + lim1
  uint32_t nE, m2, r2b, r2m;    // second sweep: #(== cut), #swaps, prefix at begin + lim1, prefix at begin + lim1 + nE
  uint32_t mid;                 // begin + idx: first point of the right child
  uint32_t slot;                // next-level slot of the left child (right = slot + 1)
};

__device__ __forceinline__ int coordOf( const Pt p, int d ) { return d == 0 ? p.x : ( d == 1 ? p.y : p.z ); }

__device__ __forceinline__ uint32_t prefixAt( const uint32_t* __restrict__ loc, const uint32_t* __restrict__ sums,
                                              const uint32_t* __restrict__ total, uint32_t i, uint32_t n ) {
  return i < n ? loc[i] + sums[i / kScanTile] : *total;
}

__global__ __launch_bounds__( 256 ) void initKernel( const Pt* __restrict__ pts, uint32_t n, Pt* __restrict__ P,
                                                      uint32_t* __restrict__ perm, uint32_t* __restrict__ seg,
                                                      BuildSeg* __restrict__ root, uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ nodeCount ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) {
    P[i]    = pts[i];
    perm[i] = i;
    seg[i]  = 0;
  }
  if ( i == 0 ) {
    BuildSeg r{};
    r.begin  = 0;
    r.end    = n;
    r.node   = 0;
    r.parent = kNone;
    for ( int d = 0; d < 3; ++d ) r.mn[d] = 0x7FFFFFFF, r.mx[d] = int32_t( 0x80000000 );
    *root      = r;
    counts[0]  = 1;  // segments of level 0
    *nodeCount = 1;  // node 0 = the root
  }
}

// level entry: move every point to its segment of THIS level (children of the previous level's segments), then
// accumulate the tight ranges.  Points of one segment are contiguous, so a segmented shuffle reduction leaves the
// range of each run in its first lane and only that lane touches memory.
template <bool FIRST>
__global__ __launch_bounds__( 256 ) void rangeKernel( const Pt* __restrict__ P, uint32_t n, uint32_t* __restrict__ seg,
                                                       const BuildSeg* __restrict__ prev, BuildSeg* __restrict__ cur ) {
  const uint32_t i    = blockIdx.x * blockDim.x + threadIdx.x;
  const int      lane = threadIdx.x & 63;
  uint32_t       s    = kNone;
  if ( i < n ) {
    if ( FIRST ) {
      s = 0;
    } else {
      const uint32_t so = seg[i];
      if ( so != kNone ) {
        if ( prev[so].split ) s = prev[so].slot + ( i >= prev[so].mid ? 1u : 0u );
        seg[i] = s;
      }
    }
  }
  int mnx = 0x7FFFFFFF, mny = 0x7FFFFFFF, mnz = 0x7FFFFFFF, mxx = int( 0x80000000 ), mxy = mxx, mxz = mxx;
  if ( s != kNone ) {
    const Pt p = P[i];
    mnx = mxx = p.x;
    mny = mxy = p.y;
    mnz = mxz = p.z;
  }
#pragma unroll
  for ( int off = 1; off < 64; off <<= 1 ) {
    const uint32_t os = __shfl_down( s, off, 64 );
    const int      a = __shfl_down( mnx, off, 64 ), b = __shfl_down( mny, off, 64 ), c = __shfl_down( mnz, off, 64 );
    const int      d = __shfl_down( mxx, off, 64 ), e = __shfl_down( mxy, off, 64 ), g = __shfl_down( mxz, off, 64 );
    if ( lane + off < 64 && os == s ) {
      mnx = min( mnx, a ), mny = min( mny, b ), mnz = min( mnz, c );
      mxx = max( mxx, d ), mxy = max( mxy, e ), mxz = max( mxz, g );
    }
  }
  const uint32_t ps = __shfl_up( s, 1, 64 );
  if ( s != kNone && ( lane == 0 || ps != s ) ) {
    BuildSeg* q = cur + s;
    atomicMin( &q->mn[0], mnx ), atomicMin( &q->mn[1], mny ), atomicMin( &q->mn[2], mnz );
    atomicMax( &q->mx[0], mxx ), atomicMax( &q->mx[1], mxy ), atomicMax( &q->mx[2], mxz );
  }
}

// per segment: report the tight range to the parent, then leaf or split rule (kdtree_build.cpp, state 0)
__global__ __launch_bounds__( 256 ) void decideKernel( BuildSeg* __restrict__ cur, const uint32_t* __restrict__ count,
                                                        KdNode* __restrict__ nodes, int32_t* __restrict__ rootBox ) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if ( s >= *count ) return;
  BuildSeg q = cur[s];
  if ( q.parent == kNone ) {
    for ( int d = 0; d < 3; ++d ) {
      q.lo[d] = int16_t( q.mn[d] ), q.hi[d] = int16_t( q.mx[d] );
      rootBox[d] = q.mn[d], rootBox[3 + d] = q.mx[d];
    }
  } else if ( q.side == 0 ) {
    nodes[q.parent].divlow = int16_t( q.mx[q.pdim] );
  } else {
    nodes[q.parent].divhigh = int16_t( q.mn[q.pdim] );
  }
  const uint32_t cnt = q.end - q.begin;
  if ( cnt <= uint32_t( kLeafMax ) ) {
    KdNode nd;
    nd.a = int32_t( q.begin ), nd.b = int32_t( q.end ), nd.divlow = nd.divhigh = 0, nd.dim = -1;
    nodes[q.node] = nd;
    q.split       = 0;
  } else {
    int32_t maxSpan = 0;
    for ( int d = 0; d < 3; ++d ) maxSpan = max( maxSpan, int32_t( q.hi[d] ) - int32_t( q.lo[d] ) );
    int     cutDim     = 0;
    int32_t bestSpread = -1;
    for ( int d = 0; d < 3; ++d ) {
      const int32_t span = int32_t( q.hi[d] ) - int32_t( q.lo[d] );
      if ( double( span ) > ( 1.0 - 0.00001 ) * double( maxSpan ) ) {
        const int32_t spread = q.mx[d] - q.mn[d];
        if ( spread > bestSpread ) {
          bestSpread = spread;
          cutDim     = d;
        }
      }
    }
    const int32_t mid = ( int32_t( q.lo[cutDim] ) + int32_t( q.hi[cutDim] ) ) / 2;
    q.cut             = min( max( mid, q.mn[cutDim] ), q.mx[cutDim] );
    q.cutDim          = uint8_t( cutDim );
    q.split           = 1;
  }
  cur[s] = q;
}

// tile-local exclusive prefix of the sweep's class flag (PASS 1: value >= cut; PASS 2: value > cut on [begin+lim1, end))
template <int PASS>
__global__ __launch_bounds__( 256 ) void flagScanTilesKernel( const Pt* __restrict__ P, const uint32_t* __restrict__ seg,
                                                               const BuildSeg* __restrict__ cur, uint32_t n,
                                                               uint32_t* __restrict__ loc, uint32_t* __restrict__ sums ) {
  __shared__ uint32_t waveSum[4];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t      base = blockIdx.x * kScanTile + threadIdx.x * 8;
  uint32_t            v[8], run = 0;
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    const uint32_t i = base + k;
    uint32_t       f = 0;
    if ( i < n ) {
      const uint32_t s = seg[i];
      if ( s != kNone ) {
        const BuildSeg* q = cur + s;
        if ( q->split ) {
          const int val = coordOf( P[i], q->cutDim );
          f             = PASS == 1 ? ( val >= q->cut ) : ( i >= q->begin + q->lim1 && val > q->cut );
        }
      }
    }
    v[k] = f;
    run += f;
  }
  uint32_t inc = run;
#pragma unroll
  for ( int off = 1; off < 64; off <<= 1 ) {
    const uint32_t t = __shfl_up( inc, off, 64 );
    if ( lane >= off ) inc += t;
  }
  if ( lane == 63 ) waveSum[wave] = inc;
  __syncthreads();
  uint32_t offset = inc - run;
  for ( int w = 0; w < wave; ++w ) offset += waveSum[w];
#pragma unroll
  for ( int k = 0; k < 8; ++k ) {
    if ( base + k < n ) loc[base + k] = offset;
    offset += v[k];
  }
  if ( threadIdx.x == 255 ) sums[blockIdx.x] = offset;
}

// one block: exclusive scan of the tile totals in place, grand total to *total
__global__ __launch_bounds__( 256 ) void scanSumsKernel( uint32_t* __restrict__ sums, uint32_t tiles,
                                                          uint32_t* __restrict__ total ) {
  __shared__ uint32_t waveSum[4];
  __shared__ uint32_t carry;
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ( threadIdx.x == 0 ) carry = 0;
  __syncthreads();
  for ( uint32_t base = 0; base < tiles; base += 256 ) {
    const uint32_t i   = base + threadIdx.x;
    const uint32_t v   = i < tiles ? sums[i] : 0u;
    uint32_t       inc = v;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 );
      if ( lane >= off ) inc += t;
    }
    if ( lane == 63 ) waveSum[wave] = inc;
    __syncthreads();
    uint32_t offset = carry + inc - v;
    for ( int w = 0; w < wave; ++w ) offset += waveSum[w];
    if ( i < tiles ) sums[i] = offset;
    __syncthreads();
    if ( threadIdx.x == 255 ) carry = offset + v;
    __syncthreads();
  }
  if ( threadIdx.x == 0 ) *total = carry;
}

// per segment, after the first prefix sum: lim1 and the number of swaps of the first sweep
__global__ __launch_bounds__( 256 ) void sweepOneKernel( BuildSeg* __restrict__ cur, const uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ loc, const uint32_t* __restrict__ sums,
                                                          const uint32_t* __restrict__ total, uint32_t n ) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if ( s >= *count ) return;
  BuildSeg* q = cur + s;
  if ( !q->split ) return;
  const uint32_t rb = prefixAt( loc, sums, total, q->begin, n ), re = prefixAt( loc, sums, total, q->end, n );
  const uint32_t nL = ( q->end - q->begin ) - ( re - rb );
  const uint32_t rm = prefixAt( loc, sums, total, q->begin + nL, n );
  q->lim1 = nL, q->r1b = rb, q->r1m = rm, q->m1 = rm - rb;
}

// misplaced right-hand elements publish their position under their rank counted from the right
template <int PASS>
__global__ __launch_bounds__( 256 ) void publishKernel( const Pt* __restrict__ P, const uint32_t* __restrict__ seg,
                                                         const BuildSeg* __restrict__ cur, uint32_t n,
                                                         const uint32_t* __restrict__ loc, const uint32_t* __restrict__ sums,
                                                         uint32_t* __restrict__ list ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint32_t s = seg[i];
  if ( s == kNone ) return;
  const BuildSeg* q = cur + s;
  if ( !q->split ) return;
  const uint32_t b    = PASS == 1 ? q->begin : q->begin + q->lim1;            // start of the swept range
  const uint32_t edge = b + ( PASS == 1 ? q->lim1 : q->nE );                   // where the left class ends
  const uint32_t m    = PASS == 1 ? q->m1 : q->m2;
  if ( i < edge || m == 0 ) return;
  const int  val  = coordOf( P[i], q->cutDim );
  const bool left = PASS == 1 ? ( val < q->cut ) : ( val <= q->cut );
  if ( !left ) return;
  const uint32_t rightBefore = ( loc[i] + sums[i / kScanTile] ) - ( PASS == 1 ? q->r1m : q->r2m );  // flagged in [edge, i)
  const uint32_t leftBefore  = ( i - edge ) - rightBefore;
  list[b + ( m - 1 - leftBefore )] = i;
}

// misplaced left-hand elements swap with the published position of equal rank
template <int PASS>
__global__ __launch_bounds__( 256 ) void swapKernel( Pt* __restrict__ P, uint32_t* __restrict__ perm,
                                                      const uint32_t* __restrict__ seg, const BuildSeg* __restrict__ cur,
                                                      uint32_t n, const uint32_t* __restrict__ loc,
                                                      const uint32_t* __restrict__ sums, const uint32_t* __restrict__ list ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint32_t s = seg[i];
  if ( s == kNone ) return;
  const BuildSeg* q = cur + s;
  if ( !q->split ) return;
  const uint32_t b    = PASS == 1 ? q->begin : q->begin + q->lim1;
  const uint32_t edge = b + ( PASS == 1 ? q->lim1 : q->nE );
  if ( i < b || i >= edge ) return;
  const Pt   pi    = P[i];
  const int  val   = coordOf( pi, q->cutDim );
  const bool right = PASS == 1 ? ( val >= q->cut ) : ( val > q->cut );
  if ( !right ) return;
  const uint32_t r  = ( loc[i] + sums[i / kScanTile] ) - ( PASS == 1 ? q->r1b : q->r2b );
  const uint32_t j  = list[b + r];
  const Pt       pj = P[j];
  P[i]              = pj;
  P[j]              = pi;
  const uint32_t t  = perm[i];
  perm[i]           = perm[j];
  perm[j]           = t;
}

// per segment, after the second prefix sum: lim2, the balance rule, the node record and the two children
__global__ __launch_bounds__( 256 ) void childrenKernel( BuildSeg* __restrict__ cur, const uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ loc, const uint32_t* __restrict__ sums,
                                                          const uint32_t* __restrict__ total, uint32_t n,
                                                          BuildSeg* __restrict__ next, uint32_t* __restrict__ nextCount,
                                                          uint32_t* __restrict__ nodeCount, KdNode* __restrict__ nodes ) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if ( s >= *count ) return;
  BuildSeg* q = cur + s;
  if ( !q->split ) return;
  const uint32_t b2 = q->begin + q->lim1;
  const uint32_t rb = prefixAt( loc, sums, total, b2, n ), re = prefixAt( loc, sums, total, q->end, n );
  const uint32_t nE = ( q->end - b2 ) - ( re - rb );
  const uint32_t rm = prefixAt( loc, sums, total, b2 + nE, n );
  q->nE = nE, q->r2b = rb, q->r2m = rm, q->m2 = rm - rb;
  const uint32_t cnt = q->end - q->begin, half = cnt / 2, lim1 = q->lim1, lim2 = q->lim1 + nE;
  const uint32_t idx = lim1 > half ? lim1 : ( lim2 < half ? lim2 : half );
  q->mid             = q->begin + idx;
  const uint32_t id  = atomicAdd( nodeCount, 2u );
  const uint32_t sl  = atomicAdd( nextCount, 2u );
  q->slot            = sl;
  KdNode nd;
  nd.a = int32_t( id ), nd.b = int32_t( id + 1 ), nd.divlow = nd.divhigh = 0, nd.dim = q->cutDim;
  nodes[q->node] = nd;
  BuildSeg c{};
  c.parent = q->node;
  c.pdim   = q->cutDim;
  for ( int d = 0; d < 3; ++d ) c.mn[d] = 0x7FFFFFFF, c.mx[d] = int32_t( 0x80000000 ), c.lo[d] = q->lo[d], c.hi[d] = q->hi[d];
  BuildSeg l = c, r = c;
  l.begin = q->begin, l.end = q->mid, l.node = id, l.side = 0, l.hi[q->cutDim] = int16_t( q->cut );
  r.begin = q->mid, r.end = q->end, r.node = id + 1, r.side = 1, r.lo[q->cutDim] = int16_t( q->cut );
  next[sl]     = l;
  next[sl + 1] = r;
}

}  // namespace

// Builds the tree of d_pts[0..n) on the context's stream.  Outputs: points and permutation in tree order, node records
// (root = node 0), root box, depth (levels).  All buffers come from the context's pool.
int buildKdTreeDevice( tmc2_ctx* ctx, const Pt* d_pts, uint64_t n64, DevBuf<Pt>& d_ptsTree, DevBuf<uint32_t>& d_perm,
                       DevBuf<KdNode>& d_nodes, int32_t lo[3], int32_t hi[3], int& depth ) {
  const uint32_t n = uint32_t( n64 );
  depth            = 0;
  for ( int d = 0; d < 3; ++d ) lo[d] = hi[d] = 0;
  if ( n == 0 ) return TMC2_OK;
  hipStream_t    s       = ctx->stream;
  const uint32_t tiles   = ( n + kScanTile - 1 ) / kScanTile;
  const size_t   maxSegs = 2 * ( size_t( n ) / ( kLeafMax + 1 ) + 1 ) + 2;
  const size_t   maxNode = 2 * size_t( n ) + 2;
  TMC2_TRY( d_ptsTree.alloc( n ) );
  TMC2_TRY( d_perm.alloc( n ) );
  TMC2_TRY( d_nodes.alloc( maxNode ) );
  DevBuf<uint32_t> d_seg, d_loc, d_sums, d_list, d_small;
  DevBuf<BuildSeg> d_segs;
  TMC2_TRY( d_seg.alloc( n ) );
  TMC2_TRY( d_loc.alloc( 2 * size_t( n ) ) );
  TMC2_TRY( d_sums.alloc( 2 * size_t( tiles ) ) );
  TMC2_TRY( d_list.alloc( n ) );
  TMC2_TRY( d_segs.alloc( 2 * maxSegs ) );
  constexpr int kMaxLevels = 64;                     // = the traversal stack of the k-NN kernels
  TMC2_TRY( d_small.alloc( kMaxLevels + 16 ) );      // [0..64] segments per level, [65] node count, [66],[67] scan totals, [72..77] root box
  uint32_t* d_counts = d_small.p;
  uint32_t* d_nodeCount = d_small.p + kMaxLevels + 1;
  uint32_t* d_total1 = d_small.p + kMaxLevels + 2, *d_total2 = d_small.p + kMaxLevels + 3;
  uint32_t *d_loc1 = d_loc.p, *d_loc2 = d_loc.p + n, *d_sums1 = d_sums.p, *d_sums2 = d_sums.p + tiles;
  int32_t*  d_rootBox = reinterpret_cast<int32_t*>( d_small.p + kMaxLevels + 8 );
  TMC2_HIP( hipMemsetAsync( d_small.p, 0, ( kMaxLevels + 16 ) * 4, s ) );
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  BuildSeg*  segA = d_segs.p;
  BuildSeg*  segB = d_segs.p + maxSegs;
  hipLaunchKernelGGL( initKernel, grdN, blk, 0, s, d_pts, n, d_ptsTree.p, d_perm.p, d_seg.p, segA, d_counts, d_nodeCount );
  uint32_t segBound = 1;  // upper bound of this level's segment count (exact once it has been read back)
  int      level    = 0;
  for ( ; level < kMaxLevels; ++level ) {
    BuildSeg*  cur = ( level & 1 ) ? segB : segA;
    BuildSeg*  nxt = ( level & 1 ) ? segA : segB;
    const dim3 grdS( ( segBound + 255 ) / 256 );
    if ( level == 0 )
      hipLaunchKernelGGL( rangeKernel<true>, grdN, blk, 0, s, d_ptsTree.p, n, d_seg.p, nxt, cur );
    else
      hipLaunchKernelGGL( rangeKernel<false>, grdN, blk, 0, s, d_ptsTree.p, n, d_seg.p, nxt, cur );
    hipLaunchKernelGGL( decideKernel, grdS, blk, 0, s, cur, d_counts + level, d_nodes.p, d_rootBox );
    hipLaunchKernelGGL( flagScanTilesKernel<1>, dim3( tiles ), blk, 0, s, d_ptsTree.p, d_seg.p, cur, n, d_loc1, d_sums1 );
    hipLaunchKernelGGL( scanSumsKernel, dim3( 1 ), blk, 0, s, d_sums1, tiles, d_total1 );
    hipLaunchKernelGGL( sweepOneKernel, grdS, blk, 0, s, cur, d_counts + level, d_loc1, d_sums1, d_total1, n );
    hipLaunchKernelGGL( publishKernel<1>, grdN, blk, 0, s, d_ptsTree.p, d_seg.p, cur, n, d_loc1, d_sums1, d_list.p );
    hipLaunchKernelGGL( swapKernel<1>, grdN, blk, 0, s, d_ptsTree.p, d_perm.p, d_seg.p, cur, n, d_loc1, d_sums1, d_list.p );
    hipLaunchKernelGGL( flagScanTilesKernel<2>, dim3( tiles ), blk, 0, s, d_ptsTree.p, d_seg.p, cur, n, d_loc2, d_sums2 );
    hipLaunchKernelGGL( scanSumsKernel, dim3( 1 ), blk, 0, s, d_sums2, tiles, d_total2 );
    hipLaunchKernelGGL( childrenKernel, grdS, blk, 0, s, cur, d_counts + level, d_loc2, d_sums2, d_total2, n, nxt,
                        d_counts + level + 1, d_nodeCount, d_nodes.p );
    hipLaunchKernelGGL( publishKernel<2>, grdN, blk, 0, s, d_ptsTree.p, d_seg.p, cur, n, d_loc2, d_sums2, d_list.p );
    hipLaunchKernelGGL( swapKernel<2>, grdN, blk, 0, s, d_ptsTree.p, d_perm.p, d_seg.p, cur, n, d_loc2, d_sums2, d_list.p );
    // a level with more than 10 * 2^level points still has a splittable segment: no need to ask
    if ( ( uint64_t( kLeafMax ) << std::min( level, 40 ) ) < n ) {
      segBound = uint32_t( std::min<uint64_t>( uint64_t( segBound ) * 2, maxSegs ) );
      continue;
    }
    uint32_t nextSegs = 0;
    TMC2_HIP( hipMemcpyAsync( &nextSegs, d_counts + level + 1, 4, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    if ( nextSegs == 0 ) break;
    if ( nextSegs > maxSegs ) {
      setError( "kdtree: internal segment bound exceeded" );
      return TMC2_E_INVALID;
    }
    segBound = nextSegs;
  }
  if ( level >= kMaxLevels ) {
    setError( "kdtree: more than %d levels", kMaxLevels );
    return TMC2_E_UNSUPPORTED;
  }
  depth = level + 1;
  int32_t box[6];
  TMC2_HIP( hipMemcpyAsync( box, d_rootBox, sizeof( box ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  for ( int d = 0; d < 3; ++d ) lo[d] = box[d], hi[d] = box[3 + d];
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2
