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
//   * all arithmetic is int32: coordinates < 2^12, squared distances < 2^26 (the reference holds the
//     same integers in float/double -- exact below 2^24 for <= 11-bit data, KDTreeVectorOfVectorsAdaptor.h:126).
#include "internal.h"

namespace tmc2 {

namespace {

constexpr int      kMaxStack = 64;
constexpr uint32_t kInf      = 0x7FFFFFFFu;

struct RootBox {
  int lo[3], hi[3];
};

template <int K>
__device__ __forceinline__ void knnInsert( uint32_t ( &bd )[K], uint32_t ( &bi )[K], uint32_t dist, uint32_t index ) {
  // precondition: dist < bd[K-1].  New entry goes after every entry with bd <= dist.
#pragma unroll
  for ( int j = K - 1; j >= 0; --j ) {
    const bool keep  = bd[j] <= dist;
    const bool place = ( j == 0 ) || ( bd[j > 0 ? j - 1 : 0] <= dist );
    if ( !keep ) {
      bd[j] = place ? dist : bd[j > 0 ? j - 1 : 0];
      bi[j] = place ? index : bi[j > 0 ? j - 1 : 0];
    }
  }
}

// SELF = true : queries are the tree-order points themselves, row j is written to out[perm[j]]
// SELF = false: queries come from `queries` (any order), row j is written to out[j]
template <int K, bool SELF>
__global__ __launch_bounds__( 256 ) void knnKernel( const Pt* __restrict__ ptsTree, const uint32_t* __restrict__ perm,
                                                     const KdNode* __restrict__ nodes, RootBox root,
                                                     const Pt* __restrict__ queries, uint32_t nq,
                                                     uint32_t* __restrict__ outIdx, uint32_t* __restrict__ outDist ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j >= nq ) return;
  const Pt  qp = SELF ? ptsTree[j] : queries[j];
  const int qx = qp.x, qy = qp.y, qz = qp.z;

  uint32_t bd[K], bi[K];
#pragma unroll
  for ( int i = 0; i < K; ++i ) {
    bd[i] = kInf;
    bi[i] = 0;
  }
  // distance of the query to the root box, per dimension (nanoflann computeInitialDistances)
  int d0 = 0, d1 = 0, d2 = 0;
  if ( qx < root.lo[0] ) d0 = ( qx - root.lo[0] ) * ( qx - root.lo[0] );
  if ( qx > root.hi[0] ) d0 = ( qx - root.hi[0] ) * ( qx - root.hi[0] );
  if ( qy < root.lo[1] ) d1 = ( qy - root.lo[1] ) * ( qy - root.lo[1] );
  if ( qy > root.hi[1] ) d1 = ( qy - root.hi[1] ) * ( qy - root.hi[1] );
  if ( qz < root.lo[2] ) d2 = ( qz - root.lo[2] ) * ( qz - root.lo[2] );
  if ( qz > root.hi[2] ) d2 = ( qz - root.hi[2] ) * ( qz - root.hi[2] );

  uint4    stack[kMaxStack];
  int      sp   = 0;
  uint32_t node = 0;
  for ( ;; ) {
    KdNode nd = nodes[node];
    while ( nd.dim >= 0 ) {
      const int  v        = nd.dim == 0 ? qx : ( nd.dim == 1 ? qy : qz );
      const int  dcur     = nd.dim == 0 ? d0 : ( nd.dim == 1 ? d1 : d2 );
      const int  diff1    = v - nd.divlow;
      const int  diff2    = v - nd.divhigh;
      const bool leftNear = ( diff1 + diff2 ) < 0;
      const int  cut      = leftNear ? diff2 * diff2 : diff1 * diff1;
      const uint32_t nearC = leftNear ? uint32_t( nd.a ) : uint32_t( nd.b );
      const uint32_t farC  = leftNear ? uint32_t( nd.b ) : uint32_t( nd.a );
      const uint32_t farMin = uint32_t( d0 + d1 + d2 + cut - dcur );
      if ( farMin <= bd[K - 1] && sp < kMaxStack ) {
        stack[sp++] = make_uint4( farC, uint32_t( nd.dim == 0 ? cut : d0 ), uint32_t( nd.dim == 1 ? cut : d1 ),
                                  uint32_t( nd.dim == 2 ? cut : d2 ) );
      }
      node = nearC;
      nd   = nodes[node];
    }
    for ( int p = nd.a; p < nd.b; ++p ) {
      const Pt       c    = ptsTree[p];
      const int      ex = qx - c.x, ey = qy - c.y, ez = qz - c.z;
      const uint32_t dist = uint32_t( ex * ex + ey * ey + ez * ez );
      if ( dist < bd[K - 1] ) knnInsert<K>( bd, bi, dist, perm[p] );
    }
    bool found = false;
    while ( sp > 0 ) {
      const uint4 e = stack[--sp];
      if ( e.y + e.z + e.w <= bd[K - 1] ) {
        node  = e.x;
        d0    = int( e.y );
        d1    = int( e.z );
        d2    = int( e.w );
        found = true;
        break;
      }
    }
    if ( !found ) break;
  }
  const size_t row = SELF ? size_t( perm[j] ) : size_t( j );
  uint32_t*    oi  = outIdx + row * K;
#pragma unroll
  for ( int i = 0; i < K; ++i ) oi[i] = bi[i];
  if ( outDist ) {
    uint32_t* od = outDist + row * K;
#pragma unroll
    for ( int i = 0; i < K; ++i ) od[i] = bd[i];
  }
}

template <bool SELF>
int dispatch( hipStream_t s, const TreeDev& t, const Pt* q, uint64_t nq, int k, uint32_t* idx, uint32_t* dist ) {
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
  const dim3 grid( uint32_t( ( nq + 255 ) / 256 ) );
#define TMC2_LAUNCH_K( KK )                                                                                   \
  hipLaunchKernelGGL( ( knnKernel<KK, SELF> ), grid, block, 0, s, t.ptsTree, t.perm, t.nodes, rb, q, uint32_t( nq ), \
                      idx, dist )
  switch ( k ) {
    case 1: TMC2_LAUNCH_K( 1 ); break;
    case 4: TMC2_LAUNCH_K( 4 ); break;
    case 8: TMC2_LAUNCH_K( 8 ); break;
    case 16: TMC2_LAUNCH_K( 16 ); break;
    default: setError( "k=%d not instantiated (1, 4, 8, 16)", k ); return TMC2_E_UNSUPPORTED;
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
  return t;
}

int launchKnnSelf( tmc2_frame* f, int k ) {
  TMC2_TRY( f->d_knn.alloc( f->n * size_t( k ) ) );
  const int sid = f->ctx->stageBegin( "knn_self" );
  const int r   = dispatch<true>( f->ctx->stream, frameTree( f ), nullptr, f->n, k, f->d_knn.p, nullptr );
  f->ctx->stageEnd( sid );
  if ( r == TMC2_OK ) {
    f->k       = k;
    f->haveKnn = true;
  }
  return r;
}

int launchKnnQueries( tmc2_frame* f, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx, uint32_t* d_dist ) {
  return launchKnnTree( f->ctx, frameTree( f ), d_queries, nq, k, d_idx, d_dist, "knn_query" );
}

int launchKnnTree( tmc2_ctx* ctx, const TreeDev& tree, const Pt* d_queries, uint64_t nq, int k, uint32_t* d_idx,
                   uint32_t* d_dist, const char* stage ) {
  const int sid = ctx->stageBegin( stage );
  const int r   = dispatch<false>( ctx->stream, tree, d_queries, nq, k, d_idx, d_dist );
  ctx->stageEnd( sid );
  return r;
}

}  // namespace tmc2
