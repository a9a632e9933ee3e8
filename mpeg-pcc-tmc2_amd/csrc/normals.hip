// normals.hip -- per-point PCA normals and the initial projection-plane assignment on gfx950.
//
// Replaces
//   PCCNormalsGenerator3::computeNormal   (PccLibEncoder/source/PCCNormalsGenerator.cpp:71-157)
//   PCCDiagonalize                        (PccLibCommon/include/PCCMath.h:505-598)
//   PCCPatchSegmenter3::initialSegmentation (PccLibEncoder/source/PCCPatchSegmenter.cpp:226-265)
//
// Bit-exactness contract: fp64, IEEE divide and sqrt, no FMA contraction
// (the library is built with -ffp-contract=off), sums in neighbour-list order, the eigen-solver's
// sequential in-place quaternion update and its three early exits reproduced literally.
//
// One point per lane.  HBM traffic per point: 4k B of neighbour ids in, 24 B normal out; the 16
// neighbour positions are gathers that stay in L1/L2 because the ids are spatial neighbours.
#include "internal.h"

namespace tmc2 {

namespace {

__device__ __forceinline__ double dsqrt( double x ) { return __dsqrt_rn( x ); }
__device__ __forceinline__ double ddiv( double a, double b ) { return __ddiv_rn( a, b ); }

// Quaternion Jacobi iteration, at most 24 steps.  Returns in Q the eigenvector matrix, in ev the
// diagonal of Q^T A Q.
__device__ void diagonalize( double a00, double a01, double a02, double a11, double a12, double a22,
                             double ( &Q )[3][3], double ( &ev )[3] ) {
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 1.0;
  double D[3][3];
  for ( int step = 0; step < 24; ++step ) {
    const double sqx = q0 * q0, sqy = q1 * q1, sqz = q2 * q2, sqw = q3 * q3;
    Q[0][0] = ( sqx - sqy - sqz + sqw );
    Q[1][1] = ( -sqx + sqy - sqz + sqw );
    Q[2][2] = ( -sqx - sqy + sqz + sqw );
    double t1 = q0 * q1, t2 = q2 * q3;
    Q[1][0]   = 2.0 * ( t1 + t2 );
    Q[0][1]   = 2.0 * ( t1 - t2 );
    t1        = q0 * q2;
    t2        = q1 * q3;
    Q[2][0]   = 2.0 * ( t1 - t2 );
    Q[0][2]   = 2.0 * ( t1 + t2 );
    t1        = q1 * q2;
    t2        = q0 * q3;
    Q[2][1]   = 2.0 * ( t1 + t2 );
    Q[1][2]   = 2.0 * ( t1 - t2 );
    double AQ[3][3];
#pragma unroll
    for ( int c = 0; c < 3; ++c ) {
      AQ[0][c] = Q[0][c] * a00 + Q[1][c] * a01 + Q[2][c] * a02;
      AQ[1][c] = Q[0][c] * a01 + Q[1][c] * a11 + Q[2][c] * a12;
      AQ[2][c] = Q[0][c] * a02 + Q[1][c] * a12 + Q[2][c] * a22;
    }
#pragma unroll
    for ( int r = 0; r < 3; ++r )
#pragma unroll
      for ( int c = 0; c < 3; ++c ) D[r][c] = AQ[0][r] * Q[0][c] + AQ[1][r] * Q[1][c] + AQ[2][r] * Q[2][c];
    ev[0] = D[0][0];
    ev[1] = D[1][1];
    ev[2] = D[2][2];
    const double o0 = D[1][2], o1 = D[0][2], o2 = D[0][1];
    const double m0 = fabs( o0 ), m1 = fabs( o1 ), m2 = fabs( o2 );
    const int    k0 = ( m0 > m1 && m0 > m2 ) ? 0 : ( m1 > m2 ) ? 1 : 2;
    const double ok = k0 == 0 ? o0 : ( k0 == 1 ? o1 : o2 );
    if ( ok == 0.0 ) break;
    // k1 = (k0+1)%3, k2 = (k0+2)%3
    const double dk1  = k0 == 0 ? D[1][1] : ( k0 == 1 ? D[2][2] : D[0][0] );
    const double dk2  = k0 == 0 ? D[2][2] : ( k0 == 1 ? D[0][0] : D[1][1] );
    double       thet = ddiv( dk2 - dk1, 2.0 * ok );
    const double sgn  = ( thet > 0.0 ) ? 1.0 : -1.0;
    thet *= sgn;
    const double t = ddiv( sgn, thet + ( ( thet < 1.E6 ) ? dsqrt( thet * thet + 1.0 ) : thet ) );
    const double c = ddiv( 1.0, dsqrt( t * t + 1.0 ) );
    if ( c == 1.0 ) break;
    double jk = sgn * dsqrt( ddiv( 1.0 - c, 2.0 ) );
    jk *= -1.0;
    const double j3 = dsqrt( 1.0 - jk * jk );
    if ( j3 == 1.0 ) break;
    const double j0 = k0 == 0 ? jk : 0.0, j1 = k0 == 1 ? jk : 0.0, j2 = k0 == 2 ? jk : 0.0;
    // sequential in-place update (q1 uses the NEW q0, ...), as in the reference
    q0 = ( q3 * j0 + q0 * j3 + q1 * j2 - q2 * j1 );
    q1 = ( q3 * j1 - q0 * j2 + q1 * j3 + q2 * j0 );
    q2 = ( q3 * j2 + q0 * j1 - q1 * j0 + q2 * j3 );
    q3 = ( q3 * j3 - q0 * j0 - q1 * j1 - q2 * j2 );
    const double mq = dsqrt( q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 );
    q0              = ddiv( q0, mq );
    q1              = ddiv( q1, mq );
    q2              = ddiv( q2, mq );
    q3              = ddiv( q3, mq );
  }
}

template <int K>
__global__ __launch_bounds__( 256 ) void normalsKernel( const Pt* __restrict__ pts, const uint32_t* __restrict__ knn,
                                                         uint32_t n, double* __restrict__ normals ) {
  const uint32_t i = chunkedIndex();  // (XCD x on the x-th eighth of the blocks: the neighbours whose points a lane gathers lie around its own)
  if ( i >= n ) return;
  uint32_t nb[K];
  {
    const uint4* row = reinterpret_cast<const uint4*>( knn + size_t( i ) * K );
#pragma unroll
    for ( int j = 0; j < K / 4; ++j ) {
      const uint4 v = row[j];
      nb[4 * j]     = v.x;
      nb[4 * j + 1] = v.y;
      nb[4 * j + 2] = v.z;
      nb[4 * j + 3] = v.w;
    }
  }
  int px[K], py[K], pz[K];
#pragma unroll
  for ( int j = 0; j < K; ++j ) {
    const Pt p = pts[nb[j]];
    px[j]      = p.x;
    py[j]      = p.y;
    pz[j]      = p.z;
  }
  // barycentre: running fp64 sum of integers (exact), then three IEEE divisions
  double bx = 0.0, by = 0.0, bz = 0.0;
#pragma unroll
  for ( int j = 0; j < K; ++j ) {
    bx = bx + double( px[j] );
    by = by + double( py[j] );
    bz = bz + double( pz[j] );
  }
  bx = ddiv( bx, double( K ) );
  by = ddiv( by, double( K ) );
  bz = ddiv( bz, double( K ) );
  double c00 = 0, c11 = 0, c22 = 0, c01 = 0, c02 = 0, c12 = 0;
#pragma unroll
  for ( int j = 0; j < K; ++j ) {
    const double x = double( px[j] ) - bx, y = double( py[j] ) - by, z = double( pz[j] ) - bz;
    c00 += x * x;
    c11 += y * y;
    c22 += z * z;
    c01 += x * y;
    c02 += x * z;
    c12 += y * z;
  }
  const double den = double( K ) - 1.0;
  double       Q[3][3], ev[3];
  diagonalize( ddiv( c00, den ), ddiv( c01, den ), ddiv( c02, den ), ddiv( c11, den ), ddiv( c12, den ),
               ddiv( c22, den ), Q, ev );
  const double e0 = fabs( ev[0] ), e1 = fabs( ev[1] ), e2 = fabs( ev[2] );
  const int    col = ( e0 < e1 && e0 < e2 ) ? 0 : ( e1 < e2 ) ? 1 : 2;
  double       nx = col == 0 ? Q[0][0] : ( col == 1 ? Q[0][1] : Q[0][2] );
  double       ny = col == 0 ? Q[1][0] : ( col == 1 ? Q[1][1] : Q[1][2] );
  double       nz = col == 0 ? Q[2][0] : ( col == 1 ? Q[2][1] : Q[2][2] );
  // towards the view point (0,0,0)
  const Pt     self = pts[i];
  const double vx = 0.0 - double( self.x ), vy = 0.0 - double( self.y ), vz = 0.0 - double( self.z );
  if ( nx * vx + ny * vy + nz * vz < 0.0 ) {
    nx = -nx;
    ny = -ny;
    nz = -nz;
  }
  normals[3 * size_t( i )]     = nx;
  normals[3 * size_t( i ) + 1] = ny;
  normals[3 * size_t( i ) + 2] = nz;
}

struct Weights6 {
  double w[6];
};

// partition[i] = argmax over the 6 axis planes (+x,+y,+z,-x,-y,-z); plane 0 is scored unweighted,
// planes 1..5 weighted (reference quirk, PCCPatchSegmenter.cpp:250-252); first maximum wins.
__global__ __launch_bounds__( 256 ) void initialSegmentationKernel( const double* __restrict__ normals, uint32_t n,
                                                                     Weights6 wv, uint8_t* __restrict__ partition ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const double nx = normals[3 * size_t( i )], ny = normals[3 * size_t( i ) + 1], nz = normals[3 * size_t( i ) + 2];
  // n . o_j written out with the zero terms kept, so signed zeros/rounding match the reference's full dot product
  const double s[6] = {nx * 1.0 + ny * 0.0 + nz * 0.0,  nx * 0.0 + ny * 1.0 + nz * 0.0,
                       nx * 0.0 + ny * 0.0 + nz * 1.0,  nx * -1.0 + ny * 0.0 + nz * 0.0,
                       nx * 0.0 + ny * -1.0 + nz * 0.0, nx * 0.0 + ny * 0.0 + nz * -1.0};
  uint32_t     best = 0;
  double       bs   = s[0];
#pragma unroll
  for ( uint32_t j = 1; j < 6; ++j ) {
    const double sc = s[j] * wv.w[j];
    if ( sc > bs ) {
      bs   = sc;
      best = j;
    }
  }
  partition[i] = uint8_t( best );
}

}  // namespace

int launchNormals( tmc2_frame* f ) {
  if ( !f->haveKnn ) {
    setError( "normals: adjacency not computed" );
    return TMC2_E_STATE;
  }
  TMC2_TRY( f->d_normals.alloc( f->n * 3 ) );
  const int  sid = f->ctx->stageBegin( "normals" );
  const char* pcOpt = ctxOption( f->ctx, "POINT_CHUNK" );  // (0: the blocks as they come, rounds 1-5)
  const dim3  block( 256 ), grid( chunkedGrid( uint32_t( ( f->n + 255 ) / 256 ), !( pcOpt && pcOpt[0] == '0' ) ) );
  if ( f->k == 16 ) {
    hipLaunchKernelGGL( normalsKernel<16>, grid, block, 0, f->ctx->stream, f->d_pts.p, f->d_knn.p, uint32_t( f->n ),
                        f->d_normals.p );
  } else if ( f->k == 8 ) {
    hipLaunchKernelGGL( normalsKernel<8>, grid, block, 0, f->ctx->stream, f->d_pts.p, f->d_knn.p, uint32_t( f->n ),
                        f->d_normals.p );
  } else {
    f->ctx->stageEnd( sid );
    setError( "normals: k=%d not instantiated (8, 16)", f->k );
    return TMC2_E_UNSUPPORTED;
  }
  f->ctx->stageEnd( sid );
  TMC2_HIP( hipGetLastError() );
  f->haveNormals = true;
  return TMC2_OK;
}

// ---- S3, device side --------------------------------------------------------------------------------------------
// The spanning-tree growth itself is sequential and stays on the host (orient_host.cpp); what it needs per edge is
// one number, n_u . n_v, and what it produces per point is one sign.  Both ends of that are data parallel:
//   edgeDotKernel      : edgeDot[u][j] = n_u . n_knn[u][j] (same operand order and rounding as the reference's dot
//                        product, PCCMath.h operator*), so that the host walk streams rows instead of chasing normals;
//   applySignsKernel   : negate the normals the walk flipped, count how many of the results look away from the origin
//                        (orientNormals' final majority test, PCCNormalsGenerator.cpp:226-241);
//   majorityFlipKernel : negate everything if that count exceeds half -- decided on the device, no round trip.
__global__ __launch_bounds__( 256 ) void edgeDotKernel( const double* __restrict__ normals, const uint32_t* __restrict__ knn,
                                                         uint32_t n, int k, double* __restrict__ edgeDot ) {
  // (a capped grid with a stride loop: 13 M edges as 52 K workgroups made the launch itself -- workgroup dispatch -- the cost,
  // sixteen frames' worth of them at once all the more)
  const size_t total = size_t( n ) * k, stride = size_t( gridDim.x ) * blockDim.x;
  for ( size_t e = size_t( blockIdx.x ) * blockDim.x + threadIdx.x; e < total; e += stride ) {
    const size_t  u = e / k, v = knn[e];
    const double* a = normals + 3 * u;
    const double* b = normals + 3 * v;
    edgeDot[e]      = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  }
}

constexpr uint32_t kNegCounters = 64;
__global__ __launch_bounds__( 256 ) void applySignsKernel( const Pt* __restrict__ pts, const int8_t* __restrict__ sign,
                                                            uint32_t n, double* __restrict__ normals,
                                                            uint32_t* __restrict__ negCount ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool           neg = false;
  if ( i < n ) {
    double nx = normals[3 * size_t( i )], ny = normals[3 * size_t( i ) + 1], nz = normals[3 * size_t( i ) + 2];
    if ( sign[i] < 0 ) {
      nx = -nx, ny = -ny, nz = -nz;
      normals[3 * size_t( i )] = nx, normals[3 * size_t( i ) + 1] = ny, normals[3 * size_t( i ) + 2] = nz;
    }
    const Pt     p  = pts[i];
    const double tx = 0.0 - double( p.x ), ty = 0.0 - double( p.y ), tz = 0.0 - double( p.z );
    neg             = nx * tx + ny * ty + nz * tz < 0.0;
  }
  // (64 counters, 128 bytes apart: 13 K wavefronts adding to ONE word queue up for 11 ns each -- that was the kernel's 142 us)
  const unsigned long long m = __ballot( neg );
  if ( ( threadIdx.x & 63 ) == 0 && m )
    atomicAdd( &negCount[( ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6 ) % kNegCounters * 32], uint32_t( __popcll( m ) ) );
}

__global__ __launch_bounds__( 256 ) void majorityFlipKernel( const uint32_t* __restrict__ negCount, uint32_t n,
                                                              double* __restrict__ normals ) {
  uint32_t neg = negCount[( threadIdx.x & 63 ) * 32];  // (kNegCounters = a wavefront's lanes)
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) neg += __shfl_xor( neg, off, 64 );
  if ( neg <= ( n + 1 ) / 2 ) return;
  const size_t i = size_t( blockIdx.x ) * blockDim.x + threadIdx.x;
  if ( i < 3 * size_t( n ) ) normals[i] = -normals[i];
}

int launchEdgeDots( tmc2_frame* f, double* d_edgeDot ) {
  const size_t edges = f->n * size_t( f->k );
  hipLaunchKernelGGL( edgeDotKernel, dim3( cappedBlocks( f->ctx, ( edges + 255 ) / 256 ) ), dim3( 256 ), 0, f->ctx->stream, f->d_normals.p,
                      f->d_knn.p, uint32_t( f->n ), f->k, d_edgeDot );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

int launchApplyOrientation( tmc2_frame* f, const int8_t* d_sign, uint32_t* d_negCount ) {
  hipStream_t    s = f->ctx->stream;
  const uint32_t n = uint32_t( f->n );
  TMC2_HIP( hipMemsetAsync( d_negCount, 0, kOrientNegCountWords * 4, s ) );
  hipLaunchKernelGGL( applySignsKernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, s, f->d_pts.p, d_sign, n, f->d_normals.p,
                      d_negCount );
  hipLaunchKernelGGL( majorityFlipKernel, dim3( uint32_t( ( 3 * size_t( n ) + 255 ) / 256 ) ), dim3( 256 ), 0, s, d_negCount, n,
                      f->d_normals.p );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

int launchInitialSegmentation( tmc2_frame* f, const double weight[3] ) {
  if ( !f->haveNormals ) {
    setError( "initialSegmentation: normals not computed" );
    return TMC2_E_STATE;
  }
  TMC2_TRY( f->d_partition.alloc( f->n ) );
  Weights6 wv;
  for ( int j = 0; j < 6; ++j ) wv.w[j] = weight[j % 3];
  const int  sid = f->ctx->stageBegin( "initial_segmentation" );
  const dim3 block( 256 ), grid( uint32_t( ( f->n + 255 ) / 256 ) );
  hipLaunchKernelGGL( initialSegmentationKernel, grid, block, 0, f->ctx->stream, f->d_normals.p, uint32_t( f->n ), wv,
                      f->d_partition.p );
  f->ctx->stageEnd( sid );
  TMC2_HIP( hipGetLastError() );
  f->havePartition = true;
  return TMC2_OK;
}

}  // namespace tmc2
