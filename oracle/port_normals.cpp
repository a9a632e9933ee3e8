// oracle/port_normals.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S2  per-point PCA normal from the 16-NN covariance
//     (PCCNormalsGenerator3::computeNormal, PCCNormalsGenerator.cpp:71-157, eigen solver PCCDiagonalize,
//      PCCMath.h:505-598 -- including its in-place quaternion update, which we must reproduce).
// S3  normal orientation by greedy directed spanning-tree growth
//     (PCCNormalsGenerator3::orientNormals SPANNING_TREE branch :198-242, addNeighbors :521-548).
// S4  initial plane assignment (PCCPatchSegmenter3::initialSegmentation, PCCPatchSegmenter.cpp:226-265).
// S0  axis weights (PCCEncoder::calculateWeightNormal, PCCEncoder.cpp:3569-3626).
// fp64 throughout, no FMA contraction (compiled with -ffp-contract=off), left-to-right sums.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>

#include "oracle.h"

namespace {

// Iterative Jacobi diagonalisation on a quaternion (<= 24 steps).  A symmetric, Q orthonormal, D = Q^T A Q.
void diagonalize( const double A[3][3], double Q[3][3], double D[3][3] ) {
  double q[4] = {0.0, 0.0, 0.0, 1.0};
  for ( int step = 0; step < 24; ++step ) {
    const double sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], sqw = q[3] * q[3];
    Q[0][0] = ( sqx - sqy - sqz + sqw );
    Q[1][1] = ( -sqx + sqy - sqz + sqw );
    Q[2][2] = ( -sqx - sqy + sqz + sqw );
    double t1 = q[0] * q[1], t2 = q[2] * q[3];
    Q[1][0] = 2.0 * ( t1 + t2 );
    Q[0][1] = 2.0 * ( t1 - t2 );
    t1 = q[0] * q[2];
    t2 = q[1] * q[3];
    Q[2][0] = 2.0 * ( t1 - t2 );
    Q[0][2] = 2.0 * ( t1 + t2 );
    t1 = q[1] * q[2];
    t2 = q[0] * q[3];
    Q[2][1] = 2.0 * ( t1 + t2 );
    Q[1][2] = 2.0 * ( t1 - t2 );
    // AQ = A*Q written with A's upper triangle only (A symmetric), term order as in the reference
    double AQ[3][3];
    const double a00 = A[0][0], a01 = A[0][1], a02 = A[0][2], a11 = A[1][1], a12 = A[1][2], a22 = A[2][2];
    for ( int c = 0; c < 3; ++c ) {
      AQ[0][c] = Q[0][c] * a00 + Q[1][c] * a01 + Q[2][c] * a02;
      AQ[1][c] = Q[0][c] * a01 + Q[1][c] * a11 + Q[2][c] * a12;
      AQ[2][c] = Q[0][c] * a02 + Q[1][c] * a12 + Q[2][c] * a22;
    }
    for ( int r = 0; r < 3; ++r )
      for ( int c = 0; c < 3; ++c ) D[r][c] = AQ[0][r] * Q[0][c] + AQ[1][r] * Q[1][c] + AQ[2][r] * Q[2][c];
    const double o[3] = {D[1][2], D[0][2], D[0][1]};
    const double m[3] = {std::fabs( o[0] ), std::fabs( o[1] ), std::fabs( o[2] )};
    const int    k0   = ( m[0] > m[1] && m[0] > m[2] ) ? 0 : ( m[1] > m[2] ) ? 1 : 2;
    const int    k1 = ( k0 + 1 ) % 3, k2 = ( k0 + 2 ) % 3;
    if ( o[k0] == 0.0 ) break;
    double       thet = ( D[k2][k2] - D[k1][k1] ) / ( 2.0 * o[k0] );
    const double sgn  = ( thet > 0.0 ) ? 1.0 : -1.0;
    thet *= sgn;
    const double t = sgn / ( thet + ( ( thet < 1.E6 ) ? std::sqrt( thet * thet + 1.0 ) : thet ) );
    const double c = 1.0 / std::sqrt( t * t + 1.0 );
    if ( c == 1.0 ) break;
    double jr[4] = {0.0, 0.0, 0.0, 0.0};
    jr[k0]       = sgn * std::sqrt( ( 1.0 - c ) / 2.0 );
    jr[k0] *= -1.0;
    jr[3] = std::sqrt( 1.0 - jr[k0] * jr[k0] );
    if ( jr[3] == 1.0 ) break;
    // NOTE: sequential in-place update -- q[1] sees the NEW q[0], q[2] the new q[0],q[1], etc.
    q[0] = ( q[3] * jr[0] + q[0] * jr[3] + q[1] * jr[2] - q[2] * jr[1] );
    q[1] = ( q[3] * jr[1] - q[0] * jr[2] + q[1] * jr[3] + q[2] * jr[0] );
    q[2] = ( q[3] * jr[2] + q[0] * jr[1] - q[1] * jr[0] + q[2] * jr[3] );
    q[3] = ( q[3] * jr[3] - q[0] * jr[0] - q[1] * jr[1] - q[2] * jr[2] );
    const double mq = std::sqrt( q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] );
    q[0] /= mq;
    q[1] /= mq;
    q[2] /= mq;
    q[3] /= mq;
  }
}

inline double dot3( const double* a, const double* b ) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

}  // namespace

extern "C" {

// S2.  knn: n*k neighbour indices in nanoflann order (self first).  normals: n*3 f64.
int orc_compute_normals( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals ) {
  for ( size_t i = 0; i < n; ++i ) {
    double nrm[3] = {0.0, 0.0, 0.0};
    if ( k > 1 ) {
      double bary[3] = {0.0, 0.0, 0.0};
      for ( int j = 0; j < k; ++j ) {
        const int16_t* p = xyz + 3 * size_t( knn[i * k + j] );
        bary[0]          = bary[0] + p[0];
        bary[1]          = bary[1] + p[1];
        bary[2]          = bary[2] + p[2];
      }
      bary[0] /= double( k );
      bary[1] /= double( k );
      bary[2] /= double( k );
      double c00 = 0, c11 = 0, c22 = 0, c01 = 0, c02 = 0, c12 = 0;
      for ( int j = 0; j < k; ++j ) {
        const int16_t* p   = xyz + 3 * size_t( knn[i * k + j] );
        const double   x = p[0] - bary[0], y = p[1] - bary[1], z = p[2] - bary[2];
        c00 += x * x;
        c11 += y * y;
        c22 += z * z;
        c01 += x * y;
        c02 += x * z;
        c12 += y * z;
      }
      const double den = ( k - 1.0 );
      double       A[3][3] = {{c00 / den, c01 / den, c02 / den}, {c01 / den, c11 / den, c12 / den},
                        {c02 / den, c12 / den, c22 / den}};
      double       Q[3][3], D[3][3];
      diagonalize( A, Q, D );
      const double e0 = std::fabs( D[0][0] ), e1 = std::fabs( D[1][1] ), e2 = std::fabs( D[2][2] );
      const int    col = ( e0 < e1 && e0 < e2 ) ? 0 : ( e1 < e2 ) ? 1 : 2;
      nrm[0]           = Q[0][col];
      nrm[1]           = Q[1][col];
      nrm[2]           = Q[2][col];
    }
    // orient towards the view point (0,0,0): flip if n . (0 - p) < 0
    const int16_t* p     = xyz + 3 * i;
    const double   toV[3] = {0.0 - p[0], 0.0 - p[1], 0.0 - p[2]};
    const bool     flip  = dot3( nrm, toV ) < 0.0;
    for ( int d = 0; d < 3; ++d ) normals[3 * i + d] = flip ? -nrm[d] : nrm[d];
  }
  return 0;
}

// S3.  In-place sign propagation.  knn lists are the same k=16 self-join as S2
// (the reference re-queries the same tree with the same k).
int orc_orient_normals( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals ) {
  struct Edge {
    double   w;
    uint32_t s, e;
    bool     operator<( const Edge& r ) const {
      if ( w == r.w ) return s == r.s ? e < r.e : s < r.s;
      return w < r.w;
    }
  };
  std::priority_queue<Edge> heap;
  std::vector<uint8_t>      visited( n, 0 );
  double                    acc[3];
  size_t                    cnt;
  auto push = [&]( uint32_t cur ) {
    acc[0] = acc[1] = acc[2] = 0.0;
    cnt                      = 0;
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = knn[size_t( cur ) * k + j];
      if ( !visited[v] ) {
        heap.push( Edge{std::fabs( dot3( normals + 3 * size_t( cur ), normals + 3 * size_t( v ) ) ), cur, v} );
      } else if ( v != cur ) {
        acc[0] += normals[3 * size_t( v )];
        acc[1] += normals[3 * size_t( v ) + 1];
        acc[2] += normals[3 * size_t( v ) + 2];
        ++cnt;
      }
    }
  };
  auto neg = [&]( size_t i ) {
    normals[3 * i]     = -normals[3 * i];
    normals[3 * i + 1] = -normals[3 * i + 1];
    normals[3 * i + 2] = -normals[3 * i + 2];
  };
  for ( size_t seed = 0; seed < n; ++seed ) {
    if ( visited[seed] ) continue;
    visited[seed] = 1;
    push( uint32_t( seed ) );
    if ( cnt == 0 ) {
      if ( seed != 0 ) {
        acc[0] = normals[3 * ( seed - 1 )];
        acc[1] = normals[3 * ( seed - 1 ) + 1];
        acc[2] = normals[3 * ( seed - 1 ) + 2];
      } else {
        acc[0] = 0.0 - xyz[0];
        acc[1] = 0.0 - xyz[1];
        acc[2] = 0.0 - xyz[2];
      }
    }
    if ( dot3( normals + 3 * seed, acc ) < 0.0 ) neg( seed );
    while ( !heap.empty() ) {
      const Edge e = heap.top();
      heap.pop();
      if ( !visited[e.e] ) {
        visited[e.e] = 1;
        if ( dot3( normals + 3 * size_t( e.s ), normals + 3 * size_t( e.e ) ) < 0.0 ) neg( e.e );
        push( e.e );
      }
    }
  }
  size_t negCount = 0;
  for ( size_t i = 0; i < n; ++i ) {
    const double toV[3] = {0.0 - xyz[3 * i], 0.0 - xyz[3 * i + 1], 0.0 - xyz[3 * i + 2]};
    negCount += dot3( normals + 3 * i, toV ) < 0.0 ? 1 : 0;
  }
  if ( negCount > ( n + 1 ) / 2 )
    for ( size_t i = 0; i < n; ++i ) neg( i );
  return 0;
}

// S0.  Axis weights from the three axis-aligned projection footprints (enhancedPP).
int orc_weight_normal( const int16_t* xyz, size_t n, int geometryBitDepth3D, double minWeightEPP, double* w ) {
  const int64_t        M = int64_t( 1 ) << geometryBitDepth3D;
  std::vector<uint8_t> face( size_t( M * M * 3 ), 0 );
  for ( size_t i = 0; i < n; ++i ) {
    const int64_t p0 = std::max<int64_t>( 0, std::min<int64_t>( M - 1, xyz[3 * i] ) );
    const int64_t p1 = std::max<int64_t>( 0, std::min<int64_t>( M - 1, xyz[3 * i + 1] ) );
    const int64_t p2 = std::max<int64_t>( 0, std::min<int64_t>( M - 1, xyz[3 * i + 2] ) );
    face[p2 * M + p1]             = 1;  // footprint along x
    face[p0 * M + p2 + M * M]     = 1;  // along y
    face[p1 * M + p0 + 2 * M * M] = 1;  // along z
  }
  struct C {
    int      idx;
    uint32_t v;
  } c[3] = {{0, 0}, {1, 0}, {2, 0}};
  for ( int64_t i = 0; i < M * M; ++i ) {
    c[0].v += face[i];
    c[1].v += face[i + M * M];
    c[2].v += face[i + 2 * M * M];
  }
  std::stable_sort( c, c + 3, []( const C& a, const C& b ) { return a.v < b.v; } );  // 3 elements: insertion sort
  double a[3];
  const double r0 = double( c[0].v ) / double( c[2].v ), r1 = double( c[1].v ) / double( c[2].v );
  if ( r0 >= minWeightEPP ) {
    a[c[0].idx] = r0;
    a[c[1].idx] = r1;
    a[c[2].idx] = 1.0;
  } else {
    a[c[0].idx] = minWeightEPP;
    a[c[2].idx] = 1.0;
    a[c[1].idx] = minWeightEPP + ( r1 - r0 ) / ( 1.0 - r0 ) * ( 1 - minWeightEPP );
  }
  w[0] = a[0];
  w[1] = a[1];
  w[2] = a[2];
  return 0;
}

// S4.  6 axis planes (+x,+y,+z,-x,-y,-z); plane 0 scored UNWEIGHTED, planes 1..5 weighted; first max wins.
int orc_initial_segmentation( const double* normals, size_t n, const double* weight, uint32_t* partition ) {
  static const double O[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
  const double        wv[6]   = {weight[0], weight[1], weight[2], weight[0], weight[1], weight[2]};
  for ( size_t i = 0; i < n; ++i ) {
    const double* nm   = normals + 3 * i;
    uint32_t      best = 0;
    double        bs   = dot3( nm, O[0] );
    for ( uint32_t j = 1; j < 6; ++j ) {
      const double s = dot3( nm, O[j] ) * wv[j];
      if ( s > bs ) {
        bs   = s;
        best = j;
      }
    }
    partition[i] = best;
  }
  return 0;
}

}  // extern "C"
