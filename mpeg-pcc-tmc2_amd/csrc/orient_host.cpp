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
#include <chrono>
#include <cmath>

#include "internal.h"

namespace tmc2 {

namespace {

struct VertexHeap {
  std::vector<uint32_t> heap;  // vertex ids
  std::vector<int32_t>  pos;   // vertex -> slot, -1 if absent
  std::vector<double>   w;     // key part 1
  std::vector<uint32_t> s;     // key part 2 (start vertex of the best in-edge)

  explicit VertexHeap( size_t n ) : pos( n, -1 ), w( n, 0.0 ), s( n, 0 ) { heap.reserve( n / 4 + 16 ); }

  // strict "a has a smaller key than b" in the reference's edge order (weight, start, end)
  bool less( uint32_t a, uint32_t b ) const {
    if ( w[a] == w[b] ) return s[a] == s[b] ? a < b : s[a] < s[b];
    return w[a] < w[b];
  }
  void siftUp( size_t i ) {
    const uint32_t v = heap[i];
    while ( i > 0 ) {
      const size_t p = ( i - 1 ) >> 1;
      if ( !less( heap[p], v ) ) break;
      heap[i]      = heap[p];
      pos[heap[i]] = int32_t( i );
      i            = p;
    }
    heap[i] = v;
    pos[v]  = int32_t( i );
  }
  void siftDown( size_t i ) {
    const size_t   n = heap.size();
    const uint32_t v = heap[i];
    for ( ;; ) {
      size_t c = 2 * i + 1;
      if ( c >= n ) break;
      if ( c + 1 < n && less( heap[c], heap[c + 1] ) ) ++c;
      if ( !less( v, heap[c] ) ) break;
      heap[i]      = heap[c];
      pos[heap[i]] = int32_t( i );
      i            = c;
    }
    heap[i] = v;
    pos[v]  = int32_t( i );
  }
  // offer in-edge (weight, start) to unvisited vertex v
  void offer( uint32_t v, double weight, uint32_t start ) {
    if ( pos[v] < 0 ) {
      w[v] = weight;
      s[v] = start;
      heap.push_back( v );
      siftUp( heap.size() - 1 );
    } else if ( weight > w[v] || ( weight == w[v] && start > s[v] ) ) {
      w[v] = weight;
      s[v] = start;
      siftUp( size_t( pos[v] ) );
    }
  }
  uint32_t popMax() {
    const uint32_t top = heap[0];
    pos[top]           = -1;
    const uint32_t last = heap.back();
    heap.pop_back();
    if ( !heap.empty() ) {
      heap[0] = last;
      siftDown( 0 );
    }
    return top;
  }
};

inline double dot( const double* a, const double* b ) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

}  // namespace

// normals: [n][3] in/out (host), knn: [n][k] (host), xyz: [n][3]
void orientNormalsSpanningTree( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals ) {
  if ( n == 0 ) return;
  VertexHeap           heap( n );
  std::vector<uint8_t> visited( n, 0 );
  double               acc[3];
  size_t               accCount = 0;
  auto expand = [&]( uint32_t cur ) {
    acc[0] = acc[1] = acc[2] = 0.0;
    accCount                 = 0;
    const double*   nc  = normals + 3 * size_t( cur );
    const uint32_t* row = knn + size_t( cur ) * k;
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = row[j];
      if ( !visited[v] ) {
        heap.offer( v, std::fabs( dot( nc, normals + 3 * size_t( v ) ) ), cur );
      } else if ( v != cur ) {
        acc[0] += normals[3 * size_t( v )];
        acc[1] += normals[3 * size_t( v ) + 1];
        acc[2] += normals[3 * size_t( v ) + 2];
        ++accCount;
      }
    }
  };
  auto flip = [&]( size_t i ) {
    normals[3 * i]     = -normals[3 * i];
    normals[3 * i + 1] = -normals[3 * i + 1];
    normals[3 * i + 2] = -normals[3 * i + 2];
  };
  for ( size_t seed = 0; seed < n; ++seed ) {
    if ( visited[seed] ) continue;
    visited[seed] = 1;
    expand( uint32_t( seed ) );
    if ( accCount == 0 ) {
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
    if ( dot( normals + 3 * seed, acc ) < 0.0 ) flip( seed );
    while ( !heap.heap.empty() ) {
      const uint32_t v     = heap.popMax();
      const uint32_t start = heap.s[v];
      visited[v]           = 1;
      if ( dot( normals + 3 * size_t( start ), normals + 3 * size_t( v ) ) < 0.0 ) flip( v );
      expand( v );
    }
  }
  size_t negCount = 0;
  for ( size_t i = 0; i < n; ++i ) {
    const double toView[3] = {0.0 - xyz[3 * i], 0.0 - xyz[3 * i + 1], 0.0 - xyz[3 * i + 2]};
    if ( dot( normals + 3 * i, toView ) < 0.0 ) ++negCount;
  }
  if ( negCount > ( n + 1 ) / 2 )
    for ( size_t i = 0; i < n; ++i ) flip( i );
}

int orientNormalsHost( tmc2_frame* f ) {
  if ( !f->haveKnn || !f->haveNormals ) {
    setError( "orientNormals: adjacency / normals not computed" );
    return TMC2_E_STATE;
  }
  const size_t          n = f->n;
  std::vector<uint32_t> knn( n * size_t( f->k ) );
  std::vector<double>   nrm( n * 3 );
  hipStream_t           s = f->ctx->stream;
  TMC2_HIP( hipMemcpyAsync( knn.data(), f->d_knn.p, knn.size() * sizeof( uint32_t ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( nrm.data(), f->d_normals.p, nrm.size() * sizeof( double ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  const auto t0 = std::chrono::steady_clock::now();
  orientNormalsSpanningTree( f->h_xyz.data(), n, knn.data(), f->k, nrm.data() );
  const auto t1 = std::chrono::steady_clock::now();
  f->ctx->stageAddHostMs( "orient_normals_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
  TMC2_HIP( hipMemcpyAsync( f->d_normals.p, nrm.data(), nrm.size() * sizeof( double ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

}  // namespace tmc2
