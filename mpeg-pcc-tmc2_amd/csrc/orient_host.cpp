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
#include <cstring>

#include "internal.h"

namespace tmc2 {

namespace {

// per-vertex state, one 16-byte record (one cache line touch per neighbour test)
struct VState {
  double   w;    // weight of the best pending in-edge
  uint32_t s;    // its start vertex
  int32_t  pos;  // slot in the heap; kAbsent = no pending in-edge yet; kVisited = already oriented
};
constexpr int32_t kAbsent = -1, kVisited = -2;

// heap entries carry their own key, so sifting compares contiguous records instead of chasing vertex ids
struct HeapEntry {
  double   w;
  uint32_t s, v;
};

struct VertexHeap {
  std::vector<HeapEntry> heap;
  VState*                st;

  VertexHeap( size_t n, VState* state ) : st( state ) {
    for ( size_t i = 0; i < n; ++i ) st[i] = VState{0.0, 0u, kAbsent};
    heap.reserve( n / 4 + 16 );
  }
  // strict "a has a smaller key than b" in the reference's edge order (weight, start, end)
  static bool less( const HeapEntry& a, const HeapEntry& b ) {
    if ( a.w == b.w ) return a.s == b.s ? a.v < b.v : a.s < b.s;
    return a.w < b.w;
  }
  void siftUp( size_t i ) {
    const HeapEntry e = heap[i];
    while ( i > 0 ) {
      const size_t p = ( i - 1 ) >> 1;
      if ( !less( heap[p], e ) ) break;
      heap[i]            = heap[p];
      st[heap[i].v].pos  = int32_t( i );
      i                  = p;
    }
    heap[i]     = e;
    st[e.v].pos = int32_t( i );
  }
  void siftDown( size_t i ) {
    const size_t    n = heap.size();
    const HeapEntry e = heap[i];
    for ( ;; ) {
      size_t c = 2 * i + 1;
      if ( c >= n ) break;
      if ( c + 1 < n && less( heap[c], heap[c + 1] ) ) ++c;
      if ( !less( e, heap[c] ) ) break;
      heap[i]           = heap[c];
      st[heap[i].v].pos = int32_t( i );
      i                 = c;
    }
    heap[i]     = e;
    st[e.v].pos = int32_t( i );
  }
  // offer in-edge (weight, start) to unvisited vertex v
  void offer( uint32_t v, double weight, uint32_t start ) {
    VState& sv = st[v];
    if ( sv.pos == kAbsent ) {
      sv.w = weight;
      sv.s = start;
      heap.push_back( HeapEntry{weight, start, v} );
      siftUp( heap.size() - 1 );
    } else if ( weight > sv.w || ( weight == sv.w && start > sv.s ) ) {
      sv.w                 = weight;
      sv.s                 = start;
      heap[sv.pos].w       = weight;
      heap[sv.pos].s       = start;
      siftUp( size_t( sv.pos ) );
    }
  }
  HeapEntry popMax() {
    const HeapEntry top  = heap[0];
    const HeapEntry last = heap.back();
    heap.pop_back();
    if ( !heap.empty() ) {
      heap[0] = last;
      siftDown( 0 );
    }
    st[top.v].pos = kVisited;
    return top;
  }
};

inline double dot( const double* a, const double* b ) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

}  // namespace

// normals: [n][3] in/out (host), knn: [n][k] (host), xyz: [n][3]; scratch: n * 16 bytes or nullptr
void orientNormalsSpanningTree( const int16_t* xyz, size_t n, const uint32_t* knn, int k, double* normals, void* scratch ) {
  if ( n == 0 ) return;
  std::vector<VState> own;
  if ( !scratch ) {
    own.resize( n );
    scratch = own.data();
  }
  VState*    st = reinterpret_cast<VState*>( scratch );
  VertexHeap heap( n, st );
  double     acc[3];
  size_t     accCount = 0;
  auto expand = [&]( uint32_t cur ) {
    acc[0] = acc[1] = acc[2] = 0.0;
    accCount                 = 0;
    const double*   nc  = normals + 3 * size_t( cur );
    const uint32_t* row = knn + size_t( cur ) * k;
    for ( int j = 0; j < k; ++j ) {
      const uint32_t v = row[j];
      if ( st[v].pos != kVisited ) {
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
    if ( st[seed].pos == kVisited ) continue;
    // a seed is never in the heap: the heap is empty whenever a new seed is picked
    st[seed].pos = kVisited;
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
      const HeapEntry e = heap.popMax();
      if ( dot( normals + 3 * size_t( e.s ), normals + 3 * size_t( e.v ) ) < 0.0 ) flip( e.v );
      expand( e.v );
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
  const size_t n   = f->n;
  tmc2_ctx*    ctx = f->ctx;
  uint32_t*    knn = ctx->hostA.get<uint32_t>( n * size_t( f->k ) );
  double*      nrm = ctx->hostB.get<double>( n * 3 );
  uint8_t*     scr = ctx->hostC.get<uint8_t>( n * 16 );
  if ( !knn || !nrm || !scr ) {
    setError( "orientNormals: hipHostMalloc failed" );
    return TMC2_E_HIP;
  }
  hipStream_t s = ctx->stream;
  TMC2_HIP( hipMemcpyAsync( knn, f->d_knn.p, n * size_t( f->k ) * sizeof( uint32_t ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipMemcpyAsync( nrm, f->d_normals.p, n * 3 * sizeof( double ), hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  std::chrono::steady_clock::time_point t0, t1;
  {
    HostGate gate;
    t0 = std::chrono::steady_clock::now();
    orientNormalsSpanningTree( f->h_xyz.data(), n, knn, f->k, nrm, scr );
    t1 = std::chrono::steady_clock::now();
  }
  ctx->stageAddHostMs( "orient_normals_host", std::chrono::duration<double, std::milli>( t1 - t0 ).count() );
  TMC2_HIP( hipMemcpyAsync( f->d_normals.p, nrm, n * 3 * sizeof( double ), hipMemcpyHostToDevice, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  return TMC2_OK;
}

}  // namespace tmc2
