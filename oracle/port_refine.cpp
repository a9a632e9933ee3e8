// oracle/port_refine.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S5  grid-based refinement of the projection-plane assignment
//     (PCCPatchSegmenter3::refineSegmentationGridBased, PCCPatchSegmenter.cpp:1386-1561;
//      AttributeOfGridCell / PointIndicesOfGridCell, PCCPatchSegmenter.h:425-510;
//      computeAdjacencyInfoInRadius :293-318).
// Points are binned into voxDim^3 voxels; every voxel carries a 6-bin histogram of its points' planes,
// an edge class and its dominant plane (ppi).  Each sweep smooths the per-point score with the summed
// histograms of the voxel's (distance-sorted, 1024-point-truncated) neighbourhood.  Only voxels that
// are "edge" voxels are re-scored; uniform voxels next to a disagreeing neighbourhood are pulled in as
// INDIRECT edges -- and that marking is visible to LATER voxels of the same sweep (index order).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "oracle.h"

struct orc_kdtree;
extern "C" {
orc_kdtree* orc_kdtree_build( const int16_t* xyz, size_t n );
void        orc_kdtree_free( orc_kdtree* t );
int         orc_radius( const orc_kdtree* t, const int16_t* q, size_t nq, double radius2, int cap, int32_t* count,
                        uint32_t* idx );
}

namespace {
enum : uint8_t { NO_EDGE = 0x00, INDIRECT_EDGE = 0x01, M_DIRECT_EDGE = 0x10, S_DIRECT_EDGE = 0x11 };

struct Voxel {
  std::vector<uint32_t> pts;
  uint16_t              hist[6];
  uint8_t               edge, ppi, updated;
};

void rescore( Voxel& v, const uint32_t* partition ) {
  for ( int k = 0; k < 6; ++k ) v.hist[k] = 0;
  for ( uint32_t j : v.pts ) ++v.hist[partition[j]];
  if ( !v.updated ) return;
  if ( v.edge != S_DIRECT_EDGE ) {
    int nz = 0;
    for ( int k = 0; k < 6; ++k ) nz += v.hist[k] != 0;
    v.edge = ( nz == 1 ) ? NO_EDGE : M_DIRECT_EDGE;
  }
  int best = 0;
  for ( int k = 1; k < 6; ++k )
    if ( v.hist[k] > v.hist[best] ) best = k;
  v.ppi     = uint8_t( best );
  v.updated = 0;
}
}  // namespace

// trace (optional): per iteration 4 words -- points whose plane differs from the state 1 and 2 iterations earlier, voxels whose
// edge class does (the state the reference carries across iterations is exactly (partition, edge): tools/refine_recurrence.py)
static int refineGrid( const int16_t* xyz, const double* normals, size_t n, uint32_t* partition, int maxNNCount, double lambda,
                       int iterationCount, int voxDim, int searchRadius, uint32_t* trace ) {
  static const double O[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
  if ( n == 0 ) return 0;
  // grid geometry exactly as the reference derives it (note: the key packs with gridDimShift bits per
  // axis although (coord + voxDim/2) >> shift can reach gridDim, so keys may alias -- identity IS the key)
  int16_t geoMax = xyz[0];
  for ( size_t i = 0; i < n; ++i )
    for ( int d = 0; d < 3; ++d ) geoMax = std::max( geoMax, xyz[3 * i + d] );
  size_t geoRange = 1;
  for ( size_t i = size_t( geoMax - 1 ); i != 0; i >>= 1, geoRange <<= 1 ) {}
  size_t voxShift = 0, gridShift = 0;
  for ( size_t i = size_t( voxDim ); i > 1; ++voxShift, i >>= 1 ) {}
  const size_t gridDim = geoRange >> voxShift;
  for ( size_t i = gridDim; i > 1; ++gridShift, i >>= 1 ) {}
  const size_t half = size_t( voxDim ) >> 1;

  std::unordered_map<size_t, uint32_t> slot;  // key -> voxel index (first-appearance order)
  std::vector<Voxel>                   vox;
  std::vector<int16_t>                 centers;
  for ( size_t i = 0; i < n; ++i ) {
    const size_t x0  = ( size_t( xyz[3 * i] ) + half ) >> voxShift;
    const size_t y0  = ( size_t( xyz[3 * i + 1] ) + half ) >> voxShift;
    const size_t z0  = ( size_t( xyz[3 * i + 2] ) + half ) >> voxShift;
    const size_t key = x0 + ( y0 << gridShift ) + ( z0 << ( 2 * gridShift ) );
    auto         it  = slot.find( key );
    uint32_t     v;
    if ( it == slot.end() ) {
      v = uint32_t( vox.size() );
      slot.emplace( key, v );
      vox.emplace_back();
      centers.push_back( int16_t( x0 ) );
      centers.push_back( int16_t( y0 ) );
      centers.push_back( int16_t( z0 ) );
    } else {
      v = it->second;
    }
    vox[v].pts.push_back( uint32_t( i ) );
  }
  const size_t V = vox.size();
  for ( auto& v : vox ) {
    v.updated = 1;
    v.edge    = ( uint8_t( v.pts.size() ) == 1 ) ? S_DIRECT_EDGE : M_DIRECT_EDGE;
    rescore( v, partition );
  }
  // neighbourhoods: radius search on voxel centres (squared radius, strict <), sorted by (dist, index)
  const int             cap = 32767;
  const double          r2  = double( size_t( searchRadius ) >> voxShift );
  orc_kdtree*           t   = orc_kdtree_build( centers.data(), V );
  std::vector<std::vector<uint32_t>> adj( V ), adjDev( V );
  std::vector<double>                weight( V );
  {
    std::vector<uint32_t> row( cap );
    const int             devRange = ( voxDim >= 4 ) ? 1 : 2;
    for ( size_t i = 0; i < V; ++i ) {
      int32_t cnt = 0;
      orc_radius( t, centers.data() + 3 * i, 1, r2, cap, &cnt, row.data() );
      size_t nn   = 0;
      size_t used = 0;
      for ( int32_t a = 0; a < cnt; ++a ) {
        const uint32_t j = row[a];
        if ( std::abs( centers[3 * i] - centers[3 * j] ) <= devRange &&
             std::abs( centers[3 * i + 1] - centers[3 * j + 1] ) <= devRange &&
             std::abs( centers[3 * i + 2] - centers[3 * j + 2] ) <= devRange )
          adjDev[i].push_back( j );
        nn += uint8_t( vox[j].pts.size() );
        used = size_t( a ) + 1;
        if ( nn >= size_t( maxNNCount ) ) break;
      }
      adj[i].assign( row.begin(), row.begin() + used );
      weight[i] = lambda / double( nn );
    }
  }
  orc_kdtree_free( t );

  uint16_t S[6];
  double   score[6];
  int      iter = 0;
  std::vector<uint32_t> prevP[2];
  std::vector<uint8_t>  prevE[2];
  auto snapshot = [&]( int slotIdx ) {
    prevP[slotIdx].assign( partition, partition + n );
    prevE[slotIdx].resize( V );
    for ( size_t i = 0; i < V; ++i ) prevE[slotIdx][i] = vox[i].edge;
  };
  if ( trace ) { snapshot( 0 ); snapshot( 1 ); }
  do {
    for ( size_t i = 0; i < V; ++i ) {
      Voxel&        v      = vox[i];
      const uint8_t edgeAt = v.edge;
      if ( edgeAt == NO_EDGE ) continue;
      for ( int k = 0; k < 6; ++k ) S[k] = 0;
      for ( uint32_t j : adj[i] )
        for ( int k = 0; k < 6; ++k ) S[k] = uint16_t( S[k] + vox[j].hist[k] );
      int arg = 0;
      for ( int k = 1; k < 6; ++k )
        if ( S[k] > S[arg] ) arg = k;
      for ( uint32_t j : adjDev[i] )
        if ( vox[j].edge == NO_EDGE && vox[j].ppi != arg ) vox[j].edge = INDIRECT_EDGE;
      if ( edgeAt != M_DIRECT_EDGE ) {
        int nz = 0;
        for ( int k = 0; k < 6; ++k ) nz += S[k] != 0;
        if ( nz == 1 && S[v.ppi] > 0 ) continue;
      }
      for ( uint32_t j : v.pts ) {
        const double* nm = normals + 3 * size_t( j );
        for ( int k = 0; k < 6; ++k )
          score[k] = ( nm[0] * O[k][0] + nm[1] * O[k][1] + nm[2] * O[k][2] ) + weight[i] * S[k];
        int best = 0;
        for ( int k = 1; k < 6; ++k )
          if ( score[k] > score[best] ) best = k;
        partition[j] = uint32_t( best );
      }
      v.updated = 1;
    }
    for ( auto& v : vox ) rescore( v, partition );
    if ( trace ) {  // prev[(iter+1)&1] = state after iteration iter-1, prev[iter&1] = state after iteration iter-2
      uint32_t* tr = trace + 4 * size_t( iter );
      tr[0] = tr[1] = tr[2] = tr[3] = 0;
      for ( size_t j = 0; j < n; ++j ) {
        tr[0] += partition[j] != prevP[( iter + 1 ) & 1][j];
        tr[1] += partition[j] != prevP[iter & 1][j];
      }
      for ( size_t i = 0; i < V; ++i ) {
        tr[2] += vox[i].edge != prevE[( iter + 1 ) & 1][i];
        tr[3] += vox[i].edge != prevE[iter & 1][i];
      }
      snapshot( iter & 1 );
    }
  } while ( ++iter < iterationCount );
  return 0;
}

extern "C" int orc_refine_grid( const int16_t* xyz, const double* normals, size_t n, uint32_t* partition,
                                int maxNNCount, double lambda, int iterationCount, int voxDim, int searchRadius ) {
  return refineGrid( xyz, normals, n, partition, maxNNCount, lambda, iterationCount, voxDim, searchRadius, nullptr );
}

extern "C" int orc_refine_grid_trace( const int16_t* xyz, const double* normals, size_t n, uint32_t* partition, int maxNNCount,
                                      double lambda, int iterationCount, int voxDim, int searchRadius, uint32_t* trace ) {
  return refineGrid( xyz, normals, n, partition, maxNNCount, lambda, iterationCount, voxDim, searchRadius, trace );
}
