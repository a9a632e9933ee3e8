// refine.hip -- grid-based refinement of the projection-plane assignment (S5) on gfx950.
//
// Replaces PCCPatchSegmenter3::refineSegmentationGridBased (reference:
// source/lib/PccLibEncoder/source/PCCPatchSegmenter.cpp:1386-1561), its voxel bookkeeping classes
// (PCCPatchSegmenter.h:425-510) and the voxel-centre radius search computeAdjacencyInfoInRadius (:293-318).
//
// Reference shape: hash maps of voxels, a k-d tree over voxel centres, and `iterationCount` sequential
// sweeps over the voxel list.  MI355X shape:
//   * voxelisation without hashing or sorting: a dense key table in HBM (2^(3s+1) words; 128 MiB at
//     vox10 -- trivial next to 288 GB, kept all-ones between frames and cleaned by scatter) receives
//     atomicMin(first point index); a flag+prefix-sum over the POINTS then numbers the voxels in
//     first-appearance order, which is exactly the reference's voxel order (:1422-1434).
//   * neighbourhoods by direct lookup of the <=1357 integer offsets of the radius ball in that table,
//     one wavefront per voxel, bitonic sort of (dist^2, voxel id) in LDS -- the radius search result is
//     canonical (sorted by (dist,index), nanoflann.hpp:945-952), so no tree is needed -- followed by a
//     wave prefix sum of member counts for the 1024-point truncation (:1484-1501).
//   * each sweep is Jacobi in the histograms (they are refreshed only at the end of a sweep); the only
//     sequential coupling is the INDIRECT_EDGE marking, visible to later voxels of the same sweep
//     (:1513, :1528-1532).  It is reachability in a static DAG (voxel-index order): one data-parallel round from the
//     voxels active at sweep start, then a single workgroup drains the dependent tail from a queue (exact, no polling).
//   * per-point re-scoring is one coalesced pass over the points (voxel id, 24 B normal, 1 B label).
// Scores are fp64: (n . o_k) + w_v * S_k with the products/sums in the reference's order, no FMA.
#include <algorithm>
#include <cstdlib>

#include "internal.h"

namespace tmc2 {
namespace {

enum : uint8_t { NO_EDGE = 0x00, INDIRECT_EDGE = 0x01, M_DIRECT_EDGE = 0x10, S_DIRECT_EDGE = 0x11 };

struct Grid {
  int      voxShift, gridShift, half;
  uint32_t tableSize;
};

__device__ __forceinline__ uint32_t cellKey( int x0, int y0, int z0, int s ) {
  return uint32_t( x0 ) + ( uint32_t( y0 ) << s ) + ( uint32_t( z0 ) << ( 2 * s ) );
}

// ---- voxelisation ---------------------------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void voxelKeyKernel( const Pt* __restrict__ pts, uint32_t n, Grid g,
                                                          uint32_t* __restrict__ key, uint32_t* __restrict__ table ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const Pt       p = pts[i];
  const uint32_t k = cellKey( ( int( p.x ) + g.half ) >> g.voxShift, ( int( p.y ) + g.half ) >> g.voxShift,
                              ( int( p.z ) + g.half ) >> g.voxShift, g.gridShift );
  key[i]           = k;
  atomicMin( &table[k], i );
}

__global__ __launch_bounds__( 256 ) void firstFlagKernel( const uint32_t* __restrict__ key,
                                                           const uint32_t* __restrict__ table, uint32_t n,
                                                           uint32_t* __restrict__ flag ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) flag[i] = ( table[key[i]] == i ) ? 1u : 0u;
}

// vid[i] = rank of the voxel's first point; member counts; centre of each voxel
__global__ __launch_bounds__( 256 ) void assignVoxelKernel( const Pt* __restrict__ pts, const uint32_t* __restrict__ key,
                                                             const uint32_t* __restrict__ table,
                                                             const uint32_t* __restrict__ rank, uint32_t n, Grid g,
                                                             uint32_t* __restrict__ vid, uint32_t* __restrict__ count,
                                                             Pt* __restrict__ centre ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint32_t first = table[key[i]];
  const uint32_t v     = rank[first];
  vid[i]               = v;
  atomicAdd( &count[v], 1u );
  if ( first == i ) {
    const Pt p = pts[i];
    centre[v]  = Pt{int16_t( ( int( p.x ) + g.half ) >> g.voxShift ), int16_t( ( int( p.y ) + g.half ) >> g.voxShift ),
                   int16_t( ( int( p.z ) + g.half ) >> g.voxShift ), 0};
  }
}

// table: first point index -> voxel id (only the voxel's first point writes)
__global__ __launch_bounds__( 256 ) void tableToVoxelKernel( const uint32_t* __restrict__ key,
                                                              const uint32_t* __restrict__ flag,
                                                              const uint32_t* __restrict__ vid, uint32_t n,
                                                              uint32_t* __restrict__ table ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n && flag[i] ) table[key[i]] = vid[i];
}

__global__ __launch_bounds__( 256 ) void tableCleanKernel( const uint32_t* __restrict__ key, uint32_t n,
                                                            uint32_t* __restrict__ table ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) table[key[i]] = 0xFFFFFFFFu;
}

// ---- histograms -----------------------------------------------------------------------------------
// hist[v] = 8 x u16 packed in a uint4 (bins 0..5 used).  Counts <= 255, so u16 halves never carry.
__global__ __launch_bounds__( 256 ) void histAccumulateKernel( const uint32_t* __restrict__ vid,
                                                                const uint8_t* __restrict__ partition,
                                                                const uint8_t* __restrict__ procMask, uint32_t n,
                                                                uint32_t* __restrict__ hist ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint32_t v = vid[i];
  if ( procMask && !procMask[v] ) return;
  const uint32_t k = partition[i];
  atomicAdd( &hist[4 * size_t( v ) + ( k >> 1 )], 1u << ( 16 * ( k & 1 ) ) );
}

__device__ __forceinline__ void unpackHist( const uint4 h, uint32_t ( &b )[6] ) {
  b[0] = h.x & 0xFFFF;
  b[1] = h.x >> 16;
  b[2] = h.y & 0xFFFF;
  b[3] = h.y >> 16;
  b[4] = h.z & 0xFFFF;
  b[5] = h.z >> 16;
}

__device__ __forceinline__ void classify( const uint32_t ( &b )[6], int& nonZero, int& arg ) {
  nonZero = 0;
  arg     = 0;
#pragma unroll
  for ( int k = 0; k < 6; ++k ) nonZero += b[k] != 0;
#pragma unroll
  for ( int k = 1; k < 6; ++k )
    if ( b[k] > b[arg] ) arg = k;
}

__global__ __launch_bounds__( 256 ) void initVoxelStateKernel( const uint4* __restrict__ hist,
                                                                const uint32_t* __restrict__ count, uint32_t V,
                                                                uint8_t* __restrict__ edge, uint8_t* __restrict__ ppi,
                                                                uint32_t* __restrict__ active ) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if ( v >= V ) return;
  uint32_t b[6];
  unpackHist( hist[v], b );
  int nz, arg;
  classify( b, nz, arg );
  uint8_t e = ( uint8_t( count[v] ) == 1 ) ? S_DIRECT_EDGE : M_DIRECT_EDGE;
  if ( e != S_DIRECT_EDGE ) e = ( nz == 1 ) ? NO_EDGE : M_DIRECT_EDGE;
  edge[v]   = e;
  ppi[v]    = uint8_t( arg );
  active[v] = e != NO_EDGE;
}

// ---- neighbourhoods ---------------------------------------------------------------------------------
// One wavefront per voxel.  offsets[] = all integer (dx,dy,dz) with d2 < radius2, packed, any order.
// Collect hits (d2 << idBits | voxel id) in LDS, bitonic-sort, cut after the cumulative member count reaches maxNN; write
// row length, weight, the row itself (rows back to back: the wave reserves its slots from a cursor -- what a frame needs
// is ~ maxNN / (points per voxel) entries per voxel, a fraction of the ball) and the DEV row: the members of the row
// within Chebyshev distance devRange (1 for voxels of 4 and more, 2 for voxels of 2: PCCPatchSegmenter.cpp:1469-1492), in
// row order, padded to devStride entries.
constexpr uint32_t kDevPad = 0xFFFFFFFFu;

template <int CAP, int WAVES>
__global__ __launch_bounds__( 64 * WAVES ) void neighbourhoodKernel(
    const Pt* __restrict__ centre, const uint32_t* __restrict__ count, const uint32_t* __restrict__ table, Grid g, uint32_t V,
    const int* __restrict__ offsets, int nOffsets, int maxNN, double lambda, int idBits, int devRange, uint32_t devStride,
    uint32_t rowCapacity, uint32_t* __restrict__ rowLen, uint32_t* __restrict__ devLen, double* __restrict__ weight,
    uint32_t* __restrict__ adjOff, uint32_t* __restrict__ adj, uint32_t* __restrict__ dev, uint32_t* __restrict__ rowCursor,
    uint32_t* __restrict__ overflow ) {
  __shared__ uint32_t keysAll[WAVES][CAP];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t      v    = blockIdx.x * WAVES + wave;
  uint32_t*           keys = keysAll[wave];
  if ( v >= V ) {  // a wave of the last workgroup without a voxel: it reserves nothing, but stays for the two barriers of the
    if ( lane == 0 ) keys[CAP - 2] = 0;  // row reservation below (leaving before them is undefined, whatever the hardware does)
    __syncthreads();
    __syncthreads();
    return;
  }
  const uint32_t idMask  = ( 1u << idBits ) - 1u;
  const Pt       c       = centre[v];
  const int      gridMax = 1 << g.gridShift;  // cell coordinates run 0..gridMax inclusive
  int            hits    = 0;
  // The ball's cells in batches of kBatch x 64: all table look-ups of a batch are issued before the first one is needed, then
  // all centre look-ups of its hits (two dependent round trips per batch instead of two per 64 cells: the kernel is bound by
  // exactly this latency).  The order of the hits does not matter: they are sorted below.
  constexpr int kBatch = 8;
  for ( int base = 0; base < nOffsets; base += 64 * kBatch ) {
    uint32_t u[kBatch], d2[kBatch], cell[kBatch];
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      const int o = base + 64 * k + lane;
      u[k]        = 0xFFFFFFFFu;
      d2[k] = cell[k] = 0;
      if ( o < nOffsets ) {
        const int packed = offsets[o];
        const int dx = ( packed & 0xFF ) - 128, dy = ( ( packed >> 8 ) & 0xFF ) - 128, dz = ( ( packed >> 16 ) & 0xFF ) - 128;
        const int x = c.x + dx, y = c.y + dy, z = c.z + dz;
        if ( x >= 0 && y >= 0 && z >= 0 && x <= gridMax && y <= gridMax && z <= gridMax ) {
          u[k]    = table[cellKey( x, y, z, g.gridShift )];
          d2[k]   = uint32_t( dx * dx + dy * dy + dz * dz );
          cell[k] = uint32_t( x ) | ( uint32_t( y ) << 10 ) | ( uint32_t( z ) << 20 );  // (cell coordinates are at most 512)
        }
      }
    }
    uint32_t key[kBatch];
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      key[k] = 0xFFFFFFFFu;
      if ( u[k] != 0xFFFFFFFFu ) {
        const Pt cu = centre[u[k]];  // aliased keys: accept only the voxel whose centre really sits here
        if ( ( uint32_t( cu.x ) | ( uint32_t( cu.y ) << 10 ) | ( uint32_t( cu.z ) << 20 ) ) == cell[k] ) key[k] = ( d2[k] << idBits ) | u[k];
      }
    }
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      const unsigned long long m = __ballot( key[k] != 0xFFFFFFFFu );
      if ( key[k] != 0xFFFFFFFFu ) keys[hits + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = key[k];
      hits += __popcll( m );
    }
  }
  // Only the head of the sorted list is ever used: the row ends where the running member count reaches maxNN.  The squared
  // distance takes few values, so the cut is found BEFORE sorting -- member totals per distance (LDS atomics into the unused
  // tail of the key array), a wave scan over them -- and the hits beyond the distance the cut falls in are dropped: the
  // sort below handles ~ 100-150 keys instead of the 300-400 voxels of the whole ball.
  {
    uint32_t* bins = keys + CAP - 160;  // (the host checks that the ball leaves this room)
    for ( int b = lane; b < 128; b += 64 ) bins[b] = 0;
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    for ( int i = lane; i < hits; i += 64 ) {
      const uint32_t key = keys[i];
      atomicAdd( &bins[min( key >> idBits, 127u )], count[key & idMask] & 0xFFu );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    const uint32_t b0 = bins[2 * lane], b1 = bins[2 * lane + 1];
    uint32_t       inc = b0 + b1;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 );
      if ( lane >= off ) inc += t;
    }
    const uint32_t           before  = inc - ( b0 + b1 );
    const unsigned long long reached = __ballot( inc >= uint32_t( maxNN ) );
    uint32_t                 cutoff  = 127;  // (fewer members than maxNN in the whole ball: everything stays)
    if ( reached ) {
      const int first = __ffsll( (long long)reached ) - 1;
      cutoff          = 2u * uint32_t( first ) + ( __shfl( before + b0, first, 64 ) >= uint32_t( maxNN ) ? 0u : 1u );
    }
    int kept = 0;
    for ( int base = 0; base < hits; base += 64 ) {  // in place: a chunk is read whole before anything lands at or below it
      const int      i    = base + lane;
      const uint32_t key  = i < hits ? keys[i] : 0xFFFFFFFFu;
      const bool     stay = i < hits && ( key >> idBits ) <= cutoff;
      const unsigned long long m = __ballot( stay );
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      if ( stay ) keys[kept + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = key;
      kept += __popcll( m );
    }
    hits = kept;
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  }
  // pad to a power of two and sort ascending (wave-private LDS: no barrier needed beyond wave lockstep,
  // but LDS visibility between lanes needs the s_waitcnt the compiler inserts for __syncthreads-free code:
  // use __builtin_amdgcn_wave_barrier to keep the order of LDS operations)
  int P = 64;
  while ( P < hits ) P <<= 1;
  for ( int i = hits + lane; i < P; i += 64 ) keys[i] = 0xFFFFFFFFu;
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  for ( int k = 2; k <= P; k <<= 1 ) {
    for ( int j = k >> 1; j > 0; j >>= 1 ) {
      for ( int i = lane; i < P; i += 64 ) {
        const int partner = i ^ j;
        if ( partner > i ) {
          const uint32_t a = keys[i], b = keys[partner];
          const bool     up = ( i & k ) == 0;
          if ( ( a > b ) == up ) {
            keys[i]       = b;
            keys[partner] = a;
          }
        }
      }
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    }
  }
  // truncation: first position where the running member count reaches maxNN (inclusive)
  uint32_t running = 0;
  int      used    = hits;
  uint32_t nn      = 0;
  bool     done    = false;
  for ( int base = 0; base < hits && !done; base += 64 ) {
    const int i   = base + lane;
    uint32_t  cnt = ( i < hits ) ? ( count[keys[i] & idMask] & 0xFFu ) : 0u;
    uint32_t  inc = cnt;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 );
      if ( lane >= off ) inc += t;
    }
    inc += running;
    const unsigned long long m = __ballot( i < hits && inc >= uint32_t( maxNN ) );
    if ( m ) {
      const int firstLane = __ffsll( (long long)m ) - 1;
      used                = base + firstLane + 1;
      nn                  = __shfl( inc, firstLane, 64 );
      done                = true;
    } else {
      running = __shfl( inc, 63, 64 );
      nn      = running;
    }
  }
  // the row's place in the table: ONE reservation per workgroup (70 K voxels queueing on a single address otherwise); the
  // bookkeeping words sit in the unused tail of the key arrays (a workgroup's LDS is exactly a fifth of the CU's)
  if ( lane == 0 ) keys[CAP - 2] = uint32_t( used );
  __syncthreads();  // (waves of the last workgroup that have no voxel have left: they wrote a zero)
  if ( threadIdx.x == 0 ) {
    uint32_t total = 0;
    for ( int w = 0; w < WAVES; ++w ) total += keysAll[w][CAP - 2];
    keysAll[0][CAP - 1] = atomicAdd( rowCursor, total );  // (the final cursor = size of the reverse rows)
  }
  __syncthreads();
  uint32_t rowBase = keysAll[0][CAP - 1];
  for ( int w = 0; w < wave; ++w ) rowBase += keysAll[w][CAP - 2];
  const bool fit = uint64_t( rowBase ) + uint32_t( used ) <= rowCapacity;
  if ( lane == 0 ) {
    rowLen[v] = uint32_t( used );
    adjOff[v] = fit ? rowBase : 0u;
    weight[v] = __ddiv_rn( lambda, double( nn ) );
    if ( !fit ) *overflow = 1u;  // the host repeats the pass with room for whole balls
  }
  if ( fit )
    for ( int i = lane; i < used; i += 64 ) adj[size_t( rowBase ) + i] = keys[i] & idMask;
  // DEV row.  Chebyshev distance <= R implies d2 <= 3 R^2: candidates are a prefix of the (sorted) row.
  const uint32_t d2Max = uint32_t( 3 * devRange * devRange );
  uint32_t*      drow  = dev + size_t( v ) * devStride;
  uint32_t       nDev  = 0;
  for ( int base = 0; base < used; base += 64 ) {
    const int i   = base + lane;
    bool      in  = false;
    uint32_t  u   = 0;
    const bool near = i < used && ( keys[i] >> idBits ) <= d2Max;
    if ( near ) {
      u           = keys[i] & idMask;
      const Pt cu = centre[u];
      in          = abs( int( cu.x ) - int( c.x ) ) <= devRange && abs( int( cu.y ) - int( c.y ) ) <= devRange &&
           abs( int( cu.z ) - int( c.z ) ) <= devRange;
    }
    const unsigned long long m = __ballot( in );
    if ( in ) {
      const uint32_t pos = nDev + uint32_t( __popcll( m & ( ( 1ull << lane ) - 1ull ) ) );
      if ( pos < devStride ) drow[pos] = u;
    }
    nDev += uint32_t( __popcll( m ) );
    if ( !__ballot( near ) ) break;
  }
  if ( nDev > devStride ) {  // (cannot happen: (2 R + 1)^3 <= devStride by construction)
    if ( lane == 0 ) *overflow = 2u;
    nDev = devStride;
  }
  for ( uint32_t i = nDev + lane; i < devStride; i += 64 ) drow[i] = kDevPad;
  if ( lane == 0 ) devLen[v] = nDev;
}

// ---- sweep kernels ----------------------------------------------------------------------------------
// S[v] = sum of the neighbourhood's histograms (u16 lanes), arg[v] = first maximum.  16 lanes per voxel (rows hold ~64
// entries: four independent gathers per lane, a four-step reduction).
// UPDATE = true also closes the PREVIOUS sweep for voxel v (refresh edge class / ppi of re-scored voxels, apply the
// INDIRECT marks, arm `active`): that only reads hist[v] and nothing here writes histograms, so it rides along.
template <bool UPDATE>
__global__ __launch_bounds__( 256 ) void smoothKernel( const uint4* __restrict__ hist, const uint32_t* __restrict__ adjOff,
                                                        const uint32_t* __restrict__ rowLen,
                                                        const uint32_t* __restrict__ adj, uint32_t V,
                                                        uint4* __restrict__ S, uint8_t* __restrict__ arg,
                                                        const uint8_t* __restrict__ proc, uint8_t* __restrict__ edge,
                                                        uint8_t* __restrict__ ppi, uint32_t* __restrict__ active,
                                                        uint8_t* __restrict__ marked, uint32_t* __restrict__ flags, int iter ) {
  const int      lane = threadIdx.x & 15;
  const uint32_t v    = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  if ( v >= V || flags[0] ) return;
  const uint32_t* row = adj + adjOff[v];
  const uint32_t  len = rowLen[v];
  uint32_t        s0 = 0, s1 = 0, s2 = 0;  // packed u16 pairs; sums <= 1024 + 255, no carry between halves
  for ( uint32_t i = lane; i < len; i += 16 ) {
    const uint4 h = hist[row[i]];
    s0 += h.x;
    s1 += h.y;
    s2 += h.z;
  }
#pragma unroll
  for ( int off = 8; off > 0; off >>= 1 ) {  // the 16 lanes of a voxel are an aligned quarter of the wave
    s0 += __shfl_xor( s0, off, 64 );
    s1 += __shfl_xor( s1, off, 64 );
    s2 += __shfl_xor( s2, off, 64 );
  }
  if ( lane == 0 ) {
    const uint4 out = make_uint4( s0, s1, s2, 0 );
    S[v]            = out;
    uint32_t b[6];
    unpackHist( out, b );
    int nz, a;
    classify( b, nz, a );
    arg[v] = uint8_t( a );
    if ( UPDATE ) {
      const uint8_t e0 = edge[v];
      uint8_t       e  = e0;
      bool          changed = false;
      if ( proc[v] ) {
        unpackHist( hist[v], b );
        classify( b, nz, a );
        if ( e != S_DIRECT_EDGE ) e = ( nz == 1 ) ? NO_EDGE : M_DIRECT_EDGE;
        changed = ppi[v] != uint8_t( a );
        ppi[v]  = uint8_t( a );
      } else if ( marked[v] && e == NO_EDGE ) {
        e = INDIRECT_EDGE;
      }
      if ( changed || e != e0 ) flags[2 * iter] = 1u;  // the voxel state the previous sweep leaves differs from what it found
      edge[v]   = e;
      active[v] = e != NO_EDGE;
      marked[v] = 0;
    }
  }
}

// INDIRECT-edge closure.  Every active voxel u marks the uniform DEV neighbours v that disagree with arg[u]; a marked
// neighbour with a LARGER index becomes active in this sweep and must mark in turn (the reference's in-order loop).
// edge / ppi / arg are frozen during the closure, so the "u marks v" relation is a static DAG:
//   round 0 (closureRoundZeroKernel, all voxels, 32 lanes per voxel): out[u] = compact list of the voxels u would
//            mark; the voxels active at sweep start issue their marks and flag what they newly activate in a bitmap;
//   tail    (closureTailKernel, ONE workgroup): walks the dependent hops level by level with the active / frontier
//            bitmaps in LDS, so a hop costs one global load (the 16-byte head of out[u]) instead of a chain of them.
//            The tail has next to no parallelism, so one workgroup loses nothing, needs no cross-workgroup polling,
//            terminates exactly when the frontier is empty, and leaves the rest of the chip to the other frames.
__global__ __launch_bounds__( 256 ) void closureRoundZeroKernel( const uint8_t* __restrict__ edge, const uint8_t* __restrict__ ppi,
                                                                  const uint8_t* __restrict__ arg,
                                                                  const uint32_t* __restrict__ dev, uint32_t V,
                                                                  uint32_t* __restrict__ active, uint8_t* __restrict__ marked,
                                                                  uint32_t* __restrict__ out, uint32_t* __restrict__ activeBits,
                                                                  uint32_t* __restrict__ frontierBits, uint32_t* __restrict__ flags,
                                                                  int iter ) {
  const uint32_t u    = blockIdx.x * 8 + ( threadIdx.x >> 5 );
  const uint32_t lane = threadIdx.x & 31;
  const int      half = ( threadIdx.x >> 5 ) & 1;
  // Fixpoint: a sweep that changed neither a label nor a voxel state leaves the next one the same input, so every
  // later sweep is a no-op too (the reference keeps iterating to its fixed count; the result is the same).
  if ( flags[0] ) return;
  if ( iter > 0 && flags[2 * iter - 1] == 0 && flags[2 * iter] == 0 ) {
    flags[0] = 1u;
    return;
  }
  if ( u >= V ) return;
  const uint32_t v    = dev[size_t( u ) * 32 + lane];
  const uint8_t  a    = arg[u];
  const bool     pred = v != kDevPad && edge[v] == NO_EDGE && ppi[v] != a;
  const uint32_t m    = uint32_t( __ballot( pred ) >> ( 32 * half ) );
  if ( pred ) out[size_t( u ) * 32 + 1 + __popc( m & ( ( 1u << lane ) - 1u ) )] = v;
  if ( lane == 0 ) out[size_t( u ) * 32] = uint32_t( __popc( m ) );
  // a voxel activated by a lower one during this very kernel may or may not be seen here; either way it is in the
  // frontier bitmap, and being handled twice is harmless -- marking is idempotent
  if ( !__hip_atomic_load( &active[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) return;
  if ( lane == 0 ) atomicOr( &activeBits[u >> 5], 1u << ( u & 31 ) );
  if ( pred ) {
    marked[v] = 1;
    if ( v > u && atomicExch( &active[v], 1u ) == 0u ) {
      atomicOr( &activeBits[v >> 5], 1u << ( v & 31 ) );
      atomicOr( &frontierBits[v >> 5], 1u << ( v & 31 ) );
    }
  }
}

__global__ __launch_bounds__( 1024 ) void closureTailKernel( const uint32_t* __restrict__ out, uint32_t W,
                                                              uint32_t* __restrict__ active, uint8_t* __restrict__ marked,
                                                              uint32_t* __restrict__ activeBits,
                                                              uint32_t* __restrict__ frontierBits, const uint32_t* __restrict__ flags ) {
  if ( flags[0] ) return;
  extern __shared__ uint32_t lds[];
  uint32_t *      act = lds, *fr = lds + W, *nx = lds + 2 * size_t( W );
  __shared__ int  any;
  for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) {
    act[w] = activeBits[w];
    fr[w]  = frontierBits[w];
    nx[w]  = 0;
    if ( fr[w] ) frontierBits[w] = 0;  // both bitmaps are handed back empty for the next sweep
    if ( act[w] ) activeBits[w] = 0;
  }
  if ( threadIdx.x == 0 ) any = 0;
  __syncthreads();
  while ( true ) {
    bool mine = false;
    for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) {
      uint32_t bits = fr[w];
      while ( bits ) {
        const uint32_t u = w * 32 + uint32_t( __ffs( int( bits ) ) - 1 );
        bits &= bits - 1;
        const uint32_t* row  = out + size_t( u ) * 32;
        const uint4     head = *reinterpret_cast<const uint4*>( row );  // count + the first three targets
        const uint32_t  cnt  = head.x;
        for ( uint32_t k = 0; k < cnt; ++k ) {
          const uint32_t v = k == 0 ? head.y : ( k == 1 ? head.z : ( k == 2 ? head.w : row[1 + k] ) );
          marked[v]        = 1;
          if ( v > u ) {
            const uint32_t bit = 1u << ( v & 31 );
            if ( !( atomicOr( &act[v >> 5], bit ) & bit ) ) {
              active[v] = 1u;
              atomicOr( &nx[v >> 5], bit );
              mine = true;
            }
          }
        }
      }
    }
    if ( mine ) any = 1;
    __syncthreads();
    const bool more = any != 0;
    __syncthreads();
    if ( !more ) break;
    for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) {
      fr[w] = nx[w];
      nx[w] = 0;
    }
    if ( threadIdx.x == 0 ) any = 0;
    __syncthreads();
  }
}

// fallback of the tail for grids whose bitmaps do not fit the LDS: same walk, bitmaps in global memory
__global__ __launch_bounds__( 1024 ) void closureTailGlobalKernel( const uint32_t* __restrict__ out, uint32_t W,
                                                                    uint32_t* __restrict__ active, uint8_t* __restrict__ marked,
                                                                    uint32_t* __restrict__ activeBits,
                                                                    uint32_t* __restrict__ frontierBits,
                                                                    uint32_t* __restrict__ nextBits, const uint32_t* __restrict__ flags ) {
  if ( flags[0] ) return;
  __shared__ int any;
  uint32_t*      fr = frontierBits;
  uint32_t*      nx = nextBits;
  if ( threadIdx.x == 0 ) any = 0;
  __syncthreads();
  while ( true ) {
    bool mine = false;
    for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) {
      uint32_t bits = __hip_atomic_load( &fr[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( bits ) __hip_atomic_store( &fr[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      while ( bits ) {
        const uint32_t u = w * 32 + uint32_t( __ffs( int( bits ) ) - 1 );
        bits &= bits - 1;
        const uint32_t* row = out + size_t( u ) * 32;
        const uint32_t  cnt = row[0];
        for ( uint32_t k = 0; k < cnt; ++k ) {
          const uint32_t v = row[1 + k];
          marked[v]        = 1;
          if ( v > u ) {
            const uint32_t bit = 1u << ( v & 31 );
            if ( !( atomicOr( &activeBits[v >> 5], bit ) & bit ) ) {
              active[v] = 1u;
              atomicOr( &nx[v >> 5], bit );
              mine = true;
            }
          }
        }
      }
    }
    if ( mine ) any = 1;
    __syncthreads();
    const bool more = any != 0;
    __syncthreads();
    if ( !more ) break;
    uint32_t* t = fr;
    fr          = nx;
    nx          = t;
    if ( threadIdx.x == 0 ) any = 0;
    __syncthreads();
  }
  for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) activeBits[w] = 0;
}

// points grouped by voxel (any order inside a voxel: re-scoring is per point, the histograms are integer sums)
__global__ __launch_bounds__( 256 ) void voxelPointListKernel( const uint32_t* __restrict__ vid, const uint32_t* __restrict__ start,
                                                                uint32_t n, uint32_t* __restrict__ cursor,
                                                                uint32_t* __restrict__ list ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j >= n ) return;
  const uint32_t v                            = vid[j];
  list[start[v] + atomicAdd( &cursor[v], 1u )] = j;
}

// decide + re-score in one pass over the VOXELS (16 lanes each): the voxels that are re-scored this sweep are few
// (patch borders), so walking their point lists beats a pass over every point, and a voxel's new histogram is built in
// registers and written once instead of through atomics.
//   proc[v] = active and (multi-plane edge, or its smoothed histogram is not unanimous for its own plane)
__global__ __launch_bounds__( 256 ) void rescoreVoxelsKernel( const uint8_t* __restrict__ edge, const uint8_t* __restrict__ ppi,
                                                               const uint32_t* __restrict__ active, const uint4* __restrict__ S,
                                                               const double* __restrict__ weight,
                                                               const uint32_t* __restrict__ pointStart,
                                                               const uint32_t* __restrict__ pointList,
                                                               const double* __restrict__ normals, uint32_t V,
                                                               uint8_t* __restrict__ proc, uint4* __restrict__ hist,
                                                               uint8_t* __restrict__ partition, uint32_t* __restrict__ flags,
                                                               int iter ) {
  const uint32_t v    = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  const int      sub  = threadIdx.x & 15;
  if ( v >= V || flags[0] ) return;
  uint32_t b[6];
  uint8_t  p = 0;
  if ( active[v] ) {
    const uint8_t edgeAt = edge[v] != NO_EDGE ? edge[v] : uint8_t( INDIRECT_EDGE );
    unpackHist( S[v], b );
    p = 1;
    if ( edgeAt != M_DIRECT_EDGE ) {
      int nz, a;
      classify( b, nz, a );
      if ( nz == 1 && b[ppi[v]] > 0 ) p = 0;
    }
  }
  if ( sub == 0 ) proc[v] = p;
  if ( !p ) return;  // (uniform over the 16 lanes of the voxel)
  const double   w     = weight[v];
  const uint32_t begin = pointStart[v], end = pointStart[v + 1];
  uint32_t       h0 = 0, h1 = 0, h2 = 0;  // packed u16 pairs like the histogram words
  uint32_t       moved = 0;
  for ( uint32_t q = begin + sub; q < end; q += 16 ) {
    const uint32_t j  = pointList[q];
    const double   nx = normals[3 * size_t( j )], ny = normals[3 * size_t( j ) + 1], nz = normals[3 * size_t( j ) + 2];
    const double d[6] = {nx * 1.0 + ny * 0.0 + nz * 0.0,  nx * 0.0 + ny * 1.0 + nz * 0.0,
                         nx * 0.0 + ny * 0.0 + nz * 1.0,  nx * -1.0 + ny * 0.0 + nz * 0.0,
                         nx * 0.0 + ny * -1.0 + nz * 0.0, nx * 0.0 + ny * 0.0 + nz * -1.0};
    int    best = 0;
    double bs   = d[0] + w * double( b[0] );
#pragma unroll
    for ( int k = 1; k < 6; ++k ) {
      const double sc = d[k] + w * double( b[k] );
      if ( sc > bs ) {
        bs   = sc;
        best = k;
      }
    }
    if ( partition[j] != uint8_t( best ) ) {
      partition[j] = uint8_t( best );
      ++moved;
    }
    const uint32_t one = 1u << ( 16 * ( best & 1 ) );
    h0 += ( best >> 1 ) == 0 ? one : 0u;
    h1 += ( best >> 1 ) == 1 ? one : 0u;
    h2 += ( best >> 1 ) == 2 ? one : 0u;
  }
#pragma unroll
  for ( int off = 8; off > 0; off >>= 1 ) {  // the 16 lanes of a voxel are an aligned quarter of the wave
    h0 += __shfl_xor( h0, off, 64 );
    h1 += __shfl_xor( h1, off, 64 );
    h2 += __shfl_xor( h2, off, 64 );
    moved += __shfl_xor( moved, off, 64 );
  }
  if ( sub == 0 ) {
    hist[v] = make_uint4( h0, h1, h2, 0 );
    if ( moved ) atomicAdd( &flags[2 * iter + 1], moved );  // this sweep moved points (the count feeds the trace hook)
  }
}


// ======================================================================================================================
// Event-driven sweeps (the default path).  What the sweeps above recompute for every voxel every sweep is maintained
// incrementally here, bit-exactly:
//   * S[v] (the smoothed histogram) only changes when the histogram of a voxel in v's neighbourhood changes.  A voxel
//     whose points moved PUSHES the difference to the voxels that list it (reverse neighbourhood rows, built once):
//     integer adds, so the order is irrelevant.  Late sweeps change a few dozen histograms, not 70 K rows of 88 gathers.
//     S is double-buffered: a sweep reads rec[cur] and pushes into rec[nxt], which the sweep's first kernel prepared as a
//     copy of rec[cur] -- nobody reads a record that is being pushed into.
//   * re-scoring a voxel is a pure function of S[v] (normals and weight are static): a voxel whose S has not changed
//     since it was last re-scored keeps its labels, so only its edge class / ppi are refreshed (epochs, below).
//   * the INDIRECT-edge closure (the one sequential coupling of a sweep): everything about it that is not sequential is
//     done chip-wide first (closurePrepareKernel: every voxel's list of DEV neighbours it would mark, the marks of the
//     voxels active at sweep start); the dependent rest runs in ONE workgroup with the active / frontier / marked
//     bitmaps in LDS (closureLevelsKernel): a level of the chain costs one 16-byte load per frontier voxel, one voxel per
//     thread, and the kernel hands the last one a compact work list (active + marked voxels).
// Three launches per sweep: closurePrepareKernel (chip-wide), closureLevelsKernel (1 workgroup), sweepKernel (chip-wide
// over the work list).
//
// rec[v] = { s0, s1, s2 : S[v] as packed u16 pairs (like hist), lastChange : id of the first S-state that holds the
// current value }.  S-state t + 1 is what sweep t reads.  lastRescore[v] = id of the S-state v's labels were computed
// from (0: never).  A voxel processed in sweep t is re-scored iff lastChange > lastRescore; pushes of sweep t stamp
// lastChange = t + 2 (in the copy the next sweep reads).
constexpr uint32_t kWorkActive = 0x80000000u;

__global__ __launch_bounds__( 256 ) void smoothInitKernel( const uint4* __restrict__ hist, const uint32_t* __restrict__ adjOff,
                                                            const uint32_t* __restrict__ rowLen,
                                                            const uint32_t* __restrict__ adj, uint32_t V,
                                                            uint4* __restrict__ rec ) {
  const int      lane = threadIdx.x & 15;
  const uint32_t v    = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  if ( v >= V ) return;
  const uint32_t* row = adj + adjOff[v];
  const uint32_t  len = rowLen[v];
  uint32_t        s0 = 0, s1 = 0, s2 = 0;
  for ( uint32_t i = lane; i < len; i += 16 ) {
    const uint4 h = hist[row[i]];
    s0 += h.x;
    s1 += h.y;
    s2 += h.z;
  }
#pragma unroll
  for ( int off = 8; off > 0; off >>= 1 ) {
    s0 += __shfl_xor( s0, off, 64 );
    s1 += __shfl_xor( s1, off, 64 );
    s2 += __shfl_xor( s2, off, 64 );
  }
  if ( lane == 0 ) rec[v] = make_uint4( s0, s1, s2, 1u );
}

// reverse neighbourhood rows: radj[roff[u] ..] = the voxels v whose (truncated) row lists u
__global__ __launch_bounds__( 256 ) void reverseCountKernel( const uint32_t* __restrict__ adjOff, const uint32_t* __restrict__ rowLen,
                                                              const uint32_t* __restrict__ adj, uint32_t V,
                                                              uint32_t* __restrict__ rcount ) {
  const int      lane = threadIdx.x & 15;
  const uint32_t v    = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  if ( v >= V ) return;
  const uint32_t* row = adj + adjOff[v];
  const uint32_t  len = rowLen[v];
  for ( uint32_t i = lane; i < len; i += 16 ) atomicAdd( &rcount[row[i]], 1u );
}

__global__ __launch_bounds__( 256 ) void reverseFillKernel( const uint32_t* __restrict__ adjOff, const uint32_t* __restrict__ rowLen,
                                                             const uint32_t* __restrict__ adj, uint32_t V,
                                                             const uint32_t* __restrict__ roff, uint32_t* __restrict__ cursor,
                                                             uint32_t* __restrict__ radj ) {
  const int      lane = threadIdx.x & 15;
  const uint32_t v    = blockIdx.x * 16 + ( threadIdx.x >> 4 );
  if ( v >= V ) return;
  const uint32_t* row = adj + adjOff[v];
  const uint32_t  len = rowLen[v];
  for ( uint32_t i = lane; i < len; i += 16 ) {
    const uint32_t u                          = row[i];
    radj[roff[u] + atomicAdd( &cursor[u], 1u )] = v;
  }
}

__device__ __forceinline__ int argOfPacked( uint32_t s0, uint32_t s1, uint32_t s2 ) {
  uint32_t b[6];
  unpackHist( make_uint4( s0, s1, s2, 0 ), b );
  int nz, a;
  classify( b, nz, a );
  return a;
}

// Chip-wide first step of a sweep, 32 lanes per voxel u (its DEV row, padded to `stride` entries: 32 or 128):
//   out[u] = { count, the DEV neighbours u marks if it is active: NO_EDGE voxels whose ppi differs from arg(S[u]) };
//   rec[nxt][u] = rec[cur][u]  (the copy this sweep's pushes go into);
//   u active at sweep start: its bit in the active bitmap, its marks, and the larger-index voxels it activates (frontier).
__device__ __forceinline__ void prepareVoxel( uint32_t u, uint32_t lane, int half, const uint8_t* __restrict__ edge,
                                              const uint8_t* __restrict__ ppi, const uint4* __restrict__ recCur,
                                              uint4* __restrict__ recNxt, const uint32_t* __restrict__ dev,
                                              const uint32_t* __restrict__ devLen, uint32_t stride, uint32_t* __restrict__ out,
                                              uint32_t* __restrict__ gAct, uint32_t* __restrict__ gFr, uint32_t* __restrict__ gMk ) {
  const uint4    r      = recCur[u];
  const uint8_t  a      = uint8_t( argOfPacked( r.x, r.y, r.z ) );
  const bool     active = edge[u] != NO_EDGE;
  const uint32_t len    = devLen[u];
  uint32_t       nOut   = 0;
  if ( lane == 0 ) {
    recNxt[u] = r;
    if ( active ) atomicOr( &gAct[u >> 5], 1u << ( u & 31 ) );
  }
  for ( uint32_t base = 0; base < len; base += 32 ) {  // (uniform over the 32 lanes of the voxel)
    const uint32_t v    = base + lane < len ? dev[size_t( u ) * stride + base + lane] : kDevPad;
    const bool     pred = v != kDevPad && edge[v] == NO_EDGE && ppi[v] != a;
    const uint32_t m    = uint32_t( __ballot( pred ) >> ( 32 * half ) );
    if ( pred ) {
      out[size_t( u ) * stride + 1 + nOut + __popc( m & ( ( 1u << lane ) - 1u ) )] = v;
      if ( active ) {
        atomicOr( &gMk[v >> 5], 1u << ( v & 31 ) );
        if ( v > u ) {  // (v is a NO_EDGE voxel: only marks activate it, so "activated" and "frontier" coincide here)
          atomicOr( &gAct[v >> 5], 1u << ( v & 31 ) );
          atomicOr( &gFr[v >> 5], 1u << ( v & 31 ) );
        }
      }
    }
    nOut += uint32_t( __popc( m ) );
  }
  if ( lane == 0 ) out[size_t( u ) * stride] = nOut;
}
__global__ __launch_bounds__( 256 ) void closurePrepareKernel( const uint8_t* __restrict__ edge, const uint8_t* __restrict__ ppi,
                                                                const uint4* __restrict__ recCur, uint4* __restrict__ recNxt,
                                                                const uint32_t* __restrict__ dev,
                                                                const uint32_t* __restrict__ devLen, uint32_t stride, uint32_t V,
                                                                uint32_t* __restrict__ out, uint32_t* __restrict__ gAct,
                                                                uint32_t* __restrict__ gFr, uint32_t* __restrict__ gMk ) {
  // (a capped grid with a stride loop: 8 860 workgroups of eight voxels each made workgroup dispatch the cost of this kernel
  // when sixteen frames issue it at the same time)
  for ( uint32_t u = blockIdx.x * 8 + ( threadIdx.x >> 5 ); u < V; u += gridDim.x * 8 )
    prepareVoxel( u, threadIdx.x & 31, ( threadIdx.x >> 5 ) & 1, edge, ppi, recCur, recNxt, dev, devLen, stride, out, gAct, gFr, gMk );
}

// Wave-level compaction of a bitmap into a voxel list (whole 64-word chunks; a chunk that does not fit stays for the next
// round and records where the complete part of the list ends: offsets are handed out in order, so everything below the
// first failure is complete).  The caller zeroes *count, sets *valid = ~0 and synchronises before; returns the length.
__device__ __forceinline__ uint32_t compactBitmap( uint32_t* __restrict__ bm, uint32_t W, uint32_t* __restrict__ list,
                                                   uint32_t cap, uint32_t* count, uint32_t* valid ) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  for ( uint32_t chunk = wave; chunk * 64 < W; chunk += waves ) {
    const uint32_t w    = chunk * 64 + lane;
    uint32_t       bits = w < W ? bm[w] : 0u;
    if ( !__ballot( bits != 0 ) ) continue;  // (most chunks of a late level are empty)
    const uint32_t cnt = uint32_t( __popc( bits ) );
    uint32_t       inc = cnt;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 );
      if ( lane >= off ) inc += t;
    }
    const uint32_t total = __shfl( inc, 63, 64 );
    uint32_t       base  = 0;
    if ( lane == 0 ) base = atomicAdd( count, total );
    base = __shfl( base, 0, 64 );
    if ( base + total <= cap ) {
      if ( bits ) {
        bm[w]        = 0;
        uint32_t off = base + inc - cnt;
        while ( bits ) {
          list[off++] = 32 * w + uint32_t( __ffs( int( bits ) ) - 1 );
          bits &= bits - 1;
        }
      }
    } else if ( lane == 0 ) {
      atomicMin( valid, base );
    }
  }
  __syncthreads();
  return min( *count, *valid );
}

// The dependent rest of the closure, one workgroup.  LDS: active | frontier | marked | scratch bitmaps (V bits each), list.
// Hands the global bitmaps back empty, writes the work list of sweepKernel: every active voxel, then the voxels that were
// only marked.
__device__ __forceinline__ void walkLevels( const uint32_t* __restrict__ out, uint32_t stride, uint32_t V, uint32_t listCap,
                                            uint32_t* __restrict__ gAct, uint32_t* __restrict__ gFr, uint32_t* __restrict__ gMk,
                                            uint32_t* __restrict__ work, uint32_t* __restrict__ workCount,
                                            unsigned long long* __restrict__ timing ) {
  extern __shared__ uint32_t lds[];
  // (test hook: timing[0..3] += ticks of load / compaction / voxel processing / list emission, [4] += rounds, [5] += voxels)
  unsigned long long tick = timing ? wall_clock64() : 0ull, tLoad = 0, tCompact = 0, tProcess = 0, tEmit = 0, rounds = 0;
#define TMC2_LAP( acc )                              \
  if ( timing && threadIdx.x == 0 ) {                \
    const unsigned long long now_ = wall_clock64();  \
    acc += now_ - tick;                              \
    tick = now_;                                     \
  }
  __shared__ uint32_t nList, nValid;
  const uint32_t      W   = ( V + 31 ) / 32;
  uint32_t *          act = lds, *fr = act + W, *mk = fr + W, *tmp = mk + W, *list = tmp + W;
  for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) {
    const uint32_t a = gAct[w], f = gFr[w], m = gMk[w];
    act[w] = a, fr[w] = f, mk[w] = m, tmp[w] = a & ~f;
    if ( a ) gAct[w] = 0;
    if ( f ) gFr[w] = 0;
    if ( m ) gMk[w] = 0;
  }
  uint32_t nWork = 0;  // (uniform: every thread keeps the same count)
  // phase 0: list the voxels active at sweep start (their marks are made already);  phase 1: the closure, level by level,
  // listing the voxels of each level (the first one: what the voxels of phase 0 activated);  phase 2: the voxels that
  // were only marked.  Every voxel is listed once, whatever the number of rounds a full list splits a level into.
  for ( int phase = 0; phase < 3; ++phase ) {
    bool first = true;
    while ( true ) {
      if ( threadIdx.x == 0 ) {
        nList  = 0;
        nValid = 0xFFFFFFFFu;
      }
      __syncthreads();
      if ( phase == 0 && first ) { TMC2_LAP( tLoad ) }
      const uint32_t n     = compactBitmap( phase == 0 ? tmp : ( phase == 1 ? fr : mk ), W, list, listCap, &nList, &nValid );
      const bool     empty = nList == 0;
      __syncthreads();  // (nList is reset at the top of the next round)
      if ( phase == 1 ) { TMC2_LAP( tCompact ) } else { TMC2_LAP( tEmit ) }
      if ( empty ) break;
      for ( uint32_t i = threadIdx.x; i < n; i += blockDim.x ) work[nWork + i] = list[i] | ( phase == 2 ? 0u : kWorkActive );
      nWork += n;
      first = false;
      if ( phase != 1 ) continue;
      ++rounds;
      for ( uint32_t i = threadIdx.x; i < n; i += blockDim.x ) {  // one voxel per thread: one 16-byte load per hop
        const uint32_t  u    = list[i];
        const uint32_t* row  = out + size_t( u ) * stride;
        const uint4 head = reinterpret_cast<const uint4*>( row )[0];  // count + the first seven targets: one round
        const uint4 more = reinterpret_cast<const uint4*>( row )[1];  // trip for all but the rarest voxels
        for ( uint32_t k = 0; k < head.x; ++k ) {
          const uint32_t v   = k == 0   ? head.y
                               : k == 1 ? head.z
                               : k == 2 ? head.w
                               : k == 3 ? more.x
                               : k == 4 ? more.y
                               : k == 5 ? more.z
                               : k == 6 ? more.w
                                        : row[1 + k];
          const uint32_t bit = 1u << ( v & 31 );
          atomicOr( &mk[v >> 5], bit );
          if ( v > u && !( atomicOr( &act[v >> 5], bit ) & bit ) ) atomicOr( &fr[v >> 5], bit );
        }
      }
      __syncthreads();
      TMC2_LAP( tProcess )
    }
    if ( phase == 1 ) {  // what is left to list: marked, but never active
      for ( uint32_t w = threadIdx.x; w < W; w += blockDim.x ) mk[w] &= ~act[w];
    }
  }
  if ( threadIdx.x == 0 ) *workCount = nWork;
  TMC2_LAP( tEmit )
  if ( timing && threadIdx.x == 0 ) {
    timing[0] += tLoad, timing[1] += tCompact, timing[2] += tProcess, timing[3] += tEmit, timing[4] += rounds, timing[5] += nWork;
  }
#undef TMC2_LAP
}
__global__ __launch_bounds__( 1024 ) void closureLevelsKernel( const uint32_t* __restrict__ out, uint32_t stride, uint32_t V,
                                                                uint32_t listCap, uint32_t* __restrict__ gAct,
                                                                uint32_t* __restrict__ gFr, uint32_t* __restrict__ gMk,
                                                                uint32_t* __restrict__ work, uint32_t* __restrict__ workCount,
                                                                unsigned long long* __restrict__ timing ) {
  walkLevels( out, stride, V, listCap, gAct, gFr, gMk, work, workCount, timing );
}

// The rest of a sweep for the voxels of the work list, 16 lanes each: decide, re-score if S changed since the labels were
// computed, push the histogram difference to the reverse row, refresh edge class / ppi.
__global__ __launch_bounds__( 256 ) void sweepKernel( const uint32_t* __restrict__ work, const uint32_t* __restrict__ workCount,
                                                       const uint4* __restrict__ recCur, uint4* __restrict__ recNxt,
                                                       uint32_t* __restrict__ lastRescore, const double* __restrict__ weight,
                                                       const uint32_t* __restrict__ pointStart,
                                                       const uint32_t* __restrict__ pointList,
                                                       const double* __restrict__ normals, const uint32_t* __restrict__ roff,
                                                       const uint32_t* __restrict__ radj, uint8_t* __restrict__ edge,
                                                       uint8_t* __restrict__ ppi, uint4* __restrict__ hist,
                                                       uint8_t* __restrict__ partition, uint32_t* __restrict__ flags, int iter ) {
  const int      sub    = threadIdx.x & 15;
  const uint32_t groups = gridDim.x * 16, count = *workCount;
  for ( uint32_t idx = blockIdx.x * 16 + ( threadIdx.x >> 4 ); idx < count; idx += groups ) {  // (uniform over the 16 lanes)
    const uint32_t entry = work[idx], v = entry & ~kWorkActive;
    const uint8_t  e0 = edge[v];
    bool           p  = false;
    uint32_t       b[6];
    uint4          s  = make_uint4( 0, 0, 0, 0 );
    if ( entry & kWorkActive ) {
      s = recCur[v];
      unpackHist( s, b );
      p = true;
      if ( ( e0 != NO_EDGE ? e0 : uint8_t( INDIRECT_EDGE ) ) != M_DIRECT_EDGE ) {
        int nz, a;
        classify( b, nz, a );
        if ( nz == 1 && b[ppi[v]] > 0 ) p = false;
      }
    }
    if ( !p ) {  // marked (every voxel of the list that was NO_EDGE at sweep start is) and not re-scored: INDIRECT from now on
      if ( sub == 0 && e0 == NO_EDGE ) edge[v] = INDIRECT_EDGE;
      continue;
    }
    uint4 h = hist[v];
    if ( s.w > lastRescore[v] ) {
      const double   w     = weight[v];
      const uint32_t begin = pointStart[v], end = pointStart[v + 1];
      uint32_t       h0 = 0, h1 = 0, h2 = 0, moved = 0;
      for ( uint32_t q = begin + sub; q < end; q += 16 ) {
        const uint32_t j  = pointList[q];
        const double   nx = normals[3 * size_t( j )], ny = normals[3 * size_t( j ) + 1], nz = normals[3 * size_t( j ) + 2];
        const double d[6] = {nx * 1.0 + ny * 0.0 + nz * 0.0,  nx * 0.0 + ny * 1.0 + nz * 0.0,
                             nx * 0.0 + ny * 0.0 + nz * 1.0,  nx * -1.0 + ny * 0.0 + nz * 0.0,
                             nx * 0.0 + ny * -1.0 + nz * 0.0, nx * 0.0 + ny * 0.0 + nz * -1.0};
        int    best = 0;
        double bs   = d[0] + w * double( b[0] );
#pragma unroll
        for ( int k = 1; k < 6; ++k ) {
          const double sc = d[k] + w * double( b[k] );
          if ( sc > bs ) {
            bs   = sc;
            best = k;
          }
        }
        if ( partition[j] != uint8_t( best ) ) {
          partition[j] = uint8_t( best );
          ++moved;
        }
        const uint32_t one = 1u << ( 16 * ( best & 1 ) );
        h0 += ( best >> 1 ) == 0 ? one : 0u;
        h1 += ( best >> 1 ) == 1 ? one : 0u;
        h2 += ( best >> 1 ) == 2 ? one : 0u;
      }
#pragma unroll
      for ( int off = 8; off > 0; off >>= 1 ) {
        h0 += __shfl_xor( h0, off, 64 );
        h1 += __shfl_xor( h1, off, 64 );
        h2 += __shfl_xor( h2, off, 64 );
        moved += __shfl_xor( moved, off, 64 );
      }
      if ( sub == 0 ) lastRescore[v] = uint32_t( iter ) + 1u;
      if ( moved ) {  // (a changed histogram needs a moved point)
        // packed u16 pairs: the difference is added modulo 2^32 per word; every true field stays within 0 .. 65535, so
        // borrows between the halves cancel in the final sums whatever the order of the adds
        const uint32_t d0 = h0 - h.x, d1 = h1 - h.y, d2 = h2 - h.z;
        if ( d0 | d1 | d2 ) {
          const uint32_t rb = roff[v], re = roff[v + 1];
          for ( uint32_t t = rb + sub; t < re; t += 16 ) {
            uint32_t* tr = reinterpret_cast<uint32_t*>( recNxt + radj[t] );
            if ( d0 ) atomicAdd( tr, d0 );
            if ( d1 ) atomicAdd( tr + 1, d1 );
            if ( d2 ) atomicAdd( tr + 2, d2 );
            tr[3] = uint32_t( iter ) + 2u;
          }
        }
        h = make_uint4( h0, h1, h2, 0 );
        if ( sub == 0 ) {
          hist[v] = h;
          if ( flags ) atomicAdd( &flags[2 * iter + 1], moved );  // (only for the trace hook: thousands of atomics on one word otherwise)
        }
      }
    }
    if ( sub == 0 ) {  // the voxel was processed: edge class and ppi follow its (new) histogram
      unpackHist( h, b );
      int nz, a;
      classify( b, nz, a );
      if ( e0 != S_DIRECT_EDGE ) edge[v] = ( nz == 1 ) ? uint8_t( NO_EDGE ) : uint8_t( M_DIRECT_EDGE );
      ppi[v] = uint8_t( a );
    }
  }
}

}  // namespace

// refineSegmentationGridBased in two halves.  geometry(): everything that depends on the points alone -- voxels, points grouped
// by voxel, neighbourhood rows -- queued without waiting for the last result; a frame's host thread runs it BEFORE the
// sequential orientation walk (S3), so the device builds the rows while the host walks.  finish(): the rest (histograms of
// the initial partition, sweeps).  The dense voxel table stays filled in between (no other stage of the frame's context
// uses it); a job dropped half-way empties it again.
struct RefineJob {
  // parameters
  int    maxNNCount = 0, iterationCount = 0, voxDim = 0, searchRadius = 0;
  double lambda = 0.0;
  // state between the halves
  tmc2_frame*      frame = nullptr;
  tmc2_ctx*        ctx   = nullptr;
  hipStream_t      s     = nullptr;
  uint32_t         n = 0, V = 0, W = 0, devStride = 32, totalLen = 0;
  int              devRange = 1, idBits = 26;
  Grid             g{};
  std::vector<int> offsets;
  uint32_t*        table = nullptr;
  bool             tableFilled = false, eventDriven = false;
  size_t           Vp = 0, ldsRoom = 0, ldsFixed = 0, ball = 0, perVoxel = 0;
  uint64_t         capacity = 0;
  uint32_t         res[2] = {0, 0};  // the neighbourhood pass' answer: row entries written, overflow flag
  DevBuf<uint32_t> d_key, d_flag, d_vid, d_small, d_count, d_rowLen, d_devLen, d_adjOff, d_hist, d_activeBuf, d_pointStart,
      d_pointList, d_cursor, d_rcount, d_rcursor, d_lastRescore, d_flags, d_gbits, d_adj, d_dev;
  DevBuf<Pt>      d_centre;
  DevBuf<double>  d_weight;
  DevBuf<uint8_t> d_state;  // edge | ppi | arg | marked | proc, V bytes each
  DevBuf<int>     d_offsets;
  DevBuf<uint4>   d_S;
  bool matches( int nn, double l, int it, int vd, int sr ) const {
    return nn == maxNNCount && l == lambda && it == iterationCount && vd == voxDim && sr == searchRadius;
  }
  int  geometry( tmc2_frame* f );
  int  finish();
  void launchNeighbourhood();
  ~RefineJob();
};

RefineJob::~RefineJob() {
  if ( tableFilled && table && d_key.p ) {  // dropped between the halves: hand the context's table back empty
    ApiScope scope( ctx );
    hipLaunchKernelGGL( tableCleanKernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, s, d_key.p, n, table );
    (void)hipStreamSynchronize( s );  // (the buffers go back to the pool when the members are destroyed)
  }
}

void RefineJob::launchNeighbourhood() {
#define TMC2_NEIGHBOURHOOD( CAP, WAVES )                                                                                       \
  hipLaunchKernelGGL( ( neighbourhoodKernel<CAP, WAVES> ), dim3( ( V + WAVES - 1 ) / WAVES ), dim3( 64 * WAVES ), 0, s, d_centre.p,      \
                      d_count.p, table, g, V, d_offsets.p, int( ball ), maxNNCount, lambda, idBits, devRange, devStride,       \
                      uint32_t( capacity ), d_rowLen.p, d_devLen.p, d_weight.p, d_adjOff.p, d_adj.p, d_dev.p, d_small.p + 1,   \
                      d_small.p + 2 )
  if ( ball <= 2048 - 160 )
    TMC2_NEIGHBOURHOOD( 2048, 4 );
  else
    TMC2_NEIGHBOURHOOD( 4096, 2 );
#undef TMC2_NEIGHBOURHOOD
}

int RefineJob::geometry( tmc2_frame* f ) {
  if ( voxDim < 2 || ( voxDim & ( voxDim - 1 ) ) ) {  // (the CTC sequences use 4 -- longdress, basketball -- and 2 -- loot, redandblack, soldier)
    setError( "refineSegmentationGridBased: voxelDimensionRefineSegmentation=%d unsupported (power of two >= 2)", voxDim );
    return TMC2_E_UNSUPPORTED;
  }
  if ( iterationCount < 1 ) iterationCount = 1;  // the reference loop is do { } while ( ++iter < count )
  frame = f;
  ctx   = f->ctx;
  s     = ctx->stream;
  n     = uint32_t( f->n );
  // grid geometry (PCCPatchSegmenter.cpp:1397-1413)
  size_t geoRange = 1;
  for ( size_t i = size_t( f->geoMax - 1 ); i != 0; i >>= 1, geoRange <<= 1 ) {}
  g.voxShift = 0;
  for ( int i = voxDim; i > 1; ++g.voxShift, i >>= 1 ) {}
  const size_t gridDim = geoRange >> g.voxShift;
  g.gridShift          = 0;
  for ( size_t i = gridDim; i > 1; ++g.gridShift, i >>= 1 ) {}
  g.half      = voxDim >> 1;
  g.tableSize = 1u << ( 3 * g.gridShift + 1 );
  if ( g.gridShift > 9 ) {
    setError( "refineSegmentationGridBased: grid of 2^%d cells per axis unsupported", g.gridShift );
    return TMC2_E_UNSUPPORTED;
  }
  const int r2 = searchRadius >> g.voxShift;
  {
    int R = 0;
    while ( R * R < r2 ) ++R;
    for ( int dz = -R; dz <= R; ++dz )
      for ( int dy = -R; dy <= R; ++dy )
        for ( int dx = -R; dx <= R; ++dx )
          if ( dx * dx + dy * dy + dz * dz < r2 )
            offsets.push_back( ( dx + 128 ) | ( ( dy + 128 ) << 8 ) | ( ( dz + 128 ) << 16 ) );
  }
  if ( offsets.size() > 4096 - 160 || r2 > 128 ) {  // (the neighbourhood kernel keeps 160 words of its 2048 / 4096 for its own use)  // (search radius 192: r2 = 48 with voxels of 4, 96 with voxels of 2 -> 3 911 cells)
    setError( "refineSegmentationGridBased: search radius %d too large for the LDS neighbourhood tile", searchRadius );
    return TMC2_E_UNSUPPORTED;
  }
  devRange  = voxDim >= 4 ? 1 : 2;  // PCCPatchSegmenter.cpp:1471
  devStride = devRange == 1 ? 32u : 128u;
  idBits    = r2 <= 64 ? 26 : 25;   // neighbourhood sort key: d2 above, voxel id below
  const int sidSetup = ctx->stageBegin( "refine_setup" );
  if ( ctx->gridTable.count < g.tableSize ) {
    TMC2_TRY( ctx->gridTable.alloc( g.tableSize ) );
    TMC2_HIP( hipMemsetAsync( ctx->gridTable.p, 0xFF, size_t( g.tableSize ) * 4, s ) );
  }
  table = ctx->gridTable.p;
  TMC2_TRY( d_key.alloc( n ) );
  TMC2_TRY( d_flag.alloc( n ) );
  TMC2_TRY( d_vid.alloc( n ) );
  TMC2_TRY( d_small.alloc( 16 ) );  // [0] voxel count, [1] adjacency size, [2] closure flag
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  tableFilled = true;
  hipLaunchKernelGGL( voxelKeyKernel, grdN, blk, 0, s, f->d_pts.p, n, g, d_key.p, table );
  hipLaunchKernelGGL( firstFlagKernel, grdN, blk, 0, s, d_key.p, table, n, d_flag.p );
  DevBuf<uint32_t> d_rank;
  TMC2_TRY( d_rank.alloc( n ) );
  TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_rank.p, n, d_small.p ) );
  TMC2_HIP( hipMemcpyAsync( &V, d_small.p, 4, hipMemcpyDeviceToHost, s ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  TMC2_TRY( d_count.alloc( size_t( V ) + 1 ) );  // (+1: scanned into the point-list offsets)
  TMC2_TRY( d_rowLen.alloc( V ) );
  TMC2_TRY( d_devLen.alloc( V ) );
  TMC2_TRY( d_adjOff.alloc( V + 1 ) );
  TMC2_TRY( d_hist.alloc( size_t( V ) * 4 ) );
  TMC2_TRY( d_centre.alloc( V ) );
  TMC2_TRY( d_weight.alloc( V ) );
  Vp = ( size_t( V ) + 63 ) & ~size_t( 63 );  // sub-arrays of the state block: whole, aligned 32-voxel words
  TMC2_TRY( d_state.alloc( Vp * 6 ) );
  TMC2_TRY( d_activeBuf.alloc( V ) );
  TMC2_TRY( d_offsets.alloc( offsets.size() ) );
  TMC2_TRY( d_S.alloc( V ) );
  // every buffer of this stage that starts from zeros, in one launch (the event-driven loop's among them)
  W = ( V + 31 ) / 32;
  // Which sweep loop: the event-driven one needs the closure's four bitmaps and a voxel list in the LDS of one workgroup
  // (160 KB: up to ~ 290 K voxels; a vox11 frame has ~ 240 K).  Larger grids take the sweep-everything loop.
  // (test hook TMC2_REFINE_SWEEPS=full forces that one)
  const char* sweepsEnv = getenv( "TMC2_REFINE_SWEEPS" );
  ldsRoom     = 160 * 1024 - 64;  // gfx950: 160 KB of LDS per workgroup (opt-in above 64 KB); static part: 8 bytes
  ldsFixed    = 16 * size_t( W );
  eventDriven = !( sweepsEnv && sweepsEnv[0] == 'f' ) && ldsFixed + 4 * 2048 <= ldsRoom;
  TMC2_TRY( d_pointStart.alloc( size_t( V ) + 1 ) );
  TMC2_TRY( d_pointList.alloc( n ) );
  TMC2_TRY( d_cursor.alloc( V ) );
  TMC2_TRY( d_flags.alloc( 2 * size_t( iterationCount ) + 2 ) );
  if ( eventDriven ) {
    TMC2_TRY( d_rcount.alloc( size_t( V ) + 1 ) );
    TMC2_TRY( d_rcursor.alloc( size_t( V ) + 1 ) );
    TMC2_TRY( d_lastRescore.alloc( V ) );
    TMC2_TRY( d_gbits.alloc( 3 * size_t( W ) ) );
  }
  TMC2_TRY( fillRegions( ctx, {{d_count.p, ( size_t( V ) + 1 ) * 4, 0},
                               {d_hist.p, size_t( V ) * 16, 0},
                               {d_state.p, Vp * 6, 0},
                               {d_cursor.p, size_t( V ) * 4, 0},
                               {d_small.p + 1, 8, 0},  // [1] row cursor, [2] overflow
                               {d_flags.p, ( 2 * size_t( iterationCount ) + 2 ) * 4, 0},
                               {d_rcount.p, eventDriven ? ( size_t( V ) + 1 ) * 4 : 0, 0},
                               {d_rcursor.p, eventDriven ? ( size_t( V ) + 1 ) * 4 : 0, 0},
                               {d_lastRescore.p, eventDriven ? size_t( V ) * 4 : 0, 0},
                               {d_gbits.p, eventDriven ? 3 * size_t( W ) * 4 : 0, 0}} ) );
  TMC2_HIP( hipMemcpyAsync( d_offsets.p, offsets.data(), offsets.size() * sizeof( int ), hipMemcpyHostToDevice, s ) );
  hipLaunchKernelGGL( assignVoxelKernel, grdN, blk, 0, s, f->d_pts.p, d_key.p, table, d_rank.p, n, g, d_vid.p,
                      d_count.p, d_centre.p );
  hipLaunchKernelGGL( tableToVoxelKernel, grdN, blk, 0, s, d_key.p, d_flag.p, d_vid.p, n, table );
  // points grouped by voxel, for the re-scoring pass
  TMC2_TRY( exclusiveScanU32( ctx, d_count.p, d_pointStart.p, size_t( V ) + 1, nullptr ) );
  hipLaunchKernelGGL( voxelPointListKernel, grdN, blk, 0, s, d_vid.p, d_pointStart.p, n, d_cursor.p, d_pointList.p );
  // neighbourhoods.  Rows back to back; room for twice the expected mean row (maxNN / points per voxel) -- the rare frame
  // that needs more repeats the pass with room for whole balls.  (test hook TMC2_REFINE_ROWCAP=tiny forces the repeat)
  if ( ( uint64_t( V ) >> idBits ) != 0 ) {
    setError( "refineSegmentationGridBased: %u voxels exceed the neighbourhood sort key", V );
    return TMC2_E_UNSUPPORTED;
  }
  TMC2_TRY( d_dev.alloc( size_t( V ) * devStride ) );
  ball                = offsets.size();
  const char* capEnv  = getenv( "TMC2_REFINE_ROWCAP" );
  perVoxel            = std::min<size_t>( ball, 2 * size_t( maxNNCount > 0 ? maxNNCount : 1 ) * V / std::max<uint32_t>( n, 1u ) + 32 );
  if ( capEnv && capEnv[0] == 't' ) perVoxel = 1;
  capacity = uint64_t( V ) * perVoxel;
  if ( capacity > 0xFFFFFFFFull ) {
    setError( "refineSegmentationGridBased: %u voxels x %zu row entries exceed the neighbourhood table", V, perVoxel );
    return TMC2_E_UNSUPPORTED;
  }
  TMC2_TRY( d_adj.alloc( size_t( capacity ) ) );
  launchNeighbourhood();
  TMC2_HIP( hipMemcpyAsync( res, d_small.p + 1, 8, hipMemcpyDeviceToHost, s ) );  // (read in finish(), after a synchronisation)
  ctx->stageEnd( sidSetup );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

int RefineJob::finish() {
  tmc2_frame* f = frame;
  if ( !f->haveNormals || !f->havePartition ) {
    setError( "refineSegmentationGridBased: normals / partition missing" );
    return TMC2_E_STATE;
  }
  const int  sidSetup = ctx->stageBegin( "refine_setup" );
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  uint8_t *d_edge = d_state.p, *d_ppi = d_state.p + Vp, *d_arg = d_state.p + 2 * Vp, *d_marked = d_state.p + 4 * Vp,
          *d_proc = d_state.p + 5 * Vp;
  uint32_t* d_active = d_activeBuf.p;
  const dim3 grdV( ( V + 255 ) / 256 ), grdV16( ( V + 15 ) / 16 );  // 16 lanes per voxel
  for ( int attempt = 0;; ++attempt ) {
    TMC2_HIP( hipStreamSynchronize( s ) );  // (usually long done: the orientation walk ran in between)
    TMC2_HIP( hipGetLastError() );
    totalLen = res[0];
    if ( res[1] == 0 ) break;
    if ( res[1] != 1 || attempt > 0 ) {
      setError( "refineSegmentationGridBased: neighbourhood pass failed (%u)", res[1] );
      return TMC2_E_HIP;
    }
    perVoxel = ball;  // room for whole balls
    capacity = uint64_t( V ) * perVoxel;
    if ( capacity > 0xFFFFFFFFull ) {
      setError( "refineSegmentationGridBased: %u voxels x %zu row entries exceed the neighbourhood table", V, perVoxel );
      return TMC2_E_UNSUPPORTED;
    }
    TMC2_TRY( d_adj.alloc( size_t( capacity ) ) );
    TMC2_HIP( hipMemsetAsync( d_small.p + 1, 0, 8, s ) );  // [1] row cursor, [2] overflow
    launchNeighbourhood();
    TMC2_HIP( hipMemcpyAsync( res, d_small.p + 1, 8, hipMemcpyDeviceToHost, s ) );
  }
  hipLaunchKernelGGL( tableCleanKernel, grdN, blk, 0, s, d_key.p, n, table );
  tableFilled = false;
  hipLaunchKernelGGL( histAccumulateKernel, grdN, blk, 0, s, d_vid.p, f->d_partition.p, (const uint8_t*)nullptr, n,
                      d_hist.p );
  hipLaunchKernelGGL( initVoxelStateKernel, grdV, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_count.p, V,
                      d_edge, d_ppi, d_active );
  const char* prepEnv = getenv( "TMC2_REFINE_PREPARE_BLOCKS" );  // (test hook: the grid of closurePrepareKernel)
  const dim3  grdV32( std::min<uint32_t>( ( V + 7 ) / 8, prepEnv ? uint32_t( std::max( 1, atoi( prepEnv ) ) ) : cappedBlocks( ctx, ( V + 7 ) / 8 ) ) );
  if ( eventDriven ) {
    // reverse rows (CSR), S records (double-buffered), epochs, closure scratch
    DevBuf<uint32_t> d_roff, d_radj, d_work, d_out;
    DevBuf<uint4>    d_rec;
    TMC2_TRY( d_roff.alloc( size_t( V ) + 1 ) );
    TMC2_TRY( d_radj.alloc( std::max<size_t>( totalLen, 1 ) ) );
    TMC2_TRY( d_work.alloc( size_t( V ) + 1 ) );  // [V]: the list's length
    TMC2_TRY( d_rec.alloc( 2 * size_t( V ) ) );
    TMC2_TRY( d_out.alloc( size_t( V ) * devStride ) );
    hipLaunchKernelGGL( reverseCountKernel, grdV16, blk, 0, s, d_adjOff.p, d_rowLen.p, d_adj.p, V, d_rcount.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_rcount.p, d_roff.p, size_t( V ) + 1, nullptr ) );
    hipLaunchKernelGGL( reverseFillKernel, grdV16, blk, 0, s, d_adjOff.p, d_rowLen.p, d_adj.p, V, d_roff.p, d_rcursor.p,
                        d_radj.p );
    hipLaunchKernelGGL( smoothInitKernel, grdV16, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_adjOff.p,
                        d_rowLen.p, d_adj.p, V, d_rec.p );
    ctx->stageEnd( sidSetup );
    TMC2_HIP( hipGetLastError() );
    const int      sidSweep = ctx->stageBegin( "refine_sweeps" );
    // (test hook TMC2_REFINE_LISTCAP: a short voxel list, so that small frames split their levels over several rounds too)
    const char*    listEnv  = getenv( "TMC2_REFINE_LISTCAP" );
    const uint32_t listCap  = uint32_t( std::min<size_t>( listEnv ? std::max( 2048, atoi( listEnv ) ) : 8192, ( ldsRoom - ldsFixed ) / 4 ) );
    const size_t   ldsBytes = ldsFixed + 4 * size_t( listCap );
    if ( ldsBytes > 48 * 1024 ) TMC2_TRY( allowLargeLds( reinterpret_cast<const void*>( closureLevelsKernel ), ldsBytes, ctx->device ) );
    const dim3 grdSweep( uint32_t( std::min<size_t>( ( size_t( V ) + 15 ) / 16, size_t( 2 ) * ctx->cuCount ) ) );
    DevBuf<unsigned long long> d_timing;  // test hook TMC2_REFINE_TIMING: where the closure's tail spends its time
    const bool                 wantTiming = getenv( "TMC2_REFINE_TIMING" ) != nullptr;
    const bool                 wantTrace  = getenv( "TMC2_REFINE_TRACE" ) != nullptr;
    if ( wantTiming ) {
      TMC2_TRY( d_timing.alloc( 8 ) );
      TMC2_HIP( hipMemsetAsync( d_timing.p, 0, 64, s ) );
    }
    uint32_t *gAct = d_gbits.p, *gFr = d_gbits.p + W, *gMk = d_gbits.p + 2 * size_t( W );
    // (test hook TMC2_REFINE_WALK_THREADS: the size of the level walk's one workgroup)
    const char* walkEnv     = getenv( "TMC2_REFINE_WALK_THREADS" );
    const int   walkThreads = walkEnv ? std::min( 1024, std::max( 64, atoi( walkEnv ) & ~63 ) ) : 1024;
    for ( int iter = 0; iter < iterationCount; ++iter ) {
      uint4 *recCur = d_rec.p + size_t( iter & 1 ) * V, *recNxt = d_rec.p + size_t( ( iter + 1 ) & 1 ) * V;
      hipLaunchKernelGGL( closurePrepareKernel, grdV32, blk, 0, s, d_edge, d_ppi, recCur, recNxt, d_dev.p, d_devLen.p, devStride,
                          V, d_out.p, gAct, gFr, gMk );
      hipLaunchKernelGGL( closureLevelsKernel, dim3( 1 ), dim3( walkThreads ), ldsBytes, s, d_out.p, devStride, V, listCap, gAct, gFr,
                          gMk, d_work.p, d_work.p + V, wantTiming ? d_timing.p : nullptr );
      hipLaunchKernelGGL( sweepKernel, grdSweep, blk, 0, s, d_work.p, d_work.p + V, recCur, recNxt, d_lastRescore.p,
                          d_weight.p, d_pointStart.p, d_pointList.p, f->d_normals.p, d_roff.p, d_radj.p, d_edge, d_ppi,
                          reinterpret_cast<uint4*>( d_hist.p ), f->d_partition.p, wantTrace ? d_flags.p : nullptr, iter );
    }
    ctx->stageEnd( sidSweep );
    TMC2_HIP( hipGetLastError() );
    ctx->stageAddHostMs( "refine_sweeps_executed", double( iterationCount ) );  // (counts, not milliseconds: what the
    ctx->stageAddHostMs( "refine_voxels", double( V ) );                        //  roofline of a sweep is quoted on,
    ctx->stageAddHostMs( "refine_row_entries", double( totalLen ) );            //  V and L = entries / V)
    if ( wantTiming ) {
      unsigned long long t[8];
      TMC2_HIP( hipMemcpyAsync( t, d_timing.p, 64, hipMemcpyDeviceToHost, s ) );
      TMC2_HIP( hipStreamSynchronize( s ) );
      const double us = 0.01 / iterationCount;  // wall_clock64 ticks at 100 MHz
      fprintf( stderr, "refine closure tail, per sweep: load %.1f us, compaction %.1f us, voxels %.1f us, lists %.1f us; %.1f levels, %.0f listed voxels (V = %u)\n",
               t[0] * us, t[1] * us, t[2] * us, t[3] * us, double( t[4] ) / iterationCount, double( t[5] ) / iterationCount, V );
    }
    if ( wantTrace ) {  // test hook: points moved per sweep
      std::vector<uint32_t> h_flags( 2 * size_t( iterationCount ) + 2 );
      TMC2_HIP( hipMemcpyAsync( h_flags.data(), d_flags.p, h_flags.size() * 4, hipMemcpyDeviceToHost, s ) );
      TMC2_HIP( hipStreamSynchronize( s ) );
      fprintf( stderr, "refine: points moved per sweep:" );
      for ( int m = 0; m < iterationCount; ++m ) fprintf( stderr, " %u", h_flags[2 * m + 1] );
      fprintf( stderr, "\n" );
    }
    // (no synchronisation: the partition stays on the device and the next stage is queued behind the sweeps; the
    // buffers go back to the context's pool, whose blocks are only ever reused by work queued on this same stream)
    return TMC2_OK;
  }
  if ( devRange != 1 ) {
    setError( "refineSegmentationGridBased: %u voxels of %d: beyond the event-driven sweep loop", V, voxDim );
    return TMC2_E_UNSUPPORTED;
  }
  ctx->stageEnd( sidSetup );
  TMC2_HIP( hipGetLastError() );

  const int sidSweep = ctx->stageBegin( "refine_sweeps" );
  DevBuf<uint32_t> d_out, d_bits;
  TMC2_TRY( d_out.alloc( size_t( V ) * 32 ) );
  TMC2_TRY( d_bits.alloc( 3 * size_t( W ) ) );
  uint32_t *d_activeBits = d_bits.p, *d_frontierBits = d_bits.p + W, *d_nextBits = d_bits.p + 2 * size_t( W );
  TMC2_HIP( hipMemsetAsync( d_bits.p, 0, 3 * size_t( W ) * 4, s ) );
  const size_t tailLds   = 3 * size_t( W ) * 4;
  // (test hook TMC2_REFINE_TAIL=global: take the global-memory tail regardless, the path of grids > 349 K voxels)
  const char*  tailEnv   = getenv( "TMC2_REFINE_TAIL" );
  const bool   tailInLds = tailLds <= 128 * 1024 && !( tailEnv && tailEnv[0] == 'g' );
  if ( tailInLds && tailLds > 48 * 1024 ) TMC2_TRY( allowLargeLds( reinterpret_cast<const void*>( closureTailKernel ), tailLds, ctx->device ) );
  // d_flags: [0] fixpoint reached; [2k + 1] sweep k moved a point; [2k + 2] sweep k changed a voxel state
  for ( int iter = 0; iter < iterationCount; ++iter ) {
    if ( iter == 0 )
      hipLaunchKernelGGL( smoothKernel<false>, grdV16, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_adjOff.p,
                          d_rowLen.p, d_adj.p, V, d_S.p, d_arg, d_proc, d_edge, d_ppi, d_active, d_marked, d_flags.p, iter );
    else
      hipLaunchKernelGGL( smoothKernel<true>, grdV16, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_adjOff.p,
                          d_rowLen.p, d_adj.p, V, d_S.p, d_arg, d_proc, d_edge, d_ppi, d_active, d_marked, d_flags.p, iter );
    hipLaunchKernelGGL( closureRoundZeroKernel, dim3( ( V + 7 ) / 8 ), blk, 0, s, d_edge, d_ppi, d_arg, d_dev.p, V, d_active, d_marked,
                        d_out.p, d_activeBits, d_frontierBits, d_flags.p, iter );
    if ( tailInLds )
      hipLaunchKernelGGL( closureTailKernel, dim3( 1 ), dim3( 1024 ), tailLds, s, d_out.p, W, d_active, d_marked,
                          d_activeBits, d_frontierBits, d_flags.p );
    else
      hipLaunchKernelGGL( closureTailGlobalKernel, dim3( 1 ), dim3( 1024 ), 0, s, d_out.p, W, d_active, d_marked,
                          d_activeBits, d_frontierBits, d_nextBits, d_flags.p );
    hipLaunchKernelGGL( rescoreVoxelsKernel, grdV16, blk, 0, s, d_edge, d_ppi, d_active, d_S.p, d_weight.p, d_pointStart.p,
                        d_pointList.p, f->d_normals.p, V, d_proc, reinterpret_cast<uint4*>( d_hist.p ), f->d_partition.p,
                        d_flags.p, iter );
    // the voxel-state update of this sweep rides in the next sweep's smoothKernel; after the last sweep nobody reads it
  }
  std::vector<uint32_t> h_flags( 2 * size_t( iterationCount ) + 2 );
  TMC2_HIP( hipMemcpyAsync( h_flags.data(), d_flags.p, h_flags.size() * 4, hipMemcpyDeviceToHost, s ) );
  ctx->stageEnd( sidSweep );
  TMC2_HIP( hipGetLastError() );
  TMC2_HIP( hipStreamSynchronize( s ) );
  {
    int executed = iterationCount;
    if ( h_flags[0] )
      for ( int m = 1; m < iterationCount; ++m )
        if ( h_flags[2 * m - 1] == 0 && h_flags[2 * m] == 0 ) {
          executed = m;
          break;
        }
    ctx->stageAddHostMs( "refine_sweeps_executed", double( executed ) );  // (a count, not milliseconds)
    if ( getenv( "TMC2_REFINE_TRACE" ) ) {  // test hook: points moved per sweep
      fprintf( stderr, "refine: points moved per sweep:" );
      for ( int m = 0; m < iterationCount; ++m ) fprintf( stderr, " %u", h_flags[2 * m + 1] );
      fprintf( stderr, "\n" );
    }
  }
  return TMC2_OK;
}

// the first half ahead of time (tmc2_segmenter_compute runs it before the orientation walk); kept with the frame
int refinePrepareGeometry( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim, int searchRadius ) {
  auto job             = std::make_shared<RefineJob>();
  job->maxNNCount      = maxNNCount;
  job->lambda          = lambda;
  job->iterationCount  = iterationCount;
  job->voxDim          = voxDim;
  job->searchRadius    = searchRadius;
  f->refineJob.reset();
  TMC2_TRY( job->geometry( f ) );
  f->refineJob = job;
  return TMC2_OK;
}

int refineGridBased( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount, int voxDim, int searchRadius ) {
  if ( !f->haveNormals || !f->havePartition ) {
    setError( "refineSegmentationGridBased: normals / partition missing" );
    return TMC2_E_STATE;
  }
  std::shared_ptr<RefineJob> job = std::static_pointer_cast<RefineJob>( f->refineJob );
  f->refineJob.reset();
  if ( !job || !job->matches( maxNNCount, lambda, iterationCount < 1 ? 1 : iterationCount, voxDim, searchRadius ) ) {
    job.reset();
    TMC2_TRY( refinePrepareGeometry( f, maxNNCount, lambda, iterationCount, voxDim, searchRadius ) );
    job = std::static_pointer_cast<RefineJob>( f->refineJob );
    f->refineJob.reset();
  }
  return job->finish();
}

}  // namespace tmc2
