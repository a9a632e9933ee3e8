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
// (bits: one bit per key -- the occupancy of the key table, which the neighbourhood passes probe 64 cells of a ball row at a
//  time; look before the atomic: a voxel's points set the same bit)
__global__ __launch_bounds__( 256 ) void voxelKeyKernel( const Pt* __restrict__ pts, uint32_t n, Grid g,
                                                          uint32_t* __restrict__ key, uint32_t* __restrict__ table,
                                                          uint2* __restrict__ bits /* .x: 32 keys' occupancy, .y: see below */ ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const Pt       p = pts[i];
  const uint32_t k = cellKey( ( int( p.x ) + g.half ) >> g.voxShift, ( int( p.y ) + g.half ) >> g.voxShift,
                              ( int( p.z ) + g.half ) >> g.voxShift, g.gridShift );
  key[i]           = k;
  if ( table ) atomicMin( &table[k], i );  // (the cell-by-cell form only: the row-wise form numbers the voxels through the ranks)
  const uint32_t bit = 1u << ( k & 31 );
  if ( !( loadStaleOk( &bits[k >> 5].x ) & bit ) ) atomicOr( &bits[k >> 5].x, bit );
}

// The occupancy words carry the number of occupied keys BELOW them (bits[w].y = exclusive prefix sum of the popcounts): the rank
// of a key among the occupied keys is then one 8-byte load and a popcount, and voxelOfRank[rank] is the voxel that owns the key
// -- a look-up chain through a few dense megabytes that neighbouring voxels share (L2), instead of one random 4-byte read of the
// gigabyte-sized key table per occupied cell (what bounded the neighbourhood pass of voxels of 2: 236 M reads, 64 bytes of HBM
// each).  Three launches per frame over the words: totals of 2 048-word blocks, their prefix sum, the words' prefix sums.
constexpr int kBitsPerThread = 8, kBitsBlock = 256 * kBitsPerThread;
__global__ __launch_bounds__( 256 ) void bitsBlockCountKernel( const uint2* __restrict__ bits, uint32_t words, uint32_t* __restrict__ blockTotal ) {
  const uint32_t base = blockIdx.x * kBitsBlock + threadIdx.x * kBitsPerThread;
  uint32_t       sum  = 0;
#pragma unroll
  for ( int j = 0; j < kBitsPerThread; ++j )
    if ( base + j < words ) sum += __popc( bits[base + j].x );
  __shared__ uint32_t waveSum[4];
#pragma unroll
  for ( int off = 32; off > 0; off >>= 1 ) sum += __shfl_xor( sum, off, 64 );
  if ( ( threadIdx.x & 63 ) == 0 ) waveSum[threadIdx.x >> 6] = sum;
  __syncthreads();
  if ( threadIdx.x == 0 ) blockTotal[blockIdx.x] = waveSum[0] + waveSum[1] + waveSum[2] + waveSum[3];
}
__global__ __launch_bounds__( 256 ) void bitsPrefixKernel( uint2* __restrict__ bits, uint32_t words, const uint32_t* __restrict__ blockBase ) {
  const uint32_t base = blockIdx.x * kBitsBlock + threadIdx.x * kBitsPerThread;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t       cnt[kBitsPerThread], sum = 0;
#pragma unroll
  for ( int j = 0; j < kBitsPerThread; ++j ) {
    cnt[j] = base + j < words ? uint32_t( __popc( bits[base + j].x ) ) : 0u;
    sum += cnt[j];
  }
  uint32_t inc = sum;
#pragma unroll
  for ( int off = 1; off < 64; off <<= 1 ) {
    const uint32_t t = __shfl_up( inc, off, 64 );
    if ( lane >= off ) inc += t;
  }
  __shared__ uint32_t waveSum[4];
  if ( lane == 63 ) waveSum[wave] = inc;
  __syncthreads();
  uint32_t at = blockBase[blockIdx.x] + inc - sum;
  for ( int w = 0; w < wave; ++w ) at += waveSum[w];
#pragma unroll
  for ( int j = 0; j < kBitsPerThread; ++j ) {
    if ( base + j < words ) bits[base + j].y = at;
    at += cnt[j];
  }
}
__device__ __forceinline__ uint32_t rankOfKey( const uint2* __restrict__ bits, uint32_t k ) {
  const uint2 e = bits[k >> 5];
  return e.y + uint32_t( __popc( e.x & ( ( 1u << ( k & 31 ) ) - 1u ) ) );
}
// voxelOfRank[rank of the voxel's key] = the voxel that owns the key, as ONE 8-byte record (written by the voxel's first point):
//   .x = its cell (x | y << 10 | z << 20: what a probe compares with the cell it looked at -- keys alias) | member count & 3 << 30
//   .y = voxel id (26 bits) | member count >> 2 << 26        (member count capped at 255, as centre[].w)
// Round 4-5 kept the id alone and read the centre through it: rank -> id -> centre, three dependent loads per hit of a ball with the
// bitmap word before them; the record makes it two.
__device__ __forceinline__ uint2 rankRecord( const Pt c, uint32_t v, uint32_t members ) {
  const uint32_t w = members & 0xFFu;
  return make_uint2( uint32_t( c.x ) | ( uint32_t( c.y ) << 10 ) | ( uint32_t( c.z ) << 20 ) | ( ( w & 3u ) << 30 ), v | ( ( w >> 2 ) << 26 ) );
}
__global__ __launch_bounds__( 256 ) void rankToVoxelKernel( const uint32_t* __restrict__ key, const uint32_t* __restrict__ flag,
                                                             const uint32_t* __restrict__ vid, uint32_t n,
                                                             const uint2* __restrict__ bits, uint2* __restrict__ voxelOfRank,
                                                             const uint32_t* __restrict__ count, Pt* __restrict__ centre ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n && flag[i] ) {
    const uint32_t v = vid[i], members = count[v];
    Pt             c = centre[v];
    voxelOfRank[rankOfKey( bits, key[i] )] = rankRecord( c, v, members );
    centre[v].w = int16_t( members & 0xFFu );  // (the DEV rows read a voxel's centre anyway: its member count rides along)
  }
}

// The first point of every voxel WITHOUT the dense key table (row-wise form: rounds 1-3 kept 2^(3s+1) words per context for this,
// 1 GiB with voxels of 2 or 11-bit geometry): firstPoint[rank of the point's key] takes the minimum over the voxel's points.
__global__ __launch_bounds__( 256 ) void firstPointKernel( const uint32_t* __restrict__ key, uint32_t n, const uint2* __restrict__ bits,
                                                            uint32_t* __restrict__ firstPoint ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  uint32_t* slot = &firstPoint[rankOfKey( bits, key[i] )];
  if ( i < loadStaleOk( slot ) ) atomicMin( slot, i );
}
// the first point of the voxel point i belongs to: through the key table, or through the ranks
__device__ __forceinline__ uint32_t firstPointOf( uint32_t k, const uint32_t* __restrict__ table, const uint2* __restrict__ bits,
                                                  const uint32_t* __restrict__ firstPoint ) {
  return table ? table[k] : firstPoint[rankOfKey( bits, k )];
}

__global__ __launch_bounds__( 256 ) void firstFlagKernel( const uint32_t* __restrict__ key,
                                                           const uint32_t* __restrict__ table, const uint2* __restrict__ bits,
                                                           const uint32_t* __restrict__ firstPoint, uint32_t n,
                                                           uint32_t* __restrict__ flag ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i < n ) flag[i] = ( firstPointOf( key[i], table, bits, firstPoint ) == i ) ? 1u : 0u;
}

// vid[i] = rank of the voxel's first point; member counts; centre of each voxel
__global__ __launch_bounds__( 256 ) void assignVoxelKernel( const Pt* __restrict__ pts, const uint32_t* __restrict__ key,
                                                             const uint32_t* __restrict__ table, const uint2* __restrict__ bits,
                                                             const uint32_t* __restrict__ firstPoint,
                                                             const uint32_t* __restrict__ rank, uint32_t n, Grid g,
                                                             uint32_t* __restrict__ vid, uint32_t* __restrict__ count,
                                                             Pt* __restrict__ centre ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint32_t first = firstPointOf( key[i], table, bits, firstPoint );
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
                                                            uint32_t* __restrict__ table, uint2* __restrict__ bits ) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  if ( table ) table[key[i]] = 0xFFFFFFFFu;
  bits[key[i] >> 5].x = 0u;  // (whole words: every key of the word belongs to this frame; the prefix sums are rebuilt per frame)
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

// The hits of every voxel's ball, kept between the two passes over the balls (round 6): kHitRegions regions with their own
// cursors (128 bytes apart), a wavefront reserves room for its voxel's hits in the region of its workgroup.
constexpr uint32_t kHitRegions = 64;
// The rows (forward, reverse) likewise: kRowRegions regions of the row tables with their own cursors, ONE reservation per
// wavefront.  Rounds 1-5 reserved per workgroup on ONE word: 65 K workgroups (voxels of 2, four wavefronts each) x ~ 11 ns per
// returning atomic on one address = 0.72 ms -- the reverse rows kernel's whole 0.77 ms and half of the forward kernel's 1.3
// (tools/gpu/r6/call34.sh: the instantiation with eight wavefronts per workgroup, half the reservations, took 0.44 ms) -- with
// two barriers around it that made every wavefront of a workgroup wait for its slowest.  Sixteen consecutive voxels share a region.
constexpr uint32_t kRowRegions = 128;
__device__ __forceinline__ uint32_t rowRegionOf( uint32_t v ) { return ( v >> 4 ) & ( kRowRegions - 1u ); }
struct BallHits {
  uint32_t* buf;        // [kHitRegions][regionCap] keys ( d2 << idBits | voxel )
  uint32_t  regionCap;
  uint32_t* cursor;     // [kHitRegions * 32]
  uint32_t* off;        // [V] where a voxel's hits start
  uint32_t* len;        // [V] how many
};

// ---- neighbourhoods, row-wise (round 4) ----------------------------------------------------------------------------------
// The ball as ROWS: for every (dy, dz) with dy^2 + dz^2 < r2 the cells dx = -xr .. xr (xr the largest with dx^2 + dy^2 + dz^2
// < r2).  Keys are linear in x, so the <= 2 xr + 1 <= 25 cells of a row are consecutive BITS of the occupancy bitmap: one lane
// probes a whole row with two 4-byte loads instead of one load per cell (voxels of 2: 301 rows against 3 911 cells, of which
// 8 % are occupied).  Only the occupied cells go on to the key table (the voxel id) and to the voxel's centre (aliased keys:
// accept only the voxel whose centre really sits there).  Leaves the hits in keys[] as ( d2 << idBits ) | id, unsorted.
//   rows[i] = ( dy + 128 ) | ( dz + 128 ) << 8 | xr << 16
// phase 1: the occupied cells of the ball as packed offsets ( dx + 16 ) | ( dy + 16 ) << 5 | ( dz + 16 ) << 10 | d2 << 15
// phase 2: id + centre of each, in place (a chunk of candidates is read whole before anything lands at or below it)
template <int CAP>
__device__ __forceinline__ int collectBall( const Pt c, const Pt* __restrict__ centre, const uint2* __restrict__ voxelOfRank,
                                            const uint2* __restrict__ bits, const Grid g, const int* __restrict__ rows, int nRows,
                                            int idBits, uint32_t* keys, int lane, uint32_t* __restrict__ overflow,
                                            uint32_t* bins /* LDS, zeroed, or null: members (low 20 bits) and hits (above) per squared distance */ ) {
  const int gridMax = 1 << g.gridShift;  // cell coordinates run 0 .. gridMax inclusive
  int       cand    = 0;
  constexpr int kRowBatch = 5;  // (5 x 64 rows: the whole ball of the CTC settings in one round of loads)
  for ( int base = 0; base < nRows; base += 64 * kRowBatch ) {
    uint32_t lo[kRowBatch], hi[kRowBatch];
    int      sh[kRowBatch], len[kRowBatch], x0[kRowBatch], dyz[kRowBatch];
#pragma unroll
    for ( int k = 0; k < kRowBatch; ++k ) {
      const int r = base + 64 * k + lane;
      lo[k] = hi[k] = 0u;
      sh[k] = len[k] = x0[k] = dyz[k] = 0;
      if ( r < nRows ) {
        const int packed = rows[r];
        const int dy = ( packed & 0xFF ) - 128, dz = ( ( packed >> 8 ) & 0xFF ) - 128, xr = packed >> 16;
        const int y = c.y + dy, z = c.z + dz;
        const int xlo = max( 0, int( c.x ) - xr ), xhi = min( gridMax, int( c.x ) + xr );
        if ( y >= 0 && z >= 0 && y <= gridMax && z <= gridMax && xlo <= xhi ) {
          const uint32_t key0 = cellKey( xlo, y, z, g.gridShift );
          lo[k]  = bits[key0 >> 5].x;
          hi[k]  = bits[( key0 >> 5 ) + 1].x;  // (the bitmap carries a spare word behind its last key)
          sh[k]  = int( key0 & 31 );
          len[k] = xhi - xlo + 1;
          x0[k]  = xlo - int( c.x );
          dyz[k] = ( ( dy + 16 ) << 5 ) | ( ( dz + 16 ) << 10 ) | ( ( dy * dy + dz * dz ) << 15 );
        }
      }
    }
#pragma unroll
    for ( int k = 0; k < kRowBatch; ++k ) {
      uint32_t w = uint32_t( ( ( (unsigned long long)hi[k] << 32 ) | lo[k] ) >> sh[k] ) & ( len[k] >= 32 ? 0xFFFFFFFFu : ( ( 1u << len[k] ) - 1u ) );
      const int mine = __popc( w );
      int       inc  = mine;
#pragma unroll
      for ( int off = 1; off < 64; off <<= 1 ) {
        const int t = __shfl_up( inc, off, 64 );
        if ( lane >= off ) inc += t;
      }
      const int total = __shfl( inc, 63, 64 );
      int       at    = cand + inc - mine;
      while ( w ) {
        const int b = __ffs( int( w ) ) - 1;
        w &= w - 1u;
        const int dx = x0[k] + b;
        if ( at < CAP - 160 ) keys[at] = uint32_t( dx + 16 ) | uint32_t( dyz[k] + ( ( dx * dx ) << 15 ) );
        ++at;
      }
      cand += total;
    }
  }
  if ( cand > CAP - 160 ) {  // more occupied cells than this instantiation has room for: the host repeats with a larger one
    if ( lane == 0 ) {
      atomicMax( overflow, 3u );  // (atomic like every other writer of this word; the fatal code is the largest)
      atomicMax( overflow + 2, uint32_t( cand ) );  // (... and which one: the room the fullest ball asks for)
    }
    cand = CAP - 160;
  }
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  int hits = 0;
  constexpr int kBatch = 4;
  for ( int base = 0; base < cand; base += 64 * kBatch ) {
    uint32_t u[kBatch], d2[kBatch], cell[kBatch];
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      const int i = base + 64 * k + lane;
      u[k]        = 0xFFFFFFFFu;
      d2[k] = cell[k] = 0;
      if ( i < cand ) {
        const uint32_t packed = keys[i];
        const int x = c.x + int( packed & 31u ) - 16, y = c.y + int( ( packed >> 5 ) & 31u ) - 16, z = c.z + int( ( packed >> 10 ) & 31u ) - 16;
        u[k]    = rankOfKey( bits, cellKey( x, y, z, g.gridShift ) );  // (the words phase 1 has just read: L1 / L2)
        d2[k]   = packed >> 15;
        cell[k] = uint32_t( x ) | ( uint32_t( y ) << 10 ) | ( uint32_t( z ) << 20 );  // (cell coordinates are at most 512)
      }
    }
    uint32_t key[kBatch];
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      key[k] = 0xFFFFFFFFu;
      if ( u[k] != 0xFFFFFFFFu ) {
        const uint2 rec = voxelOfRank[u[k]];  // aliased keys: accept only the voxel whose centre really sits here
        if ( ( rec.x & 0x3FFFFFFFu ) == cell[k] ) {
          key[k] = ( d2[k] << idBits ) | ( rec.y & 0x3FFFFFFu );
          if ( bins ) atomicAdd( &bins[min( d2[k], 127u )], ( ( rec.x >> 30 ) | ( ( rec.y >> 26 ) << 2 ) ) | ( 1u << 20 ) );
        }
      }
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );  // (every lane has read its candidates of this batch)
#pragma unroll
    for ( int k = 0; k < kBatch; ++k ) {
      const unsigned long long m = __ballot( key[k] != 0xFFFFFFFFu );
      if ( key[k] != 0xFFFFFFFFu ) keys[hits + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = key[k];
      hits += __popcll( m );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  }
  return hits;
}

// ---- neighbourhoods ---------------------------------------------------------------------------------
// One wavefront per voxel.  offsets[] = all integer (dx,dy,dz) with d2 < radius2, packed, any order.
// Collect hits (d2 << idBits | voxel id) in LDS, bitonic-sort, cut after the cumulative member count reaches maxNN; write
// row length, weight, the row itself (rows back to back: the wave reserves its slots from a cursor -- what a frame needs
// is ~ maxNN / (points per voxel) entries per voxel, a fraction of the ball) and the DEV row: the members of the row
// within Chebyshev distance devRange (1 for voxels of 4 and more, 2 for voxels of 2: PCCPatchSegmenter.cpp:1469-1492), in
// row order, padded to devStride entries.
constexpr uint32_t kDevPad = 0xFFFFFFFFu;
// wavefronts per SIMD that the LDS of a ( CAP, WAVES ) instantiation leaves room for: what the register allocation has to meet
// (round 6: the code that keeps the ball's hits took neighbourhoodKernel from 62 to 72 registers -- 7 wavefronts instead of 8)
constexpr int ldsWavesPerSimd( int cap, int waves ) {
  const int perCu = ( 160 * 1024 / ( cap * waves * 4 ) ) * waves;
  return perCu >= 32 ? 8 : ( perCu / 4 > 0 ? perCu / 4 : 1 );
}

template <int CAP, int WAVES>
__global__ __launch_bounds__( 64 * WAVES, ldsWavesPerSimd( CAP, WAVES ) ) void neighbourhoodKernel(
    const Pt* __restrict__ centre, const uint32_t* __restrict__ count, const uint32_t* __restrict__ table, Grid g, uint32_t V,
    const int* __restrict__ offsets, int nOffsets, int maxNN, double lambda, int idBits, int devRange, uint32_t devStride,
    uint32_t rowCapacity, uint32_t* __restrict__ rowLen, uint32_t* __restrict__ devLen, double* __restrict__ weight,
    uint32_t* __restrict__ adjOff, uint32_t* __restrict__ adj, uint32_t* __restrict__ dev, uint32_t* __restrict__ rowCursor,
    uint32_t* __restrict__ overflow, const uint2* __restrict__ bits /* non-null: offsets = the ball's ROWS (collectBall) */,
    const uint2* __restrict__ voxelOfRank /* record of the voxel that owns the rank-th occupied key (rankRecord) */,
    uint32_t* __restrict__ lastKey /* the last key each row keeps: what the gathered reverse rows test against */,
    BallHits saved /* .buf non-null: the ball's hits are kept for the reverse rows (round 6) */ ) {
  __shared__ uint32_t keysAll[WAVES][CAP];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t      v    = blockIdx.x * WAVES + wave;
  uint32_t*           keys = keysAll[wave];
  if ( v >= V ) return;  // (a wave of the last workgroup without a voxel; the kernel has no workgroup barrier)
  const uint32_t idMask  = ( 1u << idBits ) - 1u;
  const Pt       c       = centre[v];
  const int      gridMax = 1 << g.gridShift;  // cell coordinates run 0..gridMax inclusive
  int            hits    = 0;
  // The ball's cells in batches of kBatch x 64: all table look-ups of a batch are issued before the first one is needed, then
  // all centre look-ups of its hits (two dependent round trips per batch instead of two per 64 cells: the kernel is bound by
  // exactly this latency).  The order of the hits does not matter: they are sorted below.
  constexpr int kBatch = 8;
  uint32_t*     bins   = keys + CAP - 160;  // (the host checks that the ball leaves this room)
  for ( int b = lane; b < 128; b += 64 ) bins[b] = 0;
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  // (row-wise form: the bins -- members and hits per squared distance -- fill while the ball is collected: a voxel's member
  //  count rides in its centre record, which the collection reads anyway)
  if ( bits ) hits = collectBall<CAP>( c, centre, voxelOfRank, bits, g, offsets, nOffsets, idBits, keys, lane, overflow, bins );
  if ( saved.buf ) {  // the hits as they are now (every voxel of the ball, unsorted): the reverse rows pass reads them instead of collecting the ball again
    const uint32_t region = blockIdx.x % kHitRegions;
    uint32_t       at     = 0;
    if ( lane == 0 ) at = atomicAdd( &saved.cursor[region * 32], uint32_t( hits ) );  // (one reservation per wavefront, spread over kHitRegions words)
    at = __shfl( at, 0, 64 );
    const bool room = uint64_t( at ) + uint32_t( hits ) <= saved.regionCap;
    if ( room )
      for ( int i = lane; i < hits; i += 64 ) saved.buf[size_t( region ) * saved.regionCap + at + i] = keys[i];
    if ( lane == 0 ) {
      saved.off[v] = region * saved.regionCap + at;
      saved.len[v] = uint32_t( hits );
      if ( !room ) atomicMax( overflow + 3, 1u );  // (the reverse rows pass collects the balls itself, the next frames get more room)
    }
  }
  for ( int base = 0; !bits && base < nOffsets; base += 64 * kBatch ) {
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
  uint32_t membersBefore = 0, hitsBefore = 0, hitsCut = 0;  // of the distance the cut falls in
  bool     cutFound      = false;
  // Row-wise form: a row is a SET to everything that reads it (integer sums over it, marks through it, the reverse rows test
  // membership by the last key), so only the hits of the distance the cut falls in have to be put in order -- by voxel id, to
  // find where the row ends -- and not the whole row: ~ 100 keys to sort instead of 400-500 (the sort was two thirds of this
  // kernel).  They go to the free room above the hits; the cell-by-cell form (and a ball too full for that room) sorts everything.
  bool     partial       = false;
  int      Pc            = 64;
  uint32_t lastKeyValue  = 0;
  {
    if ( !bits )
      for ( int i = lane; i < hits; i += 64 ) {
        const uint32_t key = keys[i];
        atomicAdd( &bins[min( key >> idBits, 127u )], ( count[key & idMask] & 0xFFu ) | ( 1u << 20 ) );
      }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    const uint32_t w0 = bins[2 * lane], w1 = bins[2 * lane + 1];
    const uint32_t b0 = w0 & 0xFFFFFu, b1 = w1 & 0xFFFFFu, e0 = w0 >> 20, e1 = w1 >> 20;
    uint32_t       inc = b0 + b1, einc = e0 + e1;
#pragma unroll
    for ( int off = 1; off < 64; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 ), te = __shfl_up( einc, off, 64 );
      if ( lane >= off ) inc += t, einc += te;
    }
    const uint32_t           before  = inc - ( b0 + b1 ), ebefore = einc - ( e0 + e1 );
    const unsigned long long reached = __ballot( inc >= uint32_t( maxNN ) );
    uint32_t                 cutoff  = 127;  // (fewer members than maxNN in the whole ball: everything stays)
    if ( reached ) {
      const int  first = __ffsll( (long long)reached ) - 1;
      const bool even  = __shfl( before + b0, first, 64 ) >= uint32_t( maxNN );
      cutoff           = 2u * uint32_t( first ) + ( even ? 0u : 1u );
      membersBefore    = __shfl( even ? before : before + b0, first, 64 );
      hitsBefore       = __shfl( even ? ebefore : ebefore + e0, first, 64 );
      hitsCut          = __shfl( even ? e0 : e1, first, 64 );
      cutFound         = true;
      while ( Pc < int( hitsCut ) ) Pc <<= 1;
      partial = bits != nullptr && hits + Pc <= CAP - 160;
    }
    if ( partial ) {
      uint32_t* cut   = keys + ( CAP - 160 - Pc );  // (above every hit: nothing this loop still has to read)
      int       front = 0, tail = 0;
      for ( int base = 0; base < hits; base += 64 ) {  // the front in place: a chunk is read whole before anything lands at or below it
        const int      i   = base + lane;
        const uint32_t key = i < hits ? keys[i] : 0xFFFFFFFFu;
        const bool     isFront = i < hits && ( key >> idBits ) < cutoff, isCut = i < hits && ( key >> idBits ) == cutoff;
        const unsigned long long mF = __ballot( isFront ), mC = __ballot( isCut ), below = ( 1ull << lane ) - 1ull;
        __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
        if ( isFront ) keys[front + __popcll( mF & below )] = key;
        if ( isCut ) cut[tail + __popcll( mC & below )] = key;
        front += __popcll( mF ), tail += __popcll( mC );
      }
      for ( int i = int( hitsCut ) + lane; i < Pc; i += 64 ) cut[i] = 0xFFFFFFFFu;
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      for ( int k = 2; k <= Pc; k <<= 1 )
        for ( int j = k >> 1; j > 0; j >>= 1 ) {
          for ( int i = lane; i < Pc; i += 64 ) {
            const int partner = i ^ j;
            if ( partner > i ) {
              const uint32_t a = cut[i], b = cut[partner];
              if ( ( a > b ) == ( ( i & k ) == 0 ) ) cut[i] = b, cut[partner] = a;
            }
          }
          __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
        }
      // where the row ends inside the cut distance
      uint32_t running = membersBefore, m = hitsCut;
      for ( int base = 0; base < int( hitsCut ); base += 64 ) {
        const int i   = base + lane;
        uint32_t  inc = i < int( hitsCut ) ? ( count[cut[i] & idMask] & 0xFFu ) : 0u;
#pragma unroll
        for ( int off = 1; off < 64; off <<= 1 ) {
          const uint32_t t = __shfl_up( inc, off, 64 );
          if ( lane >= off ) inc += t;
        }
        inc += running;
        const unsigned long long reachedHere = __ballot( i < int( hitsCut ) && inc >= uint32_t( maxNN ) );
        if ( reachedHere ) {
          const int firstLane = __ffsll( (long long)reachedHere ) - 1;
          m                   = uint32_t( base + firstLane + 1 );
          running             = __shfl( inc, firstLane, 64 );
          break;
        }
        running = __shfl( inc, 63, 64 );
      }
      membersBefore = running;  // = the row's member count
      lastKeyValue  = cut[m - 1];
      for ( int i = lane; i < int( m ); i += 64 ) keys[hitsBefore + i] = cut[i];  // (the front ends below the room the cut hits sat in)
      hits = int( hitsBefore + m );
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    }
    int kept = 0;
    for ( int base = 0; !partial && base < hits; base += 64 ) {  // in place: a chunk is read whole before anything lands at or below it
      const int      i    = base + lane;
      const uint32_t key  = i < hits ? keys[i] : 0xFFFFFFFFu;
      const bool     stay = i < hits && ( key >> idBits ) <= cutoff;
      const unsigned long long m = __ballot( stay );
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      if ( stay ) keys[kept + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = key;
      kept += __popcll( m );
    }
    if ( !partial ) hits = kept;
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  }
  // pad to a power of two and sort ascending (wave-private LDS: no barrier needed beyond wave lockstep,
  // but LDS visibility between lanes needs the s_waitcnt the compiler inserts for __syncthreads-free code:
  // use __builtin_amdgcn_wave_barrier to keep the order of LDS operations)
  int P = 64;
  while ( P < hits ) P <<= 1;
  if ( partial ) P = 0;  // (nothing left to sort)
  if ( P > CAP ) {  // (the instantiations between the powers of two: a row that does not fit the padded sort goes one tier up)
    if ( lane == 0 ) {
      atomicMax( overflow, 3u );
      atomicMax( overflow + 2, uint32_t( CAP - 159 ) );
    }
    P = 0, hits = 0;
  }
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
  // truncation: first position where the running member count reaches maxNN (inclusive).  The cut falls in the last distance
  // kept (that is how it was chosen): everything before it counts whole -- members and hits are known from the bins -- and only
  // the hits of that distance, sorted by voxel id, are walked.
  uint32_t running = cutFound ? membersBefore : 0u;
  int      used    = hits;
  uint32_t nn      = 0;
  bool     done    = false;
  if ( partial ) nn = membersBefore, done = true;  // (found above: the row is keys[0 .. hits))
  if ( !cutFound ) {  // fewer members than maxNN in the whole ball: the row is the ball
    for ( int b = lane; b < 128; b += 64 ) nn += bins[b] & 0xFFFFFu;
#pragma unroll
    for ( int off = 32; off > 0; off >>= 1 ) nn += __shfl_xor( nn, off, 64 );
    done = true;
  }
  for ( int base = int( hitsBefore ); base < hits && !done; base += 64 ) {
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
  // the row's place in the table: one reservation per wavefront in the region of its voxel (kRowRegions cursors 128 bytes apart)
  const uint32_t region  = rowRegionOf( v );
  uint32_t       rowBase = 0;
  if ( lane == 0 ) rowBase = atomicAdd( &rowCursor[region * 32], uint32_t( used ) );
  rowBase        = __shfl( rowBase, 0, 64 );
  const bool fit = uint64_t( rowBase ) + uint32_t( used ) <= rowCapacity;  // (rowCapacity: of a region)
  rowBase += region * rowCapacity;
  if ( lane == 0 ) {
    rowLen[v] = uint32_t( used );
    if ( lastKey ) lastKey[v] = partial ? lastKeyValue : ( used > 0 ? keys[used - 1] : 0u );
    adjOff[v] = fit ? rowBase : 0u;
    weight[v] = __ddiv_rn( lambda, double( nn ) );
    if ( !fit ) atomicMax( overflow, 1u );  // the host repeats the pass with room for whole balls
  }
  if ( fit )
    for ( int i = lane; i < used; i += 64 ) adj[size_t( rowBase ) + i] = keys[i] & idMask;
  // DEV row.  Chebyshev distance <= R implies d2 <= 3 R^2: candidates are a prefix of the (sorted) row.
  const uint32_t d2Max = uint32_t( 3 * devRange * devRange );
  uint32_t*      drow  = dev + size_t( v ) * devStride;
  uint32_t       nDev  = 0;
  // (round 6: the candidates -- d2 <= 3 R^2, a few dozen of a row's hundreds, anywhere in a row whose front is not sorted -- are
  //  gathered first, in row order, into the free room above the row; their centres are then fetched together: one round trip to
  //  memory instead of one per 64 row entries, each waiting for the one before)
  constexpr int kNearRoom = 192;  // (179 cells have d2 <= 12)
  if ( used + kNearRoom <= CAP ) {
    uint32_t* nearList = keys + used;  // (the bins above are no longer needed)
    int       nNear    = 0;
    for ( int base = 0; base < used; base += 64 ) {
      const int  i    = base + lane;
      const bool near = i < used && ( keys[i] >> idBits ) <= d2Max;
      const unsigned long long m = __ballot( near );
      if ( near ) {
        const int at = nNear + __popcll( m & ( ( 1ull << lane ) - 1ull ) );
        if ( at < kNearRoom ) nearList[at] = keys[i] & idMask;
      }
      nNear += __popcll( m );
      if ( !partial && !m ) break;  // (a sorted row: nothing near can follow)
    }
    nNear = min( nNear, kNearRoom );  // (more than 179 cannot be near; a guard, not a case)
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    uint32_t u[3];
    bool     in[3];
#pragma unroll
    for ( int k = 0; k < 3; ++k ) {
      const int i = 64 * k + lane;
      u[k]        = i < nNear ? nearList[i] : 0u;
      in[k]       = false;
      if ( i < nNear ) {
        const Pt cu = centre[u[k]];
        in[k]       = abs( int( cu.x ) - int( c.x ) ) <= devRange && abs( int( cu.y ) - int( c.y ) ) <= devRange &&
                abs( int( cu.z ) - int( c.z ) ) <= devRange;
      }
    }
#pragma unroll
    for ( int k = 0; k < 3; ++k ) {
      const unsigned long long m = __ballot( in[k] );
      if ( in[k] ) {
        const uint32_t pos = nDev + uint32_t( __popcll( m & ( ( 1ull << lane ) - 1ull ) ) );
        if ( pos < devStride ) drow[pos] = u[k];
      }
      nDev += uint32_t( __popcll( m ) );
    }
  } else
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
    if ( !partial && !__ballot( near ) ) break;  // (a sorted row: nothing near can follow)
  }
  if ( nDev > devStride ) {  // (cannot happen: (2 R + 1)^3 <= devStride by construction)
    if ( lane == 0 ) atomicMax( overflow, 4u );  // fatal: above the two codes the host answers with a repeat, so that neither hides it
    nDev = devStride;
  }
  for ( uint32_t i = nDev + lane; i < devStride; i += 64 ) drow[i] = kDevPad;
  if ( lane == 0 ) devLen[v] = nDev;
}


// The REVERSE rows by gathering (round 4; rounds 1-3 scattered the forward rows: one returning atomic and one 4-byte store to
// a random line per entry -- 13 x the contract bytes in HBM traffic).  u is listed by v iff u sits in v's ball and survives
// v's truncation, i.e. iff ( d2( u, v ) << idBits | u ) <= the last key v kept (rows are sorted by exactly that key); the ball
// is symmetric, so the candidates are u's own ball.  One wavefront per voxel u: collect the ball as above, keep the v with
// key( u seen from v ) <= lastKey[v], write them back to back (one reservation per workgroup).  The order inside a reverse
// row is immaterial (the sweeps push integer differences over it).
template <int CAP, int WAVES>
__global__ __launch_bounds__( 64 * WAVES, ldsWavesPerSimd( CAP, WAVES ) ) void reverseRowsKernel( const Pt* __restrict__ centre, const uint2* __restrict__ voxelOfRank,
                                                                     const uint2* __restrict__ bits, Grid g, uint32_t V,
                                                                     const int* __restrict__ rows, int nRows, int idBits,
                                                                     const uint32_t* __restrict__ lastKey, uint32_t rowCapacity,
                                                                     uint32_t* __restrict__ rOff, uint32_t* __restrict__ rLen,
                                                                     uint32_t* __restrict__ radj, uint32_t* __restrict__ rowCursor,
                                                                     uint32_t* __restrict__ overflow, BallHits saved ) {
  __shared__ uint32_t keysAll[WAVES][CAP];
  const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t      u    = blockIdx.x * WAVES + wave;
  uint32_t*           keys = keysAll[wave];
  int                 kept = 0;
  // (round 6: the forward pass has kept every ball's hits -- unless some region ran out of room, which its flag says)
  const bool useSaved = saved.buf && overflow[3] == 0;
  if ( u < V && useSaved ) {
    const uint32_t idMask = ( 1u << idBits ) - 1u;
    const uint32_t len = saved.len[u];
    const uint32_t* mine = saved.buf + saved.off[u];
    for ( uint32_t base = 0; base < len; base += 64 ) {
      const uint32_t i   = base + uint32_t( lane );
      const uint32_t key = i < len ? mine[i] : 0u;
      const uint32_t v   = key & idMask;
      const bool     in  = i < len && ( ( key & ~idMask ) | u ) <= lastKey[v];
      const unsigned long long m = __ballot( in );
      if ( in ) keys[kept + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = v;
      kept += __popcll( m );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  } else if ( u < V ) {
    const uint32_t idMask = ( 1u << idBits ) - 1u;
    const int      hits   = collectBall<CAP>( centre[u], centre, voxelOfRank, bits, g, rows, nRows, idBits, keys, lane, overflow, nullptr );
    for ( int base = 0; base < hits; base += 64 ) {  // in place: a chunk is read whole before anything lands at or below it
      const int      i   = base + lane;
      const uint32_t key = i < hits ? keys[i] : 0u;
      const uint32_t v   = key & idMask;
      const bool     in  = i < hits && ( ( key & ~idMask ) | u ) <= lastKey[v];
      const unsigned long long m = __ballot( in );
      __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
      if ( in ) keys[kept + __popcll( m & ( ( 1ull << lane ) - 1ull ) )] = v;
      kept += __popcll( m );
    }
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
  }
  if ( u >= V ) return;
  const uint32_t region  = rowRegionOf( u );
  uint32_t       rowBase = 0;
  if ( lane == 0 ) rowBase = atomicAdd( &rowCursor[region * 32], uint32_t( kept ) );
  rowBase        = __shfl( rowBase, 0, 64 );
  const bool fit = uint64_t( rowBase ) + uint32_t( kept ) <= rowCapacity;  // (rowCapacity: of a region)
  rowBase += region * rowCapacity;
  if ( lane == 0 ) {
    rLen[u] = fit ? uint32_t( kept ) : 0u;
    rOff[u] = fit ? rowBase : 0u;
    if ( !fit ) atomicMax( overflow, 1u );  // (the reverse rows hold the same entries as the forward rows, region by region a little more or less)
  }
  if ( fit )
    for ( int i = lane; i < kept; i += 64 ) radj[size_t( rowBase ) + i] = keys[i];
}

// ---- sweep kernels ----------------------------------------------------------------------------------
// (Rounds 1-2 recomputed everything every sweep -- smoothKernel, a two-step closure with a one-workgroup level walk,
//  rescoreVoxelsKernel: five, then three launches per sweep.  Round 3 replaced them by the event-driven pair below and kept them
//  behind option REFINE_SWEEPS=full as a second cross-check next to the oracle; round 6 took them out of the library:
//  git show 5d8d688:mpeg-pcc-tmc2_amd/csrc/refine.hip.)
// points grouped by voxel (any order inside a voxel: re-scoring is per point, the histograms are integer sums)
__global__ __launch_bounds__( 256 ) void voxelPointListKernel( const uint32_t* __restrict__ vid, const uint32_t* __restrict__ start,
                                                                uint32_t n, uint32_t* __restrict__ cursor,
                                                                uint32_t* __restrict__ list ) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if ( j >= n ) return;
  const uint32_t v                            = vid[j];
  list[start[v] + atomicAdd( &cursor[v], 1u )] = j;
}

// ======================================================================================================================
// Event-driven sweeps.  What a sweep of the reference recomputes for every voxel is maintained incrementally here, bit-exactly:
//   * S[v] (the smoothed histogram) only changes when the histogram of a voxel in v's neighbourhood changes.  A voxel
//     whose points moved PUSHES the difference to the voxels that list it (reverse neighbourhood rows, built once):
//     integer adds, so the order is irrelevant.  Late sweeps change a few dozen histograms, not 70 K rows of 88 gathers.
//     S is double-buffered: a sweep reads rec[cur] and pushes into rec[nxt], which the sweep's first kernel prepared as a
//     copy of rec[cur] -- nobody reads a record that is being pushed into.
//   * re-scoring a voxel is a pure function of S[v] (normals and weight are static): a voxel whose S has not changed
//     since it was last re-scored keeps its labels, so only its edge class / ppi are refreshed (epochs, below).
//   * the INDIRECT-edge closure (the one sequential coupling of a sweep) is a monotone fixed point, so it needs no
//     level-by-level order: closureKernel walks it chip-wide, depth first from the voxels active at sweep start, and lists
//     every voxel the sweep has to touch (active + marked) exactly once.
// Two launches per sweep: closureKernel, sweepKernel (both chip-wide; the second over the work list).
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

// The INDIRECT-edge closure of a sweep: ONE chip-wide kernel, no levels.  edge / ppi / S are frozen until sweepKernel, so
// "u marks v" (v a NO_EDGE voxel of u's DEV row whose ppi differs from arg S[u]) is a static relation, and the voxels a sweep
// activates are the least fixed point of
//      activated( v )  <=  some u < v marks v, u active at sweep start or activated
// (the reference's in-order loop reaches v after u and finds it INDIRECT: PCCPatchSegmenter.cpp:1510-1560).  A union of
// monotone bit sets: the order in which the marks are found is irrelevant, so nothing has to wait for a level to be
// complete.  A workgroup takes a run of voxels and puts the ones active at sweep start into a ring in LDS; its 32-lane
// groups take voxels from the ring, one hop each: read the voxel's DEV row one entry per lane (padded to `stride` entries:
// 32 or 128), mark, and hand the voxels the hop is the FIRST to activate back to the ring -- except one, which the group
// keeps and hops on from at once (a chain costs no ring traffic, a fan-out spreads over the workgroup's groups).
//   state[v >> 4]: two bits per voxel (marked | activated).  One returning atomicOr per candidate answers both "first to
//                  mark" (-> v is appended to the work list, exactly once) and "first to activate" (-> v is walked, once).
//   work list:     kSubLists sub-lists with their counters 128 bytes apart (a reservation on ONE word costs 11 ns and they
//                  queue up: tools/gpu/atomic_rate.hip; spread over cache lines they are free).  The voxels active at
//                  sweep start carry kWorkActive; for the others sweepKernel reads the activated bit.
//   LDS ring:      head / tail only grow and never pass ringCap: a position is handed out once per sweep (what does not fit
//                  spills); a slot holds kNoVoxel until its giver stores, so a taker that won the slot waits for that store.
//                  `pending` = voxels given and not yet hopped on: a group retires when it finds nothing and pending == 0.
//   spill ring:    what does not fit the LDS ring goes to a ring in global memory (same protocol); a workgroup that spilled
//                  looks there whenever its own ring is empty and does not retire before it has seen the global ring empty
//                  -- whatever it put there is claimed by somebody then, so the LDS ring's size bounds nothing.
//   Nobody waits inside a divergent branch for anything a sibling half-wave has yet to do: taking is one attempt per loop
//   iteration, and a giver stores right after its reservation.
//   state and the counters are double-buffered by sweep parity: this kernel zeroes the pair the next sweep uses.
// Also rec[nxt][u] = rec[cur][u] for every voxel (the copy this sweep's pushes go into).
constexpr int      kSubLists = 32;
constexpr uint32_t kNoVoxel  = 0xFFFFFFFFu;
struct Closure {
  const uint8_t*  edge;
  const uint8_t*  ppi;
  const uint4*    rec;
  const uint32_t* dev;
  const uint32_t* devLen;
  uint32_t        stride;
  uint32_t*       state;
  uint32_t*       list;   // this workgroup's sub-list
  uint32_t*       count;  // ... and its counter
  uint32_t*       spill;  // global ring of spillCap voxels, kNoVoxel where empty
  uint32_t        spillCap;
  uint32_t*       ctl;  // [0] global ring head, [32] tail (both only ever grow)
  volatile uint32_t* ring;  // LDS ring of ringCap voxels
  uint32_t           ringCap;
  uint32_t*          wg;  // LDS: [0] head, [1] tail, [2] freed, [3] pending, [4] spilled
};
enum { kHead = 0, kTail = 1, kFreed = 2, kPending = 3, kSpilled = 4 };

// (one lane) next voxel of the global ring, kNoVoxel when everything handed to it so far has been claimed
__device__ __forceinline__ uint32_t spillPop( const Closure& c ) {
  while ( true ) {
    const uint32_t h = __hip_atomic_load( &c.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    const uint32_t t = __hip_atomic_load( &c.ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if ( int32_t( t - h ) <= 0 ) return kNoVoxel;
    if ( atomicCAS( &c.ctl[0], h, h + 1u ) != h ) continue;
    uint32_t* slot = &c.spill[h % c.spillCap];  // reserved by a giver that stores right after its reservation
    uint32_t  x;
    do { x = __hip_atomic_load( slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); } while ( x == kNoVoxel );
    __hip_atomic_store( slot, kNoVoxel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    return x;
  }
}

// (one lane) one attempt to take a voxel: the workgroup's ring, else -- if the workgroup spilled -- the global one
__device__ __forceinline__ uint32_t closureTake( const Closure& c ) {
  volatile uint32_t* wg = c.wg;
  const uint32_t     h = wg[kHead], t = wg[kTail];
  if ( h != t ) {
    if ( atomicCAS( &c.wg[kHead], h, h + 1u ) != h ) return kNoVoxel;  // (somebody else took it: pending != 0, we come back)
    volatile uint32_t* slot = &c.ring[h];  // (h < tail <= ringCap: positions are handed out once per sweep)
    uint32_t           x;
    do { x = *slot; } while ( x == kNoVoxel );  // (the giver stores right after its reservation)
    return x;
  }
  if ( !wg[kSpilled] ) return kNoVoxel;
  const uint32_t x = spillPop( c );
  if ( x != kNoVoxel ) atomicAdd( &c.wg[kPending], 1u );
  return x;
}

// One hop of the 32-lane group on voxel x; returns the activated voxel the group keeps (kNoVoxel: none).
__device__ __forceinline__ uint32_t closureHop( const Closure& c, uint32_t x, uint32_t lane, int half ) {
  const uint32_t below = ( 1u << lane ) - 1u;
  const uint4    r     = c.rec[x];
  const uint8_t  a     = uint8_t( argOfPacked( r.x, r.y, r.z ) );
  // (rows are padded with kDevPad to `stride` entries; a row of 32 is one load that waits for nothing, a row of 128 is
  // walked as far as its length says)
  const uint32_t len  = c.stride == 32 ? 32u : c.devLen[x];
  uint32_t       keep = kNoVoxel;
  // (round 6: a row of 128 is up to four chunks of 32 -- their entries, then edge / ppi of all of them, then all the returning
  //  atomics are issued together: three dependent round trips per hop whatever the row's length, not three per chunk.  The
  //  voxels of a row are distinct, so an atomic's answer does not depend on the other chunks'.)
  constexpr int kChunks = 4;  // (devStride <= 128)
  uint32_t      vAll[kChunks];
  bool          candAll[kChunks], firstAll[kChunks], freshAll[kChunks];
#pragma unroll
  for ( int k = 0; k < kChunks; ++k ) vAll[k] = uint32_t( 32 * k ) < len ? c.dev[size_t( x ) * c.stride + 32 * k + lane] : kDevPad;
#pragma unroll
  for ( int k = 0; k < kChunks; ++k ) candAll[k] = vAll[k] != kDevPad && c.edge[vAll[k]] == NO_EDGE && c.ppi[vAll[k]] != a;
#pragma unroll
  for ( int k = 0; k < kChunks; ++k ) {
    firstAll[k] = freshAll[k] = false;
    if ( candAll[k] ) {
      const uint32_t v   = vAll[k];
      const uint32_t sh  = ( v & 15u ) * 2u;
      const uint32_t old = atomicOr( &c.state[v >> 4], ( v > x ? 3u : 1u ) << sh ) >> sh;
      firstAll[k]        = !( old & 1u );
      freshAll[k]        = v > x && !( old & 2u );  // (v is a NO_EDGE voxel: only a mark activates it)
    }
  }
#pragma unroll
  for ( int ch = 0; ch < kChunks; ++ch ) {
    if ( uint32_t( 32 * ch ) >= len ) continue;  // (uniform over the 32-lane group)
    const uint32_t v     = vAll[ch];
    const bool     first = firstAll[ch];
    bool           fresh = freshAll[ch];
    const uint32_t mFirst = uint32_t( __ballot( first ) >> ( 32 * half ) );
    uint32_t       mFresh = uint32_t( __ballot( fresh ) >> ( 32 * half ) );
    if ( mFirst ) {
      const int leader = __ffs( int( mFirst ) ) - 1;
      uint32_t  at     = 0;
      if ( int( lane ) == leader ) at = atomicAdd( c.count, uint32_t( __popc( mFirst ) ) );
      at = __shfl( at, leader, 32 );
      if ( first ) c.list[at + __popc( mFirst & below )] = v;
    }
    if ( !mFresh ) continue;
    const int leader = __ffs( int( mFresh ) ) - 1;
    if ( keep == kNoVoxel ) {  // the first one stays with the group (it is pending like the others)
      keep = __shfl( v, leader, 32 );
      if ( int( lane ) == leader ) atomicAdd( &c.wg[kPending], 1u );
      mFresh &= mFresh - 1u;
      if ( int( lane ) == leader ) fresh = false;
      if ( !mFresh ) continue;
    }
    const uint32_t k  = uint32_t( __popc( mFresh ) );
    uint32_t       at = kNoVoxel;  // where the k voxels go in the workgroup's ring (kNoVoxel: no room, the global ring)
    if ( int( lane ) == __ffs( int( mFresh ) ) - 1 ) {
      volatile uint32_t* wg = c.wg;
      atomicAdd( &c.wg[kPending], k );
      // The LDS "ring" is used ONCE around per sweep: a position is never handed out twice, so a giver can never meet a slot
      // whose previous taker has claimed it and not yet read it.  (Rounds 2-3 reused the slots the takers had handed back,
      // counting them in `freed` -- but not WHICH ones; a SOLID cloud, where one seed's activations cascade through a single
      // workgroup, wraps the ring and could lose a voxel or spin on its slot for ever.  Letting the giver wait for the slot's
      // sentinel instead dead-locks: the two 32-lane groups of a wavefront leave a spin loop together, so a taker's clear can
      // be held up by its sibling's wait for a store that is itself held up behind a sibling giver's wait -- seen as a GPU hang
      // under 16 voxels-of-2 frames in flight.)  What does not fit goes through the ring in global memory, which holds every
      // voxel of the grid and so cannot wrap within a sweep either.
      while ( true ) {
        const uint32_t t = wg[kTail];
        if ( t + k > c.ringCap ) break;
        if ( atomicCAS( &c.wg[kTail], t, t + k ) == t ) {
          at = t;
          break;
        }
      }
      if ( at == kNoVoxel ) {
        atomicSub( &c.wg[kPending], k );
        wg[kSpilled] = 1u;
      }
    }
    at = __shfl( at, __ffs( int( mFresh ) ) - 1, 32 );
    if ( fresh ) {
      if ( at != kNoVoxel ) {
        c.ring[at + __popc( mFresh & below )] = v;  // (at + k <= ringCap: a slot nobody has used in this sweep)
      } else {
        const uint32_t idx = atomicAdd( &c.ctl[32], 1u );
        __hip_atomic_store( &c.spill[idx % c.spillCap], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      }
    }
  }
  return keep;
}

__global__ __launch_bounds__( 1024 ) void closureKernel( const uint8_t* __restrict__ edge, const uint8_t* __restrict__ ppi,
                                                          const uint4* __restrict__ recCur, uint4* __restrict__ recNxt,
                                                          const uint32_t* __restrict__ dev, const uint32_t* __restrict__ devLen,
                                                          uint32_t stride, uint32_t V, uint32_t run, uint32_t* __restrict__ state,
                                                          uint32_t* __restrict__ stateNext, uint32_t stateWords,
                                                          uint32_t* __restrict__ lists, uint32_t listCap,
                                                          uint32_t* __restrict__ counts, uint32_t* __restrict__ countsNext,
                                                          uint32_t* __restrict__ spill, uint32_t* __restrict__ ctl,
                                                          uint32_t ringCap, unsigned long long* __restrict__ timing ) {
  const unsigned long long   tStart = timing ? wall_clock64() : 0ull;
  extern __shared__ uint32_t lds[];  // [0, run): the run's voxels active at sweep start; then the ring
  __shared__ uint32_t        wg[8], listBase;
  uint32_t*                  ring = lds + run;
  for ( uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < stateWords; w += gridDim.x * blockDim.x ) stateNext[w] = 0;
  if ( blockIdx.x == 0 && threadIdx.x < kSubLists ) countsNext[threadIdx.x * 32] = 0;
  if ( threadIdx.x < 8 ) wg[threadIdx.x] = 0;
  for ( uint32_t i = threadIdx.x; i < ringCap; i += blockDim.x ) ring[i] = kNoVoxel;
  __syncthreads();
  const uint32_t u0 = blockIdx.x * run, u1 = min( V, u0 + run );
  for ( uint32_t u = u0 + threadIdx.x; u < u1; u += blockDim.x ) {
    recNxt[u] = recCur[u];
    if ( edge[u] != NO_EDGE ) {
      const uint32_t i = atomicAdd( &wg[kTail], 1u );
      lds[i] = u, ring[i] = u;  // (ringCap >= run)
    }
  }
  __syncthreads();
  const uint32_t n   = wg[kTail];
  const uint32_t sub = blockIdx.x % kSubLists;
  uint32_t       at  = 0;
  if ( threadIdx.x == 0 ) {
    wg[kPending] = n;
    if ( n ) at = atomicAdd( &counts[sub * 32], n );  // (the answer is not needed before the walk is done)
  }
  __syncthreads();
  Closure c;
  c.edge = edge, c.ppi = ppi, c.rec = recCur, c.dev = dev, c.devLen = devLen, c.stride = stride, c.state = state;
  c.list = lists + size_t( sub ) * listCap, c.count = &counts[sub * 32], c.spill = spill, c.spillCap = listCap, c.ctl = ctl;
  c.ring = ring, c.ringCap = ringCap, c.wg = wg;
  const uint32_t lane = threadIdx.x & 31;
  const int      half = int( ( threadIdx.x >> 5 ) & 1 );
  // (test hook: timing[0] = max ticks from kernel start to the walk, [1] = max ticks in the walk, [2] = most hops of a
  //  group, [3] += hops, [4] += workgroups that spilled)
  const unsigned long long t0   = timing ? wall_clock64() : 0ull;
  uint32_t                 hops = 0, x = kNoVoxel;
  while ( true ) {
    if ( x == kNoVoxel ) {
      const uint32_t pending = reinterpret_cast<volatile uint32_t*>( wg )[kPending];  // (read BEFORE the attempt)
      if ( lane == 0 ) x = closureTake( c );
      x = __shfl( x, 0, 32 );
      if ( x == kNoVoxel ) {
        if ( pending == 0 ) break;  // nothing in the ring, nobody hopping: nothing can turn up any more
        __builtin_amdgcn_s_sleep( 2 );  // (longer naps -- 4 x, 16 x -- measured: nothing, profiles/r06_knobs_in_flight.txt)
        continue;
      }
    }
    const uint32_t next = closureHop( c, x, lane, half );
    ++hops;
    if ( lane == 0 ) atomicSub( &wg[kPending], 1u );
    x = next;
  }
  if ( timing && lane == 0 ) {
    atomicMax( &timing[0], t0 - tStart );
    atomicMax( &timing[1], wall_clock64() - t0 );
    atomicMax( &timing[2], (unsigned long long)hops );
    atomicAdd( &timing[3], (unsigned long long)hops );
    if ( threadIdx.x == 0 && wg[kSpilled] ) atomicAdd( &timing[4], 1ull );
  }
  if ( threadIdx.x == 0 ) listBase = at;
  __syncthreads();
  for ( uint32_t i = threadIdx.x; i < n; i += blockDim.x ) c.list[listBase + i] = lds[i] | kWorkActive;
}


// The rest of a sweep for the voxels of the work list, 16 lanes each: decide, re-score if S changed since the labels were
// computed, push the histogram difference to the reverse row, refresh edge class / ppi.
__global__ __launch_bounds__( 256 ) void sweepKernel( const uint32_t* __restrict__ lists, uint32_t listCap,
                                                       const uint32_t* __restrict__ counts, const uint32_t* __restrict__ state,
                                                       const uint4* __restrict__ recCur, uint4* __restrict__ recNxt,
                                                       uint32_t* __restrict__ lastRescore, const double* __restrict__ weight,
                                                       const uint32_t* __restrict__ pointStart,
                                                       const uint32_t* __restrict__ pointList,
                                                       const double* __restrict__ normals, const uint32_t* __restrict__ roff,
                                                       const uint32_t* __restrict__ rlen /* null: roff is a CSR of V + 1 offsets */,
                                                       const uint32_t* __restrict__ radj, uint8_t* __restrict__ edge,
                                                       uint8_t* __restrict__ ppi, uint4* __restrict__ hist,
                                                       uint8_t* __restrict__ partition, uint32_t* __restrict__ flags, int iter, int pairedPush ) {
  // the sub-lists as one index space: pre[s] = entries of the sub-lists before s
  __shared__ uint32_t pre[kSubLists + 1];
  if ( threadIdx.x < 64 ) {
    uint32_t inc = threadIdx.x < kSubLists ? counts[threadIdx.x * 32] : 0u;
#pragma unroll
    for ( int off = 1; off < kSubLists; off <<= 1 ) {
      const uint32_t t = __shfl_up( inc, off, 64 );
      if ( int( threadIdx.x ) >= off ) inc += t;
    }
    if ( threadIdx.x < kSubLists ) pre[threadIdx.x + 1] = inc;
    if ( threadIdx.x == 0 ) pre[0] = 0;
  }
  __syncthreads();
  const int      sub    = threadIdx.x & 15;
  const uint32_t groups = gridDim.x * 16, count = pre[kSubLists];
  for ( uint32_t idx = blockIdx.x * 16 + ( threadIdx.x >> 4 ); idx < count; idx += groups ) {  // (uniform over the 16 lanes)
    uint32_t sl = 0;  // the last sub-list that starts at or before idx
#pragma unroll
    for ( int step = kSubLists / 2; step > 0; step >>= 1 )
      if ( pre[sl + step] <= idx ) sl += step;
    const uint32_t raw = lists[size_t( sl ) * listCap + ( idx - pre[sl] )], v = raw & ~kWorkActive;
    // active at sweep start (flagged by closureKernel), or activated by the closure
    const uint32_t entry = v | ( ( raw & kWorkActive ) || ( ( state[v >> 4] >> ( ( v & 15u ) * 2u + 1u ) ) & 1u ) ? kWorkActive : 0u );
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
          const uint32_t rb = roff[v], re = rlen ? rb + rlen[v] : roff[v + 1];
          // (round 6: the first two words in ONE 64-bit add.  A word's difference is the true integer T = dHi * 65536 + dLo modulo
          //  2^32, and with fields below 2^15 -- pairedPush: maxNN + 255 < 32 768 -- |T| < 2^31, so T is d as an int32; adding
          //  T0 + T1 * 2^32 as a SIGNED 64-bit number to x + y * 2^32 is exact integer arithmetic on the pair: every true word stays
          //  within 0 .. 2^32 - 1, so the final pair decomposes into the same two words whatever the order of the adds -- a borrow
          //  out of the low word is what the sign extension cancels.  Two atomics per target instead of three.)
          const unsigned long long d01 = (unsigned long long)( (long long)int32_t( d0 ) + ( (long long)int32_t( d1 ) << 32 ) );
          for ( uint32_t t = rb + sub; t < re; t += 16 ) {
            uint32_t* tr = reinterpret_cast<uint32_t*>( recNxt + radj[t] );
            if ( pairedPush ) {
              if ( d0 | d1 ) atomicAdd( reinterpret_cast<unsigned long long*>( tr ), d01 );
            } else {
              if ( d0 ) atomicAdd( tr, d0 );
              if ( d1 ) atomicAdd( tr + 1, d1 );
            }
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
  std::vector<int> offsets;  // the ball: its ROWS (byRows) or its cells
  uint32_t*        table = nullptr;
  uint2*           bits  = nullptr;  // occupancy bitmap of the key table (.x; kept all-zero between frames) + ranks (.y)
  bool             tableFilled = false, byRows = true;
  int              capTier = 2;  // which instantiation of the neighbourhood kernels (launchNeighbourhood)
  size_t           Vp = 0, W2 = 0, ball = 0, perVoxel = 0;
  uint64_t         capacity = 0;
  uint32_t         res[5] = {0, 0, 0, 0, 0};  // the neighbourhood pass' answer: row entries written, overflow flag, reverse row entries, room asked for, kept hits out of room
  uint32_t         hitRegionCap = 0;
  DevBuf<uint32_t> d_key, d_flag, d_vid, d_small, d_count, d_rowLen, d_devLen, d_adjOff, d_hist, d_activeBuf, d_pointStart,
      d_pointList, d_cursor, d_rcount, d_rcursor, d_lastRescore, d_flags, d_gbits, d_adj, d_dev, d_lastKey, d_roffG, d_rlenG, d_radjG,
      d_voxelOfRank, d_hitBuf, d_hitCtl, d_hitOff, d_hitLen, d_rowCtl;
  std::vector<uint32_t> rowCtlHost;  // the forward rows' region cursors, fetched with the pass' answer (their sum: the row entries)
  DevBuf<Pt>      d_centre;
  DevBuf<double>  d_weight;
  DevBuf<uint8_t> d_state;  // edge | ppi, V bytes each (padded to whole 32-voxel words)
  const int*      d_offsets = nullptr;  // the ball's rows / cells: the context's table for ( radius, form )
  bool matches( int nn, double l, int it, int vd, int sr ) const {
    return nn == maxNNCount && l == lambda && it == iterationCount && vd == voxDim && sr == searchRadius;
  }
  // entries of a row table with room for `perVoxel` entries per voxel IN EVERY REGION (rowRegionOf: sixteen consecutive voxels
  // share a region, so a region holds at most 16 * ceil( V / 16 / kRowRegions ) voxels -- V / kRowRegions for all but tiny clouds)
  uint64_t rowTableEntries( size_t perVoxel ) const {
    const uint64_t groups = ( uint64_t( V ) + 15 ) / 16, perRegion = 16 * ( ( groups + kRowRegions - 1 ) / kRowRegions );
    return uint64_t( kRowRegions ) * perRegion * perVoxel;
  }
  int  geometry( tmc2_frame* f );
  int  finish();
  void launchNeighbourhood();
  ~RefineJob();
};

RefineJob::~RefineJob() {
  if ( tableFilled && bits && d_key.p ) {  // dropped between the halves: hand the context's bitmap (and table) back empty
    ApiScope scope( ctx );
    hipLaunchKernelGGL( tableCleanKernel, dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, s, d_key.p, n, table, bits );
    (void)hipStreamSynchronize( s );  // (the buffers go back to the pool when the members are destroyed)
  }
}

// The instantiations of the neighbourhood kernels: keys of LDS room per wavefront (160 of them the kernel's own), wavefronts
// per workgroup.  32 KB / 20 KB / 24 KB / 32 KB / 32 KB per workgroup -> 8 / 8 / 6 / 5 / 2.5 wavefronts per SIMD.  Round 6: the
// tiers of 1280 and 1536 keys -- voxels of 2 on a surface (472 occupied cells of the ball's 3 911 on average, ~ 1 080 at most)
// ran in the 2048 tier at 5 wavefronts per SIMD, and the pass is bound by the latency of its dependent loads.
constexpr int kCapTiers[5] = {1024, 1280, 1536, 2048, 4096};
constexpr int kLastCapTier = 4;
inline int capTierFor( size_t cells ) {  // the smallest tier whose room holds `cells` keys
  int t = 0;
  while ( t < kLastCapTier && size_t( kCapTiers[t] - 160 ) < cells ) ++t;
  return t;
}

// forward rows and -- on the row-wise path -- the reverse rows behind them (d_small: [1] row cursor, [2] overflow, [3] reverse
// row cursor, [4] with overflow code 3: the room the fullest ball asks for)
void RefineJob::launchNeighbourhood() {
  const int nBall = int( offsets.size() );  // rows or cells
  const uint32_t rowRegionCap = uint32_t( capacity / kRowRegions );  // (the tables hold kRowRegions regions of this many entries)
  BallHits  savedHits{};
  if ( byRows && d_hitBuf.p ) {
    savedHits.buf = d_hitBuf.p, savedHits.regionCap = hitRegionCap, savedHits.cursor = d_hitCtl.p, savedHits.off = d_hitOff.p, savedHits.len = d_hitLen.p;
  }
#define TMC2_NEIGHBOURHOOD( CAP, WAVES )                                                                                       \
  hipLaunchKernelGGL( ( neighbourhoodKernel<CAP, WAVES> ), dim3( ( V + WAVES - 1 ) / WAVES ), dim3( 64 * WAVES ), 0, s, d_centre.p,      \
                      d_count.p, table, g, V, d_offsets, nBall, maxNNCount, lambda, idBits, devRange, devStride,             \
                      rowRegionCap, d_rowLen.p, d_devLen.p, d_weight.p, d_adjOff.p, d_adj.p, d_dev.p, d_rowCtl.p,            \
                      d_small.p + 2, byRows ? bits : (const uint2*)nullptr, reinterpret_cast<const uint2*>( d_voxelOfRank.p ),   \
                      byRows ? d_lastKey.p : (uint32_t*)nullptr, savedHits );                                                  \
  if ( byRows )                                                                                                                \
  hipLaunchKernelGGL( ( reverseRowsKernel<CAP, WAVES> ), dim3( ( V + WAVES - 1 ) / WAVES ), dim3( 64 * WAVES ), 0, s, d_centre.p, reinterpret_cast<const uint2*>( d_voxelOfRank.p ), \
                      bits, g, V, d_offsets, nBall, idBits, d_lastKey.p, rowRegionCap, d_roffG.p, d_rlenG.p, d_radjG.p,          \
                      d_rowCtl.p + kRowRegions * 32, d_small.p + 2, savedHits )
  // LDS per wavefront = room for the ball's OCCUPIED cells (row-wise form; a surface fills 5-10 % of a ball) or for all its
  // cells; the smaller the room, the more wavefronts a CU holds (kCapTiers) -- a frame whose balls need more raises the overflow
  // word to 3, says how much, and is repeated in the tier that holds it (remembered per context)
  switch ( capTier ) {
    case 0: TMC2_NEIGHBOURHOOD( 1024, 8 ); break;
    case 1: TMC2_NEIGHBOURHOOD( 1280, 4 ); break;
    case 2: TMC2_NEIGHBOURHOOD( 1536, 4 ); break;
    case 3: TMC2_NEIGHBOURHOOD( 2048, 4 ); break;
    default: TMC2_NEIGHBOURHOOD( 4096, 2 ); break;
  }
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
  // (test hook TMC2_REFINE_NEIGHBOURHOOD=cells: the rounds 1-3 form -- one table look-up per cell of the ball, reverse rows by
  //  scattering the forward rows -- kept as the cross-check of the row-wise form)
  const char* nbEnv = ctxOption( ctx, "REFINE_NEIGHBOURHOOD" );
  byRows            = !( nbEnv && nbEnv[0] == 'c' );
  {
    int R = 0;
    while ( R * R < r2 ) ++R;
    ball = 0;
    for ( int dz = -R; dz <= R; ++dz )
      for ( int dy = -R; dy <= R; ++dy ) {
        int xr = -1;
        for ( int dx = -R; dx <= R; ++dx )
          if ( dx * dx + dy * dy + dz * dz < r2 ) {
            ++ball;
            xr = std::max( xr, dx );
            if ( !byRows ) offsets.push_back( ( dx + 128 ) | ( ( dy + 128 ) << 8 ) | ( ( dz + 128 ) << 16 ) );
          }
        if ( byRows && xr >= 0 ) offsets.push_back( ( dy + 128 ) | ( ( dz + 128 ) << 8 ) | ( xr << 16 ) );
      }
  }
  if ( ball > 4096 - 160 || r2 > 128 ) {  // (the neighbourhood kernel keeps 160 words of its 2048 / 4096 for its own use)  // (search radius 192: r2 = 48 with voxels of 4, 96 with voxels of 2 -> 3 911 cells)
    setError( "refineSegmentationGridBased: search radius %d too large for the LDS neighbourhood tile", searchRadius );
    return TMC2_E_UNSUPPORTED;
  }
  devRange  = voxDim >= 4 ? 1 : 2;  // PCCPatchSegmenter.cpp:1471
  devStride = devRange == 1 ? 32u : 128u;
  idBits    = r2 <= 64 ? 26 : 25;   // neighbourhood sort key: d2 above, voxel id below
  const int sidSetup = ctx->stageBegin( "refine_setup" );
  table = nullptr;
  if ( !byRows ) {  // the dense key table (2^(3s+1) words: 1 GiB with voxels of 2 or 11-bit geometry) only for the cell-by-cell form
    if ( ctx->gridTable.count < g.tableSize ) {
      TMC2_TRY( ctx->gridTable.alloc( g.tableSize ) );
      TMC2_HIP( hipMemsetAsync( ctx->gridTable.p, 0xFF, size_t( g.tableSize ) * 4, s ) );
    }
    table = ctx->gridTable.p;
  }
  {
    const size_t bitWords = size_t( g.tableSize ) / 32 + 2;  // (+ spare words: a row's second word may lie behind the last key)
    if ( ctx->gridBits.count < bitWords ) {
      TMC2_TRY( ctx->gridBits.alloc( bitWords ) );
      TMC2_HIP( hipMemsetAsync( ctx->gridBits.p, 0, bitWords * sizeof( uint2 ), s ) );
    }
    bits = ctx->gridBits.p;
  }
  TMC2_TRY( d_key.alloc( n ) );
  TMC2_TRY( d_flag.alloc( n ) );
  TMC2_TRY( d_vid.alloc( n ) );
  TMC2_TRY( d_small.alloc( 16 ) );  // [0] voxel count, [1] adjacency size, [2] closure flag
  const dim3 blk( 256 ), grdN( ( n + 255 ) / 256 );
  tableFilled = true;
  hipLaunchKernelGGL( voxelKeyKernel, grdN, blk, 0, s, f->d_pts.p, n, g, d_key.p, table, bits );
  DevBuf<uint32_t> d_firstPoint;
  if ( byRows ) {
    // ranks of the occupied keys (the words' prefix sums), then the first point of every rank: the voxels are numbered through
    // them -- and the row-wise neighbourhood passes look voxels up by them
    const uint32_t   words  = g.tableSize / 32 + 2;
    const uint32_t   blocks = ( words + kBitsBlock - 1 ) / kBitsBlock;
    DevBuf<uint32_t> d_blockTotal, d_blockBase;
    TMC2_TRY( d_blockTotal.alloc( blocks ) );
    TMC2_TRY( d_blockBase.alloc( blocks ) );
    TMC2_TRY( d_firstPoint.alloc( n ) );  // (one per occupied key: at most one per point)
    TMC2_HIP( hipMemsetAsync( d_firstPoint.p, 0xFF, size_t( n ) * 4, s ) );
    hipLaunchKernelGGL( bitsBlockCountKernel, dim3( blocks ), blk, 0, s, bits, words, d_blockTotal.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_blockTotal.p, d_blockBase.p, blocks, nullptr ) );
    hipLaunchKernelGGL( bitsPrefixKernel, dim3( blocks ), blk, 0, s, bits, words, d_blockBase.p );
    hipLaunchKernelGGL( firstPointKernel, grdN, blk, 0, s, d_key.p, n, bits, d_firstPoint.p );
  }
  hipLaunchKernelGGL( firstFlagKernel, grdN, blk, 0, s, d_key.p, table, bits, d_firstPoint.p, n, d_flag.p );
  DevBuf<uint32_t> d_rank;
  TMC2_TRY( d_rank.alloc( n ) );
  volatile uint32_t* answer = ctx->answerLine( tmc2_ctx::kAnswerRefineVoxels );  // (the voxel count straight to a page-locked word: no copy)
  TMC2_TRY( exclusiveScanU32( ctx, d_flag.p, d_rank.p, n, d_small.p, ScanAnswer{answer, nullptr, 0} ) );
  TMC2_HIP( hipStreamSynchronize( s ) );
  V = answer[0];
  TMC2_TRY( d_count.alloc( size_t( V ) + 1 ) );  // (+1: scanned into the point-list offsets)
  TMC2_TRY( d_rowLen.alloc( V ) );
  TMC2_TRY( d_devLen.alloc( V ) );
  TMC2_TRY( d_adjOff.alloc( V + 1 ) );
  TMC2_TRY( d_hist.alloc( size_t( V ) * 4 ) );
  TMC2_TRY( d_centre.alloc( V ) );
  TMC2_TRY( d_weight.alloc( V ) );
  Vp = ( size_t( V ) + 63 ) & ~size_t( 63 );  // sub-arrays of the state block: whole, aligned 32-voxel words
  TMC2_TRY( d_state.alloc( Vp * 2 ) );
  TMC2_TRY( d_activeBuf.alloc( V ) );
  // every buffer of this stage that starts from zeros, in one launch (the event-driven loop's among them)
  W = ( V + 31 ) / 32;
  W2          = ( size_t( V ) + 15 ) / 16;  // closure state: two bits per voxel
  // closure scratch: state bitmaps of the two sweep parities | list counters of the two parities (one per 128 bytes) | ring
  // head, tail (64 words) | the spill ring (V voxels)
  const size_t closureZeroWords = 2 * W2 + 2 * size_t( kSubLists ) * 32 + 64;
  TMC2_TRY( d_pointStart.alloc( size_t( V ) + 1 ) );
  TMC2_TRY( d_pointList.alloc( n ) );
  TMC2_TRY( d_cursor.alloc( V ) );
  TMC2_TRY( d_flags.alloc( 2 * size_t( iterationCount ) + 2 ) );
  TMC2_TRY( d_rcount.alloc( size_t( V ) + 1 ) );
  TMC2_TRY( d_rcursor.alloc( size_t( V ) + 1 ) );
  TMC2_TRY( d_lastRescore.alloc( V ) );
  TMC2_TRY( d_gbits.alloc( closureZeroWords + V ) );
  TMC2_TRY( fillRegions( ctx, {{d_count.p, ( size_t( V ) + 1 ) * 4, 0},
                               {d_hist.p, size_t( V ) * 16, 0},
                               {d_state.p, Vp * 2, 0},
                               {d_cursor.p, size_t( V ) * 4, 0},
                               {d_small.p + 1, 20, 0},  // [1] row cursor, [2] overflow, [3] reverse row cursor, [4] room asked for, [5] kept hits out of room
                               {d_flags.p, ( 2 * size_t( iterationCount ) + 2 ) * 4, 0},
                               {d_rcount.p, ( size_t( V ) + 1 ) * 4, 0},
                               {d_rcursor.p, ( size_t( V ) + 1 ) * 4, 0},
                               {d_lastRescore.p, size_t( V ) * 4, 0},
                               {d_gbits.p, closureZeroWords * 4, 0},
                               {d_gbits.p + closureZeroWords, size_t( V ) * 4, 0xFF}} ) );  // kNoVoxel
  d_offsets = ctx->constTable( ( uint64_t( 0x5335 ) << 32 ) | ( uint64_t( r2 ) << 1 ) | ( byRows ? 1u : 0u ), offsets );
  if ( !d_offsets ) return TMC2_E_HIP;
  hipLaunchKernelGGL( assignVoxelKernel, grdN, blk, 0, s, f->d_pts.p, d_key.p, table, bits, d_firstPoint.p, d_rank.p, n, g, d_vid.p,
                      d_count.p, d_centre.p );
  if ( byRows ) {
    TMC2_TRY( d_voxelOfRank.alloc( 2 * size_t( V ) ) );  // (uint2 records)
    hipLaunchKernelGGL( rankToVoxelKernel, grdN, blk, 0, s, d_key.p, d_flag.p, d_vid.p, n, bits, reinterpret_cast<uint2*>( d_voxelOfRank.p ), d_count.p,
                        d_centre.p );
  } else {
    hipLaunchKernelGGL( tableToVoxelKernel, grdN, blk, 0, s, d_key.p, d_flag.p, d_vid.p, n, table );
  }
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
  const char* capEnv  = ctxOption( ctx, "REFINE_ROWCAP" );
  perVoxel            = std::min<size_t>( ball, 2 * size_t( maxNNCount > 0 ? maxNNCount : 1 ) * V / std::max<uint32_t>( n, 1u ) + 32 );
  if ( capEnv && capEnv[0] == 't' ) perVoxel = 1;
  capacity = rowTableEntries( perVoxel );
  if ( capacity > 0xFFFFFFFFull ) {
    setError( "refineSegmentationGridBased: %u voxels x %zu row entries exceed the neighbourhood table", V, perVoxel );
    return TMC2_E_UNSUPPORTED;
  }
  capTier = std::max( capTierFor( ball ), 3 );  // (the cell-by-cell form keeps every cell of the ball: the tiers of rounds 4-5)
  if ( byRows ) capTier = std::min( capTier, std::max( 0, ctx->refineCapTier ) );  // (test hook TMC2_REFINE_CAPTIER: start there)
  if ( const char* tierEnv = ctxOption( ctx, "REFINE_CAPTIER" ) ) capTier = std::min( kLastCapTier, std::max( byRows ? 0 : capTier, atoi( tierEnv ) ) );
  TMC2_TRY( d_adj.alloc( size_t( capacity ) ) );
  if ( byRows ) {  // the reverse rows hold the same entries as the forward rows: same room
    TMC2_TRY( d_lastKey.alloc( V ) );
    TMC2_TRY( d_roffG.alloc( V ) );
    TMC2_TRY( d_rlenG.alloc( V ) );
    TMC2_TRY( d_radjG.alloc( size_t( capacity ) ) );
    // the balls' hits, kept for the reverse rows pass (option REFINE_HITS=0: it collects the balls again, as rounds 4-5 did;
    // =tiny: a region runs out of room and says so -- the same).  Room: what this context's frames have needed, a surface's ~ 500
    // of the 3 911 cells of a ball of voxels of 2 to begin with.
    const char*  hitsEnv = ctxOption( ctx, "REFINE_HITS" );
    const size_t perHit  = hitsEnv && hitsEnv[0] == 't' ? 1 : std::min<size_t>( ball, size_t( ctx->refineHitsPerVoxel ) );
    if ( !( hitsEnv && hitsEnv[0] == '0' ) && uint64_t( V ) * perHit < 0xFFFFFFFFull ) {
      hitRegionCap = uint32_t( ( uint64_t( V ) * perHit + kHitRegions - 1 ) / kHitRegions );
      TMC2_TRY( d_hitBuf.alloc( size_t( hitRegionCap ) * kHitRegions ) );
      TMC2_TRY( d_hitCtl.alloc( kHitRegions * 32 ) );
      TMC2_TRY( d_hitOff.alloc( V ) );
      TMC2_TRY( d_hitLen.alloc( V ) );
      TMC2_TRY( fillRegions( ctx, {{d_hitCtl.p, kHitRegions * 32 * 4, 0}} ) );
    }
  }
  TMC2_TRY( d_rowCtl.alloc( 2 * kRowRegions * 32 ) );  // region cursors of the forward rows, then of the reverse rows
  TMC2_TRY( fillRegions( ctx, {{d_rowCtl.p, 2 * kRowRegions * 32 * 4, 0}} ) );
  rowCtlHost.assign( kRowRegions * 32, 0u );
  launchNeighbourhood();
  TMC2_HIP( hipMemcpyAsync( res, d_small.p + 1, 20, hipMemcpyDeviceToHost, s ) );  // (read in finish(), after a synchronisation)
  TMC2_HIP( hipMemcpyAsync( rowCtlHost.data(), d_rowCtl.p, kRowRegions * 32 * 4, hipMemcpyDeviceToHost, s ) );
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
  uint8_t *d_edge = d_state.p, *d_ppi = d_state.p + Vp;
  uint32_t* d_active = d_activeBuf.p;
  const dim3 grdV( ( V + 255 ) / 256 ), grdV16( ( V + 15 ) / 16 );  // 16 lanes per voxel
  for ( int attempt = 0;; ++attempt ) {
    TMC2_HIP( hipStreamSynchronize( s ) );  // (usually long done: the orientation walk ran in between)
    TMC2_HIP( hipGetLastError() );
    totalLen = 0;
    for ( uint32_t r = 0; r < kRowRegions; ++r ) totalLen += rowCtlHost[r * 32];
    if ( ctxOption( ctx, "REFINE_DEBUG" ) ) fprintf( stderr, "refine: neighbourhood attempt %d tier %d: %u row entries, overflow word %u (V = %u)\n", attempt, capTier, totalLen, res[1], V );
    if ( res[4] ) {  // the kept hits ran out of room (the reverse rows pass collected the balls itself): more for this context's next frames
      ctx->refineHitsPerVoxel = std::min<uint32_t>( 4096u - 160u, ctx->refineHitsPerVoxel * 7u / 4u );
      ctx->stageAddHostMs( "refine_hits_out_of_room", 0.0 );
    }
    if ( res[1] == 0 ) break;
    if ( ( res[1] != 1 && res[1] != 3 ) || attempt > 3 || ( res[1] == 3 && capTier >= kLastCapTier ) ) {
      setError( "refineSegmentationGridBased: neighbourhood pass failed (%u)", res[1] );
      return TMC2_E_HIP;
    }
    if ( res[1] == 3 ) {  // balls with more occupied cells than the instantiation holds: one tier up, for this context's next frames too
      capTier            = std::max( capTier + 1, capTierFor( res[3] ) );
      ctx->refineCapTier = capTier;
      ctx->stageAddHostMs( "refine_cap_tier_repeat", 0.0 );
    } else {
      perVoxel = ball;  // room for whole balls
    }
    capacity = rowTableEntries( perVoxel );
    if ( capacity > 0xFFFFFFFFull ) {
      setError( "refineSegmentationGridBased: %u voxels x %zu row entries exceed the neighbourhood table", V, perVoxel );
      return TMC2_E_UNSUPPORTED;
    }
    TMC2_TRY( d_adj.alloc( size_t( capacity ) ) );
    if ( byRows ) TMC2_TRY( d_radjG.alloc( size_t( capacity ) ) );
    TMC2_HIP( hipMemsetAsync( d_small.p + 1, 0, 20, s ) );  // [1] row cursor, [2] overflow, [3] reverse row cursor, [4] room asked for, [5] kept hits out of room
    if ( d_hitCtl.p ) TMC2_HIP( hipMemsetAsync( d_hitCtl.p, 0, kHitRegions * 32 * 4, s ) );
    TMC2_HIP( hipMemsetAsync( d_rowCtl.p, 0, 2 * kRowRegions * 32 * 4, s ) );
    launchNeighbourhood();
    TMC2_HIP( hipMemcpyAsync( res, d_small.p + 1, 20, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipMemcpyAsync( rowCtlHost.data(), d_rowCtl.p, kRowRegions * 32 * 4, hipMemcpyDeviceToHost, s ) );
  }
  d_hitBuf.release(), d_hitCtl.release(), d_hitOff.release(), d_hitLen.release();  // (both passes over the balls are done: the stream was synchronised above)
  hipLaunchKernelGGL( tableCleanKernel, grdN, blk, 0, s, d_key.p, n, table, bits );
  tableFilled = false;
  hipLaunchKernelGGL( histAccumulateKernel, grdN, blk, 0, s, d_vid.p, f->d_partition.p, (const uint8_t*)nullptr, n,
                      d_hist.p );
  hipLaunchKernelGGL( initVoxelStateKernel, grdV, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_count.p, V,
                      d_edge, d_ppi, d_active );
  // reverse rows (CSR), S records (double-buffered), epochs, closure scratch
  DevBuf<uint32_t> d_roff, d_radj, d_lists;
  DevBuf<uint4>    d_rec;
  TMC2_TRY( d_lists.alloc( size_t( kSubLists ) * V ) );  // (a voxel is listed once per sweep: any sub-list can hold them all)
  TMC2_TRY( d_rec.alloc( 2 * size_t( V ) ) );
  const uint32_t *revOff = d_roffG.p, *revLen = d_rlenG.p, *revAdj = d_radjG.p;  // gathered behind the forward rows
  if ( !byRows ) {  // the rounds 1-3 form: count, prefix sum, scatter
    TMC2_TRY( d_roff.alloc( size_t( V ) + 1 ) );
    TMC2_TRY( d_radj.alloc( std::max<size_t>( totalLen, 1 ) ) );
    hipLaunchKernelGGL( reverseCountKernel, grdV16, blk, 0, s, d_adjOff.p, d_rowLen.p, d_adj.p, V, d_rcount.p );
    TMC2_TRY( exclusiveScanU32( ctx, d_rcount.p, d_roff.p, size_t( V ) + 1, nullptr ) );
    hipLaunchKernelGGL( reverseFillKernel, grdV16, blk, 0, s, d_adjOff.p, d_rowLen.p, d_adj.p, V, d_roff.p, d_rcursor.p,
                        d_radj.p );
    revOff = d_roff.p, revLen = nullptr, revAdj = d_radj.p;
  }
  hipLaunchKernelGGL( smoothInitKernel, grdV16, blk, 0, s, reinterpret_cast<const uint4*>( d_hist.p ), d_adjOff.p,
                      d_rowLen.p, d_adj.p, V, d_rec.p );
  ctx->stageEnd( sidSetup );
  TMC2_HIP( hipGetLastError() );
  const int sidSweep = ctx->stageBegin( "refine_sweeps" );
  // closureKernel: a run of voxels per workgroup; LDS = the run's active voxels + the ring
  // (test hooks: TMC2_REFINE_CLOSURE_BLOCKS = its grid, TMC2_REFINE_CLOSURE_THREADS = its workgroup; TMC2_REFINE_RING = room
  // of the LDS ring beyond the run -- 1 sends nearly every fan-out through the spill ring)
  const char*    gridEnv    = ctxOption( ctx, "REFINE_CLOSURE_BLOCKS" );
  const char*    threadsEnv = ctxOption( ctx, "REFINE_CLOSURE_THREADS" );
  const char*    ringEnv    = ctxOption( ctx, "REFINE_RING" );
  // (round 6, sixteen in flight, this round's kernels around it -- profiles/r06_knobs_in_flight.txt: two workgroups per CU of 512 /
  //  256 / 128 threads -> 186.2-188.4 / 191.3-192.4 / 187.5 frames/s; one per CU of 256: 190.4; four of 256: 190.0 -- half the
  //  idle waves of rounds 3-5 for the many-in-flight regime, the walk's deepest chain is what a launch takes either way)
  const int      closureThreads = threadsEnv ? std::min( 1024, std::max( 64, atoi( threadsEnv ) & ~63 ) ) : ( refineOverlap( ctx ) ? 512 : 256 );
  const uint32_t perGroup   = 4;  // voxels of the run per 32-lane group
  const uint32_t wantGrid   = gridEnv ? uint32_t( std::max( 1, atoi( gridEnv ) ) )
                                      : std::min<uint32_t>( ( refineOverlap( ctx ) ? 4u : 2u ) * uint32_t( ctx->cuCount ),
                                                            ( V + perGroup * ( closureThreads / 32 ) - 1 ) / ( perGroup * ( closureThreads / 32 ) ) );
  // (two workgroups per CU: 8 % slower alone than four and 3 % more frames per second with sixteen frames in flight -- a
  // workgroup's groups idle through most of the walk, and idle waves are in the way of the other frames' kernels)
  const uint32_t run        = std::min<uint32_t>( 8192u, std::max<uint32_t>( 1u, ( V + wantGrid - 1 ) / wantGrid ) );
  const dim3     grdClosure( ( V + run - 1 ) / run );
  const uint32_t ringCap    = run + ( ringEnv ? uint32_t( std::min( 8192, std::max( 1, atoi( ringEnv ) ) ) ) : 1024u );
  const size_t   closureLds = 4 * ( size_t( run ) + ringCap );
  if ( closureLds > 48 * 1024 ) TMC2_TRY( allowLargeLds( reinterpret_cast<const void*>( closureKernel ), closureLds, ctx->device ) );
  // (test hook TMC2_REFINE_SWEEP_BLOCKS: the sweep kernel's grid)
  const char* sweepGridEnv = ctxOption( ctx, "REFINE_SWEEP_BLOCKS" );
  const dim3  grdSweep( uint32_t( std::min<size_t>( ( size_t( V ) + 15 ) / 16, sweepGridEnv ? size_t( std::max( 1, atoi( sweepGridEnv ) ) ) : size_t( refineOverlap( ctx ) ? 8 : 2 ) * ctx->cuCount ) ) );
  // (tmc2_set_refine_overlap( 1 ) = "few frames in flight": the chip has room, so both kernels of a sweep take the grids that
  //  are fastest with the GPU to themselves -- four closure workgroups and eight sweep workgroups per CU)
  // (round 4 sweep over the grids, 16 frames in flight / one sweep alone: sweep kernel 2 / 4 / 8 / 16 workgroups per CU ->
  //  loot 109.0 / 108.1 / 107.1 / 106.6 frames/s, 338 / 327 / 305 / 289 us; longdress 174.1 / 174.9 frames/s, 54.8 / 51.9 us;
  //  closure 1 / 2 / 4 / 8 per CU -> loot 109.3 / 109.0 / 107.8 / 107.1 frames/s, 434 / 338 / 308 / 291 us: all within
  //  +- 1.5 % of each other in flight -- four per CU for the sweep, two for the closure;
  //  round 6, three rounds alternating with the closure at two workgroups of 256 per CU: sweep kernel 4 / 2 / 1 per CU ->
  //  longdress 189.2 / 191.7 / 191.8 frames/s -- two per CU)
  // (the sweep's pushes: two words of a target in one 64-bit add where the histogram fields stay below 2^15 -- a row holds at most
  //  maxNN + 255 members; option REFINE_PUSH=words: three 32-bit adds, as rounds 2-5)
  const char* pushEnv    = ctxOption( ctx, "REFINE_PUSH" );
  const int   pairedPush = maxNNCount + 255 < 32768 && !( pushEnv && pushEnv[0] == 'w' ) ? 1 : 0;
  const bool wantTrace = ctxOption( ctx, "REFINE_TRACE" ) != nullptr;
  DevBuf<unsigned long long> d_timing;  // test hook TMC2_REFINE_TIMING: where the closure spends its time, per sweep
  const bool                 wantTiming = ctxOption( ctx, "REFINE_TIMING" ) != nullptr;
  if ( wantTiming ) {
    TMC2_TRY( d_timing.alloc( 8 * size_t( iterationCount ) ) );
    TMC2_HIP( hipMemsetAsync( d_timing.p, 0, 64 * size_t( iterationCount ), s ) );
  }
  // d_gbits: state bitmaps of the two sweep parities | list counters of the two parities (one per 128 bytes) | ring head,
  // tail | the spill ring
  uint32_t*  state[2]  = {d_gbits.p, d_gbits.p + W2};
  uint32_t*  counts[2] = {d_gbits.p + 2 * size_t( W2 ), d_gbits.p + 2 * size_t( W2 ) + kSubLists * 32};
  uint32_t*  ctl       = d_gbits.p + 2 * size_t( W2 ) + 2 * kSubLists * 32;
  uint32_t*  spill     = ctl + 64;
  // (Measured and dropped, twice: the 2 I launches of the loop as ONE hipGraph per frame -- round 5, option REFINE_GRAPH, removed in
  //  round 6 -- and as ONE kernel whose phases take their work by ticket -- round 6, profiles/r06_one_launch_sweeps.txt.  The loop has no
  //  host decision in it, but what sixteen frames in flight share is wave-slot time, not launches: DESIGN.md section 5.)
  const bool      debugSweeps = ctxOption( ctx, "REFINE_DEBUG" ) != nullptr;
  for ( int iter = 0; iter < iterationCount; ++iter ) {
    const int cur    = iter & 1, nxt = cur ^ 1;
    uint4 *   recCur = d_rec.p + size_t( cur ) * V, *recNxt = d_rec.p + size_t( nxt ) * V;
    hipLaunchKernelGGL( closureKernel, grdClosure, dim3( closureThreads ), closureLds, s, d_edge, d_ppi, recCur, recNxt, d_dev.p, d_devLen.p,
                        devStride, V, run, state[cur], state[nxt], uint32_t( W2 ), d_lists.p, V, counts[cur], counts[nxt], spill,
                        ctl, ringCap, wantTiming ? d_timing.p + 8 * size_t( iter ) : nullptr );
    if ( debugSweeps ) {
      const hipError_t e = hipStreamSynchronize( s );
      fprintf( stderr, "refine: sweep %d closure done (%d), %u workgroups of %d, run %u, ring %u\n", iter, int( e ), grdClosure.x, closureThreads, run, ringCap );
    }
    hipLaunchKernelGGL( sweepKernel, grdSweep, blk, 0, s, d_lists.p, V, counts[cur], state[cur], recCur, recNxt,
                        d_lastRescore.p, d_weight.p, d_pointStart.p, d_pointList.p, f->d_normals.p, revOff, revLen, revAdj, d_edge,
                        d_ppi, reinterpret_cast<uint4*>( d_hist.p ), f->d_partition.p, wantTrace ? d_flags.p : nullptr, iter, pairedPush );
    if ( debugSweeps ) {
      const hipError_t e = hipStreamSynchronize( s );
      fprintf( stderr, "refine: sweep %d sweep done (%d)\n", iter, int( e ) );
    }
  }
  ctx->stageEnd( sidSweep );
  TMC2_HIP( hipGetLastError() );
  ctx->stageAddHostMs( "refine_sweeps_executed", double( iterationCount ) );  // (counts, not milliseconds: what the
  ctx->stageAddHostMs( "refine_voxels", double( V ) );                        //  roofline of a sweep is quoted on,
  ctx->stageAddHostMs( "refine_row_entries", double( totalLen ) );            //  V and L = entries / V)
  if ( wantTiming ) {
    std::vector<unsigned long long> t( 8 * size_t( iterationCount ) );
    TMC2_HIP( hipMemcpyAsync( t.data(), d_timing.p, t.size() * 8, hipMemcpyDeviceToHost, s ) );
    TMC2_HIP( hipStreamSynchronize( s ) );
    fprintf( stderr, "refine closure (V = %u, %u workgroups, runs of %u), per sweep: us to the walk | us walking | most hops of a group | hops | spilling workgroups\n", V, grdClosure.x, run );
    for ( int m = 0; m < iterationCount; ++m )  // wall_clock64 ticks at 100 MHz
      fprintf( stderr, "  %2d: %5.1f %5.1f %4llu %6llu %3llu\n", m, t[8 * m] * 0.01, t[8 * m + 1] * 0.01, t[8 * m + 2], t[8 * m + 3], t[8 * m + 4] );
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
