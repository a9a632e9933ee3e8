// lex_order.h -- the stable (x, y, z) order of a cloud of 16-bit positions, by counting sort.
//
// PCCPointSet3::removeDuplicate / reorder (PccLibCommon/source/PCCPointSet.cpp:169-220) sort the points by position before
// they merge duplicates; the metric (S23) and the conformance checksum both start from that order.  Three stable counting
// passes (z, then y, then x) over the index array replace a comparison sort: 0.8 M points take a few ms instead of ~100.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace tmc2 {

// order[k] = index of the k-th point in (x, y, z) order (signed comparison per coordinate); equal positions keep input order
inline void lexOrderStable( const int16_t* xyz, size_t n, std::vector<uint32_t>& order ) {
  order.resize( n );
  for ( size_t i = 0; i < n; ++i ) order[i] = uint32_t( i );
  if ( n < 2 ) return;
  std::vector<uint32_t> other( n );
  std::vector<uint32_t> count;
  for ( int d = 2; d >= 0; --d ) {
    int lo = 32767, hi = -32768;
    for ( size_t i = 0; i < n; ++i ) {
      const int v = xyz[3 * i + size_t( d )];
      lo          = v < lo ? v : lo;
      hi          = v > hi ? v : hi;
    }
    if ( lo == hi ) continue;
    count.assign( size_t( hi - lo ) + 2, 0u );
    for ( size_t i = 0; i < n; ++i ) ++count[size_t( xyz[3 * i + size_t( d )] - lo ) + 1];
    for ( size_t b = 1; b < count.size(); ++b ) count[b] += count[b - 1];
    for ( size_t i = 0; i < n; ++i ) {
      const uint32_t p = order[i];
      other[count[size_t( xyz[3 * size_t( p ) + size_t( d )] - lo )]++] = p;
    }
    order.swap( other );
  }
}

}  // namespace tmc2
