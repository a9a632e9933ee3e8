// CPU check of csrc/ordered_sum.h (test infrastructure): the block form of an ordered fp64 sum, staged as the kernels of
// metrics.hip stage it -- approximate block sums, a guessed binade per block, the blocks' steps reduced in a tree, the chain over
// the blocks with the term-by-term fallback -- against the plain loop `acc += t[i]` (PCCMetrics.cpp:73-229).
#include <cstdint>
#include <cstring>
#include <vector>

#include "ordered_sum.h"

using namespace tmc2::osum;

extern "C" double osum_sequential( const double* t, uint64_t n, uint64_t stride ) {
  volatile double acc = 0.0;  // (volatile: no vectorisation, no re-association whatever the flags)
  for ( uint64_t i = 0; i < n; ++i ) acc = acc + t[i * stride];
  return acc;
}

static double pairwise( const double* t, uint64_t n, uint64_t stride ) {  // an order the sequential loop does not use
  if ( n <= 4 ) {
    double s = 0.0;
    for ( uint64_t i = n; i-- > 0; ) s += t[i * stride];
    return s;
  }
  return pairwise( t, n / 2, stride ) + pairwise( t + ( n / 2 ) * stride, n - n / 2, stride );
}

static Step treeReduce( const std::vector<Step>& s, size_t lo, size_t hi ) {  // [lo, hi), in order, bracketed as a tree
  if ( hi - lo == 1 ) return s[lo];
  const size_t mid = lo + ( hi - lo + 1 ) / 2;
  return then( treeReduce( s, lo, mid ), treeReduce( s, mid, hi ) );
}

// returns the sum; *fallbackBlocks = blocks added term by term
extern "C" double osum_block_form( const double* t, uint64_t n, uint64_t stride, uint64_t B, uint64_t* fallbackBlocks ) {
  const uint64_t      nb = ( n + B - 1 ) / B;
  std::vector<double> approx( nb );
  for ( uint64_t k = 0; k < nb; ++k ) approx[k] = pairwise( t + k * B * stride, std::min<uint64_t>( B, n - k * B ), stride );
  std::vector<int>  E( nb );
  std::vector<Step> st( nb );
  double            prefix = 0.0;
  for ( uint64_t k = 0; k < nb; ++k ) {
    const double lo = prefix, hi = prefix + approx[k];
    prefix          = hi;
    const uint64_t cnt = std::min<uint64_t>( B, n - k * B );
    E[k]               = approx[k] == 0.0 ? kExpIdentity : guessExponent( lo, hi );
    bool allZero = true;
    std::vector<Step> steps( cnt );
    for ( uint64_t i = 0; i < cnt; ++i ) {
      unsigned long long bits;
      memcpy( &bits, &t[( k * B + i ) * stride], 8 );
      if ( bits << 1 ) allZero = false;
      if ( E[k] >= 1 && !stepOf( bits, E[k], steps[i] ) ) E[k] = kExpUnsafe;
    }
    if ( E[k] == kExpIdentity && !allZero ) E[k] = kExpUnsafe;  // (a sum of zero that is not a sum of zeros: negative terms)
    if ( E[k] >= 1 ) st[k] = treeReduce( steps, 0, cnt );
  }
  unsigned long long bits = 0;
  uint64_t           fb   = 0;
  for ( uint64_t k = 0; k < nb; ++k ) {
    if ( apply( bits, st[k], E[k] ) ) continue;
    ++fb;
    double acc;
    memcpy( &acc, &bits, 8 );
    const uint64_t cnt = std::min<uint64_t>( B, n - k * B );
    volatile double a = acc;
    for ( uint64_t i = 0; i < cnt; ++i ) a = a + t[( k * B + i ) * stride];
    acc = a;
    memcpy( &bits, &acc, 8 );
  }
  if ( fallbackBlocks ) *fallbackBlocks = fb;
  double out;
  memcpy( &out, &bits, 8 );
  return out;
}
