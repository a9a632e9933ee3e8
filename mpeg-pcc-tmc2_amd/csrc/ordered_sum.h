// ordered_sum.h -- the exact block form of an ORDERED fp64 sum of non-negative terms (S23: the D2 and colour sums of
// PCCMetrics: `sse += dist` over the points in index order, reference PCCMetrics.cpp:73-229).
//
// acc = fl( fl( fl( a0 + a1 ) + a2 ) + ... ) rounds after every add, so it cannot be re-associated -- but it can be re-stated.
// While the running sum s stays inside ONE binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-52), S = s / u is an integer of
// 53 bits, and an add of a >= 0 is integer arithmetic:  S' = S + q + round,  q = floor( a / u ),  f = a / u - q,
//   round = 1 if f > 1/2,  0 if f < 1/2,  and on the tie f == 1/2 whatever makes S' even (round-to-nearest-even).
// The only thing an add needs to know of the sum so far is therefore the PARITY of S, and that only on ties.  A term -- and,
// by composition, any run of consecutive terms -- is a map "parity of S on entry -> increment of S": two integers (Step).
// Composition is associative, so a block of terms is reduced IN PARALLEL (in order, any bracketing), and the chain over the
// blocks is one integer add per block: S lives in the mantissa of s, so the increment is added to the BIT PATTERN of the double
// (no carry into the exponent as long as the sum stays in the binade).  The terms are non-negative, so the partial sums only
// grow: a block stays inside binade e iff its entry is in it and its exit still is -- checked on the exact values when the
// blocks are chained, which makes the form self-verifying: the binade a block's steps were computed for is a GUESS (from an
// approximate prefix sum), and a block whose guess does not hold -- the few that straddle a power of two, the start of the
// sum -- is added term by term, as before.
//
// Shared by the kernels (metrics.hip) and by the CPU check of the arithmetic (tests/helpers/ordered_sum_check.cpp).
#pragma once
#include <cstdint>
#include <cstring>

#if defined( __HIPCC__ )
#define TMC2_OSUM_HD __host__ __device__ __forceinline__
#else
#define TMC2_OSUM_HD inline
#endif

namespace tmc2 {
namespace osum {

// increment of the sum's bit pattern over a run of terms, for an even (d0) / odd (d1) mantissa on entry
struct Step {
  unsigned long long d0, d1;
};

// f, then g
TMC2_OSUM_HD Step then( const Step f, const Step g ) {
  Step h;
  h.d0 = f.d0 + ( ( f.d0 & 1ull ) ? g.d1 : g.d0 );
  h.d1 = f.d1 + ( ( ( f.d1 + 1ull ) & 1ull ) ? g.d1 : g.d0 );
  return h;
}

constexpr int kExpUnsafe   = -1;  // no binade guessed for the block: its terms are added one by one
constexpr int kExpIdentity = -2;  // every term of the block is zero

// the step of ONE term (its bit pattern) while the sum is in the binade of biased exponent E (1 .. 2046); false: the term does
// not fit the model (negative, not finite, or too large for the sum to stay in the binade) -- the block is not to be trusted
TMC2_OSUM_HD bool stepOf( unsigned long long bits, int E, Step& out ) {
  out.d0 = out.d1 = 0;
  if ( bits >> 63 ) return bits == 0x8000000000000000ull;  // (-0: s + -0 = s)
  int Ea = int( bits >> 52 );
  if ( Ea == 2047 ) return false;
  unsigned long long m = bits & 0xFFFFFFFFFFFFFull;
  if ( Ea )
    m |= 1ull << 52;
  else
    Ea = 1;  // subnormal: m * 2^(1 - 1075)
  if ( m == 0 ) return true;
  const int sh = E - Ea;  // a / u = m * 2^-sh
  if ( sh <= 0 ) return false;  // a >= 2^e: the sum leaves the binade
  if ( sh > 54 ) return true;   // a < u / 2
  const unsigned long long q = m >> sh, r = m & ( ( 1ull << sh ) - 1ull ), half = 1ull << ( sh - 1 );
  if ( r > half ) {
    out.d0 = out.d1 = q + 1ull;
  } else if ( r < half ) {
    out.d0 = out.d1 = q;
  } else {  // tie: to even
    out.d0 = q + ( q & 1ull );
    out.d1 = q + ( ( q + 1ull ) & 1ull );
  }
  return true;
}

// a block's step applied to the sum (bit pattern of a non-negative double); E = the binade the step was computed for.
// false: the guess does not hold for this entry / exit -- the sum is left as it was
TMC2_OSUM_HD bool apply( unsigned long long& sumBits, const Step st, int E ) {
  if ( E == kExpIdentity ) return true;
  if ( E < 1 || int( sumBits >> 52 ) != E ) return false;
  const unsigned long long next = sumBits + ( ( sumBits & 1ull ) ? st.d1 : st.d0 );
  if ( int( next >> 52 ) != E ) return false;  // (the exit would be in the next binade, or beyond)
  sumBits = next;
  return true;
}

// the binade to guess for a block whose entry / exit are about lo / hi (approximate prefix sums, lo <= hi): the exponent the
// two share once a relative margin is taken off / put on, else none
TMC2_OSUM_HD int guessExponent( double lo, double hi ) {
  const double margin = 1.0 / 16777216.0;  // 2^-24: far above what an approximate sum of 10^7 terms is off by (~ n 2^-53)
  lo = lo - lo * margin, hi = hi + hi * margin;
  if ( !( lo > 0.0 ) || !( hi >= lo ) ) return kExpUnsafe;
  unsigned long long bl, bh;
  memcpy( &bl, &lo, 8 ), memcpy( &bh, &hi, 8 );
  const int El = int( bl >> 52 ), Eh = int( bh >> 52 );
  if ( El != Eh || El < 64 || El > 2000 ) return kExpUnsafe;
  return El;
}

}  // namespace osum
}  // namespace tmc2
