// metrics_text.cpp -- the text PCCMetrics::display() / QualityMetrics::print() write for one frame (reference:
// source/lib/PccLibMetrics/source/PCCMetrics.cpp:230-279, 376-391), from the numbers tmc2_metrics_compute returns.  The CTC
// log parsers read these lines ("mseF,PSNR (p2point): ..."), so an application that swaps in the device metric keeps its logs
// byte for byte.  Host only.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "internal.h"

namespace {
// operator<<( double ) of a stream in its default float format with the given precision
std::string num( double v, int precision ) {
  char buf[64];
  std::snprintf( buf, sizeof( buf ), "%.*g", precision, v );
  return buf;
}
}  // namespace

extern "C" int tmc2_metrics_display( const double* out, uint64_t sourcePoints, uint64_t reconstructPoints, const int64_t* counts,
                                     uint64_t resolution, int withC2p, int precision, char* text, uint64_t capacity,
                                     uint64_t* needed ) {
  if ( !out || !counts || precision < 1 || precision > 17 ) return TMC2_E_INVALID;
  std::string s = "Metrics results \n";
  s += "WARNING: " + std::to_string( (unsigned long long)( reconstructPoints - uint64_t( counts[1] ) ) ) +
       " points with same coordinates found\n";
  s += "Imported intrinsic resoluiton: " + std::to_string( (unsigned long long)resolution ) + "\n";  // (sic)
  s += "Peak distance for PSNR: " + std::to_string( (unsigned long long)resolution ) + "\n";
  s += "Point cloud sizes for org version, dec version, and the scaling ratio: " + std::to_string( (unsigned long long)sourcePoints ) +
       ", " + std::to_string( (long long)counts[1] ) + ", " +
       num( double( static_cast<float>( counts[1] ) / static_cast<float>( sourcePoints ) ), precision ) + "\n";
  const char* head[3] = {"1. Use infile1 (A) as reference, loop over A, use normals on B. (A->B).\n",
                         "2. Use infile2 (B) as reference, loop over B, use normals on A. (B->A).\n", "3. Final (symmetric).\n"};
  const char  code[3] = {'1', '2', 'F'};
  // colour PSNR of U and V: getPSNR( mse, 1.0 ) per direction, the smaller one for the symmetric row
  double psnr[3][3];
  for ( int r = 0; r < 2; ++r )
    for ( int c = 0; c < 3; ++c ) psnr[r][c] = c == 0 ? out[8 * r + 7] : 10 * std::log10( ( 1.0 * ( 1.0 * 1.0 ) ) / out[8 * r + 4 + c] );
  for ( int c = 0; c < 3; ++c ) psnr[2][c] = c == 0 ? out[8 * 2 + 7] : std::min( psnr[0][c], psnr[1][c] );
  for ( int r = 0; r < 3; ++r ) {
    const double*     q = out + 8 * r;
    const std::string k( 1, code[r] );
    s += head[r];
    s += "   mse" + k + "      (p2point): " + num( q[0], precision ) + "\n";
    s += "   mse" + k + ",PSNR (p2point): " + num( q[1], precision ) + "\n";
    // (the symmetric row is a default-constructed QualityMetrics: it prints its point-to-plane lines even when that metric
    // was not computed -- as zeros)
    if ( withC2p || r == 2 ) {
      s += "   mse" + k + "      (p2plane): " + num( withC2p ? q[2] : 0.0, precision ) + "\n";
      s += "   mse" + k + ",PSNR (p2plane): " + num( withC2p ? q[3] : 0.0, precision ) + "\n";
    }
    for ( int c = 0; c < 3; ++c ) s += "   c[" + std::to_string( c ) + "],    " + k + "         : " + num( q[4 + c], precision ) + "\n";
    for ( int c = 0; c < 3; ++c ) s += "   c[" + std::to_string( c ) + "],PSNR" + k + "         : " + num( psnr[r][c], precision ) + "\n";
  }
  if ( needed ) *needed = s.size() + 1;
  if ( !text || capacity < s.size() + 1 ) {
    if ( text ) tmc2::setError( "metrics_display: %zu bytes needed", s.size() + 1 );
    return text ? TMC2_E_INVALID : TMC2_OK;
  }
  std::memcpy( text, s.c_str(), s.size() + 1 );
  return TMC2_OK;
}
