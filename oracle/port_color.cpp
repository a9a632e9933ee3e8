// oracle/port_color.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// Colour-space conversion around the attribute video codec: what PCCVideoEncoder::compress
// (PccLibEncoder/source/PCCVideoEncoder.cpp:326-413) asks of PCCInternalColorConverter (PccLibColorConverter/source/
// PCCInternalColorConverter.cpp) when no external converter is configured:
//   "RGB444ToYUV420_8_4"  before the codec: convertRGB44ToYUV420 (:406-424) = RGBtoFloatRGB (:559), convertRGBToYUV (:570),
//                         downsampling (:649) with filter 4 (DF_GS; downsamplingHorizontal / Vertical,
//                         PCCInternalColorConverter.h:153-185), floatYUVToYUV (:589)
//   "YUV420ToYUV444_8_0"  after it: convertYUV420ToYUV444 (:462-482) = YUVtoFloatYUV (:603), upsampling (:675) with filter 0
//                         (UF_F0; upsamplingVertical0/1, upsamplingHorizontal0/1, .h:187-249), floatYUVToYUV with 16 bits
// The filter taps are those of the reference's tables g_filter444to420[4] / g_filter420to444[0] (:37-327).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "oracle.h"

namespace {
inline int    clampi( int v, int a, int b ) { return v < a ? a : ( v > b ? b : v ); }
inline float  clampf( float v, float a, float b ) { return v < a ? a : ( v > b ? b : v ); }
inline double clampd( double v, double a, double b ) { return v < a ? a : ( v > b ? b : v ); }

// DF_GS, as the reference writes it: normalised taps times 512, rounded to float; shift 9
const double kGsHorizontal[15] = {-0.01716352771649, 0.0, +0.04066666714886, 0.0, -0.09154810319329, 0.0, 0.31577823859943,
                                  0.50453345032298,  0.31577823859943, 0.0, -0.09154810319329, 0.0, 0.04066666714886, 0.0,
                                  -0.01716352771649};
const double kGsVertical[16]   = {-0.00945406160902, -0.01539537217249, 0.02360533018213,  0.03519540819902,
                                  -0.05254456550808, -0.08189331229717, 0.14630826357715,  0.45417830962846,
                                  0.45417830962846,  0.14630826357715,  -0.08189331229717, -0.05254456550808,
                                  0.03519540819902,  0.02360533018213,  -0.01539537217249, -0.00945406160902};
// UF_F0; shift 8
const float kF0Vertical0[4]   = {-8.0f, +64.0f, +216.0f, -16.0f};
const float kF0Vertical1[4]   = {-16.0f, +216.0f, +64.0f, -8.0f};
const float kF0Horizontal0[2] = {0.0f, +256.0f};
const float kF0Horizontal1[4] = {-16.0f, +144.0f, +144.0f, -16.0f};

template <typename T>
T quantise( float v, bool chroma, int nbyte ) {  // floatYUVToYUV
  const double offset = chroma ? ( nbyte == 1 ? 128. : 32768. ) : 0;
  const double scale  = nbyte == 1 ? 255. : 65535.;
  const float  r      = std::round( (float)( scale * (double)v + offset ) );
  return static_cast<T>( std::min( std::max( r, 0.f ), (float)scale ) );
}
}  // namespace

extern "C" {

// rgb: u8 [3][H][W] (R, G, B planes).  y: u8 [H][W]; u, v: u8 [H/2][W/2].  filter must be 4 (DF_GS).
int orc_convert_rgb444_to_yuv420( const uint8_t* rgb, int W, int H, int filter, uint8_t* y, uint8_t* u, uint8_t* v ) {
  if ( filter != 4 ) return -1;
  const size_t       area = size_t( W ) * H;
  std::vector<float> Y( area ), C[2];
  C[0].resize( area ), C[1].resize( area );
  for ( size_t i = 0; i < area; ++i ) {
    const float R = (float)rgb[i] / 255.f, G = (float)rgb[area + i] / 255.f, B = (float)rgb[2 * area + i] / 255.f;
    Y[i]    = (float)( clampd( 0.212600 * R + 0.715200 * G + 0.072200 * B, 0.0, 1.0 ) );
    C[0][i] = (float)( clampd( -0.114572 * R - 0.385428 * G + 0.500000 * B, -0.5, 0.5 ) );
    C[1][i] = (float)( clampd( 0.500000 * R - 0.454153 * G - 0.045847 * B, -0.5, 0.5 ) );
  }
  float hz[15], vt[16];
  for ( int k = 0; k < 15; ++k ) hz[k] = (float)( kGsHorizontal[k] * 512 );
  for ( int k = 0; k < 16; ++k ) vt[k] = (float)( kGsVertical[k] * 512 );
  const float scale = 1.0f / ( (float)( 1 << 9 ) );
  const int   Wo = W / 2, Ho = H / 2;
  for ( int c = 0; c < 2; ++c ) {
    std::vector<float> temp( size_t( Wo ) * H );
    for ( int i = 0; i < H; ++i )
      for ( int j = 0; j < Wo; ++j ) {
        double value = 0;
        for ( int k = 0; k < 15; ++k ) value += (double)hz[k] * (double)( C[c][size_t( i ) * W + clampi( 2 * j + k - 7, 0, W - 1 )] );
        temp[size_t( i ) * Wo + j] = (float)( ( value + 0.0 ) * (double)scale );
      }
    uint8_t* out = c == 0 ? u : v;
    for ( int i = 0; i < Ho; ++i )
      for ( int j = 0; j < Wo; ++j ) {
        double value = 0;
        for ( int k = 0; k < 16; ++k ) value += (double)vt[k] * (double)( temp[size_t( clampi( 2 * i + k - 7, 0, H - 1 ) ) * Wo + j] );
        out[size_t( i ) * Wo + j] = quantise<uint8_t>( (float)( ( value + 0.0 ) * (double)scale ), true, 1 );
      }
  }
  for ( size_t i = 0; i < area; ++i ) y[i] = quantise<uint8_t>( Y[i], false, 1 );
  return 0;
}

// y: u8 [H][W]; u, v: u8 [H/2][W/2].  out: u16 [3][H][W] (16-bit YUV 4:4:4).  filter must be 0 (UF_F0).
int orc_convert_yuv420_to_yuv444( const uint8_t* y, const uint8_t* u, const uint8_t* v, int W, int H, int filter, uint16_t* out ) {
  if ( filter != 0 ) return -1;
  const size_t area = size_t( W ) * H;
  const int    Wi = W / 2, Hi = H / 2;
  const double weight = 1.0 / 255.;
  for ( size_t i = 0; i < area; ++i )
    out[i] = quantise<uint16_t>( clampf( (float)( weight * (double)( int( y[i] ) - 0 ) ), 0.f, 1.f ), false, 2 );
  const float scale = 1.0f / ( (float)( 1 << 8 ) );
  for ( int c = 0; c < 2; ++c ) {
    const uint8_t*     src = c == 0 ? u : v;
    std::vector<float> in( size_t( Wi ) * Hi ), temp( size_t( Wi ) * H );
    for ( size_t i = 0; i < in.size(); ++i ) in[i] = clampf( (float)( weight * (double)( int( src[i] ) - 128 ) ), -0.5f, 0.5f );
    for ( int i = 0; i < Hi; ++i )
      for ( int j = 0; j < Wi; ++j ) {
        float v0 = 0, v1 = 0;
        for ( int k = 0; k < 4; ++k ) v0 += kF0Vertical0[k] * (float)( in[size_t( clampi( i + k - 2, 0, Hi - 1 ) ) * Wi + j] );
        for ( int k = 0; k < 4; ++k ) v1 += kF0Vertical1[k] * (float)( in[size_t( clampi( i + 1 + k - 2, 0, Hi - 1 ) ) * Wi + j] );
        temp[size_t( 2 * i ) * Wi + j]     = (float)( ( v0 + 0.f ) * scale );
        temp[size_t( 2 * i + 1 ) * Wi + j] = (float)( ( v1 + 0.f ) * scale );
      }
    uint16_t* dst = out + size_t( c + 1 ) * area;
    for ( int i = 0; i < H; ++i )
      for ( int j = 0; j < Wi; ++j ) {
        float h0 = 0, h1 = 0;
        for ( int k = 0; k < 2; ++k ) h0 += kF0Horizontal0[k] * (float)( temp[size_t( i ) * Wi + clampi( j + k - 1, 0, Wi - 1 )] );
        for ( int k = 0; k < 4; ++k ) h1 += kF0Horizontal1[k] * (float)( temp[size_t( i ) * Wi + clampi( j + 1 + k - 2, 0, Wi - 1 )] );
        dst[size_t( i ) * W + 2 * j]     = quantise<uint16_t>( (float)( ( h0 + 0.f ) * scale ), true, 2 );
        dst[size_t( i ) * W + 2 * j + 1] = quantise<uint16_t>( (float)( ( h1 + 0.f ) * scale ), true, 2 );
      }
  }
  return 0;
}
}
