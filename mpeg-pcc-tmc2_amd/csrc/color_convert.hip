// color_convert.hip -- colour-space conversion around the attribute video codec on gfx950.
//
// Replaces what PCCVideoEncoder::compress (reference: source/lib/PccLibEncoder/source/PCCVideoEncoder.cpp:326-413) asks of
// PCCInternalColorConverter (source/lib/PccLibColorConverter/source/PCCInternalColorConverter.cpp) when no external
// converter is configured -- the CTC attribute path:
//   before the codec  convert( "RGB444ToYUV420_8_4" )  = convertRGB44ToYUV420 (:406-424): RGBtoFloatRGB (:559),
//                     convertRGBToYUV (:570), downsampling (:649) with g_filter444to420[4] (DF_GS: 15 taps across, 16
//                     down; downsamplingHorizontal / Vertical, PCCInternalColorConverter.h:153-185), floatYUVToYUV (:589)
//   after the codec   convert( "YUV420ToYUV444_8_0" )  = convertYUV420ToYUV444 (:462-482): YUVtoFloatYUV (:603), upsampling
//                     (:675) with g_filter420to444[0] (UF_F0: upsamplingVertical0/1, upsamplingHorizontal0/1, .h:187-249),
//                     floatYUVToYUV to 16 bits -- the frames PCCCodec::colorPointCloud reads
//
// Image-parallel byte / float work: one thread per output sample, every tap a clamped neighbour read (rows of a 1280-wide
// plane stay in L2).  The arithmetic keeps the reference's types and order (float image, double accumulation going down,
// float accumulation going up, one rounding per stage, no FMA contraction), so the 8-bit 4:2:0 frames handed to the video
// encoder and the 16-bit 4:4:4 frames handed to the reconstruction are the reference's bit for bit.
#include <algorithm>

#include "internal.h"

namespace tmc2 {
namespace {

struct DownFilter {  // DF_GS
  float across[15], down[16];
  float scale;
};
// the reference stores the taps as (float)( normalised tap * 512 ), shift 9
DownFilter makeDownGS() {
  static const double a[15] = {-0.01716352771649, 0.0, +0.04066666714886, 0.0, -0.09154810319329, 0.0, 0.31577823859943,
                               0.50453345032298,  0.31577823859943, 0.0, -0.09154810319329, 0.0, 0.04066666714886, 0.0,
                               -0.01716352771649};
  static const double d[16] = {-0.00945406160902, -0.01539537217249, 0.02360533018213,  0.03519540819902,
                               -0.05254456550808, -0.08189331229717, 0.14630826357715,  0.45417830962846,
                               0.45417830962846,  0.14630826357715,  -0.08189331229717, -0.05254456550808,
                               0.03519540819902,  0.02360533018213,  -0.01539537217249, -0.00945406160902};
  DownFilter f;
  for ( int k = 0; k < 15; ++k ) f.across[k] = float( a[k] * 512 );
  for ( int k = 0; k < 16; ++k ) f.down[k] = float( d[k] * 512 );
  f.scale = 1.0f / float( 1 << 9 );
  return f;
}

__device__ __forceinline__ float quantiseF( float v, double scale, double offset ) {  // floatYUVToYUV before the cast
  const float r = roundf( float( scale * double( v ) + offset ) );
  return fminf( fmaxf( r, 0.f ), float( scale ) );
}
__device__ __forceinline__ double clampD( double v, double a, double b ) { return v < a ? a : ( v > b ? b : v ); }
__device__ __forceinline__ float  clampF( float v, float a, float b ) { return v < a ? a : ( v > b ? b : v ); }

// rgb: three u8 planes of W*H.  luma out as u8; the two chroma planes as float images
__global__ __launch_bounds__( 256 ) void rgbToYuvKernel( const uint8_t* __restrict__ rgb, size_t area, uint8_t* __restrict__ luma,
                                                          float* __restrict__ chroma ) {
  const size_t i = size_t( blockIdx.x ) * blockDim.x + threadIdx.x;
  if ( i >= area ) return;
  const float R = __fdiv_rn( float( rgb[i] ), 255.f ), G = __fdiv_rn( float( rgb[area + i] ), 255.f ),
              B = __fdiv_rn( float( rgb[2 * area + i] ), 255.f );
  const float Y = float( clampD( 0.212600 * R + 0.715200 * G + 0.072200 * B, 0.0, 1.0 ) );
  chroma[i]        = float( clampD( -0.114572 * R - 0.385428 * G + 0.500000 * B, -0.5, 0.5 ) );
  chroma[area + i] = float( clampD( 0.500000 * R - 0.454153 * G - 0.045847 * B, -0.5, 0.5 ) );
  luma[i]          = uint8_t( quantiseF( Y, 255., 0. ) );
}

// blockIdx.y = chroma plane.  in: [2][H][W] float, out: [2][H][W/2] float
__global__ __launch_bounds__( 256 ) void downAcrossKernel( const float* __restrict__ in, int W, int H, DownFilter f,
                                                            float* __restrict__ out ) {
  const int Wo = W / 2;
  const int t  = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= Wo * H ) return;
  const int    i = t / Wo, j = t % Wo;
  const float* row   = in + size_t( blockIdx.y ) * W * H + size_t( i ) * W;
  double       value = 0;
#pragma unroll
  for ( int k = 0; k < 15; ++k ) value += double( f.across[k] ) * double( row[min( max( 2 * j + k - 7, 0 ), W - 1 )] );
  out[size_t( blockIdx.y ) * Wo * H + t] = float( ( value + 0.0 ) * double( f.scale ) );
}
// in: [2][H][Wo] float, out: u8 [2][H/2][Wo] with stride planeStride between the two chroma planes
__global__ __launch_bounds__( 256 ) void downDownKernel( const float* __restrict__ in, int Wo, int H, DownFilter f,
                                                          uint8_t* __restrict__ out, size_t planeStride ) {
  const int Ho = H / 2;
  const int t  = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= Wo * Ho ) return;
  const int    i = t / Wo, j = t % Wo;
  const float* col   = in + size_t( blockIdx.y ) * Wo * H + j;
  double       value = 0;
#pragma unroll
  for ( int k = 0; k < 16; ++k ) value += double( f.down[k] ) * double( col[size_t( min( max( 2 * i + k - 7, 0 ), H - 1 ) ) * Wo] );
  out[size_t( blockIdx.y ) * planeStride + t] = uint8_t( quantiseF( float( ( value + 0.0 ) * double( f.scale ) ), 255., 128. ) );
}

// ---- 4:2:0, 8 bits  ->  4:4:4, 16 bits (UF_F0) ------------------------------------------------------------
__global__ __launch_bounds__( 256 ) void lumaTo16Kernel( const uint8_t* __restrict__ y, size_t area, uint16_t* __restrict__ out ) {
  const size_t i = size_t( blockIdx.x ) * blockDim.x + threadIdx.x;
  if ( i >= area ) return;
  const float v = clampF( float( __ddiv_rn( 1.0, 255. ) * double( int( y[i] ) ) ), 0.f, 1.f );
  out[i]        = uint16_t( quantiseF( v, 65535., 0. ) );
}
// blockIdx.y = chroma plane.  in: u8 [Hi][Wi] per plane (planeStride apart), out: float [2][2*Hi][Wi]
__global__ __launch_bounds__( 256 ) void upDownKernel( const uint8_t* __restrict__ in, size_t planeStride, int Wi, int Hi,
                                                        float* __restrict__ out ) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= Wi * Hi ) return;
  const int      i = t / Wi, j = t % Wi;
  const uint8_t* src    = in + size_t( blockIdx.y ) * planeStride;
  const double   weight = __ddiv_rn( 1.0, 255. );
  float          s[5];  // rows i-2 .. i+2 (clamped) as float samples
#pragma unroll
  for ( int k = 0; k < 5; ++k )
    s[k] = clampF( float( weight * double( int( src[size_t( min( max( i + k - 2, 0 ), Hi - 1 ) ) * Wi + j] ) - 128 ) ), -0.5f, 0.5f );
  const float scale = 1.0f / float( 1 << 8 );
  float       v0 = 0.f, v1 = 0.f;
  v0 += -8.0f * s[0];
  v0 += 64.0f * s[1];
  v0 += 216.0f * s[2];
  v0 += -16.0f * s[3];
  v1 += -16.0f * s[1];
  v1 += 216.0f * s[2];
  v1 += 64.0f * s[3];
  v1 += -8.0f * s[4];
  float* dst = out + size_t( blockIdx.y ) * Wi * 2 * Hi;
  dst[size_t( 2 * i ) * Wi + j]     = ( v0 + 0.f ) * scale;
  dst[size_t( 2 * i + 1 ) * Wi + j] = ( v1 + 0.f ) * scale;
}
// in: float [2][H][Wi], out: u16 planes [2][H][2*Wi]
__global__ __launch_bounds__( 256 ) void upAcrossKernel( const float* __restrict__ in, int Wi, int H, uint16_t* __restrict__ out ) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= Wi * H ) return;
  const int    i = t / Wi, j = t % Wi;
  const float* row = in + size_t( blockIdx.y ) * Wi * H + size_t( i ) * Wi;
  const float  scale = 1.0f / float( 1 << 8 );
  float        h0 = 0.f, h1 = 0.f;
  h0 += 0.0f * row[max( j - 1, 0 )];
  h0 += 256.0f * row[j];
  h1 += -16.0f * row[max( j - 1, 0 )];
  h1 += 144.0f * row[j];
  h1 += 144.0f * row[min( j + 1, Wi - 1 )];
  h1 += -16.0f * row[min( j + 2, Wi - 1 )];
  uint16_t* dst = out + size_t( blockIdx.y ) * size_t( 2 * Wi ) * H + size_t( i ) * 2 * Wi + 2 * j;
  dst[0]        = uint16_t( quantiseF( ( h0 + 0.f ) * scale, 65535., 32768. ) );
  dst[1]        = uint16_t( quantiseF( ( h1 + 0.f ) * scale, 65535., 32768. ) );
}

int checkGeometry( int W, int H, const char* who ) {
  if ( W <= 0 || H <= 0 || ( W & 1 ) || ( H & 1 ) ) {
    setError( "%s: image %dx%d unsupported (even dimensions required)", who, W, H );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}
}  // namespace

// d_rgb: u8 [3][H][W] on the device -> d_yuv: Y [H][W], U [H/2][W/2], V [H/2][W/2] back to back (I420 frame)
int rgb444ToYuv420Device( tmc2_ctx* ctx, const uint8_t* d_rgb, int W, int H, int filter, uint8_t* d_yuv ) {
  TMC2_TRY( checkGeometry( W, H, "RGB444ToYUV420" ) );
  if ( filter != 4 ) {
    setError( "RGB444ToYUV420: downsampling filter %d unsupported (4 = DF_GS, the reference's default)", filter );
    return TMC2_E_UNSUPPORTED;
  }
  hipStream_t   s    = ctx->stream;
  const size_t  area = size_t( W ) * H;
  const int     Wo = W / 2, Ho = H / 2;
  DevBuf<float> d_chroma, d_temp;
  TMC2_TRY( d_chroma.alloc( 2 * area ) );
  TMC2_TRY( d_temp.alloc( 2 * size_t( Wo ) * H ) );
  const DownFilter f = makeDownGS();
  const dim3       blk( 256 );
  hipLaunchKernelGGL( rgbToYuvKernel, dim3( uint32_t( ( area + 255 ) / 256 ) ), blk, 0, s, d_rgb, area, d_yuv, d_chroma.p );
  hipLaunchKernelGGL( downAcrossKernel, dim3( ( Wo * H + 255 ) / 256, 2 ), blk, 0, s, d_chroma.p, W, H, f, d_temp.p );
  hipLaunchKernelGGL( downDownKernel, dim3( ( Wo * Ho + 255 ) / 256, 2 ), blk, 0, s, d_temp.p, Wo, H, f, d_yuv + area,
                      size_t( Wo ) * Ho );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

// d_yuv: I420 frame (8 bits) on the device -> d_out: u16 [3][H][W]
int yuv420ToYuv444Device( tmc2_ctx* ctx, const uint8_t* d_yuv, int W, int H, int filter, uint16_t* d_out ) {
  TMC2_TRY( checkGeometry( W, H, "YUV420ToYUV444" ) );
  if ( filter != 0 ) {
    setError( "YUV420ToYUV444: upsampling filter %d unsupported (0 = UF_F0, the reference's default)", filter );
    return TMC2_E_UNSUPPORTED;
  }
  hipStream_t   s    = ctx->stream;
  const size_t  area = size_t( W ) * H;
  const int     Wi = W / 2, Hi = H / 2;
  DevBuf<float> d_temp;
  TMC2_TRY( d_temp.alloc( 2 * size_t( Wi ) * H ) );
  const dim3 blk( 256 );
  hipLaunchKernelGGL( lumaTo16Kernel, dim3( uint32_t( ( area + 255 ) / 256 ) ), blk, 0, s, d_yuv, area, d_out );
  hipLaunchKernelGGL( upDownKernel, dim3( ( Wi * Hi + 255 ) / 256, 2 ), blk, 0, s, d_yuv + area, size_t( Wi ) * Hi, Wi, Hi, d_temp.p );
  hipLaunchKernelGGL( upAcrossKernel, dim3( ( Wi * H + 255 ) / 256, 2 ), blk, 0, s, d_temp.p, Wi, H, d_out + area );
  TMC2_HIP( hipGetLastError() );
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

int tmc2_color_convert_rgb444_to_yuv420( tmc2_ctx* ctx, const uint8_t* rgb, int width, int height, int downsamplingFilter,
                                         uint8_t* yuv420 ) {
  if ( !ctx || !rgb || !yuv420 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  const size_t   area = size_t( width ) * size_t( height );
  tmc2::DevBuf<uint8_t> d_rgb, d_yuv;
  TMC2_TRY( d_rgb.alloc( 3 * area ) );
  TMC2_TRY( d_yuv.alloc( area * 3 / 2 ) );
  TMC2_HIP( hipMemcpyAsync( d_rgb.p, rgb, 3 * area, hipMemcpyHostToDevice, ctx->stream ) );
  const int sid = ctx->stageBegin( "rgb444_to_yuv420" );
  const int r   = tmc2::rgb444ToYuv420Device( ctx, d_rgb.p, width, height, downsamplingFilter, d_yuv.p );
  ctx->stageEnd( sid );
  TMC2_TRY( r );
  TMC2_HIP( hipMemcpyAsync( yuv420, d_yuv.p, area * 3 / 2, hipMemcpyDeviceToHost, ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  return TMC2_OK;
}

int tmc2_color_convert_yuv420_to_yuv444( tmc2_ctx* ctx, const uint8_t* yuv420, int width, int height, int upsamplingFilter,
                                         uint16_t* yuv444 ) {
  if ( !ctx || !yuv420 || !yuv444 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( ctx );
  const size_t   area = size_t( width ) * size_t( height );
  tmc2::DevBuf<uint8_t>  d_yuv;
  tmc2::DevBuf<uint16_t> d_out;
  TMC2_TRY( d_yuv.alloc( area * 3 / 2 ) );
  TMC2_TRY( d_out.alloc( 3 * area ) );
  TMC2_HIP( hipMemcpyAsync( d_yuv.p, yuv420, area * 3 / 2, hipMemcpyHostToDevice, ctx->stream ) );
  const int sid = ctx->stageBegin( "yuv420_to_yuv444" );
  const int r   = tmc2::yuv420ToYuv444Device( ctx, d_yuv.p, width, height, upsamplingFilter, d_out.p );
  ctx->stageEnd( sid );
  TMC2_TRY( r );
  TMC2_HIP( hipMemcpyAsync( yuv444, d_out.p, 3 * area * sizeof( uint16_t ), hipMemcpyDeviceToHost, ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  return TMC2_OK;
}

// the frame's two attribute canvases (resident after tmc2_encoder_generate_attribute_images) as the two I420 frames the
// attribute video encoder reads: yuv420 = [map][ Y H*W | U H*W/4 | V H*W/4 ]
int tmc2_encoder_attribute_to_yuv420( tmc2_frame* f, int downsamplingFilter, uint8_t* yuv420 ) {
  if ( !f || !yuv420 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  if ( !f->haveAttributeImages ) {
    tmc2::setError( "attribute_to_yuv420: no attribute images" );
    return TMC2_E_STATE;
  }
  tmc2_ctx*    ctx   = f->ctx;
  const int    W = f->canvasW, H = f->canvasH;
  const size_t area = size_t( W ) * H, frame = area * 3 / 2;
  tmc2::DevBuf<uint8_t> d_yuv;
  TMC2_TRY( d_yuv.alloc( 2 * frame ) );
  const int sid = ctx->stageBegin( "rgb444_to_yuv420" );
  int       r   = TMC2_OK;
  for ( int m = 0; m < 2 && r == TMC2_OK; ++m )
    r = tmc2::rgb444ToYuv420Device( ctx, f->d_attr.p + size_t( m ) * 3 * area, W, H, downsamplingFilter, d_yuv.p + size_t( m ) * frame );
  ctx->stageEnd( sid );
  TMC2_TRY( r );
  TMC2_HIP( hipMemcpyAsync( yuv420, d_yuv.p, 2 * frame, hipMemcpyDeviceToHost, ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  return TMC2_OK;
}

// the two decoded I420 attribute frames of the point-cloud frame -> 16-bit 4:4:4 planes, kept on the device for
// tmc2_codec_color_point_cloud( f, NULL )
int tmc2_codec_set_decoded_attribute_yuv420( tmc2_frame* f, const uint8_t* yuv420, int upsamplingFilter ) {
  if ( !f || !yuv420 ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  if ( !f->haveGeometryImages ) {
    tmc2::setError( "set_decoded_attribute_yuv420: the frame has no canvas yet" );
    return TMC2_E_STATE;
  }
  tmc2_ctx*    ctx   = f->ctx;
  const int    W = f->canvasW, H = f->canvasH;
  const size_t area = size_t( W ) * H, frame = area * 3 / 2;
  tmc2::DevBuf<uint8_t> d_yuv;
  TMC2_TRY( d_yuv.alloc( 2 * frame ) );
  TMC2_TRY( f->d_attr16.alloc( 6 * area ) );
  TMC2_HIP( hipMemcpyAsync( d_yuv.p, yuv420, 2 * frame, hipMemcpyHostToDevice, ctx->stream ) );
  const int sid = ctx->stageBegin( "yuv420_to_yuv444" );
  int       r   = TMC2_OK;
  for ( int m = 0; m < 2 && r == TMC2_OK; ++m )
    r = tmc2::yuv420ToYuv444Device( ctx, d_yuv.p + size_t( m ) * frame, W, H, upsamplingFilter, f->d_attr16.p + size_t( m ) * 3 * area );
  ctx->stageEnd( sid );
  TMC2_TRY( r );
  TMC2_HIP( hipStreamSynchronize( ctx->stream ) );
  f->haveAttr16 = true;
  return TMC2_OK;
}

int tmc2_frame_get_decoded_attribute( tmc2_frame* f, uint16_t* planes ) {
  if ( !f || !planes ) return TMC2_E_INVALID;
  tmc2::ApiScope scope( f->ctx );
  if ( !f->haveAttr16 ) {
    tmc2::setError( "get_decoded_attribute: no decoded attribute frames on the device" );
    return TMC2_E_STATE;
  }
  const size_t area = size_t( f->canvasW ) * f->canvasH;
  TMC2_HIP( hipMemcpyAsync( planes, f->d_attr16.p, 6 * area * sizeof( uint16_t ), hipMemcpyDeviceToHost, f->ctx->stream ) );
  TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  return TMC2_OK;
}
}
