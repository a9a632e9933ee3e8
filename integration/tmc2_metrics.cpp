// integration/tmc2_metrics.cpp -- PccAppMetrics over the C-ABI: D1 / D2 / colour PSNR between two PLY sequences on the device,
// printed as PCCMetrics::display() prints them (the lines the CTC log parsers read).
//
//   tmc2_metrics --uncompressedDataPath src_%04d.ply --reconstructedDataPath rec_%04d.ply [--normalDataPath nrm_%04d.ply]
//                --startFrameNumber N --frameCount N [--resolution 1023] [--device 0]
// Exits non-zero with the library's message when no MI355X is visible -- there is no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "tmc2hip.h"

namespace {
[[noreturn]] void die( const char* what ) {
  std::fprintf( stderr, "tmc2_metrics: %s: %s\n", what, tmc2_last_error() );
  std::exit( 2 );
}
#define CHECK( call )                        \
  do {                                       \
    if ( ( call ) != TMC2_OK ) die( #call ); \
  } while ( 0 )
struct Cloud {
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  std::vector<double>  normals;
  uint64_t             n = 0;
};
void load( const std::string& pattern, int frame, bool wantNormals, Cloud& c ) {
  char path[4096];
  std::snprintf( path, sizeof( path ), pattern.c_str(), frame );
  int hasColors = 0, hasNormals = 0;
  CHECK( tmc2_ply_info( path, wantNormals ? 1 : 0, &c.n, &hasColors, &hasNormals ) );
  c.xyz.assign( 3 * c.n, 0 );
  c.rgb.assign( 3 * c.n, 0 );
  if ( wantNormals ) {
    if ( !hasNormals ) {
      std::fprintf( stderr, "tmc2_metrics: %s carries no float normals\n", path );
      std::exit( 2 );
    }
    c.normals.assign( 3 * c.n, 0.0 );
  }
  CHECK( tmc2_ply_read( path, c.xyz.data(), hasColors ? c.rgb.data() : nullptr, wantNormals ? c.normals.data() : nullptr, c.n, 8, &c.n ) );
}
}  // namespace

int main( int argc, char** argv ) {
  std::string src, rec, nrm;
  int         start = 0, count = 1, device = 0;
  uint64_t    resolution = 1023;
  for ( int i = 1; i < argc; ++i ) {
    const std::string a    = argv[i];
    auto              next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if ( a == "--uncompressedDataPath" ) src = next();
    else if ( a == "--reconstructedDataPath" ) rec = next();
    else if ( a == "--normalDataPath" ) nrm = next();
    else if ( a == "--startFrameNumber" ) start = std::atoi( next() );
    else if ( a == "--frameCount" ) count = std::atoi( next() );
    else if ( a == "--resolution" ) resolution = uint64_t( std::atoll( next() ) );
    else if ( a == "--device" ) device = std::atoi( next() );
    else {
      std::puts( "usage: tmc2_metrics --uncompressedDataPath src_%04d.ply --reconstructedDataPath rec_%04d.ply [--normalDataPath n_%04d.ply]\n"
                 "                    [--startFrameNumber N] [--frameCount N] [--resolution 1023] [--device N]" );
      return 1;
    }
  }
  if ( src.empty() || rec.empty() || count <= 0 ) {
    std::puts( "tmc2_metrics: --uncompressedDataPath and --reconstructedDataPath are required" );
    return 1;
  }
  tmc2_ctx* ctx = nullptr;
  CHECK( tmc2_ctx_create( device, &ctx ) );
  for ( int f = start; f < start + count; ++f ) {
    Cloud a, b, n;
    load( src, f, false, a );
    load( rec, f, false, b );
    if ( !nrm.empty() ) {
      load( nrm, f, true, n );
      if ( n.n != a.n ) {
        std::fprintf( stderr, "tmc2_metrics: frame %d: %llu normals for %llu source points\n", f, (unsigned long long)n.n,
                      (unsigned long long)a.n );
        return 2;
      }
    }
    double  out[3][8];
    int64_t counts[2];
    CHECK( tmc2_metrics_compute( ctx, a.xyz.data(), a.rgb.data(), a.n, b.xyz.data(), b.rgb.data(), b.n,
                                 nrm.empty() ? nullptr : n.normals.data(), double( resolution ), &out[0][0], counts ) );
    uint64_t need = 0;
    CHECK( tmc2_metrics_display( &out[0][0], a.n, b.n, counts, resolution, nrm.empty() ? 0 : 1, 9, nullptr, 0, &need ) );
    std::vector<char> text( need );
    CHECK( tmc2_metrics_display( &out[0][0], a.n, b.n, counts, resolution, nrm.empty() ? 0 : 1, 9, text.data(), need, nullptr ) );
    std::fputs( text.data(), stdout );
  }
  tmc2_ctx_destroy( ctx );
  return 0;
}
