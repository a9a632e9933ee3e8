// integration/tmc2_encode_gof.cpp -- a native host front end over the C-ABI of libtmc2hip.so (no reference code, no Python):
// the image-generation half of PccAppEncoder for one GOF, with an identity video codec.
//
//   tmc2_encode_gof --in frame_%04d.ply --start 1051 --frames 32 --out /tmp/gof [--condition ai|ld|ra] [--devices 0-7]
//                   [--workers 16] [--iterations 10] [--voxel 4] [--bits 10] [--precision 4] [--min-width 1280]
//                   [--min-height 1280] [--repeat N] [--no-tail] [--no-files]
//
// SEVERAL GPUs from ONE process (--devices 0-7 / 0,1,2 / 0,0: the frame loop PCCEncoder::encode runs as tbb::parallel_for,
// PCCEncoder.cpp:4729-4750, sharded over the node): frame f lives on device f mod D for its whole life, `workers` host threads
// and contexts per device; the only things the frames of a GOF share are the axis weights (S0, frame 0: 24 bytes), the common
// canvas size (a maximum over the packed heights) and -- under the low-delay / random-access conditions -- the packing chain,
// which runs on the host over every frame's patch records wherever the frame lives.  Every GPU copies its frames' finished
// canvases straight into page-locked host memory (tmc2_host_alloc: portable, one node = one address space) over its own PCIe
// link, where the video encoder reads them; no frame data crosses xGMI.  (One process per GPU with RCCL, the other way to run
// the node: bench.py / tmc2_amd/gof.py.)
//
// Per frame (one host thread + one tmc2_ctx per in-flight frame and device, as the reference runs one TBB task per frame):
//   PLY ingest -> S1-S9 patch generation -> packing (all-intra: per frame; low delay / random access: the chained packer, then
//   the global patch allocation over the GOF) -> occupancy / geometry canvases -> reconstruction, colour transfer, attribute
//   canvases -> I420 attribute frames (what the video encoder reads) -> identity codec -> 16-bit 4:4:4 -> post-reconstruction
//   tail -> reconstructed PLY + conformance checksum.
// Written to <out>_occupancy_WxH.yuv (8-bit 4:0:0 samples), <out>_geometry_WxH_16bit.yuv (two maps per frame, luma only),
// <out>_attribute_WxH_8bit_p420.yuv (two maps per frame), <out>_rec_%04d.ply, <out>_checksums.txt, <out>.checksum (PCCChecksum::write).
// Exits non-zero with the library's message when no MI355X is visible -- there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fstream>
#include <map>
#include <pthread.h>
#include <sched.h>

#include "tmc2hip.h"

namespace {
struct Options {
  std::string in, out = "gof";
  std::string condition = "ai";
  int         start = 0, frames = 1, workers = 8, iterations = 10, voxel = 4, bits = 10, precision = 4, minW = 1280, minH = 1280;
  int         repeat = 1;
  bool        tail = true, files = true;
  std::vector<int> devices{0};
};
// "0-7", "0,1,2", "3", "0,0" (the same device twice: two shards on one GPU, for trying the sharded path on a one-GPU box)
bool parseDevices( const std::string& spec, std::vector<int>& out ) {
  out.clear();
  size_t at = 0;
  while ( at < spec.size() ) {
    size_t end = spec.find( ',', at );
    if ( end == std::string::npos ) end = spec.size();
    const std::string item = spec.substr( at, end - at );
    const size_t      dash = item.find( '-' );
    if ( item.empty() ) return false;
    if ( dash == std::string::npos ) {
      out.push_back( std::atoi( item.c_str() ) );
    } else {
      const int a = std::atoi( item.substr( 0, dash ).c_str() ), b = std::atoi( item.substr( dash + 1 ).c_str() );
      if ( b < a ) return false;
      for ( int d = a; d <= b; ++d ) out.push_back( d );
    }
    at = end + 1;
  }
  return !out.empty();
}
[[noreturn]] void die( const std::string& what ) {
  std::fprintf( stderr, "tmc2_encode_gof: %s: %s\n", what.c_str(), tmc2_last_error() );
  std::exit( 2 );
}
#define CHECK( call )                    \
  do {                                   \
    if ( ( call ) != TMC2_OK ) die( #call ); \
  } while ( 0 )

void usage() {
  std::puts( "usage: tmc2_encode_gof --in frame_%04d.ply [--start N] [--frames N] [--out prefix] [--condition ai|ld|ra]\n"
             "                       [--devices 0-7 | --device N] [--workers N (per device)] [--iterations N] [--voxel N] [--bits N]\n"
             "                       [--precision N] [--min-width N] [--min-height N] [--repeat N] [--no-tail] [--no-files]" );
}
bool parse( int argc, char** argv, Options& o ) {
  for ( int i = 1; i < argc; ++i ) {
    const std::string a = argv[i];
    auto              next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if ( a == "--in" ) o.in = next();
    else if ( a == "--out" ) o.out = next();
    else if ( a == "--condition" ) o.condition = next();
    else if ( a == "--start" ) o.start = std::atoi( next() );
    else if ( a == "--frames" ) o.frames = std::atoi( next() );
    else if ( a == "--device" || a == "--devices" ) {
      if ( !parseDevices( next(), o.devices ) ) return false;
    }
    else if ( a == "--voxel" ) o.voxel = std::atoi( next() );
    else if ( a == "--repeat" ) o.repeat = std::max( 1, std::atoi( next() ) );
    else if ( a == "--no-tail" ) o.tail = false;
    else if ( a == "--no-files" ) o.files = false;
    else if ( a == "--workers" ) o.workers = std::atoi( next() );
    else if ( a == "--iterations" ) o.iterations = std::atoi( next() );
    else if ( a == "--bits" ) o.bits = std::atoi( next() );
    else if ( a == "--precision" ) o.precision = std::atoi( next() );
    else if ( a == "--min-width" ) o.minW = std::atoi( next() );
    else if ( a == "--min-height" ) o.minH = std::atoi( next() );
    else if ( a == "--help" || a == "-h" ) return false;
    else {
      std::fprintf( stderr, "unknown option %s\n", a.c_str() );
      return false;
    }
  }
  return !o.in.empty() && o.frames > 0 && ( o.condition == "ai" || o.condition == "ld" || o.condition == "ra" );
}

// the CTC lossy settings (cfg/common/ctc-common.cfg + sequence cfg), as tmc2_amd.ctc_params
tmc2_segmenter_params ctcParams( int iterations, int voxel, int bits3D, const double w[3] ) {
  tmc2_segmenter_params p{};
  p.nnNormalEstimation = 16, p.normalOrientation = 1, p.gridBasedRefineSegmentation = 1, p.maxNNCountRefineSegmentation = 1024;
  p.iterationCountRefineSegmentation = iterations, p.voxelDimensionRefineSegmentation = voxel, p.searchRadiusRefineSegmentation = 192;
  p.occupancyResolution = 16, p.enablePatchSplitting = 1, p.maxPatchSize = 1024, p.quantizerSizeX = 16, p.quantizerSizeY = 16;
  p.minPointCountPerCCPatchSegmentation = 16, p.maxNNCountPatchSegmentation = 16, p.surfaceThickness = 4, p.mapCountMinus1 = 1;
  p.minLevel = 64, p.maxAllowedDepth = 255, p.geometryBitDepth2D = 8, p.geometryBitDepth3D = bits3D;
  p.maxAllowedDist2RawPointsDetection = 9, p.maxAllowedDist2RawPointsSelection = 1, p.lambdaRefineSegmentation = 3;
  for ( int c = 0; c < 3; ++c ) p.weightNormal[c] = w[c];
  return p;
}

struct Frame {
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  uint64_t             n = 0;
  tmc2_frame*          f = nullptr;
  int32_t              height = 0;
};

// One core per slot, spread over the last-level caches (one entry per CCD on EPYC; SMT siblings dropped): the host-resident steps
// of a frame (the orientation walk, the packers) are cache- and latency-bound, so sixteen of them should not share two L3s.
// (What tmc2_amd/gof.py does for bench.py's workers.)  Empty when the topology cannot be read: the threads stay unpinned.
std::vector<int> coresByCacheDomain() {
  cpu_set_t allowed;
  CPU_ZERO( &allowed );
  if ( sched_getaffinity( 0, sizeof( allowed ), &allowed ) != 0 ) return {};
  std::map<std::string, std::vector<int>> domains;
  auto firstOf = []( const std::string& list ) { return std::atoi( list.c_str() ); };  // "0,128" / "0-7,128-135" -> 0
  for ( int cpu = 0; cpu < CPU_SETSIZE; ++cpu ) {
    if ( !CPU_ISSET( cpu, &allowed ) ) continue;
    const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string( cpu ) + "/";
    std::ifstream     sib( base + "topology/thread_siblings_list" ), l3( base + "cache/index3/shared_cpu_list" );
    std::string       s, d;
    if ( !std::getline( sib, s ) || !std::getline( l3, d ) ) return {};
    if ( firstOf( s ) != cpu ) continue;  // the second hardware thread of a core already listed
    domains[d].push_back( cpu );
  }
  std::vector<int> order;  // round-robin over the domains
  for ( size_t k = 0;; ++k ) {
    bool any = false;
    for ( auto& kv : domains )
      if ( k < kv.second.size() ) order.push_back( kv.second[k] ), any = true;
    if ( !any ) break;
  }
  return order;
}
void pinThisThread( const std::vector<int>& cores, int slot ) {
  if ( cores.empty() ) return;
  cpu_set_t one;
  CPU_ZERO( &one );
  CPU_SET( cores[size_t( slot ) % cores.size()], &one );
  (void)pthread_setaffinity_np( pthread_self(), sizeof( one ), &one );
}

// run fn( frameIndex ) for every frame on `workers` threads; worker w owns context w
template <typename Fn>
void forFrames( int frames, int workers, Fn fn ) {
  std::atomic<int>         next( 0 );
  std::vector<std::thread> pool;
  for ( int w = 0; w < workers; ++w )
    pool.emplace_back( [&, w] {
      for ( int i = next++; i < frames; i = next++ ) fn( i, w );
    } );
  for ( auto& t : pool ) t.join();
}
}  // namespace

// page-locked host memory the finished canvases land in (tmc2_host_alloc)
template <typename T>
struct Pinned {
  T*     p = nullptr;
  size_t n = 0;
  void   resize( size_t count ) {
    if ( count == n ) return;
    tmc2_host_free( p );
    p = nullptr, n = 0;
    void* q = nullptr;
    if ( count ) {
      if ( tmc2_host_alloc( count * sizeof( T ), &q ) != TMC2_OK ) die( "tmc2_host_alloc" );
      p = static_cast<T*>( q ), n = count;
    }
  }
  ~Pinned() { tmc2_host_free( p ); }
};

int main( int argc, char** argv ) {
  Options o;
  if ( !parse( argc, argv, o ) ) {
    usage();
    return 1;
  }
  // one hardware queue per in-flight frame: the HIP runtime multiplexes streams onto 4 hardware queues by default, and streams that
  // share a queue serialise behind each other (must be set before the runtime initialises; a setting of the user wins)
  setenv( "GPU_MAX_HW_QUEUES", "16", 0 );
  // slots: ( device shard d, worker w ) -> one host thread + one context (a HIP stream + allocator each); frame i lives on
  // shard i % D, worker ( i / D ) % workers of that shard, for its whole life
  const int D       = int( o.devices.size() );
  const int workers = std::max( 1, std::min( o.workers, ( o.frames + D - 1 ) / D ) );
  const int slots   = D * workers;
  auto      slotOf  = [&]( int i ) { return ( i % D ) * workers + ( i / D ) % workers; };
  std::vector<tmc2_ctx*> ctx( size_t( slots ), nullptr );
  {  // a slot's context is created by a thread pinned to the slot's core: its page-locked staging lands on that core's NUMA node
    const std::vector<int>   cores = coresByCacheDomain();
    std::vector<std::thread> makers;
    for ( int s = 0; s < slots; ++s )
      makers.emplace_back( [&, s] {
        pinThisThread( cores, s );
        CHECK( tmc2_ctx_create( o.devices[size_t( s / workers )], &ctx[size_t( s )] ) );
      } );
    for ( auto& t : makers ) t.join();
  }
  tmc2_set_host_parallelism( 16 );
  for ( tmc2_ctx* c : ctx ) {  // options of THESE contexts (nothing process-wide: another encoder of the process keeps its own)
    CHECK( tmc2_ctx_set_option( c, "REFINE_OVERLAP", workers <= 4 ? "1" : "0" ) );  // few frames in flight per device: shorten a frame's chain
    CHECK( tmc2_ctx_set_option( c, "KDTREE_HOST", "0" ) );  // device trees (the device build beats the host build at every number of frames in flight)
  }
  // ... and with few frames in flight the second half of a device's frames start 2 ms after the first: frames that start together
  // reach S3's host walk together and leave the GPU idle meanwhile (profiles/r06_rank_stagger.txt)
  if ( workers >= 2 && workers <= 4 )
    for ( size_t s = 0; s < ctx.size(); ++s )
      if ( int( s % size_t( workers ) ) >= ( workers + 1 ) / 2 ) CHECK( tmc2_ctx_set_option( ctx[s], "FRAME_START_DELAY_US", "2000" ) );

  std::vector<Frame> gof( size_t( o.frames ) );
  // ingest + upload (all later calls on a frame are ordered on its slot's context)
  forFrames( o.frames, std::min( o.frames, 16 ), [&]( int i, int ) {
    char path[4096];
    std::snprintf( path, sizeof( path ), o.in.c_str(), o.start + i );
    Frame& fr = gof[size_t( i )];
    int    hasColors = 0;
    CHECK( tmc2_ply_info( path, 0, &fr.n, &hasColors, nullptr ) );
    if ( !hasColors || fr.n == 0 ) {
      std::fprintf( stderr, "tmc2_encode_gof: %s has no colours or no points\n", path );
      std::exit( 2 );
    }
    fr.xyz.resize( 3 * fr.n ), fr.rgb.resize( 3 * fr.n );
    CHECK( tmc2_ply_read( path, fr.xyz.data(), fr.rgb.data(), nullptr, fr.n, 4, &fr.n ) );
  } );
  {  // the sequence's bounds are known now: every context reserves its worst case before the first frame arrives (no hipMalloc --
     // a device-wide synchronisation under all frames in flight -- inside the GOFs; a context that cannot is left to grow on demand)
    uint64_t most = 0;
    for ( const Frame& fr : gof ) most = std::max<uint64_t>( most, fr.n );
    for ( tmc2_ctx* c : ctx )
      if ( tmc2_ctx_reserve( c, most, o.voxel, o.bits + 1, o.minW, std::max( o.minW, o.minH ) ) != TMC2_OK )
        std::fprintf( stderr, "tmc2_encode_gof: no reservation (%s): the pool grows on demand\n", tmc2_last_error() );
  }
  for ( int i = 0; i < o.frames; ++i )
    CHECK( tmc2_frame_create( ctx[size_t( slotOf( i ) )], gof[size_t( i )].xyz.data(), gof[size_t( i )].rgb.data(), gof[size_t( i )].n,
                              &gof[size_t( i )].f ) );

  std::vector<std::thread> pool;
  const std::vector<int>   cores = coresByCacheDomain();
  auto perFrame = [&]( auto fn ) {  // frames of one slot in order, slots in parallel
    pool.clear();
    for ( int sl = 0; sl < slots; ++sl )
      pool.emplace_back( [&, sl] {
        pinThisThread( cores, sl );
        for ( int i = 0; i < o.frames; ++i )
          if ( slotOf( i ) == sl ) fn( gof[size_t( i )], i );
      } );
    for ( auto& t : pool ) t.join();
  };
  const bool   chained = o.condition != "ai";
  const size_t p       = size_t( o.precision );
  int32_t      W = 0, H = 0;
  std::vector<Pinned<uint8_t>>  occVideo( size_t( o.frames ) ), attribute( size_t( o.frames ) ), i420( size_t( o.frames ) );
  std::vector<Pinned<uint16_t>> geometry( size_t( o.frames ) );

  // One pass of the path S0-S22 over the GOF: everything from the k-d trees to the finished canvases in host memory.
  auto encodeGof = [&]() {
    for ( auto& fr : gof ) CHECK( tmc2_frame_reset( fr.f ) );
    // S0 once per GOF on frame 0 (what a sharded run broadcasts: three doubles), then S1-S9 per frame
    double w[3];
    CHECK( tmc2_weight_normal( gof[0].f, o.bits + 1, 0.6, w ) );
    const tmc2_segmenter_params params = ctcParams( o.iterations, o.voxel, o.bits + 1, w );
    perFrame( [&]( Frame& fr, int ) {
      CHECK( tmc2_segmenter_compute( fr.f, &params ) );
      if ( !chained ) CHECK( tmc2_encoder_pack_flexible( fr.f, o.minW, 2, 1.0, &fr.height ) );
    } );
    int32_t tileW = o.minW, gofH = 0;
    if ( chained ) {  // a sequential chain over the GOF: microseconds per frame on the host, whatever device a frame lives on
      CHECK( tmc2_encoder_pack_flexible( gof[0].f, o.minW, 2, 1.0, &gof[0].height ) );
      for ( int i = 1; i < o.frames; ++i )
        CHECK( tmc2_encoder_pack_spatial_consistency( gof[size_t( i )].f, gof[size_t( i - 1 )].f, o.minW, 2, 1.0, &gof[size_t( i )].height ) );
      if ( o.condition == "ra" ) {
        std::vector<tmc2_frame*> fs;
        for ( auto& fr : gof ) fs.push_back( fr.f );
        std::vector<int32_t> widths( size_t( o.frames ) ), heights( size_t( o.frames ) );
        CHECK( tmc2_encoder_global_patch_allocation( fs.data(), o.frames, o.minW, o.minH, widths.data(), heights.data() ) );
        for ( int i = 0; i < o.frames; ++i ) gof[size_t( i )].height = heights[size_t( i )];
      }
      for ( auto& fr : gof ) {
        int32_t pw = 0;
        CHECK( tmc2_frame_get_packed_size( fr.f, &pw, nullptr ) );
        tileW = std::max( tileW, pw );
      }
    }
    for ( auto& fr : gof ) gofH = std::max( gofH, fr.height );  // (the all-reduce(max) of a run with one process per GPU)
    CHECK( tmc2_encoder_canvas_size( &gofH, 1, tileW, o.minW, o.minH, &W, &H ) );
    const size_t area = size_t( W ) * H;
    perFrame( [&]( Frame& fr, int i ) {
      CHECK( tmc2_encoder_generate_geometry_images( fr.f, W, H, o.precision ) );
      occVideo[size_t( i )].resize( area / ( p * p ) );
      geometry[size_t( i )].resize( 2 * area );
      attribute[size_t( i )].resize( 6 * area );
      // (a real encoder codes occupancy + geometry here and hands the decoded frames back: tmc2_frame_set_decoded_geometry)
      CHECK( tmc2_encoder_generate_attribute_images( fr.f ) );
      // the finished canvases of THIS frame leave its GPU as soon as they exist: DMA into page-locked host memory
      CHECK( tmc2_frame_get_geometry_images( fr.f, nullptr, occVideo[size_t( i )].p, nullptr, geometry[size_t( i )].p,
                                             geometry[size_t( i )].p + area ) );
      CHECK( tmc2_frame_get_attribute_images( fr.f, attribute[size_t( i )].p ) );
    } );
  };

  encodeGof();
  std::printf( "GOF canvas %d x %d, %d frames on %d device shard(s) x %d worker(s), condition %s\n", W, H, o.frames, D, workers,
               o.condition.c_str() );
  if ( o.repeat > 1 ) {  // throughput of the path from this (native) host: repeat the pass over the resident input
    const auto t0 = std::chrono::steady_clock::now();
    for ( int r = 1; r < o.repeat; ++r ) encodeGof();
    const double dt = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
    std::printf( "{\"driver\": \"tmc2_encode_gof\", \"frames_per_s\": %.2f, \"ms_per_gof\": %.2f, \"frames\": %d, \"passes\": %d, "
                 "\"device_shards\": %d, \"workers_per_shard\": %d, \"condition\": \"%s\", \"canvas\": [%d, %d]}\n",
                 double( o.frames ) * ( o.repeat - 1 ) / dt, 1e3 * dt / ( o.repeat - 1 ), o.frames, o.repeat - 1, D, workers,
                 o.condition.c_str(), W, H );
  }

  const size_t area = size_t( W ) * H, frame420 = area * 3 / 2;
  std::vector<std::string> checksums( size_t( o.frames ) );
  std::vector<uint8_t>     digests( size_t( o.frames ) * 16 );
  if ( o.tail )
    perFrame( [&]( Frame& fr, int i ) {
      i420[size_t( i )].resize( 2 * frame420 );
      CHECK( tmc2_encoder_attribute_to_yuv420( fr.f, 4, i420[size_t( i )].p ) );
      // (attribute video codec here) -- identity: the frames come straight back
      CHECK( tmc2_codec_set_decoded_attribute_yuv420( fr.f, i420[size_t( i )].p, 0 ) );
      CHECK( tmc2_codec_identify_boundary_points( fr.f ) );
      CHECK( tmc2_codec_color_point_cloud( fr.f, nullptr ) );
      CHECK( tmc2_codec_smooth_point_cloud_postprocess( fr.f, 8, 64.0 ) );
      CHECK( tmc2_codec_transfer_colors_16bit_bp( fr.f ) );
      CHECK( tmc2_codec_convert_yuv16_to_rgb8( fr.f ) );
      const size_t         M = size_t( tmc2_frame_recon_count( fr.f ) );
      std::vector<int16_t> xyz( 3 * M );
      std::vector<uint8_t> rgb( 3 * M );
      CHECK( tmc2_frame_get_post_reconstruction( fr.f, xyz.data(), nullptr, rgb.data(), nullptr ) );
      if ( o.files ) {
        char path[4096];
        std::snprintf( path, sizeof( path ), "%s_rec_%04d.ply", o.out.c_str(), o.start + i );
        CHECK( tmc2_ply_write( path, xyz.data(), rgb.data(), nullptr, M, 1 ) );
      }
      uint8_t* digest = digests.data() + 16 * size_t( i );
      CHECK( tmc2_point_set_checksum( xyz.data(), rgb.data(), M, 0, digest ) );
      char hex[33];
      for ( int k = 0; k < 16; ++k ) std::snprintf( hex + 2 * k, 3, "%02x", digest[k] );
      checksums[size_t( i )] = hex;
    } );

  auto writeAll = [&]( const std::string& name, auto&& writer ) {
    FILE* fp = std::fopen( name.c_str(), "wb" );
    if ( !fp ) {
      std::fprintf( stderr, "tmc2_encode_gof: cannot create %s\n", name.c_str() );
      std::exit( 2 );
    }
    writer( fp );
    std::fclose( fp );
  };
  if ( o.files ) {
    const std::string dims = std::to_string( W ) + "x" + std::to_string( H );
    writeAll( o.out + "_occupancy_" + std::to_string( W / o.precision ) + "x" + std::to_string( H / o.precision ) + "_8bit_p400.yuv",
              [&]( FILE* fp ) { for ( auto& v : occVideo ) std::fwrite( v.p, 1, v.n, fp ); } );
    writeAll( o.out + "_geometry_" + dims + "_16bit_p400.yuv",
              [&]( FILE* fp ) { for ( auto& v : geometry ) std::fwrite( v.p, 2, v.n, fp ); } );
    if ( o.tail ) {
      writeAll( o.out + "_attribute_" + dims + "_8bit_p420.yuv",
                [&]( FILE* fp ) { for ( auto& v : i420 ) std::fwrite( v.p, 1, v.n, fp ); } );
      writeAll( o.out + "_checksums.txt", [&]( FILE* fp ) {
        for ( int i = 0; i < o.frames; ++i ) std::fprintf( fp, "%04d %s\n", o.start + i, checksums[size_t( i )].c_str() );
      } );
      CHECK( tmc2_checksum_file_write( ( o.out + ".checksum" ).c_str(), digests.data(), uint64_t( o.frames ) ) );  // as PccAppEncoder
    }
  }
  for ( auto& fr : gof ) tmc2_frame_destroy( fr.f );
  for ( auto& c : ctx ) tmc2_ctx_destroy( c );
  return 0;
}
