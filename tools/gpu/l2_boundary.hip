// l2_boundary.hip -- do the kernel boundaries of OTHER streams cost a running kernel its L2 contents?
// (test infrastructure, round 6.  On this eight-XCD part the L2s are not coherent with each other: a kernel's end writes dirty
//  lines back, a kernel's start invalidates -- device-wide operations, not per stream.  Sixteen frames in flight put ~ 77 000
//  kernel boundaries per second on the chip.  If every one of them empties the L2s under the kernels that are running, every
//  kernel of the GOF runs out of HBM / the memory-side cache instead of its L2 -- which would be why a launch that takes 3 us alone
//  takes 110-160 us in flight with 6-7 % memory activity, and why the NUMBER of launches is the lever.)
//
// Victim: W workgroups x 64 lanes chase pointers through a buffer of S bytes (every hop a different 128-byte line; S fits the L2 of
// one XCD several times over), H hops each, after a warm-up lap -- the time per hop says where the lines come from.
// Alone; next to a stream of empty kernels launched back to back; next to a stream of kernels that each dirty 1 MB; next to ONE long
// ALU-only kernel (control: the command processor is busy with another queue, but there are no boundaries).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

#define CHECK( x )                                                                  \
  do {                                                                              \
    hipError_t e_ = ( x );                                                          \
    if ( e_ != hipSuccess ) {                                                       \
      fprintf( stderr, "%s: %s\n", #x, hipGetErrorString( e_ ) );                   \
      exit( 1 );                                                                    \
    }                                                                               \
  } while ( 0 )

__global__ void chaseKernel( const uint32_t* __restrict__ next, uint32_t lines, uint32_t hops, uint32_t* out ) {
  // lane l of workgroup w starts at line ( w * 64 + l ) * stride: 64 independent chains per workgroup, all hitting distinct lines
  uint32_t at = ( ( blockIdx.x * 64u + threadIdx.x ) * 977u ) % lines;
  for ( uint32_t h = 0; h < hops; ++h ) at = next[size_t( at ) * 32u];  // (32 words = one 128-byte line per entry)
  if ( at == 0xFFFFFFFFu ) out[0] = at;
}
__global__ void emptyKernel( int* p ) {
  if ( p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFF ) *p = 1;
}
__global__ void dirtyKernel( uint32_t* p, uint32_t words, uint32_t v ) {
  for ( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x ) p[i] = v;
}
__global__ void spinKernel( float* x, int iters ) {
  float v = x[threadIdx.x];
  for ( int k = 0; k < iters; ++k ) v = v * 1.0001f + 0.5f;
  x[threadIdx.x] = v;
}

int main( int argc, char** argv ) {
  const uint32_t hops = argc > 1 ? uint32_t( atoi( argv[1] ) ) : 4000;
  for ( size_t bytes : {size_t( 1 ) << 20, size_t( 2 ) << 20, size_t( 16 ) << 20, size_t( 512 ) << 20} ) {
    const uint32_t lines = uint32_t( bytes / 128 );
    std::vector<uint32_t> perm( lines ), h_next( size_t( lines ) * 32, 0 );
    std::iota( perm.begin(), perm.end(), 0u );
    std::mt19937 rng( 7 );
    std::shuffle( perm.begin(), perm.end(), rng );
    for ( uint32_t i = 0; i < lines; ++i ) h_next[size_t( perm[i] ) * 32] = perm[( i + 1 ) % lines];  // one cycle through all lines
    uint32_t *d_next, *d_out, *d_dirty;
    float*    d_spin;
    CHECK( hipMalloc( &d_next, h_next.size() * 4 ) );
    CHECK( hipMalloc( &d_out, 64 ) );
    CHECK( hipMalloc( &d_dirty, 1 << 20 ) );
    CHECK( hipMalloc( &d_spin, 4096 ) );
    CHECK( hipMemcpy( d_next, h_next.data(), h_next.size() * 4, hipMemcpyHostToDevice ) );
    CHECK( hipMemset( d_spin, 0, 4096 ) );
    hipStream_t victim, other;
    CHECK( hipStreamCreateWithFlags( &victim, hipStreamNonBlocking ) );
    CHECK( hipStreamCreateWithFlags( &other, hipStreamNonBlocking ) );
    hipEvent_t e0, e1;
    CHECK( hipEventCreate( &e0 ) );
    CHECK( hipEventCreate( &e1 ) );
    for ( uint32_t groups : {8u, 256u} ) {
      for ( int scenario = 0; scenario < 4; ++scenario ) {
        std::atomic<bool> stop{false};
        std::atomic<long> launched{0};
        std::thread       noise;
        if ( scenario == 1 )
          noise = std::thread( [&] {
            while ( !stop.load() ) {
              for ( int k = 0; k < 64; ++k ) hipLaunchKernelGGL( emptyKernel, dim3( 1 ), dim3( 64 ), 0, other, (int*)nullptr );
              launched += 64;
              hipStreamSynchronize( other );
            }
          } );
        if ( scenario == 2 )
          noise = std::thread( [&] {
            uint32_t v = 0;
            while ( !stop.load() ) {
              for ( int k = 0; k < 64; ++k ) hipLaunchKernelGGL( dirtyKernel, dim3( 64 ), dim3( 256 ), 0, other, d_dirty, 1u << 18, ++v );
              launched += 64;
              hipStreamSynchronize( other );
            }
          } );
        if ( scenario == 3 ) hipLaunchKernelGGL( spinKernel, dim3( 1 ), dim3( 64 ), 0, other, d_spin, 40000000 );
        std::this_thread::sleep_for( std::chrono::milliseconds( 20 ) );
        // warm-up lap, then the timed one
        hipLaunchKernelGGL( chaseKernel, dim3( groups ), dim3( 64 ), 0, victim, d_next, lines, hops, d_out );
        CHECK( hipStreamSynchronize( victim ) );
        const long before = launched.load();
        CHECK( hipEventRecord( e0, victim ) );
        hipLaunchKernelGGL( chaseKernel, dim3( groups ), dim3( 64 ), 0, victim, d_next, lines, hops, d_out );
        CHECK( hipEventRecord( e1, victim ) );
        CHECK( hipStreamSynchronize( victim ) );
        float ms = 0;
        CHECK( hipEventElapsedTime( &ms, e0, e1 ) );
        const long during = launched.load() - before;
        stop.store( true );
        if ( noise.joinable() ) noise.join();
        CHECK( hipStreamSynchronize( other ) );
        static const char* what[4] = {"alone", "next to a stream of empty kernels", "next to a stream of kernels that dirty 1 MB each",
                                      "next to ONE long ALU-only kernel"};
        printf( "buffer %4zu MB, %3u workgroups x 64 chains, %u hops: %8.3f ms = %6.1f ns per hop  %s", bytes >> 20, groups, hops, ms,
                1e6 * ms / hops, what[scenario] );
        if ( scenario == 1 || scenario == 2 ) printf( "  (~ %ld boundaries during the lap)", during );
        printf( "\n" );
      }
    }
    hipFree( d_next ), hipFree( d_out ), hipFree( d_dirty ), hipFree( d_spin );
    hipStreamDestroy( victim ), hipStreamDestroy( other );
  }
  return 0;
}
