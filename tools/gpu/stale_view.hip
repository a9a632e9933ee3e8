// stale_view.hip -- does a workgroup-scope (sc0) or plain load ever see data OLDER than its kernel's start on this 8-XCD part?
//
// The union-finds of S3 / S7 climb parent links with workgroup-scope loads (served by the CU's L1 / the XCD's L2: a view that
// may lag behind the other XCDs' during the kernel).  Their soundness argument needs the view to be no older than the start of
// the kernel: the parent array lives in a pool block that held the PREVIOUS round's forest, so a line that survived a kernel
// boundary would be a valid-looking link of the wrong graph.  This program tests exactly that, alone and with other streams
// keeping the chip busy (16 frames in flight):
//     W(A)  plain stores of pattern A           (kernel 1)
//     R(A)  every word read by a DIFFERENT workgroup than wrote it, sc0 and plain: warms L1 / L2 of the readers with A
//     W(B)  plain stores of pattern B, written by yet another assignment of words to workgroups
//     R(B)  sc0 / plain loads: any A seen here is a view older than the kernel's start
// Test infrastructure; not linked into the product.  Build: hipcc --offload-arch=gfx950 -O2 -o stale_view stale_view.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK( x )                                                                      \
  do {                                                                                  \
    hipError_t e_ = ( x );                                                              \
    if ( e_ != hipSuccess ) {                                                           \
      fprintf( stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString( e_ ) );     \
      exit( 2 );                                                                        \
    }                                                                                   \
  } while ( 0 )

__global__ void writeKernel( uint32_t* b, uint32_t n, uint32_t pattern, uint32_t rot ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= n ) return;
  const uint32_t i = ( t + rot * 256u * 37u ) % n;  // another workgroup (and XCD: blocks go round-robin) than last time
  b[i]             = pattern ^ i;
}

// mode 0: workgroup-scope relaxed atomic load (sc0), 1: plain load, 2: agent-scope load (sc1, the control)
__global__ void readKernel( uint32_t* b, uint32_t n, uint32_t pattern, uint32_t rot, int mode, uint32_t* stale ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= n ) return;
  const uint32_t i = ( t + rot * 256u * 101u ) % n;
  uint32_t       v;
  if ( mode == 0 )
    v = __hip_atomic_load( &b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
  else if ( mode == 1 )
    v = b[i];
  else
    v = __hip_atomic_load( &b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  if ( v != ( pattern ^ i ) ) atomicAdd( stale, 1u );
}

// in-kernel control: the reader spins until a flag says the writer kernel (another stream) is done, then reads sc0 -- what it
// may see there IS allowed to be stale (same launch, no acquire): shows that the test can detect staleness at all
__global__ void busyKernel( float* x, uint32_t n, int iters ) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ( t >= n ) return;
  float v = x[t];
  for ( int k = 0; k < iters; ++k ) v = v * 1.0001f + 0.5f;
  x[t] = v;
}

int main( int argc, char** argv ) {
  const uint32_t n     = argc > 1 ? uint32_t( atoi( argv[1] ) ) : ( 1u << 20 );
  const int      reps  = argc > 2 ? atoi( argv[2] ) : 200;
  const int      noise = argc > 3 ? atoi( argv[3] ) : 8;
  uint32_t *     b, *stale;
  CHECK( hipMalloc( &b, size_t( n ) * 4 ) );
  CHECK( hipMalloc( &stale, 64 ) );
  hipStream_t s;
  CHECK( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ) );
  std::vector<hipStream_t> ns( noise );
  std::vector<float*>      nb( noise );
  for ( int i = 0; i < noise; ++i ) {
    CHECK( hipStreamCreateWithFlags( &ns[i], hipStreamNonBlocking ) );
    CHECK( hipMalloc( &nb[i], size_t( 1 << 22 ) * 4 ) );
    CHECK( hipMemsetAsync( nb[i], 0, size_t( 1 << 22 ) * 4, ns[i] ) );
  }
  const dim3 blk( 256 ), grd( ( n + 255 ) / 256 );
  for ( int busy = 0; busy < 2; ++busy ) {
    for ( int mode = 0; mode < 3; ++mode ) {
      CHECK( hipMemsetAsync( stale, 0, 64, s ) );
      for ( int r = 0; r < reps; ++r ) {
        if ( busy )
          for ( int i = 0; i < noise; ++i )
            hipLaunchKernelGGL( busyKernel, dim3( ( 1 << 22 ) / 256 ), blk, 0, ns[i], nb[i], 1u << 22, 64 );
        const uint32_t A = 0xA0000000u + 2 * r, B = 0xB0000000u + 2 * r + 1;
        hipLaunchKernelGGL( writeKernel, grd, blk, 0, s, b, n, A, uint32_t( 3 * r ) );
        hipLaunchKernelGGL( readKernel, grd, blk, 0, s, b, n, A, uint32_t( 3 * r + 1 ), mode, stale );      // counts into [0]
        hipLaunchKernelGGL( writeKernel, grd, blk, 0, s, b, n, B, uint32_t( 3 * r + 2 ) );
        hipLaunchKernelGGL( readKernel, grd, blk, 0, s, b, n, B, uint32_t( 3 * r + 1 ), mode, stale + 1 );  // same readers as before
      }
      uint32_t h[2];
      CHECK( hipMemcpyAsync( h, stale, 8, hipMemcpyDeviceToHost, s ) );
      CHECK( hipStreamSynchronize( s ) );
      for ( int i = 0; i < noise; ++i ) CHECK( hipStreamSynchronize( ns[i] ) );
      printf( "n=%u reps=%d %s load=%s: words older than the kernel's start: after W(A) %u, after W(B) %u of %llu reads\n", n, reps,
              busy ? "busy(other streams)" : "idle", mode == 0 ? "sc0" : ( mode == 1 ? "plain" : "sc1" ), h[0], h[1],
              (unsigned long long)n * reps );
    }
  }
  return 0;
}
