// launch_rate.hip -- how many kernel launches per second does the runtime take from T host threads, one stream each?
// (test infrastructure: settles whether the 32-frame GOF is bound by the host-side launch path rather than by the GPU)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void emptyKernel( int* p ) {
  if ( p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFF ) *p = 1;
}
__global__ void busyKernel( float* x, int iters ) {
  float v = x[threadIdx.x];
  for ( int k = 0; k < iters; ++k ) v = v * 1.0001f + 0.5f;
  x[threadIdx.x] = v;
}

int main( int argc, char** argv ) {
  const int launches = argc > 1 ? atoi( argv[1] ) : 20000;
  const int blocks   = argc > 2 ? atoi( argv[2] ) : 1;
  const int iters    = argc > 3 ? atoi( argv[3] ) : 0;  // > 0: each kernel runs ~ iters * 4 cycles (one workgroup per block)
  for ( int T : {1, 2, 4, 8, 16} ) {
    std::vector<hipStream_t> s( T );
    std::vector<float*>      buf( T );
    for ( int t = 0; t < T; ++t ) {
      hipStreamCreateWithFlags( &s[t], hipStreamNonBlocking );
      hipMalloc( &buf[t], 4096 );
      hipMemset( buf[t], 0, 4096 );
    }
    hipDeviceSynchronize();
    const auto               t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for ( int t = 0; t < T; ++t )
      th.emplace_back( [&, t] {
        for ( int i = 0; i < launches; ++i ) {
          if ( iters )
            hipLaunchKernelGGL( busyKernel, dim3( blocks ), dim3( 256 ), 0, s[t], buf[t], iters );
          else
            hipLaunchKernelGGL( emptyKernel, dim3( blocks ), dim3( 256 ), 0, s[t], (int*)nullptr );
        }
        hipStreamSynchronize( s[t] );
      } );
    for ( auto& x : th ) x.join();
    const double sec = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
    printf( "threads %2d: %d launches each (%d blocks, iters %d): %.0f launches/s total, %.2f us per launch per stream\n", T, launches, blocks,
            iters, T * launches / sec, 1e6 * sec / launches );
    for ( int t = 0; t < T; ++t ) hipStreamDestroy( s[t] ), hipFree( buf[t] );
  }
  return 0;
}
