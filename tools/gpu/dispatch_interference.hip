// dispatch_interference.hip -- what does a tiny kernel of one stream wait for while other streams run big grids?
// (test infrastructure, round 6.)  Victim stream: a chain of one-workgroup kernels of ~ 5 us each, timed per launch (events around
// 200 of them).  Noise: K other streams, each running back-to-back kernels of G workgroups x 256 threads that each do ~ T us of
// dependent ALU work and nothing else (no memory traffic, no LDS): G = one wave of the chip (2 048 workgroups), a fraction, or a
// multiple of it.  If the victim's time per launch grows with G at constant K, the workgroup dispatcher is what the streams share;
// if it only grows with K, it is the number of hardware queues in service.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void spin( float* x, int iters ) {
  float v = x[threadIdx.x & 63];
  for ( int k = 0; k < iters; ++k ) v = v * 1.0001f + 0.5f;
  if ( v == 12345.678f ) x[0] = v;
}

int main() {
  float* d;
  hipMalloc( &d, 4096 );
  hipMemset( d, 0, 4096 );
  hipStream_t victim;
  hipStreamCreateWithFlags( &victim, hipStreamNonBlocking );
  hipEvent_t e0, e1;
  hipEventCreate( &e0 );
  hipEventCreate( &e1 );
  const int tinyIters = 3000;  // ~ 5 us
  struct Case { int K, G, iters; };
  const Case cases[] = {{0, 0, 0},       {1, 256, 12000},  {1, 2048, 12000},  {1, 16384, 12000}, {4, 256, 12000},  {4, 2048, 12000},
                        {4, 16384, 12000}, {8, 256, 12000},  {8, 2048, 12000},  {8, 16384, 12000}, {15, 256, 12000}, {15, 2048, 12000},
                        {15, 16384, 3000}, {15, 1, 12000},   {15, 1, 60000},    {7, 1, 60000}};
  for ( const Case& c : cases ) {
    std::vector<hipStream_t> noise( c.K );
    for ( auto& s : noise ) hipStreamCreateWithFlags( &s, hipStreamNonBlocking );
    std::atomic<bool>        stop{false};
    std::vector<std::thread> th;
    for ( int k = 0; k < c.K; ++k )
      th.emplace_back( [&, k] {
        while ( !stop.load() ) {
          for ( int i = 0; i < 8; ++i ) hipLaunchKernelGGL( spin, dim3( c.G ), dim3( 256 ), 0, noise[k], d, c.iters );
          hipStreamSynchronize( noise[k] );
        }
      } );
    std::this_thread::sleep_for( std::chrono::milliseconds( 30 ) );
    const int launches = 200;
    hipEventRecord( e0, victim );
    for ( int i = 0; i < launches; ++i ) hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, victim, d, tinyIters );
    hipEventRecord( e1, victim );
    hipStreamSynchronize( victim );
    float ms = 0;
    hipEventElapsedTime( &ms, e0, e1 );
    stop.store( true );
    for ( auto& t : th ) t.join();
    for ( auto& s : noise ) hipStreamSynchronize( s ), hipStreamDestroy( s );
    // how long one noise kernel takes alone (for the table)
    float alone = 0;
    if ( c.K ) {
      hipEventRecord( e0, victim );
      hipLaunchKernelGGL( spin, dim3( c.G ), dim3( 256 ), 0, victim, d, c.iters );
      hipEventRecord( e1, victim );
      hipStreamSynchronize( victim );
      hipEventElapsedTime( &alone, e0, e1 );
    }
    printf( "%2d other streams x kernels of %5d workgroups (%7.1f us each alone): the one-workgroup chain takes %7.1f us per launch\n", c.K, c.G,
            1e3 * alone, 1e3 * ms / launches );
  }
  return 0;
}
