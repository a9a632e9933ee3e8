// atomic_rate.hip -- what does a returning atomicAdd on ONE word cost when W waves issue one each (list reservations)?
// (test infrastructure: sizes the list appends of the S5 closure kernel)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void reserveKernel( unsigned* counter, unsigned* out, int spread ) {
  const unsigned wave = ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6;
  if ( ( threadIdx.x & 63 ) == 0 ) out[wave] = atomicAdd( &counter[spread ? ( wave % unsigned( spread ) ) * 32 : 0], 1u );
}
__global__ void noAtomicKernel( unsigned* counter, unsigned* out ) {
  const unsigned wave = ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6;
  if ( ( threadIdx.x & 63 ) == 0 ) out[wave] = counter[0] + wave;
}

int main() {
  unsigned *counter, *out;
  hipMalloc( &counter, 4096 * 32 );
  hipMalloc( &out, 4 << 20 );
  hipMemset( counter, 0, 4096 * 32 );
  hipEvent_t a, b;
  hipEventCreate( &a ), hipEventCreate( &b );
  for ( int blocks : {128, 512, 2048, 8192, 32768} ) {
    for ( int spread : {-1, 0, 8, 64} ) {
      float best = 1e9f;
      for ( int rep = 0; rep < 20; ++rep ) {
        hipEventRecord( a );
        if ( spread < 0 )
          hipLaunchKernelGGL( noAtomicKernel, dim3( blocks ), dim3( 256 ), 0, 0, counter, out );
        else
          hipLaunchKernelGGL( reserveKernel, dim3( blocks ), dim3( 256 ), 0, 0, counter, out, spread );
        hipEventRecord( b );
        hipEventSynchronize( b );
        float ms;
        hipEventElapsedTime( &ms, a, b );
        if ( ms < best ) best = ms;
      }
      printf( "%6d waves, %s: %.1f us (%.2f ns per wave)\n", blocks * 4,
              spread < 0 ? "no atomic      " : ( spread == 0 ? "one word       " : ( spread == 8 ? "8 words (lines)" : "64 words       " ) ),
              best * 1e3, best * 1e6 / ( blocks * 4 ) );
    }
  }
  return 0;
}
