// cand_sort.h -- libstdc++'s std::sort replayed on the device, for the candidate lists of the colour transfers.
#pragma once
#include <hip/hip_runtime.h>

namespace tmc2 {

// The reference orders a target's backward candidates with std::sort( ..., dist < dist ): libstdc++'s introsort
// (median-of-3 quicksort down to 16-element runs, then one insertion sort).  It is not stable, so for equal
// distances the outcome depends on the algorithm's exact moves -- reproduced here move for move on the list that
// is first brought into source-index order (the order in which the reference appended the candidates).
struct CandSort {
  uint2* a;
  __device__ bool less( int i, int j ) const { return a[i].x < a[j].x; }
  __device__ void swap( int i, int j ) const {
    const uint2 t = a[i];
    a[i]          = a[j];
    a[j]          = t;
  }
  __device__ void unguardedLinearInsert( int last ) const {
    const uint2 val  = a[last];
    int         next = last - 1;
    while ( val.x < a[next].x ) {
      a[last] = a[next];
      last    = next;
      --next;
    }
    a[last] = val;
  }
  __device__ void insertionSort( int first, int last ) const {
    for ( int i = first + 1; i < last; ++i ) {
      if ( a[i].x < a[first].x ) {
        const uint2 val = a[i];
        for ( int k = i; k > first; --k ) a[k] = a[k - 1];
        a[first] = val;
      } else {
        unguardedLinearInsert( i );
      }
    }
  }
  __device__ void moveMedianToFirst( int result, int x, int y, int z ) const {
    if ( less( x, y ) ) {
      if ( less( y, z ) )
        swap( result, y );
      else if ( less( x, z ) )
        swap( result, z );
      else
        swap( result, x );
    } else if ( less( x, z ) )
      swap( result, x );
    else if ( less( y, z ) )
      swap( result, z );
    else
      swap( result, y );
  }
  __device__ int unguardedPartition( int first, int last, int pivot ) const {
    for ( ;; ) {
      while ( less( first, pivot ) ) ++first;
      --last;
      while ( less( pivot, last ) ) --last;
      if ( !( first < last ) ) return first;
      swap( first, last );
      ++first;
    }
  }
  // returns false if the depth limit was reached (libstdc++ would switch to heapsort; not reproduced)
  __device__ bool sort( int n ) const {
    if ( n < 2 ) return true;
    int lg = 0;
    while ( ( n >> ( lg + 1 ) ) > 0 ) ++lg;
    // __introsort_loop: recursion on the right part, iteration on the left -> explicit stack of right parts
    int  stackFirst[64], stackLast[64], stackDepth[64], sp = 0;
    int  first = 0, last = n, depth = 2 * lg;
    bool ok = true;
    for ( ;; ) {
      while ( last - first > 16 ) {
        if ( depth == 0 ) {
          ok = false;
          break;
        }
        --depth;
        const int mid = first + ( last - first ) / 2;
        moveMedianToFirst( first, first + 1, mid, last - 1 );
        const int cut = unguardedPartition( first + 1, last, first );
        // the reference recurses into [cut,last) FIRST and then continues with [first,cut); the two ranges are
        // disjoint, so the order in which they are processed does not change the result
        if ( sp < 64 ) {
          stackFirst[sp] = cut, stackLast[sp] = last, stackDepth[sp] = depth;
          ++sp;
        } else {
          ok = false;
        }
        last = cut;
      }
      if ( sp == 0 ) break;
      --sp;
      first = stackFirst[sp], last = stackLast[sp], depth = stackDepth[sp];
    }
    // __final_insertion_sort
    if ( n > 16 ) {
      insertionSort( 0, 16 );
      for ( int i = 16; i < n; ++i ) unguardedLinearInsert( i );
    } else {
      insertionSort( 0, n );
    }
    return ok;
  }
};

}  // namespace tmc2
