// Entry points declared in tmc2hip.h whose kernels are not written yet: they fail loudly.
#include "internal.h"
using namespace tmc2;
extern "C" {
int tmc2_segmenter_segment_patches( tmc2_frame*, const tmc2_segmenter_params* ) {
  setError( "segmentPatches: not implemented yet" );
  return TMC2_E_UNSUPPORTED;
}
int tmc2_segmenter_compute( tmc2_frame*, const tmc2_segmenter_params* ) {
  setError( "PCCPatchSegmenter3::compute: not implemented yet" );
  return TMC2_E_UNSUPPORTED;
}
int tmc2_segmenter_params_check( const tmc2_segmenter_params* p ) { return p ? TMC2_OK : TMC2_E_INVALID; }
int tmc2_frame_patch_count( tmc2_frame* f ) { return f ? int( f->patches.size() ) : 0; }
int tmc2_frame_patch_pool_sizes( tmc2_frame* f, int64_t* d, int64_t* o ) {
  if ( !f || !d || !o ) return TMC2_E_INVALID;
  *d = int64_t( f->depth0.size() );
  *o = int64_t( f->occupancy.size() );
  return TMC2_OK;
}
int tmc2_frame_get_patches( tmc2_frame* f, tmc2_patch* patches, int16_t* depth0, int16_t* depth1, uint8_t* occ ) {
  if ( !f ) return TMC2_E_INVALID;
  if ( !f->patches.empty() ) memcpy( patches, f->patches.data(), f->patches.size() * sizeof( tmc2_patch ) );
  if ( !f->depth0.empty() ) {
    memcpy( depth0, f->depth0.data(), f->depth0.size() * 2 );
    memcpy( depth1, f->depth1.data(), f->depth1.size() * 2 );
  }
  if ( !f->occupancy.empty() ) memcpy( occ, f->occupancy.data(), f->occupancy.size() );
  return TMC2_OK;
}
}
