// integration/tmc2hip_adaptor.h -- the reference-side binding of libtmc2hip.so, as compilable code.
//
// INTEGRATION.md describes the adaptors a TMC2 maintainer adds at each seam; this is the same code, kept compiling against
// the reference's own headers (oracle/Makefile builds it into oracle/_ref/libtmc2adaptor.so wherever the reference tree is
// present) and exercised by the tests: the conversions between the reference's containers (PCCPointSet3, PCCPatch) and the
// plain buffers of include/tmc2hip.h are checked on the CPU against what the reference's own segmenter produces.
// It is NOT part of the product library and holds no algorithm: flatten, call the C entry point, write back.
#pragma once
#include <cstdint>
#include <vector>

#include "PCCCommon.h"
#include "PCCPatch.h"
#include "PCCPatchSegmenter.h"
#include "PCCPointSet.h"
#include "tmc2hip.h"

namespace tmc2hip {

// PCCPointSet3 -> xyz int16[n][3], rgb uint8[n][3] (rgb empty if the cloud has no colours)
void flatten( const pcc::PCCPointSet3& cloud, std::vector<int16_t>& xyz, std::vector<uint8_t>& rgb );

// PCCPatchSegmenter3Parameters -> tmc2_segmenter_params; false if the parameter set uses something the library does not
// mirror (the caller then keeps the reference's own body)
bool toParams( const pcc::PCCPatchSegmenter3Parameters& params, tmc2_segmenter_params& out );

// patch records + pools (tmc2_frame_get_patches) -> the PCCPatch objects PCCPatchSegmenter3::compute would have appended
void toPCCPatches( const tmc2_patch* records, int count, const int16_t* depth0, const int16_t* depth1, const uint8_t* occupancy,
                   size_t occupancyResolution, size_t frameIndex, std::vector<pcc::PCCPatch>& patches );

// PCCPatch list (e.g. a decoder's, in list order) -> patch records for tmc2_decoder_frame_create
void toRecords( const std::vector<pcc::PCCPatch>& patches, std::vector<tmc2_patch>& records );

// the placement a packer of the library decided, written back into the tile's patch list: reorders `patches` into list order
// and sets u0 / v0 / patchOrientation / bestMatchIdx as packFlexible / spatialConsistencyPackFlexible leave them
void applyPacking( const tmc2_patch* recordsByIndex, const int32_t* order, const int32_t* matches, int count,
                   std::vector<pcc::PCCPatch>& patches );

// drop-in body of PCCPatchSegmenter3::compute (PCCPatchSegmenter.cpp:53-224) for the CTC lossy conditions: S1-S9 on the
// device; the frame stays resident in *keep for the image-generation calls that follow.  Returns a tmc2 status.
int segmenterCompute( tmc2_ctx* ctx, const pcc::PCCPointSet3& geometry, size_t frameIndex,
                      const pcc::PCCPatchSegmenter3Parameters& params, std::vector<pcc::PCCPatch>& patches, tmc2_frame** keep );

}  // namespace tmc2hip
