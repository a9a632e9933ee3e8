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
#include "PCCContext.h"
#include "PCCEncoderParameters.h"
#include "PCCFrameContext.h"
#include "PCCGroupOfFrames.h"
#include "PCCImage.h"
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

// the packed LIST a GOF-level packer of the library returns for a frame (tmc2_host_place_segments, or tmc2_frame_get_patches
// after tmc2_encoder_global_patch_allocation: records in list order, block occupancy at occupancy + occOffset): rebuilds
// `patches` (in: creation order, as the segmenter left it) in list order with what performDataAdaptiveGPAMethod rewrites --
// index, block box, block occupancy -- and the placement / best-match index.  A record names its patch by its depthOffset.
void applyPackedList( const tmc2_patch* list, const int32_t* matches, const uint8_t* occupancy, int count,
                      std::vector<pcc::PCCPatch>& patches );

// S11-S16 of one frame (tmc2_frame_get_geometry_images) -> the reference's containers: PCCFrameContext::occupancyMap_ and
// blockToPatch_, the frame of the occupancy video, the two frames of the geometry video (formats and untouched planes as
// generateOccupancyMapVideo / generateIntraImage leave them)
void toFrameImages( const uint8_t* occupancy, const uint8_t* occVideo, const uint32_t* blockToPatch, const uint16_t* geometryD0,
                    const uint16_t* geometryD1, size_t W, size_t H, size_t occupancyPrecision, std::vector<uint32_t>& occupancyMap,
                    std::vector<size_t>& blockToPatchOut, pcc::PCCImage<uint8_t, 3>& occupancyFrame, pcc::PCCImage<uint16_t, 3>& d0,
                    pcc::PCCImage<uint16_t, 3>& d1 );
// S20-S22 (tmc2_frame_get_attribute_images, uint8 [2][3][H][W]) -> the two frames of the attribute video
void toAttributeFrames( const uint8_t* attribute, size_t W, size_t H, pcc::PCCImage<uint16_t, 3>& t0, pcc::PCCImage<uint16_t, 3>& t1 );
// S17 / S18 (tmc2_frame_get_reconstruction) -> the reconstructed cloud and PCCFrameContext::pointToPixel_
void toReconstruction( const int16_t* xyz, const uint8_t* rgb, const uint32_t* pointToPixel, size_t n, pcc::PCCPointSet3& cloud,
                       std::vector<pcc::PCCVector3<size_t>>& pointToPixelOut );

// drop-in body of PCCPatchSegmenter3::compute (PCCPatchSegmenter.cpp:53-224) for the CTC lossy conditions: S1-S9 on the
// device; the frame stays resident in *keep for the image-generation calls that follow.  Returns a tmc2 status.
int segmenterCompute( tmc2_ctx* ctx, const pcc::PCCPointSet3& geometry, size_t frameIndex,
                      const pcc::PCCPatchSegmenter3Parameters& params, std::vector<pcc::PCCPatch>& patches, tmc2_frame** keep );

// ---- the seams of PCCEncoder::encode (PCCEncoder.cpp:85-424) for the CTC lossy conditions, over the reference's own
// containers: the bodies a maintainer puts behind generateSegments / placeSegments / generateGeometryVideo (with the occupancy
// steps before it) / generateAttributeVideo (with generatePointCloud before and the padding after it).  encode() keeps calling
// them in its order; every frame of the GOF stays resident in HBM between the calls.  One object per encode() call.
// (Shown on one library context; an encoder that runs its frames as TBB tasks holds one context per task, INTEGRATION.md.)
class EncoderDropIn {
 public:
  explicit EncoderDropIn( int device );
  ~EncoderDropIn();
  EncoderDropIn( const EncoderDropIn& ) = delete;
  EncoderDropIn& operator=( const EncoderDropIn& ) = delete;
  // false: no device / parameter set not mirrored -- the caller keeps the reference's own bodies
  bool accepts( const pcc::PCCEncoderParameters& params );
  // generateSegments( sources, context ) :4672-4760: S0 on frame 0, S1-S9 per frame; patches appended to every frame's context
  int generateSegments( const pcc::PCCGroupOfFrames& sources, pcc::PCCContext& context, const pcc::PCCEncoderParameters& params );
  // placeSegments( sources, context ) :4762-4840: all-intra, low-delay (constrainedPack) and random-access (+ globalPatchAllocation
  // 1) conditions; lists reordered, placements / matches written, tile and atlas frame sizes set
  int placeSegments( pcc::PCCContext& context, const pcc::PCCEncoderParameters& params );
  // generateOccupancyMap + generateOccupancyMapVideo + generateBlockToPatchFromOccupancyMapVideo + generateGeometryVideo
  // (:3767, :806, PCCCodec.cpp:1736, :3894 with padding and group dilation): S11-S16
  int generateGeometryVideo( pcc::PCCContext& context, const pcc::PCCEncoderParameters& params );
  // generatePointCloud per frame + generateAttributeVideo + dilateSmoothedPushPull + attribute group dilation (encode()
  // :313-424): S17-S22, on the resident (= losslessly "decoded") geometry; decodedGeometry: see tmc2_frame_set_decoded_geometry
  int generateAttributeVideo( pcc::PCCContext& context, pcc::PCCGroupOfFrames& reconstructs, const pcc::PCCEncoderParameters& params );
  const char* lastError() const { return tmc2_last_error(); }

 private:
  void                     release();
  tmc2_ctx*                ctx_ = nullptr;
  std::vector<tmc2_frame*> frames_;
  int                      width_ = 0, height_ = 0;
};

// PCCEncoderParameters -> tmc2_segmenter_params (weightNormal left 1, 1, 1); false if the set uses what the library refuses
bool toParams( const pcc::PCCEncoderParameters& params, tmc2_segmenter_params& out );

}  // namespace tmc2hip

// ---- the remaining seams of SURVEY.md 8(b), as classes with the interface of the object they stand in for -----------------
#include <array>
#include <string>

#include "PCCKdTree.h"
#include "PCCMetrics.h"
#include "PCCMetricsParameters.h"

namespace tmc2hip {

// Stands in for the PCCMetrics object of PccAppEncoder / PccAppDecoder / PccAppMetrics (the same three calls:
// setParameters, compute( sources, reconstructs, normals ), display -- PCCMetrics.h:93-101, PCCMetrics.cpp:324-391) for
// the default PCCMetricsParameters (dropDuplicates 2, neighborsProc 1, no Hausdorff, no reflectance): every frame through
// tmc2_metrics_compute on the device, the text of display() through tmc2_metrics_display.
class MetricsDropIn {
 public:
  explicit MetricsDropIn( int device );
  ~MetricsDropIn();
  MetricsDropIn( const MetricsDropIn& ) = delete;
  MetricsDropIn& operator=( const MetricsDropIn& ) = delete;
  bool accepts( const pcc::PCCMetricsParameters& params ) const;  // false: keep the reference's object
  void setParameters( const pcc::PCCMetricsParameters& params ) { params_ = params; }
  int  compute( const pcc::PCCGroupOfFrames& sources, const pcc::PCCGroupOfFrames& reconstructs, const pcc::PCCGroupOfFrames& normals );
  int  display();  // the reference's text, on stdout
  // per frame: q[3][8] (rows A->B, B->A, symmetric; columns c2cMse, c2cPsnr, c2pMse, c2pPsnr, colorMse Y U V, colorPsnr Y)
  const std::vector<std::array<double, 24>>& results() const { return q_; }

 private:
  tmc2_ctx*                            ctx_ = nullptr;
  pcc::PCCMetricsParameters            params_;
  std::vector<std::array<double, 24>>  q_;
  std::vector<std::array<int64_t, 2>>  counts_;
  std::vector<std::array<uint64_t, 2>> points_;
  std::vector<bool>                    withC2p_;
};

// The per-frame finish of PCCDecoder::decode (PCCDecoder.cpp:325-470; the encoder's own reconstruction loop :571-719 is the
// same code) for the CTC lossy conditions, one tile per frame: occupancy map and blockToPatch from the decoded occupancy
// video, generatePointCloud, colorPointCloud from the decoded (colour-converted, 16-bit 4:4:4) attribute frames, grid
// geometry smoothing, transferColors16bitBP onto the moved points, convertYUV16ToRGB8 -- on the device, from the reference's
// own containers, into `reconstruct` (positions, 16-bit and 8-bit colours, boundary point types).
class DecoderDropIn {
 public:
  explicit DecoderDropIn( int device );
  ~DecoderDropIn();
  DecoderDropIn( const DecoderDropIn& ) = delete;
  DecoderDropIn& operator=( const DecoderDropIn& ) = delete;
  int reconstructFrame( pcc::PCCContext& context, size_t frameIdx, size_t occupancyPrecision, size_t gridSize, double thresholdSmoothing,
                        pcc::PCCPointSet3& reconstruct );
  const char* lastError() const { return tmc2_last_error(); }

 private:
  tmc2_ctx* ctx_ = nullptr;
};

// Stands in for PCCKdTree (PCCKdTree.h:85-100) where a caller asks for the neighbours of MANY points of one cloud: the tree
// stays in HBM (tmc2_frame), a batch of queries is one tmc2_kdtree_search.  search() keeps the reference's one-point
// signature (a batch of one); searchBatch is what a ported caller uses.  Same results, same order on ties.
class KdTreeDropIn {
 public:
  explicit KdTreeDropIn( int device );
  ~KdTreeDropIn();
  KdTreeDropIn( const KdTreeDropIn& ) = delete;
  KdTreeDropIn& operator=( const KdTreeDropIn& ) = delete;
  int  init( const pcc::PCCPointSet3& pointCloud );
  int  search( const pcc::PCCPoint3D& point, size_t num_results, pcc::PCCNNResult& results ) const;
  int  searchBatch( const std::vector<pcc::PCCPoint3D>& points, size_t num_results, std::vector<pcc::PCCNNResult>& results ) const;

 private:
  tmc2_ctx*   ctx_   = nullptr;
  tmc2_frame* frame_ = nullptr;
};

}  // namespace tmc2hip
