/* tmc2hip.h -- C-ABI of the MI355X-native TMC2 patch-generation / image-generation hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI around this path: the
 * seams are C++ member functions of PCCPatchSegmenter3 / PCCNormalsGenerator3 / PCCKdTree / PCCEncoder
 * / PCCMetrics.  Each entry point below names the reference member it replaces (file:line relative
 * to the reference tree); INTEGRATION.md shows the adaptor a maintainer adds at each call site.
 *
 * Conventions
 *   - plain C, POD only; every function returns 0 on success or a negative TMC2_E_* code and never
 *     throws or exits (the reference exit()s on fatal conditions, e.g. PCCPatch.cpp:242);
 *     tmc2_last_error() returns a human-readable message for the calling thread's last failure.
 *   - a tmc2_ctx is bound to ONE HIP device and owns one stream; it is not thread-safe, create one
 *     per host thread (the reference calls the segmenter from a tbb::parallel_for over frames,
 *     PCCEncoder.cpp:4729-4750 -- one ctx per in-flight frame).
 *   - a tmc2_frame holds the device-resident state of one point-cloud frame; stage functions chain
 *     on it without bouncing through the host.  Host arrays are AoS exactly like the reference
 *     containers: xyz = int16[n][3] (PCCPoint3D, PCCMath.h:450), rgb = uint8[n][3] (PCCColor3B :453),
 *     normals = double[n][3].
 *   - there is NO CPU fallback: without a HIP device tmc2_ctx_create fails with TMC2_E_NO_DEVICE.
 */
#ifndef TMC2HIP_H
#define TMC2HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMC2_OK 0
#define TMC2_E_NO_DEVICE -1
#define TMC2_E_HIP -2
#define TMC2_E_INVALID -3
#define TMC2_E_UNSUPPORTED -4
#define TMC2_E_STATE -5

typedef struct tmc2_ctx   tmc2_ctx;
typedef struct tmc2_frame tmc2_frame;

/* Flat mirror of the live fields of PCCPatchSegmenter3Parameters (PCCPatchSegmenter.h:48-100) as
 * filled by PCCEncoder::generateSegments (PCCEncoder.cpp:4672-4723).  Options the CTC lossy
 * conditions keep off are not representable; tmc2_segmenter_params_check() documents them.        */
typedef struct tmc2_segmenter_params {
  int32_t nnNormalEstimation;               /* 16  */
  int32_t normalOrientation;                /* 1 = PCC_NORMALS_GENERATOR_ORIENTATION_SPANNING_TREE */
  int32_t gridBasedRefineSegmentation;      /* 1   */
  int32_t maxNNCountRefineSegmentation;     /* 1024 */
  int32_t iterationCountRefineSegmentation; /* 10 / 50 longdress / 20 basketball */
  int32_t voxelDimensionRefineSegmentation; /* 4   */
  int32_t searchRadiusRefineSegmentation;   /* 192 */
  int32_t occupancyResolution;              /* 16  */
  int32_t enablePatchSplitting;             /* 1   */
  int32_t maxPatchSize;                     /* 1024 */
  int32_t quantizerSizeX;                   /* 1 << log2QuantizerSizeX = 16 */
  int32_t quantizerSizeY;                   /* 16  */
  int32_t minPointCountPerCCPatchSegmentation; /* 16 */
  int32_t maxNNCountPatchSegmentation;      /* 16  */
  int32_t surfaceThickness;                 /* 4   */
  int32_t mapCountMinus1;                   /* 1   */
  int32_t minLevel;                         /* 64  */
  int32_t maxAllowedDepth;                  /* (1 << geometryNominal2dBitdepth) - 1 = 255 */
  int32_t geometryBitDepth2D;               /* 8   */
  int32_t geometryBitDepth3D;               /* geometry3dCoordinatesBitdepth + 1 = 11 / 12 */
  double  maxAllowedDist2RawPointsDetection; /* 9 */
  double  maxAllowedDist2RawPointsSelection; /* 1 */
  double  lambdaRefineSegmentation;          /* 3 */
  double  weightNormal[3];                   /* PCCEncoder::calculateWeightNormal, see tmc2_weight_normal */
} tmc2_segmenter_params;

/* One patch record = the PCCPatch fields the hot path produces (PCCPatch.h:352-409).
 * depthOffset / occOffset index the frame's pools (see tmc2_frame_get_patches).                     */
typedef struct tmc2_patch {
  int32_t index, viewId;
  int32_t normalAxis, tangentAxis, bitangentAxis, projectionMode;
  int32_t u1, v1, d1;
  int32_t sizeU, sizeV, sizeD, sizeDPixel;
  int32_t sizeU0, sizeV0;
  int32_t size2DXInPixel, size2DYInPixel;
  int32_t d0Count, eomAndD1Count;
  int32_t u0, v0, patchOrientation; /* set by packing */
  int64_t depthOffset;
  int64_t occOffset;
} tmc2_patch;

/* ---- context ------------------------------------------------------------------------------- */
int         tmc2_ctx_create( int device, tmc2_ctx** out );
void        tmc2_ctx_destroy( tmc2_ctx* ctx );
const char* tmc2_last_error( void );
int         tmc2_ctx_synchronize( tmc2_ctx* ctx );
/* Page-locked host memory that every device of the node can DMA into (hipHostMalloc, portable): where a native front end lands
 * the finished canvases -- what the reference keeps in PCCVideo<T> frames for the video encoder (PCCVideo.h) -- so that the
 * copies out of HBM are plain DMA from each GPU over its own PCIe link (integration/tmc2_encode_gof.cpp).          */
int         tmc2_host_alloc( size_t bytes, void** out );
void        tmc2_host_free( void* p );
/* ... and the same for memory the caller already has (a shared-memory segment that several processes of the node map: every
 * rank of a one-process-per-GPU run lands its frames' canvases where the process that runs the video encoder reads them) */
int         tmc2_host_register( void* p, size_t bytes );
int         tmc2_host_unregister( void* p );
/* Limit on concurrently running host-resident steps (k-d tree builds of option KDTREE_HOST, the normal orientation's walk); 0 =
 * none.  Frames of a GOF run on separate host threads; the gate keeps the cache-hungry host steps at the core-complex count while
 * the GPU phases of the other frames proceed.  A gate is an object an encoder shares among ITS contexts (round 6: two encoders of
 * one process do not draw on one count): tmc2_host_gate_create, tmc2_ctx_set_host_gate( ctx, gate ) on each (gate NULL: back to the
 * default), tmc2_host_gate_destroy when the encoder is done (contexts that still hold it keep it alive).  A context without a gate
 * uses the process' default one, whose limit tmc2_set_host_parallelism presets -- the last of the old process-wide setters, kept
 * for callers that have a single encoder.                                                                              */
typedef struct tmc2_host_gate tmc2_host_gate;
int         tmc2_host_gate_create( int maxConcurrentHostSteps, tmc2_host_gate** out );
void        tmc2_host_gate_destroy( tmc2_host_gate* gate );
int         tmc2_ctx_set_host_gate( tmc2_ctx* ctx, tmc2_host_gate* gate );
void        tmc2_set_host_parallelism( int maxConcurrentHostSteps );
/* Per-context options: everything that tunes or cross-checks the path is a property of ONE context -- no process-global state, and
 * the library does not look at the environment while it runs.  key = the knob's name (DESIGN.md section 5 lists them):
 * "REFINE_OVERLAP" (0 / 1, below), "KDTREE_HOST" (0 / 1 / 2, below), "UF_PRECHECK" / "UF_SCOPE" / "UF_CHECK" (the union passes'
 * conservative forms and debug invariants), "KD_HUGEMAX" / "KD_PIECE_PER" / "KD_DECIDE" (forms of the device tree build),
 * "REFINE_*" (forms and grids of the refinement), "METRICS_K", "ORIENT_*"; a "TMC2_" prefix is accepted and dropped.  When a
 * context is created its options start as the TMC2_* variables of the process environment (read once, there); value NULL
 * unsets an option.  None of them ever changes a result.  Options may be set from any thread while frames of the context are
 * in flight (a mutex guards the table; a stage reads an option when it starts).  tmc2_ctx_get_option returns a COPY that belongs
 * to the calling thread and stays valid until that thread's eighth later look-up -- never a pointer into the table.      */
int         tmc2_ctx_set_option( tmc2_ctx* ctx, const char* key, const char* value );
const char* tmc2_ctx_get_option( tmc2_ctx* ctx, const char* key );
/* Reserve, at context creation time, the device memory the frames of a sequence will need (upper bounds of the sequence: points
 * per frame, voxelDimensionRefineSegmentation, geometry3dCoordinatesBitdepth + 1, the largest canvas).  The reference sizes its
 * containers per frame as it goes (std::vector growth inside PCCPatchSegmenter3::compute / PCCEncoder::generateGeometryVideo); here
 * every temporary comes from the context's caching pool, whose FIRST use of a size is a hipMalloc -- a device-wide synchronisation
 * under all frames in flight: the first GOFs of a process ran at half speed.  With the reservation the pool carves from memory it
 * already holds.  Optional; a frame that needs more than was reserved still works (hipMalloc).                        */
int         tmc2_ctx_reserve( tmc2_ctx* ctx, uint64_t maxPoints, int voxelDimRefine, int bits3d, int maxCanvasWidth, int maxCanvasHeight );
int         tmc2_ctx_pool_stats( tmc2_ctx* ctx, uint64_t* bytesHeld, uint64_t* mallocCalls, double* mallocMs, uint64_t* carvedBlocks );
/* Device staging for a host that exchanges small records between the ranks of a sharded GOF itself (libtmc2gof.so over RCCL,
 * include/tmc2gof.h: the collectives need device buffers and the stream they become ready on): memory of the context's device,
 * copies ordered on the context's stream (download waits for it), the stream (a hipStream_t) and the device ordinal.      */
int         tmc2_ctx_device_alloc( tmc2_ctx* ctx, size_t bytes, void** out );
int         tmc2_ctx_device_free( tmc2_ctx* ctx, void* p );
int         tmc2_ctx_upload( tmc2_ctx* ctx, void* deviceDst, const void* hostSrc, size_t bytes );
int         tmc2_ctx_download( tmc2_ctx* ctx, void* hostDst, const void* deviceSrc, size_t bytes );
void*       tmc2_ctx_stream( tmc2_ctx* ctx );
int         tmc2_ctx_device( tmc2_ctx* ctx );
int         tmc2_ctx_make_current( tmc2_ctx* ctx );
/* per-stage GPU time of the last frame operation, milliseconds (hipEvent); name list via index */
int         tmc2_ctx_stage_count( tmc2_ctx* ctx );
const char* tmc2_ctx_stage_name( tmc2_ctx* ctx, int i );
double      tmc2_ctx_stage_ms( tmc2_ctx* ctx, int i );
long        tmc2_ctx_stage_calls( tmc2_ctx* ctx, int i );
void        tmc2_ctx_stage_reset( tmc2_ctx* ctx );
void        tmc2_ctx_set_timing( tmc2_ctx* ctx, int enabled ); /* hipEvent timing of stages on/off (default on) */

/* ---- frame: upload + PCCKdTree ------------------------------------------------------------- */
/* replaces: PCCKdTree::PCCKdTree(const PCCPointSet3&) / init  (PccLibCommon/source/PCCKdTree.cpp:44-59).
 * Copies the points to HBM; the nanoflann-identical k-d tree (leaf size 10) is built on first use.   */
int  tmc2_frame_create( tmc2_ctx* ctx, const int16_t* xyz, const uint8_t* rgb, uint64_t n, tmc2_frame** out );
void tmc2_frame_destroy( tmc2_frame* f );
/* the tree is built (host) and uploaded on first use; this forces it (PCCKdTree::init, PCCKdTree.cpp:56-59) */
int  tmc2_kdtree_build( tmc2_frame* f );
/* where PCCKdTree::init runs: option "KDTREE_HOST" of the frame's context; this call presets what contexts WITHOUT the option do
 * (rounds 1-4's process-wide switch).  0 (default): on the device, lowest latency for a frame on its own;
 * 1: on the host (same algorithm, same tree) -- with many frames in flight and idle host cores this leaves the
 * GPU to the stages only it can run; 2: adaptive -- on the host while one of the host slots
 * (tmc2_set_host_parallelism) is free at that moment, else on the device.  Both builders are exact; the choice
 * never changes a result.                                                                                   */
void tmc2_set_kdtree_placement( int mode );
/* option "REFINE_OVERLAP" of the frame's context (this call presets what contexts WITHOUT the option do):
 * tmc2_segmenter_compute queues the geometry of the refinement (S5: voxels, neighbourhood rows -- it needs the
 * points only) before the host-resident walk of the normal orientation (S3) and builds it while the host walks.  Shortens a
 * frame's chain (few frames in flight: one rank of a many-GPU run); with the chip full of other frames it only competes with
 * them -- off by default.  The same setting makes the two kernels of a refinement sweep take the grids that are fastest with
 * the GPU to themselves (twice the workgroups: idle waves are in nobody's way then).  Never changes a result.        */
void tmc2_set_refine_overlap( int on );
/* inspection: the permutation nanoflann's build leaves in vind (tree order -> point index), uint32[n], and the
 * number of tree levels; the search order under distance ties is a function of exactly this permutation */
int  tmc2_frame_get_kdtree_order( tmc2_frame* f, uint32_t* perm, int32_t* depth );
uint64_t tmc2_frame_point_count( const tmc2_frame* f );
/* drop every derived result (tree, adjacency, normals, partition, patches, canvases); the points stay in HBM */
int  tmc2_frame_reset( tmc2_frame* f );

/* replaces: PCCKdTree::search (PCCKdTree.cpp:61-66) for a batch of queries against the frame's tree.
 * idx = uint32[nq][k] in nanoflann result order; dist2 = uint32[nq][k] squared distances or NULL.   */
int tmc2_kdtree_search( tmc2_frame* f, const int16_t* queries, uint64_t nq, int k, uint32_t* idx, uint32_t* dist2 );

/* ---- PCCNormalsGenerator3 ------------------------------------------------------------------- */
/* replaces: PCCNormalsGenerator3::computeNormals (PccLibEncoder/source/PCCNormalsGenerator.cpp:158-185)
 * and, sharing its k-NN lists, PCCPatchSegmenter3::computeAdjacencyInfo (PCCPatchSegmenter.cpp:267-291). */
int tmc2_normals_compute_normals( tmc2_frame* f, int k );
/* replaces: PCCNormalsGenerator3::orientNormals, SPANNING_TREE branch (PCCNormalsGenerator.cpp:198-242).  */
int tmc2_normals_orient( tmc2_frame* f );
/* replaces: PCCNormalsGenerator3::compute (PCCNormalsGenerator.cpp:61-70) = both of the above.          */
int tmc2_normals_compute( tmc2_frame* f, int k, int orientation );
int tmc2_frame_get_normals( tmc2_frame* f, double* normals /* [n][3] */ );
int tmc2_frame_set_normals( tmc2_frame* f, const double* normals );
int tmc2_frame_get_adjacency( tmc2_frame* f, uint32_t* adj /* [n][k] */ );

/* ---- PCCEncoder::calculateWeightNormal (PCCEncoder.cpp:3569-3626) ---------------------------- */
int tmc2_weight_normal( tmc2_frame* f, int geometryBitDepth3D, double minWeightEPP, double weight[3] );

/* ---- PCCPatchSegmenter3 stages --------------------------------------------------------------- */
/* replaces: PCCPatchSegmenter3::initialSegmentation (PCCPatchSegmenter.cpp:226-265), 6 planes.        */
int tmc2_segmenter_initial_segmentation( tmc2_frame* f, const double weight[3] );
/* replaces: PCCPatchSegmenter3::refineSegmentationGridBased (PCCPatchSegmenter.cpp:1386-1561).       */
int tmc2_segmenter_refine_grid_based( tmc2_frame* f, int maxNNCount, double lambda, int iterationCount,
                                      int voxDim, int searchRadius );
int tmc2_frame_get_partition( tmc2_frame* f, uint32_t* partition /* [n] */ );
int tmc2_frame_set_partition( tmc2_frame* f, const uint32_t* partition );
/* replaces: PCCPatchSegmenter3::segmentPatches (PCCPatchSegmenter.cpp:542-1320).                      */
int tmc2_segmenter_segment_patches( tmc2_frame* f, const tmc2_segmenter_params* p );
/* replaces: PCCPatchSegmenter3::compute (PCCPatchSegmenter.cpp:53-150) = S1..S9 end to end.           */
int tmc2_segmenter_compute( tmc2_frame* f, const tmc2_segmenter_params* p );
int tmc2_segmenter_params_check( const tmc2_segmenter_params* p );

/* patch list of the frame (after segment_patches / compute) */
int tmc2_frame_patch_count( tmc2_frame* f );
int tmc2_frame_patch_pool_sizes( tmc2_frame* f, int64_t* depthCount, int64_t* occCount );
int tmc2_frame_get_patches( tmc2_frame* f, tmc2_patch* patches, int16_t* depth0, int16_t* depth1, uint8_t* occupancy );

/* ---- PCCEncoder image generation, phase A (all-intra packing) ---------------------------------- */
/* replaces: PCCEncoder::packFlexible (PccLibEncoder/source/PCCEncoder.cpp:2306-2449) for one frame, as called by
 * PCCEncoder::placeSegments (:4762-4835) with constrainedPack=0, packingStrategy=1, two orientations.
 * Sorts the patch list (PCCPatch::gt), assigns u0/v0/patchOrientation, returns the frame's canvas height.  */
int tmc2_encoder_pack_flexible( tmc2_frame* f, int presetWidth, int numTilesHor, double tileHeightToWidthRatio,
                                int32_t* height );
/* replaces: PCCEncoder::spatialConsistencyPackFlexible (PccLibEncoder/source/PCCEncoder.cpp:1183-1412) for one frame, as called
 * by placeSegments (:4790-4795) for frames after the first when constrainedPack = 1 (the program default and the CTC
 * low-delay condition; globalPatchAllocation 0), packingStrategy = 1, two orientations: patches are matched to those of
 * `previous` (same view, bounding-box IoU > 0.2, pcc::computeIOU PCCPatchSegmenter.cpp:1563), matched ones go first,
 * in the previous frame's order and if possible at the previous position.  `previous` must be packed already.       */
int tmc2_encoder_pack_spatial_consistency( tmc2_frame* f, tmc2_frame* previous, int presetWidth, int numTilesHor,
                                           double tileHeightToWidthRatio, int32_t* height );
/* the tile size the packers left behind (what resizeTileGeometryVideo / resizeGeometryVideo look at): packFlexible keeps the
 * preset width, spatialConsistencyPackFlexible and the global patch allocation set the width of the canvas they packed on */
int tmc2_frame_get_packed_size( tmc2_frame* f, int32_t* width, int32_t* height );
/* per list position: the position of the matched patch in the previous frame's list (PCCPatch::getBestMatchIdx), -1 = none
 * (all -1 after tmc2_encoder_pack_flexible) */
int tmc2_frame_get_patch_matches( tmc2_frame* f, int32_t* matches );
/* replaces: PCCEncoder::performDataAdaptiveGPAMethod (PccLibEncoder/source/PCCEncoder.cpp:6821-6971) with the members it
 * drives (generateGlobalPatches :7022, unionPatchGenerationAndPacking :7059, packingFirstFrame :7228, performGPAPacking
 * :7531, updatePatchInformation :7366), as placeSegments runs it after the per-frame packing chain when constrainedPack = 1
 * and globalPatchAllocation = 1 (the CTC random-access condition).  frames[0..count): the frames of the GOF in order, each
 * packed (tmc2_encoder_pack_flexible for the first, tmc2_encoder_pack_spatial_consistency for the others).  Patches tracked
 * across the frames of a sub-context take the block size of the union of their track and one common position; every
 * frame's list is reordered (tracked patches first, aligned across frames), patch.index becomes the list position,
 * sizeU0 / sizeV0 / u0 / v0 / patchOrientation, the matches and the block-occupancy pool are rewritten; afterwards
 * tmc2_frame_get_patches returns the records in list order and tmc2_frame_get_patch_order the identity.
 * widths / heights (int32[count], may be NULL): the tile size of every frame for tmc2_encoder_canvas_size.          */
int tmc2_encoder_global_patch_allocation( tmc2_frame** frames, int count, int minimumImageWidth, int minimumImageHeight,
                                          int32_t* widths, int32_t* heights );
/* A frame sharded onto another rank takes part in the inter-frame packers through its patch RECORDS: they travel to the
 * rank that runs the chain (tmc2_host_pack_flexible / _spatial_consistency / _global_patch_allocation on plain records,
 * "gather to rank 0 of the per-frame patch table before packing"), the packed list comes back and is installed
 * here -- what PCCEncoder::placeSegments leaves in the frame's PCCFrameContext (PCCEncoder.cpp:4790-4840).
 * list[count]: the frame's patches in LIST order with u0 / v0 / patchOrientation (and, after the global patch allocation,
 * rewritten index / sizeU0 / sizeV0) and occOffset into occupancy[occupancyBytes] (pool in list order); depthOffset must
 * still point into this frame's own depth pools.  matches: int32[count] or NULL (none).  packedWidth / packedHeight: the
 * tile size the packers left.  Afterwards the frame is in the state tmc2_encoder_global_patch_allocation leaves.      */
int tmc2_frame_set_packing( tmc2_frame* f, const tmc2_patch* list, int count, const int32_t* matches, const uint8_t* occupancy,
                            int64_t occupancyBytes, int packedWidth, int packedHeight );
/* packing order of the frame: order[listPosition] = patch index (the reference reorders the list itself) */
int tmc2_frame_get_patch_order( tmc2_frame* f, int32_t* order );
/* replaces: resizeTileGeometryVideo + resizeGeometryVideo (PCCEncoder.cpp:5593-5632, 5546-5591): common GOF canvas */
int tmc2_encoder_canvas_size( const int32_t* frameHeights, int frames, int tileWidth, int minimumImageWidth,
                              int minimumImageHeight, int32_t* width, int32_t* height );
/* replaces, for one frame: generateOccupancyMap (:3767-3784), generateOccupancyMapVideo (:806-861),
 * PCCCodec::generateBlockToPatchFromOccupancyMapVideo (PccLibCommon/source/PCCCodec.cpp:1736-1775),
 * generateGeometryVideo -> generateIntraImage (:3929-3992) + dilate3DPadding (:5951-6130, geometryPadding=0)
 * + dilateGroupGeometryVideo (:3717-3739).                                                              */
int tmc2_encoder_generate_geometry_images( tmc2_frame* f, int width, int height, int occupancyPrecision );
/* canvases of the frame; any pointer may be NULL.  occupancy u8[W*H], occVideo u8[(W/p)*(H/p)],
 * blockToPatch u32[(W/16)*(H/16)] (list position + 1), geometry D0 / D1 luma u16[W*H] (chroma planes are zero) */
int tmc2_frame_get_geometry_images( tmc2_frame* f, uint8_t* occupancy, uint8_t* occVideo, uint32_t* blockToPatch,
                                    uint16_t* geometryD0, uint16_t* geometryD1 );
/* ---- PCCEncoder image generation, phase B (after the geometry video has been coded and decoded) -------- */
/* replaces, for one frame: PCCCodec::generatePointCloud (PccLibCommon/source/PCCCodec.cpp:519-980, lossy CTC branch),
 * PCCPointSet3::transferColors source -> reconstruction (PccLibCommon/source/PCCPointSet.cpp:807-1124, CTC settings),
 * presmoothPointCloudColor (PCCEncoder.cpp:6593-6655, a no-op in the reference build), generateAttributeVideo
 * (:6736-6794), dilateSmoothedPushPull (:6542-6591) and the attribute group dilation (encode() :380-402).
 * Works on the frame's resident occupancy / geometry canvases (decoded == generated when the codec is lossless;
 * upload decoded planes with tmc2_frame_set_decoded_geometry first when it is not).                         */
int     tmc2_encoder_generate_attribute_images( tmc2_frame* f );
/* replaces: PCCPointSet3::transferColors (PccLibCommon/source/PCCPointSet.cpp:807-1124) alone, on two host clouds, with the
 * arguments PCCEncoder::generateAttributeVideo passes under the CTC (PCCEncoder.cpp:6679-6697); tgtRgb = uint8[m][3]      */
int     tmc2_transfer_colors( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n, const int16_t* tgtXyz,
                              uint64_t m, uint8_t* tgtRgb );
int64_t tmc2_frame_recon_count( tmc2_frame* f );
/* reconstruction: xyz int16[M][3], rgb uint8[M][3], pointToPixel uint32[M][3] = (x, y, layer); any may be NULL */
int     tmc2_frame_get_reconstruction( tmc2_frame* f, int16_t* xyz, uint8_t* rgb, uint32_t* pointToPixel );
/* attribute images: uint8[2 maps][3 channels][H][W] (the reference holds the same values in uint16 planes) */
int     tmc2_frame_get_attribute_images( tmc2_frame* f, uint8_t* attribute );
/* replace the resident occupancy video / geometry planes by decoded ones (uint8[(W/p)*(H/p)], uint16[2][H][W]) */
int     tmc2_frame_set_decoded_geometry( tmc2_frame* f, const uint8_t* occVideo, const uint16_t* geometry );

/* device addresses of the frame's canvases (valid until the next generate call / frame destroy), for
 * zero-copy hand-off to a collective (RCCL gather of finished frames) or to a device-side consumer       */
int tmc2_frame_device_images( tmc2_frame* f, void** occupancy, void** occVideo, void** blockToPatch, void** geometry );
int tmc2_frame_device_attribute( tmc2_frame* f, void** attribute );

/* ---- colour-space conversion around the attribute video codec (PCCVideoEncoder::compress) -------------------------- */
/* replaces: PCCInternalColorConverter<T>::convert( "RGB444ToYUV420_8_<f>" ) (PccLibColorConverter/source/
 * PCCInternalColorConverter.cpp:355-378 -> convertRGB44ToYUV420 :406-424), which PCCVideoEncoder::compress
 * (PccLibEncoder/source/PCCVideoEncoder.cpp:353) runs on the attribute video before the codec when no external converter is
 * configured.  rgb: uint8 [3][H][W] (R, G, B planes); yuv420: one I420 frame, Y [H][W] then U, V [H/2][W/2].  Only the
 * reference's default filter (4 = DF_GS) is built; others return TMC2_E_UNSUPPORTED.                                 */
int tmc2_color_convert_rgb444_to_yuv420( tmc2_ctx* ctx, const uint8_t* rgb, int width, int height, int downsamplingFilter,
                                         uint8_t* yuv420 );
/* replaces: convert( "YUV420ToYUV444_8_<f>" ) (-> convertYUV420ToYUV444 :462-482) on the decoded video (PCCVideoEncoder.cpp:413;
 * the decoder does the same): 8-bit I420 frame -> uint16 [3][H][W], 16-bit 4:4:4.  Only filter 0 (UF_F0, the default). */
int tmc2_color_convert_yuv420_to_yuv444( tmc2_ctx* ctx, const uint8_t* yuv420, int width, int height, int upsamplingFilter,
                                         uint16_t* yuv444 );
/* the same two steps on a frame, without moving the source through the host: the resident attribute canvases as the two
 * I420 frames the attribute video encoder reads (yuv420: [2 maps] x I420 frame), and the two decoded I420 frames as the
 * 16-bit 4:4:4 planes that tmc2_codec_color_point_cloud( f, NULL ) then reads on the device                          */
int tmc2_encoder_attribute_to_yuv420( tmc2_frame* f, int downsamplingFilter, uint8_t* yuv420 );
int tmc2_codec_set_decoded_attribute_yuv420( tmc2_frame* f, const uint8_t* yuv420, int upsamplingFilter );
int tmc2_frame_get_decoded_attribute( tmc2_frame* f, uint16_t* planes ); /* uint16 [2][3][H][W] */

/* ---- decoder side: reconstruction from what the bitstream carries --------------------------------------------------- */
/* replaces what PCCDecoder::decode does per frame before the reconstruction (PccLibDecoder/source/PCCDecoder.cpp:333-349:
 * generateOccupancyMap, generateBlockToPatchFromOccupancyMapVideo -- PCCCodec.cpp:1736-1775): a frame WITHOUT a source
 * cloud, made of the decoded patch records in list order (u0, v0, sizeU0, sizeV0, patchOrientation, u1, v1, d1, the three
 * axes and projectionMode are used), the decoded occupancy video (uint8 [(H/p)][(W/p)]) and the two decoded geometry
 * maps (uint16 [2][H][W]).  Continue with tmc2_codec_generate_point_cloud and the tmc2_codec_* tail.                 */
int tmc2_decoder_frame_create( tmc2_ctx* ctx, const tmc2_patch* patches, int count, int width, int height, int occupancyPrecision,
                               const uint8_t* occVideo, const uint16_t* geometry, tmc2_frame** out );
/* replaces: PCCCodec::generatePointCloud (PccLibCommon/source/PCCCodec.cpp:519-980) alone -- the reconstruction from the
 * resident canvases without the encoder's colour transfer / attribute images (tmc2_encoder_generate_attribute_images runs it
 * as its first step).  tmc2_frame_get_reconstruction( f, xyz, NULL, pointToPixel ) returns the points.             */
int tmc2_codec_generate_point_cloud( tmc2_frame* f );

/* ---- post-reconstruction tail (PCCEncoder::encode :571-719, PCCDecoder::decode :330-470) -------------------- */
/* All of these work on the reconstruction left by tmc2_encoder_generate_attribute_images (PCCCodec::generatePointCloud on
 * the resident, or decoded, occupancy / geometry canvases).  CTC settings: two maps in one stream, lossy attributes,
 * flagGeometrySmoothing 1, gridSmoothing 1, attrTransferFilterType 1, flagColorSmoothing 0.
 * replaces: PCCCodec::identifyBoundaryPoints (PccLibCommon/source/PCCCodec.cpp:268-327), as generatePointCloud runs it
 * over all points when flagGeometrySmoothing is set (:955-976): PCCPointSet3::boundaryPointTypes_ becomes 0 / 1.     */
int tmc2_codec_identify_boundary_points( tmc2_frame* f );
/* replaces: PCCCodec::colorPointCloud (PCCCodec.cpp:1319-1460), branch "f < mapCount": every point takes the 16-bit colour
 * of its pixel in the decoded attribute frame of its map.  attribute: uint16 [2 maps][3 channels][H][W] (host memory),
 * i.e. context.getVideoAttributesMultiple()[0] frames 2f and 2f+1 after decoding + colour conversion.              */
int tmc2_codec_color_point_cloud( tmc2_frame* f, const uint16_t* attribute );
/* attribute == NULL above: the frames tmc2_codec_set_decoded_attribute_yuv420 left on the device are used.             */
/* replaces: PCCCodec::smoothPointCloudPostprocess (PCCCodec.cpp:54-148, gridSmoothing branch: addGridCentroid :982,
 * gridFiltering :1002, smoothPointCloudGrid :1067).  Boundary points that the trilinear blend of the cell centroids
 * pulls further than the threshold are moved there; their boundary type becomes 3.  Runs
 * tmc2_codec_identify_boundary_points first if that has not happened.                                                */
int tmc2_codec_smooth_point_cloud_postprocess( tmc2_frame* f, int gridSize, double thresholdSmoothing );
/* replaces: PCCPointSet3::transferColors16bitBP (PccLibCommon/source/PCCPointSet.cpp:1126-1470) with filterType 1 and the
 * arguments of PCCEncoder.cpp:657-672 / PCCDecoder.cpp:416-431: source = the cloud before smoothing with its 16-bit
 * colours, target = the smoothed cloud; only the moved points (boundary type 3) are recoloured.                     */
int tmc2_codec_transfer_colors_16bit_bp( tmc2_frame* f );
/* replaces: PCCPointSet3::convertYUV16ToRGB8 (PccLibCommon/include/PCCPointSet.h:133-166) */
int tmc2_codec_convert_yuv16_to_rgb8( tmc2_frame* f );
/* the finished cloud (any pointer may be NULL): positions int16[M][3] (smoothed once the smoothing ran), 16-bit colours
 * uint16[M][3], 8-bit colours uint8[M][3], boundary types uint16[M]; M = tmc2_frame_recon_count                     */
int tmc2_frame_get_post_reconstruction( tmc2_frame* f, int16_t* xyz, uint16_t* colors16, uint8_t* rgb, uint16_t* boundaryType );

/* ---- PCCMetrics ------------------------------------------------------------------------------------ */
/* replaces: PCCMetrics::compute for one frame (PccLibMetrics/source/PCCMetrics.cpp:324-375) with the defaults of
 * PCCMetricsParameters (dropDuplicates 2, neighborsProc 1, no Hausdorff): duplicate removal, normal copy / scaling,
 * QualityMetrics::compute both ways (:73-229) and their symmetric combination (:289-322).
 * srcNormals (double[n][3], the normals of the SOURCE cloud, e.g. from tmc2_frame_get_normals) may be NULL: no D2.
 * out[3][8]: rows A->B, B->A, symmetric; columns c2cMse, c2cPsnr, c2pMse, c2pPsnr, colorMse Y,U,V, colorPsnr Y.
 * counts[2]: points after duplicate removal (source, reconstruction).                                     */
int tmc2_metrics_compute( tmc2_ctx* ctx, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n, const int16_t* recXyz,
                          const uint8_t* recRgb, uint64_t m, const double* srcNormals, double resolution, double* out,
                          int64_t* counts );

/* The same for a frame whose clouds are resident in HBM already (nothing is uploaded): source = the frame's points and colours
 * (and, with useNormals, its normals: tmc2_normals_compute / tmc2_frame_set_normals); reconstruction = which 0: what
 * tmc2_encoder_generate_attribute_images left (PCCCodec::generatePointCloud + transferColors), which 1: the finished cloud
 * of the post-reconstruction tail (tmc2_codec_smooth_point_cloud_postprocess .. tmc2_codec_convert_yuv16_to_rgb8: smoothed positions, final colours) -- what PccAppEncoder / PccAppDecoder hand to
 * PCCMetrics::compute (PccAppEncoder.cpp, metrics.compute( sources, reconstructs, normals ): PCCMetrics.cpp:324-375).        */
int tmc2_metrics_compute_frame( tmc2_frame* frame, int which, int useNormals, double resolution, double* out, int64_t* counts );
/* ... and for a frame that has no source cloud (the decoder side, tmc2_decoder_frame_create: PccAppDecoder reads the uncompressed
 * frames and their normals from PLY files and calls PCCMetrics::compute on them and the decoded clouds, PccAppDecoder.cpp): the
 * source comes from the host (as for tmc2_metrics_compute), the reconstruction is the frame's resident one (which: as above).   */
int tmc2_metrics_compute_frame_source( tmc2_frame* frame, int which, const int16_t* srcXyz, const uint8_t* srcRgb, uint64_t n,
                                       const double* srcNormals, double resolution, double* out, int64_t* counts );

/* The accumulation of QualityMetrics::compute alone (`sseC2c += ..`, `sseC2p += ..`, `sseColor[i] += ..` over the points in
 * index order, PCCMetrics.cpp:187-198) for per-point terms the caller holds on the host: termsA[nA][5] / termsB[nB][5] = the
 * squared distance (an integer), the point-to-plane term and the three colour terms of every point, one direction each (either
 * may be empty).  out[10]: the five sums of A, then of B -- the doubles the reference's loop leaves, bit for bit (the ordered
 * fp64 sums are evaluated block-wise in exact integer arithmetic: csrc/ordered_sum.h).  What tmc2_metrics_compute* run
 * internally on the device-resident terms; exported so that the form can be checked on arbitrary terms.                */
int tmc2_metrics_ordered_sums( tmc2_ctx* ctx, const double* termsA, uint64_t nA, const double* termsB, uint64_t nB, double* out );

/* replaces: PCCMetrics::display / QualityMetrics::print (PCCMetrics.cpp:376-391, 230-279) for one frame: the text the CTC log
 * parsers read, from the numbers of tmc2_metrics_compute (out[3][8], counts[2]), the point counts before duplicate removal
 * and the peak value.  precision: that of the application's std::cout (PccAppEncoder / PccAppMetrics set 9).  text may be
 * NULL to query the size (*needed, including the terminating 0).  Host only.                                       */
int tmc2_metrics_display( const double* out, uint64_t sourcePoints, uint64_t reconstructPoints, const int64_t* counts,
                          uint64_t resolution, int withC2p, int precision, char* text, uint64_t capacity, uint64_t* needed );

/* ---- point-cloud ingest and conformance checksums (host; no device needed) ------------------------------------------- */
/* replaces: PCCPointSet3::read (PccLibCommon/source/PCCPointSet.cpp:464-757) as PCCGroupOfFrames::load (PCCGroupOfFrames.cpp:
 * 46-80) calls it per frame: ASCII and binary_little_endian PLY, x / y / z of 2, 4 or 8 bytes, uchar red / green / blue, float
 * nx / ny / nz (binary bodies only, as in the reference); other vertex properties are skipped.  The cloud lands directly in
 * the caller's buffers -- xyz int16[capacity][3], rgb uint8[capacity][3] (may be NULL), normals double[capacity][3] (NULL:
 * not read) -- e.g. the page-locked staging the upload reads.  threads: parser threads (0 = 8).
 * tmc2_ply_info only reads the header (pointCount, whether red/green/blue resp. float normals are present).          */
int tmc2_ply_info( const char* path, int readNormals, uint64_t* pointCount, int* hasColors, int* hasNormals );
int tmc2_ply_read( const char* path, int16_t* xyz, uint8_t* rgb, double* normals, uint64_t capacity, int threads,
                   uint64_t* pointCount );
/* replaces: PCCPointSet3::write (PCCPointSet.cpp:359-462): the PLY file the reference writes for a reconstructed or decoded
 * frame, byte for byte (ASCII or binary_little_endian; rgb / normals may be NULL)                                  */
int tmc2_ply_write( const char* path, const int16_t* xyz, const uint8_t* rgb, const double* normals, uint64_t n, int asAscii );
/* replaces: PCCPointSet3::computeChecksum( reorderPoints ) / computeMd5 / reorder (PCCPointSet.cpp:222-305): the MD5 the
 * conformance logs carry, over int16 positions then uint8 colours (rgb may be NULL); reorderPoints sorts by (x, y, z) and
 * merges points that share a position (mean colour) first.                                                        */
int tmc2_point_set_checksum( const int16_t* xyz, const uint8_t* rgb, uint64_t n, int reorderPoints, uint8_t digest[16] );

/* replaces: PCCChecksum::write / read (PccLibMetrics/source/PCCChecksum.cpp:112-139): the ".checksum" file kept next to the
 * bitstream, one line per frame.  digests: uint8[frames][16] (tmc2_point_set_checksum).  read: digests may be NULL to query
 * the frame count.                                                                                                  */
int tmc2_checksum_file_write( const char* path, const uint8_t* digests, uint64_t frames );
int tmc2_checksum_file_read( const char* path, uint8_t* digests, uint64_t capacity, uint64_t* frames );

/* ---- host-only pieces of the path (no device needed; used by the CPU test tier) -------------- */
/* the nanoflann-identical tree builder behind tmc2_frame_create: perm = tree order -> original index */
int tmc2_host_kdtree_build( const int16_t* xyz, uint64_t n, uint32_t* perm, uint64_t* nodeCount, int32_t* depth );
/* the placement logic behind tmc2_encoder_pack_flexible on plain records: patches by index (u0 / v0 / patchOrientation out),
 * their block-occupancy pool; order = list order out */
int tmc2_host_pack_flexible( tmc2_patch* patches, int count, const uint8_t* occupancy, int presetWidth, int numTilesHor,
                             double tileHeightToWidthRatio, int32_t* order, int32_t* height );
/* the placement logic behind tmc2_encoder_pack_spatial_consistency on plain records: patches by index (u0 / v0 /
 * patchOrientation out), their block-occupancy pool, the previous frame's patches in list order */
/* the allocation behind tmc2_encoder_global_patch_allocation on plain records.  counts[frames]; patches / matches: all
 * frames back to back, each IN LIST ORDER (in / out); occupancy + occupancyBase[f]: frame f's block-occupancy pool
 * (patch.occOffset is relative to it); tileWidth / tileHeight: the common tile size after resizeTileGeometryVideo.
 * Out: the rebuilt pools back to back in occupancyOut (frame f at occupancyOutBase[f]; occupancyOutBase[frames] = bytes
 * needed -- TMC2_E_INVALID if that exceeds occupancyOutCapacity), the tile size of every frame.                      */
int tmc2_host_global_patch_allocation( int frames, int32_t* counts, tmc2_patch* patches, const uint8_t* occupancy,
                                       const int64_t* occupancyBase, int32_t* matches, int tileWidth, int tileHeight,
                                       int minimumImageWidth, int minimumImageHeight, uint8_t* occupancyOut,
                                       int64_t occupancyOutCapacity, int64_t* occupancyOutBase, int32_t* widths,
                                       int32_t* heights );
int tmc2_host_pack_spatial_consistency( tmc2_patch* patches, int count, const uint8_t* occupancy, const tmc2_patch* previous,
                                        int previousCount, int presetWidth, int numTilesHor, double tileHeightToWidthRatio,
                                        int32_t* order, int32_t* matches, int32_t* height );
/* replaces: PCCEncoder::placeSegments (PccLibEncoder/source/PCCEncoder.cpp:4790-4840) over the patch RECORDS of a whole GOF
 * -- what the rank that holds the records of all frames runs when the frames themselves live on several GPUs.
 * mode 0: packFlexible per frame; 1: frame f > 0 packed against frame f-1 (spatialConsistencyPackFlexible); 2: that chain
 * followed by performDataAdaptiveGPAMethod.  counts[frames]; patches: all frames back to back, in: by index (occOffset
 * into occupancy + occupancyBase[f]), out: in LIST order with placements (occOffset into occupancyOut +
 * occupancyOutBase[f]); matches: out, per list position; widths / heights: the tile each frame is left with (may be NULL).
 * The number of records per frame does not change.  occupancyOutBase has frames + 1 entries (the last = bytes used).    */
int tmc2_host_place_segments( int frames, const int32_t* counts, tmc2_patch* patches, const uint8_t* occupancy,
                              const int64_t* occupancyBase, int mode, int minimumImageWidth, int minimumImageHeight, int numTilesHor,
                              double tileHeightToWidthRatio, int32_t* matches, uint8_t* occupancyOut, int64_t occupancyOutCapacity,
                              int64_t* occupancyOutBase, int32_t* widths, int32_t* heights );
/* the exact spanning-tree orientation behind tmc2_normals_orient (normals in/out, knn = [n][k]) */
int tmc2_host_orient_normals( const int16_t* xyz, uint64_t n, const uint32_t* knn, int k, double* normals );

#ifdef __cplusplus
}
#endif
#endif /* TMC2HIP_H */
