/* oracle/oracle.h -- TEST INFRASTRUCTURE.  Shared POD types of the two CPU checkers:
 *
 *   oracle/port_*.cpp      our own CPU restatement of the TMC2 hot path  -> oracle/liboracle.so
 *   oracle/ref_harness.cpp a driver around the UNMODIFIED reference      -> oracle/_ref/libtmc2ref.so
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these libraries.
 * The product (mpeg-pcc-tmc2_amd/) never includes this header and never links either library.
 */
#ifndef TMC2_ORACLE_H
#define TMC2_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flat mirror of the fields of PCCPatchSegmenter3Parameters
 * (reference: source/lib/PccLibEncoder/include/PCCPatchSegmenter.h:48-100) that are live under the
 * CTC lossy configurations.  Flags that the CTC configs keep off (EOM, surface separation, patch
 * expansion, high-gradient separation, point-cloud partitioning, additional projection planes,
 * grid-based segmentation, createSubPointCloud) are fixed to off and absoluteD1 to on.          */
typedef struct orc_seg_params {
  int32_t nnNormalEstimation;               /* 16  */
  int32_t normalOrientation;                /* 1 = spanning tree */
  int32_t gridBasedRefineSegmentation;      /* 1   */
  int32_t maxNNCountRefineSegmentation;     /* 1024 */
  int32_t iterationCountRefineSegmentation; /* 10 (50 longdress, 20 basketball) */
  int32_t voxelDimensionRefineSegmentation; /* 4   */
  int32_t searchRadiusRefineSegmentation;   /* 192 */
  int32_t occupancyResolution;              /* 16  */
  int32_t enablePatchSplitting;             /* 1   */
  int32_t maxPatchSize;                     /* 1024 */
  int32_t quantizerSizeX;                   /* 16  */
  int32_t quantizerSizeY;                   /* 16  */
  int32_t minPointCountPerCC;               /* 16  */
  int32_t maxNNCountPatchSegmentation;      /* 16  */
  int32_t surfaceThickness;                 /* 4   */
  int32_t mapCountMinus1;                   /* 1   */
  int32_t minLevel;                         /* 64  */
  int32_t maxAllowedDepth;                  /* 255 */
  int32_t geometryBitDepth2D;               /* 8   */
  int32_t geometryBitDepth3D;               /* 11 (vox10) / 12 (vox11) */
  double  maxAllowedDist2RawPointsDetection; /* 9 */
  double  maxAllowedDist2RawPointsSelection; /* 1 */
  double  lambdaRefineSegmentation;          /* 3 */
  double  weightNormal[3];                   /* from calculateWeightNormal (S0) */
} orc_seg_params;

/* One patch as produced by PCCPatchSegmenter3::segmentPatches
 * (reference: PCCPatchSegmenter.cpp:910-1290, fields of PCCPatch.h).  depthOffset indexes the int16
 * depth pools (sizeU*sizeV entries per map), occOffset the per-block occupancy pool.               */
typedef struct orc_patch {
  int32_t index, viewId;
  int32_t normalAxis, tangentAxis, bitangentAxis, projectionMode;
  int32_t u1, v1, d1;
  int32_t sizeU, sizeV, sizeD, sizeDPixel;
  int32_t sizeU0, sizeV0;
  int32_t size2DXInPixel, size2DYInPixel;
  int32_t d0Count, eomAndD1Count;
  int32_t u0, v0, patchOrientation;   /* filled by packing (S10) */
  int64_t depthOffset;
  int64_t occOffset;
} orc_patch;

#ifdef __cplusplus
}
#endif
#endif
