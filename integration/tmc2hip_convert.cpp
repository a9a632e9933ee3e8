// integration/tmc2hip_convert.cpp -- see tmc2hip_adaptor.h: the conversions between the reference's containers and the plain
// buffers of include/tmc2hip.h.  No call into the library here (tmc2hip_adaptor.cpp holds the bodies that call it).
#include <algorithm>

#include "tmc2hip_adaptor.h"

namespace tmc2hip {
using namespace pcc;

void flatten( const PCCPointSet3& cloud, std::vector<int16_t>& xyz, std::vector<uint8_t>& rgb ) {
  const size_t n = cloud.getPointCount();
  xyz.resize( 3 * n );
  rgb.resize( cloud.hasColors() ? 3 * n : 0 );
  for ( size_t i = 0; i < n; ++i )
    for ( int c = 0; c < 3; ++c ) {
      xyz[3 * i + c] = cloud[i][c];
      if ( cloud.hasColors() ) rgb[3 * i + c] = cloud.getColor( i )[c];
    }
}

void toPCCPatches( const tmc2_patch* records, int count, const int16_t* depth0, const int16_t* depth1, const uint8_t* occupancy,
                   size_t occupancyResolution, size_t frameIndex, std::vector<PCCPatch>& patches ) {
  const size_t base = patches.size();  // compute() appends
  patches.resize( base + size_t( count ) );
  for ( int i = 0; i < count; ++i ) {
    const tmc2_patch& r = records[i];
    PCCPatch&         q = patches[base + size_t( i )];
    q.setIndex( size_t( r.index ) );
    (void)frameIndex;  // compute() receives it but leaves PCCPatch::frameIndex_ alone (checked against the reference)
    q.setViewId( size_t( r.viewId ) );  // normal / tangent / bitangent axes and the projection mode follow from the view
    q.setU1( size_t( r.u1 ) ), q.setV1( size_t( r.v1 ) ), q.setD1( size_t( r.d1 ) );
    q.setSizeU( size_t( r.sizeU ) ), q.setSizeV( size_t( r.sizeV ) ), q.setSizeD( size_t( r.sizeD ) );
    q.setSizeDPixel( size_t( r.sizeDPixel ) );
    q.setSizeU0( size_t( r.sizeU0 ) ), q.setSizeV0( size_t( r.sizeV0 ) );
    q.setPatchSize2DXInPixel( size_t( r.size2DXInPixel ) ), q.setPatchSize2DYInPixel( size_t( r.size2DYInPixel ) );
    q.setOccupancyResolution( occupancyResolution );
    q.setD0Count( size_t( r.d0Count ) ), q.setEOMandD1Count( size_t( r.eomAndD1Count ) ), q.setEOMCount( 0 );
    const size_t px = size_t( r.sizeU ) * size_t( r.sizeV ), bl = size_t( r.sizeU0 ) * size_t( r.sizeV0 );
    q.setDepth( 0, std::vector<int16_t>( depth0 + r.depthOffset, depth0 + r.depthOffset + px ) );
    q.setDepth( 1, std::vector<int16_t>( depth1 + r.depthOffset, depth1 + r.depthOffset + px ) );
    std::vector<bool> occ( bl );
    for ( size_t k = 0; k < bl; ++k ) occ[k] = occupancy[r.occOffset + int64_t( k )] != 0;
    q.setOccupancy( occ );
  }
}

void toRecords( const std::vector<PCCPatch>& patches, std::vector<tmc2_patch>& records ) {
  records.assign( patches.size(), tmc2_patch{} );
  int64_t depthOffset = 0, occOffset = 0;
  for ( size_t i = 0; i < patches.size(); ++i ) {
    const PCCPatch& q = patches[i];
    tmc2_patch&     r = records[i];
    r.index           = int32_t( q.getIndex() );
    r.viewId          = int32_t( q.getViewId() );
    r.normalAxis = int32_t( q.getNormalAxis() ), r.tangentAxis = int32_t( q.getTangentAxis() );
    r.bitangentAxis = int32_t( q.getBitangentAxis() ), r.projectionMode = int32_t( q.getProjectionMode() );
    r.u1 = int32_t( q.getU1() ), r.v1 = int32_t( q.getV1() ), r.d1 = int32_t( q.getD1() );
    r.sizeU = int32_t( q.getSizeU() ), r.sizeV = int32_t( q.getSizeV() ), r.sizeD = int32_t( q.getSizeD() );
    r.sizeDPixel = int32_t( q.getSizeDPixel() );
    r.sizeU0 = int32_t( q.getSizeU0() ), r.sizeV0 = int32_t( q.getSizeV0() );
    r.size2DXInPixel = int32_t( q.getPatchSize2DXInPixel() ), r.size2DYInPixel = int32_t( q.getPatchSize2DYInPixel() );
    r.d0Count = int32_t( q.getD0Count() ), r.eomAndD1Count = int32_t( q.getEOMandD1Count() );
    r.u0 = int32_t( q.getU0() ), r.v0 = int32_t( q.getV0() ), r.patchOrientation = int32_t( q.getPatchOrientation() );
    r.depthOffset = depthOffset, r.occOffset = occOffset;
    depthOffset += int64_t( q.getSizeU() * q.getSizeV() );
    occOffset += int64_t( q.getSizeU0() * q.getSizeV0() );
  }
}

void applyPacking( const tmc2_patch* recordsByIndex, const int32_t* order, const int32_t* matches, int count,
                   std::vector<PCCPatch>& patches ) {
  std::vector<PCCPatch> byIndex;
  byIndex.swap( patches );
  patches.reserve( size_t( count ) );
  for ( int k = 0; k < count; ++k ) {
    const tmc2_patch& r = recordsByIndex[order[k]];
    patches.push_back( byIndex[size_t( order[k] )] );
    PCCPatch& q = patches.back();
    q.setU0( size_t( r.u0 ) ), q.setV0( size_t( r.v0 ) ), q.setPatchOrientation( size_t( r.patchOrientation ) );
    q.setBestMatchIdx( matches ? matches[k] : -1 );
  }
}

void applyPackedList( const tmc2_patch* list, const int32_t* matches, const uint8_t* occupancy, int count,
                      std::vector<PCCPatch>& patches ) {
  std::vector<PCCPatch> created;
  created.swap( patches );
  std::vector<int64_t> depthOffset( created.size() );  // as toRecords / tmc2_frame_get_patches number the depth pools
  int64_t              at = 0;
  for ( size_t i = 0; i < created.size(); ++i ) {
    depthOffset[i] = at;
    at += int64_t( created[i].getSizeU() * created[i].getSizeV() );
  }
  patches.reserve( size_t( count ) );
  for ( int k = 0; k < count; ++k ) {
    const tmc2_patch& r = list[k];
    const size_t      i = size_t( std::lower_bound( depthOffset.begin(), depthOffset.end(), r.depthOffset ) - depthOffset.begin() );
    patches.push_back( created[i] );
    PCCPatch& q = patches.back();
    q.setIndex( size_t( r.index ) );
    q.setSizeU0( size_t( r.sizeU0 ) ), q.setSizeV0( size_t( r.sizeV0 ) );
    q.setU0( size_t( r.u0 ) ), q.setV0( size_t( r.v0 ) ), q.setPatchOrientation( size_t( r.patchOrientation ) );
    q.setBestMatchIdx( matches ? matches[k] : -1 );
    std::vector<bool> blocks( size_t( r.sizeU0 ) * size_t( r.sizeV0 ) );
    for ( size_t b = 0; b < blocks.size(); ++b ) blocks[b] = occupancy[r.occOffset + int64_t( b )] != 0;
    q.setOccupancy( blocks );
  }
}

void toFrameImages( const uint8_t* occupancy, const uint8_t* occVideo, const uint32_t* blockToPatch, const uint16_t* geometryD0,
                    const uint16_t* geometryD1, size_t W, size_t H, size_t occupancyPrecision, std::vector<uint32_t>& occupancyMap,
                    std::vector<size_t>& blockToPatchOut, PCCImage<uint8_t, 3>& occupancyFrame, PCCImage<uint16_t, 3>& d0,
                    PCCImage<uint16_t, 3>& d1 ) {
  occupancyMap.assign( occupancy, occupancy + W * H );                          // PCCFrameContext::occupancyMap_ (0 / 1)
  blockToPatchOut.assign( blockToPatch, blockToPatch + ( W / 16 ) * ( H / 16 ) );  // list position + 1, 0 = none
  const size_t Wv = W / occupancyPrecision, Hv = H / occupancyPrecision;
  occupancyFrame.resize( Wv, Hv, PCCCOLORFORMAT::YUV420 );  // generateOccupancyMapVideo: luma 0 / 1, chroma untouched (0)
  std::copy( occVideo, occVideo + Wv * Hv, occupancyFrame.getChannel( 0 ).begin() );
  PCCImage<uint16_t, 3>* maps[2] = {&d0, &d1};
  const uint16_t*        src[2]  = {geometryD0, geometryD1};
  for ( int m = 0; m < 2; ++m ) {  // generateIntraImage: YUV444, depth in the first channel, the others zero
    maps[m]->resize( W, H, PCCCOLORFORMAT::YUV444 );
    std::copy( src[m], src[m] + W * H, maps[m]->getChannel( 0 ).begin() );
    std::fill( maps[m]->getChannel( 1 ).begin(), maps[m]->getChannel( 1 ).end(), uint16_t( 0 ) );
    std::fill( maps[m]->getChannel( 2 ).begin(), maps[m]->getChannel( 2 ).end(), uint16_t( 0 ) );
  }
}

void toAttributeFrames( const uint8_t* attribute, size_t W, size_t H, PCCImage<uint16_t, 3>& t0, PCCImage<uint16_t, 3>& t1 ) {
  PCCImage<uint16_t, 3>* maps[2] = {&t0, &t1};
  for ( int m = 0; m < 2; ++m ) {  // generateAttributeVideo: RGB444, 8-bit values in 16-bit samples
    maps[m]->resize( W, H, PCCCOLORFORMAT::RGB444 );
    for ( int c = 0; c < 3; ++c ) {
      const uint8_t* src = attribute + ( size_t( m ) * 3 + size_t( c ) ) * W * H;
      std::copy( src, src + W * H, maps[m]->getChannel( size_t( c ) ).begin() );
    }
  }
}

void toReconstruction( const int16_t* xyz, const uint8_t* rgb, const uint32_t* pointToPixel, size_t n, PCCPointSet3& cloud,
                       std::vector<PCCVector3<size_t>>& pointToPixelOut ) {
  cloud.clear();
  cloud.addColors();
  cloud.resize( n );
  pointToPixelOut.resize( n );
  for ( size_t i = 0; i < n; ++i ) {
    cloud[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    if ( rgb ) cloud.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
    pointToPixelOut[i] = PCCVector3<size_t>( pointToPixel[3 * i], pointToPixel[3 * i + 1], pointToPixel[3 * i + 2] );
  }
}

}  // namespace tmc2hip
