// integration/tmc2hip_dropins.cpp -- reference-side binding, part 2: the metric object, the decoder's per-frame finish and the
// k-d tree.  Flatten, call the C entry points of include/tmc2hip.h, write back; no algorithm here.
#include <cstdio>
#include <cstring>

#include "tmc2hip_adaptor.h"

using namespace pcc;

namespace tmc2hip {

// ---- PCCMetrics ---------------------------------------------------------------------------------------------------------
MetricsDropIn::MetricsDropIn( int device ) { tmc2_ctx_create( device, &ctx_ ); }
MetricsDropIn::~MetricsDropIn() {
  if ( ctx_ ) tmc2_ctx_destroy( ctx_ );
}
bool MetricsDropIn::accepts( const PCCMetricsParameters& p ) const {
  return ctx_ && p.computeMetrics_ && p.computeC2c_ && p.computeColor_ && !p.computeLidar_ && !p.computeReflectance_ &&
         !p.computeHausdorff_ && p.dropDuplicates_ == 2 && p.neighborsProc_ == 1;
}
int MetricsDropIn::compute( const PCCGroupOfFrames& sources, const PCCGroupOfFrames& reconstructs, const PCCGroupOfFrames& normals ) {
  if ( !ctx_ ) return TMC2_E_NO_DEVICE;
  bool c2p = params_.computeC2p_;
  if ( normals.getFrameCount() != 0 && sources.getFrameCount() != normals.getFrameCount() ) c2p = false;  // PCCMetrics.cpp:328-330
  if ( sources.getFrameCount() != reconstructs.getFrameCount() ) return TMC2_E_INVALID;
  for ( size_t i = 0; i < sources.getFrameCount(); ++i ) {
    std::vector<int16_t> sx, rx;
    std::vector<uint8_t> sc, rc;
    flatten( sources[i], sx, sc );
    flatten( reconstructs[i], rx, rc );
    std::vector<double> nrm;
    const bool          withNormals = normals.getFrameCount() != 0 && normals[i].getPointCount() > 0;  // :364-367
    if ( withNormals ) {
      // copyNormals looks every source point up in the normal cloud by position (PCCPointSet.cpp:2282-2320); the applications
      // load both from the same frame, point for point
      const PCCPointSet3& nc = normals[i];
      if ( nc.getPointCount() != sources[i].getPointCount() ) return TMC2_E_INVALID;
      nrm.resize( 3 * nc.getPointCount() );
      for ( size_t k = 0; k < nc.getPointCount(); ++k ) {
        if ( nc[k][0] != sources[i][k][0] || nc[k][1] != sources[i][k][1] || nc[k][2] != sources[i][k][2] ) return TMC2_E_INVALID;
        for ( int d = 0; d < 3; ++d ) nrm[3 * k + d] = nc.getNormals()[k][d];
      }
    }
    std::array<double, 24>  q{};
    std::array<int64_t, 2>  counts{};
    const int rc2 = tmc2_metrics_compute( ctx_, sx.data(), sc.data(), sx.size() / 3, rx.data(), rc.data(), rx.size() / 3,
                                          withNormals ? nrm.data() : nullptr, double( params_.resolution_ ), q.data(), counts.data() );
    if ( rc2 != TMC2_OK ) return rc2;
    q_.push_back( q );
    counts_.push_back( counts );
    points_.push_back( {uint64_t( sx.size() / 3 ), uint64_t( rx.size() / 3 )} );
    withC2p_.push_back( c2p );
  }
  return TMC2_OK;
}
int MetricsDropIn::display() {
  printf( "Metrics results \n" );
  for ( size_t i = 0; i < q_.size(); ++i ) {
    uint64_t needed = 0;
    int      rc = tmc2_metrics_display( q_[i].data(), points_[i][0], points_[i][1], counts_[i].data(), uint64_t( params_.resolution_ ),
                                        withC2p_[i] ? 1 : 0, int( std::cout.precision() ), nullptr, 0, &needed );
    if ( rc != TMC2_OK ) return rc;
    std::string text( size_t( needed ), '\0' );
    rc = tmc2_metrics_display( q_[i].data(), points_[i][0], points_[i][1], counts_[i].data(), uint64_t( params_.resolution_ ),
                               withC2p_[i] ? 1 : 0, int( std::cout.precision() ), &text[0], needed, &needed );
    if ( rc != TMC2_OK ) return rc;
    // (tmc2_metrics_display writes one frame's report, heading included: the group's heading went out above)
    const char*  body = text.c_str();
    const char   head[] = "Metrics results \n";
    if ( strncmp( body, head, sizeof( head ) - 1 ) == 0 ) body += sizeof( head ) - 1;
    fputs( body, stdout );
  }
  return TMC2_OK;
}

// ---- PCCDecoder::decode, per frame ----------------------------------------------------------------------------------------
DecoderDropIn::DecoderDropIn( int device ) { tmc2_ctx_create( device, &ctx_ ); }
DecoderDropIn::~DecoderDropIn() {
  if ( ctx_ ) tmc2_ctx_destroy( ctx_ );
}
int DecoderDropIn::reconstructFrame( PCCContext& context, size_t frameIdx, size_t occupancyPrecision, size_t gridSize,
                                     double thresholdSmoothing, PCCPointSet3& reconstruct ) {
  if ( !ctx_ ) return TMC2_E_NO_DEVICE;
  auto&                   tile = context[frameIdx].getTile( 0 );
  std::vector<tmc2_patch> records;
  toRecords( tile.getPatches(), records );
  const size_t W = tile.getWidth(), H = tile.getHeight(), p = occupancyPrecision;
  // the decoded occupancy video frame (luma) and the two decoded geometry frames (luma), as plain planes
  const auto&           occ = context.getVideoOccupancyMap().getFrame( frameIdx );
  std::vector<uint8_t>  occVideo( ( W / p ) * ( H / p ) );
  for ( size_t v = 0; v < H / p; ++v )
    for ( size_t u = 0; u < W / p; ++u ) occVideo[v * ( W / p ) + u] = uint8_t( occ.getValue( 0, u, v ) );
  std::vector<uint16_t> geo( 2 * W * H );
  for ( size_t m = 0; m < 2; ++m ) {
    const auto& g = context.getVideoGeometryMultiple()[0].getFrame( 2 * frameIdx + m );
    for ( size_t v = 0; v < H; ++v )
      for ( size_t u = 0; u < W; ++u ) geo[( m * H + v ) * W + u] = g.getValue( 0, u, v );
  }
  tmc2_frame* f  = nullptr;
  int         rc = tmc2_decoder_frame_create( ctx_, records.data(), int( records.size() ), int( W ), int( H ), int( p ), occVideo.data(),
                                              geo.data(), &f );
  if ( rc != TMC2_OK ) return rc;
  struct Guard {
    tmc2_frame* f;
    ~Guard() { tmc2_frame_destroy( f ); }
  } guard{f};
  if ( ( rc = tmc2_codec_generate_point_cloud( f ) ) != TMC2_OK ) return rc;
  // the decoded attribute frames after the colour conversion: 16-bit 4:4:4, frames 2f and 2f + 1 of the attribute video
  std::vector<uint16_t> att( 2 * 3 * W * H );
  for ( size_t m = 0; m < 2; ++m ) {
    const auto& a = context.getVideoAttributesMultiple()[0].getFrame( 2 * frameIdx + m );
    for ( size_t c = 0; c < 3; ++c )
      for ( size_t v = 0; v < H; ++v )
        for ( size_t u = 0; u < W; ++u ) att[( ( m * 3 + c ) * H + v ) * W + u] = a.getValue( c, u, v );
  }
  if ( ( rc = tmc2_codec_identify_boundary_points( f ) ) != TMC2_OK ) return rc;
  if ( ( rc = tmc2_codec_color_point_cloud( f, att.data() ) ) != TMC2_OK ) return rc;
  if ( ( rc = tmc2_codec_smooth_point_cloud_postprocess( f, int( gridSize ), thresholdSmoothing ) ) != TMC2_OK ) return rc;
  if ( ( rc = tmc2_codec_transfer_colors_16bit_bp( f ) ) != TMC2_OK ) return rc;
  if ( ( rc = tmc2_codec_convert_yuv16_to_rgb8( f ) ) != TMC2_OK ) return rc;
  const int64_t M = tmc2_frame_recon_count( f );
  if ( M < 0 ) return TMC2_E_STATE;
  const size_t          count = size_t( M );
  std::vector<int16_t>  xyz( 3 * count );
  std::vector<uint16_t> c16( 3 * count ), bt( count );
  std::vector<uint8_t>  rgb( 3 * count );
  if ( ( rc = tmc2_frame_get_post_reconstruction( f, xyz.data(), c16.data(), rgb.data(), bt.data() ) ) != TMC2_OK ) return rc;
  reconstruct.clear();
  reconstruct.addColors();
  reconstruct.addColors16bit();
  reconstruct.resize( count );
  for ( size_t i = 0; i < count; ++i ) {
    reconstruct[i] = PCCPoint3D( xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] );
    reconstruct.setColor( i, PCCColor3B( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] ) );
    reconstruct.setColor16bit( i, PCCColor16bit( c16[3 * i], c16[3 * i + 1], c16[3 * i + 2] ) );
    reconstruct.setBoundaryPointType( i, bt[i] );
  }
  return TMC2_OK;
}

// ---- PCCKdTree ------------------------------------------------------------------------------------------------------------
KdTreeDropIn::KdTreeDropIn( int device ) { tmc2_ctx_create( device, &ctx_ ); }
KdTreeDropIn::~KdTreeDropIn() {
  if ( frame_ ) tmc2_frame_destroy( frame_ );
  if ( ctx_ ) tmc2_ctx_destroy( ctx_ );
}
int KdTreeDropIn::init( const PCCPointSet3& pointCloud ) {
  if ( !ctx_ ) return TMC2_E_NO_DEVICE;
  if ( frame_ ) tmc2_frame_destroy( frame_ ), frame_ = nullptr;
  std::vector<int16_t> xyz;
  std::vector<uint8_t> rgb;
  flatten( pointCloud, xyz, rgb );
  const int rc = tmc2_frame_create( ctx_, xyz.data(), nullptr, xyz.size() / 3, &frame_ );
  return rc != TMC2_OK ? rc : tmc2_kdtree_build( frame_ );
}
int KdTreeDropIn::searchBatch( const std::vector<PCCPoint3D>& points, size_t k, std::vector<PCCNNResult>& results ) const {
  if ( !frame_ ) return TMC2_E_STATE;
  std::vector<int16_t> q( 3 * points.size() );
  for ( size_t i = 0; i < points.size(); ++i )
    for ( int d = 0; d < 3; ++d ) q[3 * i + d] = int16_t( points[i][d] );
  std::vector<uint32_t> idx( points.size() * k ), d2( points.size() * k );
  const int rc = tmc2_kdtree_search( frame_, q.data(), points.size(), int( k ), idx.data(), d2.data() );
  if ( rc != TMC2_OK ) return rc;
  results.resize( points.size() );
  for ( size_t i = 0; i < points.size(); ++i ) {
    results[i].resize( k );
    for ( size_t j = 0; j < k; ++j ) results[i].indices( j ) = idx[i * k + j], results[i].dist( j ) = double( d2[i * k + j] );
  }
  return TMC2_OK;
}
int KdTreeDropIn::search( const PCCPoint3D& point, size_t k, PCCNNResult& results ) const {
  std::vector<PCCNNResult> one;
  const int                rc = searchBatch( {point}, k, one );
  if ( rc == TMC2_OK ) results = one[0];
  return rc;
}

}  // namespace tmc2hip
