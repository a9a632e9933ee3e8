// gpa.cpp -- global patch allocation over the frames of a GOF (S10', random-access condition), host side.
//
// Replaces PCCEncoder::performDataAdaptiveGPAMethod (reference: source/lib/PccLibEncoder/source/PCCEncoder.cpp:6821-6971)
// and the members it drives: initializeSubContext / clearCurrentGPAPatchDataInfor / generateGlobalPatches (:6973-7057),
// unionPatchGenerationAndPacking (:7059-7226), packingFirstFrame (:7228-7364), updatePatchInformation /
// updateGPAPatchInformation (:7366-7529), performGPAPacking and packingWith(out)RefForFirstFrameNoglobalPatch (:7531-7840),
// with GPAPatchData (PccLibCommon/include/PCCPatch.h:42-71) and checkFitPatchCanvasForGPA / patchBlock2CanvasBlockForGPA
// (PccLibCommon/source/PCCPatch.cpp:617-692), as placeSegments runs them when constrainedPack = 1 and
// globalPatchAllocation = 1 (packingStrategy 1, two orientations, safeguard 0, one tile per frame).
//
// What it does: the frames arrive packed one by one (S10 / S10' chain).  They are then grouped greedily into
// "sub-contexts" of consecutive frames.  Inside a sub-context a patch of the opening frame starts a TRACK that is
// extended frame by frame to the patch of the same view whose bounding box overlaps the track's last patch most (IoU >
// 0.2).  The patches of a track are placed at ONE position in all frames: the position found for the UNION of their block
// occupancies on a canvas that holds only the unions; every frame is then completed with its untracked patches around
// the unions.  A sub-context stops growing -- and the last accepted layout of its frames is committed -- when fewer than
// 15 % of the tracks survive, when the unions alone are taller than the minimum image, or when a completed frame is.
//
// Like the per-frame packers (packing.cpp) this is sequential first-fit search over a few hundred boxes on a canvas of a
// few thousand blocks: host work of well under a millisecond per frame, between the segmentation kernels and the raster
// kernels.  Canvases are bit rows, a probe is a handful of ANDs; the reference probes block by block through a copy of
// the canvas.
#include <algorithm>
#include <cstring>

#include "internal.h"

namespace tmc2 {

namespace {
constexpr int kUnset = -1;

// a canvas of blocks, one bit each; rows of 64-bit words
class BitCanvas {
 public:
  BitCanvas( int w, int h ) : w_( w ), h_( h ), words_( ( w + 63 ) / 64 ), bits_( size_t( words_ ) * size_t( h ), 0 ) {}
  int  width() const { return w_; }
  int  height() const { return h_; }
  void doubleHeight() {
    h_ = std::max( 1, 2 * h_ );
    bits_.resize( size_t( words_ ) * size_t( h_ ), 0 );
  }
  // a box of sizeU0 x sizeV0 blocks laid down with an orientation at (u0, v0): inside the canvas and on free blocks only
  bool accepts( int sizeU0, int sizeV0, int u0, int v0, int orient ) const {
    if ( orient != 0 && orient != 1 ) return false;
    const int bw = orient == 0 ? sizeU0 : sizeV0, bh = orient == 0 ? sizeV0 : sizeU0;
    if ( u0 < 0 || v0 < 0 || u0 + bw > w_ || v0 + bh > h_ ) return false;
    for ( int y = v0; y < v0 + bh; ++y ) {
      const uint64_t* row = &bits_[size_t( y ) * words_];
      for ( int x = u0; x < u0 + bw; ) {
        const int      bit = x & 63, span = std::min( 64 - bit, u0 + bw - x );
        const uint64_t m   = ( span == 64 ? ~0ull : ( ( 1ull << span ) - 1ull ) ) << bit;
        if ( row[x >> 6] & m ) return false;
        x += span;
      }
    }
    return true;
  }
  // occupy the blocks of the box that the occupancy (rows of `stride` blocks) marks
  void stamp( const uint8_t* occ, int stride, int sizeU0, int sizeV0, int u0, int v0, int orient ) {
    for ( int vb = 0; vb < sizeV0; ++vb )
      for ( int ub = 0; ub < sizeU0; ++ub )
        if ( occ[size_t( vb ) * stride + ub] ) {
          const int x = orient == 0 ? u0 + ub : u0 + vb, y = orient == 0 ? v0 + vb : v0 + ub;
          bits_[size_t( y ) * words_ + ( x >> 6 )] |= 1ull << ( x & 63 );
        }
  }

 private:
  int                   w_, h_, words_;
  std::vector<uint64_t> bits_;
};

// one layout hypothesis of a patch: the box it takes part in the packing with and where that box went
struct Layout {
  bool                 claimed = false;  // a track took this patch in the current round
  int                  track   = kUnset;  // the track (= union) it belongs to; kUnset: packed on its own
  int                  sizeU0 = 0, sizeV0 = 0;
  int                  u0 = kUnset, v0 = kUnset, orient = kUnset;
  std::vector<uint8_t> occ;
  bool                 tracked() const { return track != kUnset; }
};
struct Item {
  tmc2_patch           rec;
  std::vector<uint8_t> occ;     // rec.sizeU0 x rec.sizeV0
  int32_t              match;   // list position of the matched patch in the previous frame, -1
  bool                 global = false;
  Layout               trial, kept;
};
struct Tile {
  std::vector<Item> items;
  int               width = 0, height = 0;
  int               trialW = 0, trialH = 0, keptW = 0, keptH = 0;
};
struct UnionBox {
  bool                 alive  = false;
  int                  sizeU0 = 0, sizeV0 = 0, u0 = 0, v0 = 0, orient = 0;
  std::vector<uint8_t> occ;
};
using Track = std::vector<std::pair<int, int>>;  // (frame, list position) per frame of the sub-context

inline int preferredOrientation( int sizeU0, int sizeV0, int attempt ) {  // wide boxes are tried turned first
  return ( sizeU0 > sizeV0 ) == ( attempt == 0 ) ? 1 : 0;
}
inline float boxIoU( const tmc2_patch& a, const tmc2_patch& b ) {  // pcc::computeIOU, PCCPatchSegmenter.cpp:1563-1570
  const int x1 = std::max( a.u1, b.u1 ), y1 = std::max( a.v1, b.v1 );
  int       w  = std::min( a.u1 + a.sizeU, b.u1 + b.sizeU ) - x1, h = std::min( a.v1 + a.sizeV, b.v1 + b.sizeV ) - y1;
  if ( w <= 0 || h <= 0 ) w = h = 0;
  const int inter = w * h;
  return static_cast<float>( inter ) / ( a.sizeU * a.sizeV + b.sizeU * b.sizeV - inter );
}

class Allocator {
 public:
  Allocator( std::vector<Tile>& tiles, int minW, int minH, int occRes ) : T( tiles ), minW_( minW ), minH_( minH ), res_( occRes ) {}
  int run();

 private:
  std::vector<Tile>& T;
  const int          minW_, minH_, res_;
  bool               runaway_ = false, outside_ = false;
  static constexpr int kRunawayRows = 1 << 20;

  void extent( Tile& t, const Layout& l ) const {
    const int spanU = l.orient == 0 ? l.sizeU0 : l.sizeV0, spanV = l.orient == 0 ? l.sizeV0 : l.sizeU0;
    t.trialW = std::max( t.trialW, ( l.u0 + spanU ) * res_ );
    t.trialH = std::max( t.trialH, ( l.v0 + spanV ) * res_ );
  }
  // first free position in raster order for a box whose orientation is fixed, or for both orientations per position
  static bool scanFixed( const BitCanvas& c, Layout& l ) {
    for ( int v = 0; v < c.height(); ++v )
      for ( int u = 0; u < c.width(); ++u )
        if ( c.accepts( l.sizeU0, l.sizeV0, u, v, l.orient ) ) {
          l.u0 = u, l.v0 = v;
          return true;
        }
    return false;
  }
  static bool scanFree( const BitCanvas& c, Layout& l, int prefU0, int prefV0 ) {
    for ( int v = 0; v < c.height(); ++v )
      for ( int u = 0; u < c.width(); ++u )
        for ( int k = 0; k < 2; ++k ) {
          const int o = preferredOrientation( prefU0, prefV0, k );
          if ( c.accepts( l.sizeU0, l.sizeV0, u, v, o ) ) {
            l.u0 = u, l.v0 = v, l.orient = o;
            return true;
          }
        }
    return false;
  }
  // a patch packed on its own: at its match's place (and in any case with its orientation) when there is one
  void placeAlone( BitCanvas& c, Layout& l, const tmc2_patch& rec, bool haveAnchor, int anchorU0, int anchorV0, int anchorOrient ) {
    for ( ;; ) {
      bool done;
      if ( haveAnchor ) {
        l.orient = anchorOrient, l.u0 = anchorU0, l.v0 = anchorV0;
        done     = c.accepts( l.sizeU0, l.sizeV0, l.u0, l.v0, l.orient ) || scanFixed( c, l );
      } else {
        done = scanFree( c, l, rec.sizeU0, rec.sizeV0 );
      }
      if ( done ) return;
      if ( c.height() > kRunawayRows ) {  // nothing can be placed (an anchor without orientation): the reference spins here
        runaway_ = true;
        l.u0 = l.v0 = 0, l.orient = 0;
        return;
      }
      c.doubleHeight();
    }
  }
  void openSubContext( int fi, bool hasRef );
  void extendTracks( int fi, std::vector<Track>& tracks, int last );
  int  packUnions( const std::vector<Track>& tracks, int frameWidth, std::vector<UnionBox>& unions, int refFrame, bool useRef,
                   int& alive );
  void adoptUnionBoxes( int first, int end, const std::vector<UnionBox>& unions );
  bool completeFrames( int first, int end, const std::vector<UnionBox>& unions, int frameWidth, int unionsHeight, bool useRef );
  int  commit( int first, int end );
  void keepTrial( int first, int end ) {
    for ( int j = first; j < end; ++j ) {
      T[j].keptW = T[j].trialW, T[j].keptH = T[j].trialH;
      for ( auto& it : T[j].items ) it.kept = it.trial;
    }
  }
  void dropTrial( int first, int end ) {
    for ( int j = first; j < end; ++j )
      for ( auto& it : T[j].items ) it.trial = Layout();
  }
};

// the opening frame of a sub-context on its own: every patch is a track of its own and keeps its own box
void Allocator::openSubContext( int fi, bool hasRef ) {
  Tile& t     = T[fi];
  int   sizeU = t.width / res_, sizeV = 0;
  for ( auto& it : t.items ) {
    sizeV = std::max( sizeV, std::max( it.rec.sizeU0, it.rec.sizeV0 ) );
    sizeU = std::max( sizeU, it.rec.sizeU0 + 1 );
  }
  t.trialW = sizeU * res_, t.trialH = sizeV * res_;
  BitCanvas canvas( sizeU, sizeV );
  for ( auto& it : t.items ) {
    Layout& l = it.trial;
    l.occ     = it.occ;
    l.sizeU0 = it.rec.sizeU0, l.sizeV0 = it.rec.sizeV0;
    const bool anchored = hasRef && it.match != -1;
    const tmc2_patch* a = anchored ? &T[fi - 1].items[size_t( it.match )].rec : nullptr;
    placeAlone( canvas, l, it.rec, anchored, a ? a->u0 : 0, a ? a->v0 : 0, a ? a->patchOrientation : 0 );
    canvas.stamp( it.occ.data(), it.rec.sizeU0, l.sizeU0, l.sizeV0, l.u0, l.v0, l.orient );
    extent( t, l );
  }
}

// every live track looks in frame fi for its continuation; a track that finds none dies for this sub-context
void Allocator::extendTracks( int fi, std::vector<Track>& tracks, int last ) {
  auto& cur = T[fi].items;
  for ( auto& tr : tracks ) {
    if ( tr.empty() ) continue;
    const tmc2_patch& tail = T[tr[size_t( last )].first].items[size_t( tr[size_t( last )].second )].rec;
    float             best = 0.0F;
    int               pick = -1;
    for ( size_t c = 0; c < cur.size(); ++c ) {
      if ( cur[c].rec.viewId != tail.viewId || cur[c].trial.claimed ) continue;
      const float iou = boxIoU( tail, cur[c].rec );
      if ( iou > best ) best = iou, pick = int( c );
    }
    if ( best > 0.2F ) {
      cur[size_t( pick )].trial.claimed = true;
      tr.emplace_back( fi, pick );
    } else {
      tr.clear();
    }
  }
  for ( size_t k = 0; k < tracks.size(); ++k )
    for ( auto& e : tracks[k] ) T[e.first].items[size_t( e.second )].trial.track = int( k );
}

// unions of the surviving tracks, packed among themselves; returns the height (pixels) they need
int Allocator::packUnions( const std::vector<Track>& tracks, int frameWidth, std::vector<UnionBox>& unions, int refFrame,
                           bool useRef, int& alive ) {
  unions.assign( tracks.size(), UnionBox() );
  alive     = 0;
  int sizeU = frameWidth / res_, sizeV = 0;
  for ( size_t k = 0; k < tracks.size(); ++k ) {
    if ( tracks[k].empty() ) continue;
    UnionBox& U = unions[k];
    U.alive     = true;
    ++alive;
    for ( auto& e : tracks[k] ) {
      const tmc2_patch& r = T[e.first].items[size_t( e.second )].rec;
      U.sizeU0 = std::max( U.sizeU0, r.sizeU0 ), U.sizeV0 = std::max( U.sizeV0, r.sizeV0 );
    }
    U.occ.assign( size_t( U.sizeU0 ) * U.sizeV0, 0 );
    for ( auto& e : tracks[k] ) {
      const Item& it = T[e.first].items[size_t( e.second )];
      for ( int v = 0; v < it.rec.sizeV0; ++v )
        for ( int u = 0; u < it.rec.sizeU0; ++u ) U.occ[size_t( v ) * U.sizeU0 + u] |= it.occ[size_t( v ) * it.rec.sizeU0 + u] ? 1 : 0;
    }
    // a sub-context that follows another one inherits orientations from the frame before it, through the opening patch's match
    U.orient = kUnset;
    if ( useRef ) {
      const Item& head = T[tracks[k][0].first].items[size_t( tracks[k][0].second )];
      if ( head.match != -1 ) U.orient = T[refFrame].items[size_t( head.match )].rec.patchOrientation;
    }
    sizeU = std::max( sizeU, U.sizeU0 + 1 );
    sizeV = std::max( sizeV, U.sizeV0 + 1 );
  }
  int       height = sizeV * res_;
  BitCanvas canvas( sizeU, sizeV );
  for ( auto& U : unions ) {
    if ( !U.alive ) continue;
    Layout l;
    l.sizeU0 = U.sizeU0, l.sizeV0 = U.sizeV0, l.orient = U.orient;
    for ( ;; ) {
      // In a sub-context with a reference frame an orientation, once set, is binding -- and the reference SETS it by merely
      // trying: a union without an inherited orientation gets both orientations at the first position only; if neither
      // fits there, the one tried last stays and is the only one tried from the next position on, also after the canvas
      // has been doubled.  Without a reference frame both orientations are tried everywhere.
      bool found = false;
      if ( useRef && l.orient == kUnset && canvas.width() > 0 && canvas.height() > 0 ) {
        for ( int k = 0; k < 2 && !found; ++k ) {
          const int o = preferredOrientation( U.sizeU0, U.sizeV0, k );
          if ( canvas.accepts( l.sizeU0, l.sizeV0, 0, 0, o ) ) l.u0 = 0, l.v0 = 0, l.orient = o, found = true;
        }
        if ( !found ) l.orient = preferredOrientation( U.sizeU0, U.sizeV0, 1 );
      }
      if ( !found ) found = ( useRef && l.orient != kUnset ) ? scanFixed( canvas, l ) : scanFree( canvas, l, U.sizeU0, U.sizeV0 );
      if ( found ) break;
      if ( canvas.height() > kRunawayRows ) {
        runaway_ = true;
        l.u0 = l.v0 = 0, l.orient = 0;
        break;
      }
      canvas.doubleHeight();
    }
    U.u0 = l.u0, U.v0 = l.v0, U.orient = l.orient;
    canvas.stamp( U.occ.data(), U.sizeU0, U.sizeU0, U.sizeV0, U.u0, U.v0, U.orient );
    height = std::max( height, ( U.v0 + ( U.orient == 0 ? U.sizeV0 : U.sizeU0 ) ) * res_ );
  }
  return height;
}

// tracked patches take the box of their union (their own occupancy, re-laid on the union's row length)
void Allocator::adoptUnionBoxes( int first, int end, const std::vector<UnionBox>& unions ) {
  for ( int j = first; j < end; ++j )
    for ( auto& it : T[j].items ) {
      Layout& l = it.trial;
      if ( !l.tracked() ) {
        l.sizeU0 = it.rec.sizeU0, l.sizeV0 = it.rec.sizeV0, l.occ = it.occ;
        continue;
      }
      const UnionBox& U = unions[size_t( l.track )];
      l.sizeU0 = U.sizeU0, l.sizeV0 = U.sizeV0;
      l.occ.assign( size_t( U.sizeU0 ) * U.sizeV0, 0 );
      for ( int v = 0; v < it.rec.sizeV0; ++v )
        std::memcpy( &l.occ[size_t( v ) * U.sizeU0], &it.occ[size_t( v ) * it.rec.sizeU0], size_t( it.rec.sizeU0 ) );
    }
}

// every frame of the sub-context: tracked patches at their union's place, then the others around them.
// Returns false when the result is not acceptable (a frame taller than the minimum image; three frames that grew 10 %).
bool Allocator::completeFrames( int first, int end, const std::vector<UnionBox>& unions, int frameWidth, int unionsHeight,
                                bool useRef ) {
  int grown = 0;
  for ( int j = first; j < end; ++j ) {
    Tile& t = T[j];
    if ( t.items.empty() ) return true;  // (the reference leaves the whole step here)
    int sizeU = frameWidth / res_, sizeV = unionsHeight / res_;
    for ( auto& it : t.items ) sizeU = std::max( sizeU, it.trial.sizeU0 + 1 );
    t.trialW = sizeU * res_, t.trialH = sizeV * res_;
    BitCanvas canvas( sizeU, sizeV );
    for ( auto& it : t.items ) {
      Layout& l = it.trial;
      if ( !l.tracked() ) continue;
      const UnionBox& U = unions[size_t( l.track )];
      l.u0 = U.u0, l.v0 = U.v0, l.orient = U.orient;
      {  // the unions were packed on the GOF-wide tile, this frame's canvas can be narrower: the reference writes outside its map
        const int bw = l.orient == 0 ? l.sizeU0 : l.sizeV0, bh = l.orient == 0 ? l.sizeV0 : l.sizeU0;
        if ( l.u0 + bw > canvas.width() || l.v0 + bh > canvas.height() ) {
          outside_ = true;
          return false;
        }
      }
      canvas.stamp( l.occ.data(), l.sizeU0, l.sizeU0, l.sizeV0, l.u0, l.v0, l.orient );
      extent( t, l );
    }
    // anchors: the frame before, as committed (for the opening frame of a sub-context that follows another one) or as
    // laid out in this very pass
    const bool opening = j == first, anchors = j != 0 && ( !opening || useRef );
    for ( auto& it : t.items ) {
      Layout& l = it.trial;
      if ( l.tracked() ) continue;
      bool anchored = anchors && it.match != -1;
      int  au = 0, av = 0, ao = 0;
      if ( anchored ) {
        const Item& a = T[j - 1].items[size_t( it.match )];
        if ( opening )
          au = a.rec.u0, av = a.rec.v0, ao = a.rec.patchOrientation;
        else
          au = a.trial.u0, av = a.trial.v0, ao = a.trial.orient;
      }
      placeAlone( canvas, l, it.rec, anchored, au, av, ao );
      canvas.stamp( it.occ.data(), it.rec.sizeU0, it.rec.sizeU0, it.rec.sizeV0, l.u0, l.v0, l.orient );
      extent( t, l );
    }
    if ( t.trialH > minH_ ) return false;
    if ( double( t.trialH ) / double( t.height ) >= 1.10 ) ++grown;
  }
  return grown <= 2;
}

// the kept layout of frames [first, end) becomes the frames' packing; lists are reordered so that tracked patches lead,
// aligned across the frames, and the matches are rewritten to the new positions
int Allocator::commit( int first, int end ) {
  int globals = 0;
  for ( int j = first; j < end; ++j ) {
    Tile& t  = T[j];
    t.width  = t.keptW;
    t.height = t.keptH;
    globals  = 0;
    for ( auto& it : t.items ) {
      const Layout& l = it.kept;
      it.rec.sizeU0 = l.sizeU0, it.rec.sizeV0 = l.sizeV0;
      it.rec.u0 = l.u0, it.rec.v0 = l.v0, it.rec.patchOrientation = l.orient;
      it.occ    = l.occ;
      it.global = l.tracked();
      globals += it.global ? 1 : 0;
    }
  }
  if ( end - first == 1 ) {
    for ( auto& it : T[first].items ) it.match = -1;
    return TMC2_OK;
  }
  for ( int j = first; j < end; ++j ) {
    auto& items = T[j].items;
    for ( size_t i = 0; i < items.size(); ++i ) items[i].rec.index = int32_t( i );
    std::vector<Item> old;
    old.swap( items );
    if ( j == first ) {
      for ( auto& it : old )
        if ( it.global ) items.push_back( it );
    } else {
      // tracked patches in the order of the positions their matches had in the frame before
      const int before = int( T[j - 1].items.size() );
      for ( int pos = 0; pos < before; ++pos )
        for ( auto& it : old )
          if ( it.global && it.match == pos ) {
            items.push_back( it );
            break;
          }
    }
    for ( auto& it : old )
      if ( !it.global ) items.push_back( it );
  }
  for ( int j = first; j < end; ++j )
    if ( int( T[j].items.size() ) < globals ) {
      // a tracked patch whose per-frame match disagrees with its track has no place in the aligned lists; the reference
      // indexes past the end of its list here
      setError( "global patch allocation: a tracked patch of frame %d has no matched predecessor (undefined in the reference)", j );
      return TMC2_E_UNSUPPORTED;
    }
  for ( int j = first; j < end; ++j ) {
    auto& items = T[j].items;
    for ( int i = 0; i < globals; ++i ) {
      if ( j > first ) items[size_t( i )].match = i;
      items[size_t( i )].rec.index = i;
    }
    if ( j == end - 1 ) {
      for ( size_t i = size_t( globals ); i < items.size(); ++i ) items[i].rec.index = int32_t( i );
      continue;
    }
    auto&             next = T[j + 1].items;
    std::vector<char> moved( next.size(), 0 );
    for ( size_t i = size_t( globals ); i < items.size(); ++i ) {
      for ( size_t k = size_t( globals ); k < next.size(); ++k )
        if ( next[k].match == items[i].rec.index && !moved[k] ) {
          next[k].match = int32_t( i );
          moved[k]      = 1;
          break;
        }
      items[i].rec.index = int32_t( i );
    }
  }
  for ( auto& it : T[first].items ) it.match = -1;
  return TMC2_OK;
}

int Allocator::run() {
  const int             F = int( T.size() );
  std::vector<Track>    tracks;
  std::vector<UnionBox> unions;
  int                   first = 0, end = 0;  // the sub-context whose layout is kept
  bool                  opening = true;
  for ( int fi = 0; fi < F; ++fi ) {
    if ( opening ) {
      first = fi, end = fi + 1;
      tracks.assign( T[fi].items.size(), Track() );
      for ( size_t k = 0; k < tracks.size(); ++k ) {
        tracks[k].emplace_back( fi, int( k ) );
        T[fi].items[k].trial.track = int( k );
      }
      openSubContext( fi, first != 0 );
      if ( runaway_ ) {
        setError( "global patch allocation: a patch of frame %d cannot be placed at any canvas height", fi );
        return TMC2_E_INVALID;
      }
      keepTrial( first, end );
      dropTrial( first, end );
      T[fi].trialW = T[fi].trialH = 0;
      if ( fi == F - 1 ) return commit( first, end );
      opening = false;
      continue;
    }
    const int  trialEnd = fi + 1;
    const bool useRef   = first != 0;
    dropTrial( first, trialEnd );
    extendTracks( fi, tracks, fi - first - 1 );
    int       alive        = 0;
    const int unionsHeight = packUnions( tracks, T[fi].width, unions, first - 1, useRef, alive );
    bool      ok           = unionsHeight != 0 && !( double( alive ) / double( tracks.size() ) < 0.15 ) && unionsHeight <= minH_;
    if ( ok ) {
      adoptUnionBoxes( first, trialEnd, unions );
      ok = completeFrames( first, trialEnd, unions, minW_, unionsHeight, useRef );
    }
    if ( runaway_ ) {
      setError( "global patch allocation: a patch of frame %d cannot be placed at any canvas height", fi );
      return TMC2_E_INVALID;
    }
    if ( outside_ ) {
      setError( "global patch allocation: a union placed on the GOF-wide tile lies outside a frame's own canvas (sub-context ending "
                "at frame %d; undefined in the reference)", fi );
      return TMC2_E_UNSUPPORTED;
    }
    if ( !ok ) {
      // the kept layout of [first, end) stands; this frame opens the next sub-context
      dropTrial( first, trialEnd );
      TMC2_TRY( commit( first, end ) );
      opening = true;
      --fi;
      continue;
    }
    keepTrial( first, trialEnd );
    dropTrial( first, trialEnd );
    end = trialEnd;
    if ( fi == F - 1 ) return commit( first, end );
  }
  return TMC2_OK;
}
}  // namespace

// frames: per frame the patches IN LIST ORDER with their block-occupancy pool (rec.occOffset) and matches; tile sizes.
// On return the lists are reordered, box sizes / placements / indices / matches rewritten, pools rebuilt.
int globalPatchAllocationCore( std::vector<GpaFrameIO>& frames, int minW, int minH, int occRes ) {
  // placeSegments runs the allocation only if the FIRST frame of the tile has patches (PCCEncoder.cpp:4812): otherwise the
  // per-frame packing stands as it is (the pools still come back in list order, as on the allocating path)
  if ( frames.empty() || frames[0].list.empty() ) {
    for ( auto& io : frames ) {
      std::vector<uint8_t> pool;
      for ( auto& rec : io.list ) {
        const uint8_t* o = io.occ.data() + rec.occOffset;
        rec.occOffset    = int64_t( pool.size() );
        pool.insert( pool.end(), o, o + size_t( rec.sizeU0 ) * rec.sizeV0 );
      }
      io.occ.swap( pool );
    }
    return TMC2_OK;
  }
  std::vector<Tile> tiles( frames.size() );
  for ( size_t f = 0; f < frames.size(); ++f ) {
    GpaFrameIO& io = frames[f];
    Tile&       t  = tiles[f];
    t.width = io.width, t.height = io.height;
    t.items.resize( io.list.size() );
    for ( size_t i = 0; i < io.list.size(); ++i ) {
      Item& it = t.items[i];
      it.rec   = io.list[i];
      it.match = io.match[i];
      if ( it.match >= int32_t( f ? frames[f - 1].list.size() : 0 ) || it.match < -1 ) {
        setError( "global patch allocation: frame %zu, list position %zu: match %d out of range", f, i, it.match );
        return TMC2_E_INVALID;
      }
      const uint8_t* o = io.occ.data() + it.rec.occOffset;
      it.occ.assign( o, o + size_t( it.rec.sizeU0 ) * it.rec.sizeV0 );
    }
  }
  Allocator A( tiles, minW, minH, occRes );
  TMC2_TRY( A.run() );
  for ( size_t f = 0; f < frames.size(); ++f ) {
    GpaFrameIO& io = frames[f];
    Tile&       t  = tiles[f];
    io.width = t.width, io.height = t.height;
    io.list.resize( t.items.size() );
    io.match.resize( t.items.size() );
    io.occ.clear();
    for ( size_t i = 0; i < t.items.size(); ++i ) {
      io.list[i]           = t.items[i].rec;
      io.list[i].occOffset = int64_t( io.occ.size() );
      io.match[i]          = t.items[i].match;
      io.occ.insert( io.occ.end(), t.items[i].occ.begin(), t.items[i].occ.end() );
    }
  }
  return TMC2_OK;
}

// a packed patch list (list order, pool in list order) becomes the frame's state: records on the host, pool re-uploaded
int installPacking( tmc2_frame* f, const GpaFrameIO& g ) {
  ApiScope scope( f->ctx );
  f->patches   = g.list;  // list order from here on: the reference rewrites the patch indices to list positions
  f->packMatch = g.match;
  f->packOrder.resize( g.list.size() );
  for ( size_t k = 0; k < g.list.size(); ++k ) f->packOrder[k] = int32_t( k );
  f->occCount = int64_t( g.occ.size() );
  TMC2_TRY( f->growPools() );
  if ( f->occCount ) {
    TMC2_HIP( hipMemcpyAsync( f->d_occupancy.p, g.occ.data(), g.occ.size(), hipMemcpyHostToDevice, f->ctx->stream ) );
    TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  }
  f->packedHeight       = g.height;
  f->packedWidth        = g.width;
  f->haveGeometryImages = f->haveAttributeImages = f->haveReconstruction = false;
  return TMC2_OK;
}

// the frames of a GOF, each packed by the per-frame chain: pools come back from the devices (a few KB each), the
// allocation runs, the rewritten lists / pools go back
int globalPatchAllocationFrames( tmc2_frame** fr, int count, int minW, int minH, int occRes, int32_t* widths, int32_t* heights ) {
  std::vector<GpaFrameIO> io;
  io.resize( size_t( count ) );
  int                     tileW = minW, tileH = minH;  // resizeTileGeometryVideo (PCCEncoder.cpp:5593-5632) ran before
  for ( int i = 0; i < count; ++i ) {
    tmc2_frame* f = fr[i];
    if ( !f->havePatches || !f->havePacking ) {
      setError( "global patch allocation: frame %d is not packed", i );
      return TMC2_E_STATE;
    }
    ApiScope    scope( f->ctx );
    GpaFrameIO& g = io[size_t( i )];
    g.occ.resize( size_t( f->occCount ) );
    if ( f->occCount ) TMC2_HIP( hipMemcpyAsync( g.occ.data(), f->d_occupancy.p, g.occ.size(), hipMemcpyDeviceToHost, f->ctx->stream ) );
    g.list.resize( f->patches.size() );
    g.match = f->packMatch;
    g.match.resize( f->patches.size(), -1 );
    for ( size_t k = 0; k < g.list.size(); ++k ) g.list[k] = f->patches[size_t( f->packOrder[k] )];
    tileW = std::max( tileW, f->packedWidth );  // (a frame packed by packFlexible kept the preset width, whatever its patches need)
    tileH = std::max( tileH, f->packedHeight );
    TMC2_HIP( hipStreamSynchronize( f->ctx->stream ) );
  }
  for ( auto& g : io ) g.width = tileW, g.height = tileH;
  TMC2_TRY( globalPatchAllocationCore( io, minW, minH, occRes ) );
  for ( int i = 0; i < count; ++i ) {
    GpaFrameIO& g = io[size_t( i )];
    TMC2_TRY( installPacking( fr[i], g ) );
    if ( widths ) widths[i] = g.width;
    if ( heights ) heights[i] = g.height;
  }
  return TMC2_OK;
}

}  // namespace tmc2

extern "C" {

int tmc2_encoder_global_patch_allocation( tmc2_frame** frames, int count, int minimumImageWidth, int minimumImageHeight,
                                          int32_t* widths, int32_t* heights ) {
  if ( !frames || count <= 0 || minimumImageWidth <= 0 || minimumImageHeight <= 0 ) return TMC2_E_INVALID;
  for ( int i = 0; i < count; ++i )
    if ( !frames[i] ) return TMC2_E_INVALID;
  return tmc2::globalPatchAllocationFrames( frames, count, minimumImageWidth, minimumImageHeight, 16, widths, heights );
}

int tmc2_frame_set_packing( tmc2_frame* f, const tmc2_patch* list, int count, const int32_t* matches, const uint8_t* occupancy,
                            int64_t occupancyBytes, int packedWidth, int packedHeight ) {
  if ( !f || count < 0 || occupancyBytes < 0 || ( count && ( !list || !occupancy ) ) || packedWidth <= 0 || packedHeight < 0 )
    return TMC2_E_INVALID;
  if ( !f->havePatches ) {
    tmc2::setError( "set_packing: the frame has no patches" );
    return TMC2_E_STATE;
  }
  if ( size_t( count ) != f->patches.size() ) {
    tmc2::setError( "set_packing: %d records for a frame of %zu patches (packing reorders and places, it does not add or drop)", count,
                    f->patches.size() );
    return TMC2_E_INVALID;
  }
  tmc2::GpaFrameIO g;
  g.list.assign( list, list + count );
  if ( matches )
    g.match.assign( matches, matches + count );
  else
    g.match.assign( size_t( count ), -1 );
  g.occ.assign( occupancy, occupancy + occupancyBytes );
  g.width = packedWidth, g.height = packedHeight;
  for ( int i = 0; i < count; ++i ) {
    const tmc2_patch& p = list[i];
    const int64_t     blocks = int64_t( p.sizeU0 ) * p.sizeV0, samples = int64_t( p.sizeU ) * p.sizeV;
    if ( p.sizeU0 <= 0 || p.sizeV0 <= 0 || p.occOffset < 0 || p.occOffset + blocks > occupancyBytes || p.depthOffset < 0 ||
         p.depthOffset + samples > f->depthCount || p.u0 < 0 || p.v0 < 0 ) {
      tmc2::setError( "set_packing: record %d does not belong to this frame's pools", i );
      return TMC2_E_INVALID;
    }
  }
  TMC2_TRY( tmc2::installPacking( f, g ) );
  f->havePacking = true;
  return TMC2_OK;
}

int tmc2_host_global_patch_allocation( int frames, int32_t* counts, tmc2_patch* patches, const uint8_t* occupancy,
                                       const int64_t* occupancyBase, int32_t* matches, int tileWidth, int tileHeight,
                                       int minimumImageWidth, int minimumImageHeight, uint8_t* occupancyOut,
                                       int64_t occupancyOutCapacity, int64_t* occupancyOutBase, int32_t* widths,
                                       int32_t* heights ) {
  if ( frames <= 0 || !counts || !occupancyBase || !occupancyOutBase || minimumImageWidth <= 0 || minimumImageHeight <= 0 )
    return TMC2_E_INVALID;
  std::vector<tmc2::GpaFrameIO> io;
  io.resize( size_t( frames ) );
  size_t                        at = 0;
  for ( int f = 0; f < frames; ++f ) {
    if ( counts[f] < 0 || ( counts[f] && ( !patches || !occupancy || !matches ) ) ) return TMC2_E_INVALID;
    tmc2::GpaFrameIO& g = io[size_t( f )];
    g.list.assign( patches + at, patches + at + counts[f] );
    g.match.assign( matches + at, matches + at + counts[f] );
    int64_t bytes = 0;
    for ( auto& p : g.list ) bytes = std::max( bytes, p.occOffset + int64_t( p.sizeU0 ) * p.sizeV0 );
    g.occ.assign( occupancy + occupancyBase[f], occupancy + occupancyBase[f] + bytes );
    g.width = tileWidth, g.height = tileHeight;
    at += size_t( counts[f] );
  }
  TMC2_TRY( tmc2::globalPatchAllocationCore( io, minimumImageWidth, minimumImageHeight, 16 ) );
  int64_t total = 0;
  for ( auto& g : io ) total += int64_t( g.occ.size() );
  occupancyOutBase[frames] = total;
  if ( total > occupancyOutCapacity || ( total && !occupancyOut ) ) {
    tmc2::setError( "host_global_patch_allocation: the rebuilt occupancy pools need %lld bytes", (long long)total );
    return TMC2_E_INVALID;
  }
  at            = 0;
  int64_t base  = 0;
  for ( int f = 0; f < frames; ++f ) {
    tmc2::GpaFrameIO& g = io[size_t( f )];
    occupancyOutBase[f] = base;
    std::copy( g.list.begin(), g.list.end(), patches + at );
    std::copy( g.match.begin(), g.match.end(), matches + at );
    if ( !g.occ.empty() ) std::memcpy( occupancyOut + base, g.occ.data(), g.occ.size() );
    base += int64_t( g.occ.size() );
    at += g.list.size();
    counts[f] = int32_t( g.list.size() );
    if ( widths ) widths[f] = g.width;
    if ( heights ) heights[f] = g.height;
  }
  return TMC2_OK;
}
}
