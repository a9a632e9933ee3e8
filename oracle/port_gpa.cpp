// oracle/port_gpa.cpp -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
//
// S10' (random-access condition: constrainedPack = 1, globalPatchAllocation = 1): the data-adaptive global patch
// allocation that placeSegments runs after the per-frame spatial-consistency packing
//      PCCEncoder::performDataAdaptiveGPAMethod        (PccLibEncoder/source/PCCEncoder.cpp:6821-6971)
//      initializeSubContext / clearCurrentGPAPatchDataInfor / generateGlobalPatches   (:6973-7057)
//      unionPatchGenerationAndPacking                  (:7059-7226)
//      packingFirstFrame                               (:7228-7364)
//      updatePatchInformation / updateGPAPatchInformation  (:7366-7529)
//      performGPAPacking, packingWith(out)RefForFirstFrameNoglobalPatch   (:7531-7840)
//      GPAPatchData, PCCPatch::checkFitPatchCanvasForGPA / patchBlock2CanvasBlockForGPA (PCCPatch.h:42-71, PCCPatch.cpp:617-692)
// with packingStrategy = 1, two orientations, safeguard 0, lowDelayEncoding off, one tile per frame, no raw / EOM patches.
// Frames are grown into "sub-contexts": patches tracked across the frames of a sub-context (IoU matching) are packed as
// the UNION of their block occupancies at one common place; a sub-context ends when too few patches can be tracked or the
// packing gets too tall, and the last good state is committed.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "oracle.h"

namespace {
enum { ORIENT_DEFAULT = 0, ORIENT_SWAP = 1 };
constexpr size_t kNone = size_t( -1 );
struct Runaway {};
struct OutOfCanvas {};

struct GpaData {
  bool                 isMatched = false, isGlobal = false;
  int                  globalIdx = -1;
  size_t               sizeU0 = 0, sizeV0 = 0, u0 = kNone, v0 = kNone, orient = kNone;
  std::vector<uint8_t> occ;
  void                 init() { *this = GpaData(); }
  bool                 switched() const { return orient != ORIENT_DEFAULT; }
};
struct WPatch {
  orc_patch            p;
  std::vector<uint8_t> occ;  // sizeU0 x sizeV0 blocks
  int                  bestMatch = -1;
  bool                 isGlobal  = false;
  GpaData              cur, pre;
};
struct WFrame {
  std::vector<WPatch> patches;
  size_t              width = 0, height = 0, preW = 0, preH = 0, curW = 0, curH = 0;
  int                 globalCount = 0;
};
struct Gpa {
  std::vector<WFrame> frames;
  int                 minW, minH, occRes;
  bool                undefined = false;
};
using Tracks = std::map<size_t, std::vector<std::pair<size_t, size_t>>>;  // track -> [(frame, patch position)]
struct Union {
  size_t               sizeU0 = 0, sizeV0 = 0, u0 = 0, v0 = 0, orient = ORIENT_DEFAULT;
  std::vector<uint8_t> occ;
};
using Unions = std::map<size_t, Union>;

// canvas position of block (ub, vb) of a box placed at (u0, v0) with an orientation; -1 outside
int canvasBlock( size_t ub, size_t vb, size_t u0, size_t v0, size_t orient, size_t stride, size_t rows ) {
  size_t x, y;
  if ( orient == ORIENT_DEFAULT ) {
    x = ub + u0, y = vb + v0;
  } else if ( orient == ORIENT_SWAP ) {
    x = vb + u0, y = ub + v0;
  } else {
    return -1;
  }
  if ( x >= stride || y >= rows ) return -1;
  return int( x + stride * y );
}
bool fits( const std::vector<uint8_t>& map, size_t stride, size_t rows, size_t sizeU0, size_t sizeV0, size_t u0, size_t v0,
           size_t orient ) {
  for ( size_t vb = 0; vb < sizeV0; ++vb )
    for ( size_t ub = 0; ub < sizeU0; ++ub ) {
      const int pos = canvasBlock( ub, vb, u0, v0, orient, stride, rows );
      if ( pos < 0 || map[size_t( pos )] ) return false;
    }
  return true;
}
bool fitsGpa( const std::vector<uint8_t>& map, size_t stride, size_t rows, const GpaData& g ) {
  return fits( map, stride, rows, g.sizeU0, g.sizeV0, g.u0, g.v0, g.orient );
}

float iouOf( const orc_patch& a, const orc_patch& b ) {
  const int x1 = std::max( a.u1, b.u1 ), y1 = std::max( a.v1, b.v1 );
  int       w  = std::min( a.u1 + a.sizeU, b.u1 + b.sizeU ) - x1, h = std::min( a.v1 + a.sizeV, b.v1 + b.sizeV ) - y1;
  if ( w <= 0 || h <= 0 ) w = h = 0;
  const int inter = w * h, uni = a.sizeU * a.sizeV + b.sizeU * b.sizeV - inter;
  return static_cast<float>( inter ) / uni;
}

// the two best-effort orientations of a box (wide ones are tried swapped first)
size_t orientationOf( size_t sizeU0, size_t sizeV0, int k ) {
  return sizeU0 > sizeV0 ? ( k == 0 ? ORIENT_SWAP : ORIENT_DEFAULT ) : ( k == 0 ? ORIENT_DEFAULT : ORIENT_SWAP );
}

void grow( size_t& w, size_t& h, size_t& maxRow, const GpaData& g, size_t occRes ) {
  const size_t spanV = g.switched() ? g.sizeU0 : g.sizeV0, spanU = g.switched() ? g.sizeV0 : g.sizeU0;
  h      = std::max( h, ( g.v0 + spanV ) * occRes );
  w      = std::max( w, ( g.u0 + spanU ) * occRes );
  maxRow = std::max( maxRow, g.v0 + spanV );
}

// marks the blocks of g on the map with the block occupancy occ (row length occStride)
void mark( std::vector<uint8_t>& map, size_t stride, size_t rows, const GpaData& g, const std::vector<uint8_t>& occ,
           size_t occStride ) {
  for ( size_t vb = 0; vb < g.sizeV0; ++vb )
    for ( size_t ub = 0; ub < g.sizeU0; ++ub ) {
      const int pos = canvasBlock( ub, vb, g.u0, g.v0, g.orient, stride, rows );
      if ( pos < 0 ) throw OutOfCanvas();  // the reference writes occupancyMap[-1] here
      map[size_t( pos )] = map[size_t( pos )] || occ[vb * occStride + ub];
    }
}

void packingFirstFrame( Gpa& G, size_t fi, size_t frameWidth, bool hasRef ) {
  WFrame& F     = G.frames[fi];
  size_t  sizeU = frameWidth / G.occRes, sizeV = 0;
  for ( auto& q : F.patches ) sizeV = std::max( sizeV, size_t( std::max( q.p.sizeU0, q.p.sizeV0 ) ) );
  for ( auto& q : F.patches ) sizeU = std::max( sizeU, size_t( q.p.sizeU0 + 1 ) );
  F.curW = sizeU * G.occRes;
  F.curH = sizeV * G.occRes;
  size_t               maxRow = 0;
  std::vector<uint8_t> map( sizeU * sizeV, 0 );
  for ( auto& q : F.patches ) {
    GpaData& g = q.cur;
    g.occ      = q.occ;
    g.sizeU0   = size_t( q.p.sizeU0 );
    g.sizeV0   = size_t( q.p.sizeV0 );
    bool found = false;
    while ( !found ) {
      if ( q.bestMatch != -1 && hasRef ) {
        const WPatch& r = G.frames[fi - 1].patches[size_t( q.bestMatch )];
        g.orient        = size_t( r.p.patchOrientation );
        g.u0            = size_t( r.p.u0 );
        g.v0            = size_t( r.p.v0 );
        found           = fitsGpa( map, sizeU, sizeV, g );
        for ( size_t v = 0; v <= sizeV && !found; ++v )
          for ( size_t u = 0; u <= sizeU && !found; ++u ) {
            g.u0  = u;
            g.v0  = v;
            found = fitsGpa( map, sizeU, sizeV, g );
          }
      } else {
        for ( size_t v = 0; v < sizeV && !found; ++v )
          for ( size_t u = 0; u < sizeU && !found; ++u ) {
            g.u0 = u;
            g.v0 = v;
            for ( int k = 0; k < 2 && !found; ++k ) {
              g.orient = orientationOf( g.sizeU0, g.sizeV0, k );
              found    = fitsGpa( map, sizeU, sizeV, g );
            }
          }
      }
      if ( !found ) {
        if ( sizeV > ( size_t( 1 ) << 20 ) ) throw Runaway();  // nothing can be placed: the reference spins here
        sizeV *= 2;
        map.resize( sizeU * sizeV, 0 );
      }
    }
    mark( map, sizeU, sizeV, g, q.occ, size_t( q.p.sizeU0 ) );
    grow( F.curW, F.curH, maxRow, g, size_t( G.occRes ) );
  }
}

void generateGlobalPatches( Gpa& G, size_t fi, Tracks& tracks, size_t preIndex ) {
  auto& cur = G.frames[fi].patches;
  for ( auto& t : tracks ) {
    auto& tp = t.second;
    if ( tp.empty() ) continue;
    const auto&   pg  = tp[preIndex];
    const WPatch& pre = G.frames[pg.first].patches[pg.second];
    float         maxIou = 0.0F;
    int           best   = -1;
    for ( size_t c = 0; c < cur.size(); ++c ) {
      if ( pre.p.viewId != cur[c].p.viewId || cur[c].cur.isMatched ) continue;
      const float iou = iouOf( pre.p, cur[c].p );
      if ( iou > maxIou ) {
        maxIou = iou;
        best   = int( c );
      }
    }
    if ( maxIou > 0.2F ) {
      cur[size_t( best )].cur.isMatched = true;
      tp.emplace_back( fi, size_t( best ) );
    } else {
      tp.clear();
    }
  }
  for ( auto& t : tracks )
    for ( auto& e : t.second ) {
      GpaData& g  = G.frames[e.first].patches[e.second].cur;
      g.isGlobal  = true;
      g.globalIdx = int( t.first );
    }
}

size_t unionPatchGenerationAndPacking( Gpa& G, const Tracks& tracks, size_t frameWidth, Unions& unions, size_t refFrame,
                                       bool useRef ) {
  unions.clear();
  for ( auto& t : tracks ) {
    if ( t.second.empty() ) continue;
    Union  U;
    for ( auto& e : t.second ) {
      const WPatch& q = G.frames[e.first].patches[e.second];
      U.sizeU0        = std::max( U.sizeU0, size_t( q.p.sizeU0 ) );
      U.sizeV0        = std::max( U.sizeV0, size_t( q.p.sizeV0 ) );
    }
    U.occ.assign( U.sizeU0 * U.sizeV0, 0 );
    if ( useRef ) {
      const WPatch& first = G.frames[t.second[0].first].patches[t.second[0].second];
      U.orient = first.bestMatch == -1 ? kNone : size_t( G.frames[refFrame].patches[size_t( first.bestMatch )].p.patchOrientation );
    }
    for ( auto& e : t.second ) {
      const WPatch& q = G.frames[e.first].patches[e.second];
      for ( int v = 0; v < q.p.sizeV0; ++v )
        for ( int u = 0; u < q.p.sizeU0; ++u )
          if ( q.occ[size_t( v ) * q.p.sizeU0 + u] ) U.occ[size_t( v ) * U.sizeU0 + u] = 1;
    }
    unions[t.first] = U;
  }
  size_t sizeU = frameWidth / G.occRes, sizeV = 0;
  for ( auto& u : unions ) {
    sizeU = std::max( sizeU, u.second.sizeU0 + 1 );
    sizeV = std::max( sizeV, u.second.sizeV0 + 1 );
  }
  size_t               width = sizeU * G.occRes, height = sizeV * G.occRes, maxRow = 0;
  std::vector<uint8_t> map( sizeU * sizeV, 0 );
  for ( auto& it : unions ) {
    Union& U     = it.second;
    bool   found = false;
    while ( !found ) {
      for ( size_t v = 0; v < sizeV && !found; ++v )
        for ( size_t u = 0; u < sizeU && !found; ++u ) {
          U.u0 = u;
          U.v0 = v;
          if ( useRef && U.orient != kNone ) {
            found = fits( map, sizeU, sizeV, U.sizeU0, U.sizeV0, U.u0, U.v0, U.orient );
          } else {
            for ( int k = 0; k < 2 && !found; ++k ) {
              U.orient = orientationOf( U.sizeU0, U.sizeV0, k );
              found    = fits( map, sizeU, sizeV, U.sizeU0, U.sizeV0, U.u0, U.v0, U.orient );
            }
          }
        }
      if ( !found ) {
        if ( sizeV > ( size_t( 1 ) << 20 ) ) throw Runaway();  // nothing can be placed: the reference spins here
        sizeV *= 2;
        map.resize( sizeU * sizeV, 0 );
      }
    }
    GpaData g;
    g.sizeU0 = U.sizeU0, g.sizeV0 = U.sizeV0, g.u0 = U.u0, g.v0 = U.v0, g.orient = U.orient;
    mark( map, sizeU, sizeV, g, U.occ, U.sizeU0 );
    grow( width, height, maxRow, g, size_t( G.occRes ) );
  }
  return height;
}

void updateGPAPatchInformation( Gpa& G, size_t first, size_t second, Unions& unions ) {
  for ( size_t i = first; i < second; ++i )
    for ( auto& q : G.frames[i].patches ) {
      GpaData& g = q.cur;
      if ( g.isGlobal ) {
        const Union&         U = unions[size_t( g.globalIdx )];
        std::vector<uint8_t> occ( U.sizeU0 * U.sizeV0, 0 );
        for ( int v = 0; v < q.p.sizeV0; ++v )
          for ( int u = 0; u < q.p.sizeU0; ++u )
            if ( q.occ[size_t( v ) * q.p.sizeU0 + u] ) occ[size_t( v ) * U.sizeU0 + u] = 1;
        g.sizeU0 = U.sizeU0;
        g.sizeV0 = U.sizeV0;
        g.occ    = occ;
      } else {
        g.sizeU0 = size_t( q.p.sizeU0 );
        g.sizeV0 = size_t( q.p.sizeV0 );
        g.occ    = q.occ;
      }
    }
}

void packNonGlobal( Gpa& G, WPatch& q, const std::vector<WPatch>* prePatches, bool preIsStart, size_t& sizeU, size_t& sizeV,
                    std::vector<uint8_t>& map, size_t& heightGPA, size_t& widthGPA, size_t& maxRow ) {
  GpaData& g     = q.cur;
  bool     found = false;
  while ( !found ) {
    if ( prePatches && q.bestMatch != -1 ) {
      const WPatch& r = ( *prePatches )[size_t( q.bestMatch )];
      if ( preIsStart ) {
        g.orient = size_t( r.p.patchOrientation );
        g.u0     = size_t( r.p.u0 );
        g.v0     = size_t( r.p.v0 );
      } else {
        g.orient = r.cur.orient;
        g.u0     = r.cur.u0;
        g.v0     = r.cur.v0;
      }
      found = fitsGpa( map, sizeU, sizeV, g );
      for ( size_t v = 0; v <= sizeV && !found; ++v )
        for ( size_t u = 0; u <= sizeU && !found; ++u ) {
          g.u0  = u;
          g.v0  = v;
          found = fitsGpa( map, sizeU, sizeV, g );
        }
    } else {
      for ( size_t v = 0; v < sizeV && !found; ++v )
        for ( size_t u = 0; u < sizeU && !found; ++u ) {
          g.u0 = u;
          g.v0 = v;
          for ( int k = 0; k < 2 && !found; ++k ) {
            g.orient = orientationOf( size_t( q.p.sizeU0 ), size_t( q.p.sizeV0 ), k );
            found    = fitsGpa( map, sizeU, sizeV, g );
          }
        }
    }
    if ( !found ) {
      if ( sizeV > ( size_t( 1 ) << 20 ) ) throw Runaway();
      sizeV *= 2;
      map.resize( sizeU * sizeV, 0 );
    }
  }
  mark( map, sizeU, sizeV, g, q.occ, size_t( q.p.sizeU0 ) );
  grow( widthGPA, heightGPA, maxRow, g, size_t( G.occRes ) );
}

void performGPAPacking( Gpa& G, size_t first, size_t second, Unions& unions, size_t frameWidth, bool& bad, size_t unionsHeight,
                        bool useRef ) {
  bool   exceed       = false;
  size_t badCondition = 0;
  for ( size_t i = first; i < second; ++i ) {
    WFrame& F = G.frames[i];
    if ( F.patches.empty() ) return;
    const size_t preIndex = i > 0 ? i - 1 : 0;
    size_t       sizeU = frameWidth / G.occRes, sizeV = unionsHeight / G.occRes;
    for ( auto& q : F.patches ) sizeU = std::max( sizeU, q.cur.sizeU0 + 1 );
    F.curW = sizeU * G.occRes;
    F.curH = sizeV * G.occRes;
    size_t               maxRow = 0;
    std::vector<uint8_t> map( sizeU * sizeV, 0 );
    for ( auto& q : F.patches ) {
      GpaData& g = q.cur;
      if ( !g.isGlobal ) continue;
      const Union& U = unions[size_t( g.globalIdx )];
      g.u0 = U.u0, g.v0 = U.v0, g.orient = U.orient;
      mark( map, sizeU, sizeV, g, g.occ, g.sizeU0 );
      grow( F.curW, F.curH, maxRow, g, size_t( G.occRes ) );
    }
    for ( auto& q : F.patches ) {
      if ( q.cur.isGlobal ) continue;
      if ( i == 0 || ( i == first && !useRef ) )
        packNonGlobal( G, q, nullptr, false, sizeU, sizeV, map, F.curH, F.curW, maxRow );
      else
        packNonGlobal( G, q, &G.frames[preIndex].patches, i == first, sizeU, sizeV, map, F.curH, F.curW, maxRow );
    }
    if ( F.curH > size_t( G.minH ) ) {
      exceed = true;
      break;
    }
    if ( double( F.curH ) / double( F.height ) >= 1.10 ) ++badCondition;
  }
  if ( exceed || badCondition > 2 ) bad = true;
}

void updatePatchInformation( Gpa& G, size_t first, size_t second ) {
  for ( size_t fi = first; fi < second; ++fi ) {
    WFrame& F     = G.frames[fi];
    F.globalCount = 0;
    F.width       = F.preW;
    F.height      = F.preH;
    for ( auto& q : F.patches ) {
      const GpaData& g     = q.pre;
      q.p.sizeU0           = int32_t( g.sizeU0 );
      q.p.sizeV0           = int32_t( g.sizeV0 );
      q.occ                = g.occ;
      q.p.u0               = int32_t( g.u0 );
      q.p.v0               = int32_t( g.v0 );
      q.p.patchOrientation = int32_t( g.orient );
      q.isGlobal           = g.isGlobal;
      if ( q.isGlobal ) ++F.globalCount;
    }
  }
  if ( second - first == 1 ) {
    for ( auto& q : G.frames[first].patches ) q.bestMatch = -1;
    return;
  }
  int globalCount = 0;
  for ( size_t fi = first; fi < second; ++fi ) {
    auto& cur = G.frames[fi].patches;
    for ( size_t i = 0; i < cur.size(); ++i ) cur[i].p.index = int32_t( i );
    std::vector<WPatch> re = cur;
    globalCount            = G.frames[fi].globalCount;
    cur.clear();
    if ( fi == first ) {
      for ( auto& q : re )
        if ( q.isGlobal ) cur.push_back( q );
      for ( auto& q : re )
        if ( !q.isGlobal ) cur.push_back( q );
    } else {
      const size_t prevCount = G.frames[fi - 1].patches.size();
      for ( int32_t index = 0; index < int32_t( prevCount ); ++index )
        for ( auto& q : re )
          if ( index == q.bestMatch && q.isGlobal ) {
            cur.push_back( q );
            break;
          }
      for ( auto& q : re )
        if ( !q.isGlobal ) cur.push_back( q );
    }
  }
  for ( size_t fi = first; fi < second; ++fi )
    if ( int32_t( G.frames[fi].patches.size() ) < globalCount ) {
      G.undefined = true;  // the reference indexes past the end of the list here (a tracked patch lost its place)
      return;
    }
  for ( size_t fi = first; fi < second; ++fi ) {
    auto& cur = G.frames[fi].patches;
    for ( int32_t i = 0; i < globalCount; ++i ) {
      if ( fi > first ) cur[size_t( i )].bestMatch = i;
      cur[size_t( i )].p.index = i;
    }
    if ( fi == second - 1 ) {
      for ( size_t i = size_t( globalCount ); i < cur.size(); ++i ) cur[i].p.index = int32_t( i );
      continue;
    }
    auto&             next = G.frames[fi + 1].patches;
    std::vector<bool> updated( next.size(), false );
    for ( size_t i = size_t( globalCount ); i < cur.size(); ++i ) {
      for ( size_t j = size_t( globalCount ); j < next.size(); ++j )
        if ( cur[i].p.index == next[j].bestMatch && !updated[j] ) {
          next[j].bestMatch = int( i );
          updated[j]        = true;
          break;
        }
      cur[i].p.index = int32_t( i );
    }
  }
  for ( auto& q : G.frames[first].patches ) q.bestMatch = -1;
}

void run( Gpa& G ) {
  const size_t F = G.frames.size();
  size_t       preFirst = 0, preSecond = 0, curFirst = 0, curSecond = 0;
  Unions       unionCur;
  Tracks       tracks;
  bool         startSub = true;
  for ( size_t fi = 0; fi < F; ++fi ) {
    WFrame& tile   = G.frames[fi];
    bool    useRef = true;
    if ( startSub ) {
      preFirst = fi, preSecond = fi + 1;
      tracks.clear();
      for ( size_t k = 0; k < tile.patches.size(); ++k ) {
        tracks[k].emplace_back( fi, k );
        tile.patches[k].cur.isGlobal  = true;
        tile.patches[k].cur.globalIdx = int( k );
      }
      if ( preFirst == 0 ) useRef = false;
      packingFirstFrame( G, fi, tile.width, useRef );
      WFrame& S = G.frames[preFirst];
      S.preW = S.curW, S.preH = S.curH, S.curW = S.curH = 0;
      for ( auto& q : S.patches ) {
        q.pre = q.cur;
        q.cur.init();
      }
      if ( fi == F - 1 ) {
        updatePatchInformation( G, preFirst, preSecond );
        break;
      }
      curFirst = preFirst, curSecond = preSecond;
      startSub = false;
      continue;
    }
    curFirst  = preFirst;
    curSecond = fi + 1;
    size_t preSubFrame = curFirst - 1;
    if ( curFirst == 0 ) {
      useRef      = false;
      preSubFrame = kNone;
    }
    for ( size_t j = curFirst; j < curSecond; ++j )
      for ( auto& q : G.frames[j].patches ) q.cur.init();
    generateGlobalPatches( G, fi, tracks, fi - curFirst - 1 );
    const size_t unionsHeight = unionPatchGenerationAndPacking( G, tracks, tile.width, unionCur, preSubFrame, useRef );
    bool         badCount = double( unionCur.size() ) / double( tracks.size() ) < 0.15, badHeight = unionsHeight > size_t( G.minH );
    bool         badPacking = false;
    if ( unionsHeight == 0 ) badCount = true;
    if ( !badCount && !badHeight ) {
      updateGPAPatchInformation( G, curFirst, curSecond, unionCur );
      performGPAPacking( G, curFirst, curSecond, unionCur, size_t( G.minW ), badPacking, unionsHeight, useRef );
    }
    if ( getenv( "ORC_GPA_TRACE" ) )
      fprintf( stderr, "gpa: frame %zu sub [%zu,%zu) unions %zu tracks %zu height %zu -> badCount %d badHeight %d badPacking %d\n", fi,
               curFirst, curSecond, unionCur.size(), tracks.size(), unionsHeight, int( badCount ), int( badHeight ), int( badPacking ) );
    if ( badCount || badHeight || badPacking ) {
      for ( size_t j = curFirst; j < curSecond; ++j )
        for ( auto& q : G.frames[j].patches ) q.cur.init();
      unionCur.clear();
      tracks.clear();
      startSub = true;
      --fi;  // this frame opens the next sub-context
      updatePatchInformation( G, preFirst, preSecond );
      if ( G.undefined ) return;
    } else {
      for ( size_t j = curFirst; j < curSecond; ++j ) {
        WFrame& T = G.frames[j];
        T.preW = T.curW, T.preH = T.curH;
        for ( auto& q : T.patches ) q.pre = q.cur;
      }
      preFirst = curFirst, preSecond = curSecond;
      for ( size_t j = curFirst; j < curSecond; ++j )
        for ( auto& q : G.frames[j].patches ) q.cur.init();
      unionCur.clear();
      if ( fi == F - 1 ) {
        updatePatchInformation( G, preFirst, preSecond );
        break;
      }
    }
  }
}
}  // namespace

extern "C" {
void* orc_gpa_begin( int frames, int minW, int minH, int occRes ) {
  Gpa* G = new Gpa();
  G->frames.resize( size_t( frames ) );
  G->minW = minW, G->minH = minH, G->occRes = occRes;
  return G;
}
// list: the frame's patches in list order (after the per-frame packing); occPool indexed by occOffset
void orc_gpa_set_frame( void* h, int f, const orc_patch* list, int P, const uint8_t* occPool, const int32_t* bestMatch,
                        int width, int height ) {
  WFrame& F = static_cast<Gpa*>( h )->frames[size_t( f )];
  F.patches.resize( size_t( P ) );
  for ( int i = 0; i < P; ++i ) {
    WPatch& q   = F.patches[size_t( i )];
    q.p         = list[i];
    q.bestMatch = bestMatch[i];
    q.occ.assign( occPool + list[i].occOffset, occPool + list[i].occOffset + size_t( list[i].sizeU0 ) * list[i].sizeV0 );
  }
  F.width  = size_t( width );
  F.height = size_t( height );
}
// returns 1 where the reference's behaviour is undefined (see updatePatchInformation), 0 otherwise
int orc_gpa_run( void* h ) {
  Gpa& G = *static_cast<Gpa*>( h );
  // placeSegments runs the allocation only if the FIRST frame of the tile has patches (PCCEncoder.cpp:4812)
  if ( G.frames.empty() || G.frames[0].patches.empty() ) return 0;
  try {
    run( G );
  } catch ( const Runaway& ) { return 2; }  // the reference never returns (a patch that fits at no canvas height)
  catch ( const OutOfCanvas& ) { return 1; }  // a union placed on the wider GOF canvas does not fit this frame's own: undefined there
  return G.undefined ? 1 : 0;
}
int64_t orc_gpa_occ_bytes( void* h, int f ) {
  int64_t n = 0;
  for ( auto& q : static_cast<Gpa*>( h )->frames[size_t( f )].patches ) n += int64_t( q.occ.size() );
  return n;
}
// patches in the new list order (occOffset re-based into occPool), matches per list position, tile width / height
void orc_gpa_get_frame( void* h, int f, orc_patch* list, uint8_t* occPool, int32_t* bestMatch, int32_t* wh ) {
  WFrame& F = static_cast<Gpa*>( h )->frames[size_t( f )];
  int64_t o = 0;
  for ( size_t i = 0; i < F.patches.size(); ++i ) {
    list[i]           = F.patches[i].p;
    list[i].occOffset = o;
    std::memcpy( occPool + o, F.patches[i].occ.data(), F.patches[i].occ.size() );
    o += int64_t( F.patches[i].occ.size() );
    bestMatch[i] = F.patches[i].bestMatch;
  }
  wh[0] = int32_t( F.width );
  wh[1] = int32_t( F.height );
}
void orc_gpa_free( void* h ) { delete static_cast<Gpa*>( h ); }
}
