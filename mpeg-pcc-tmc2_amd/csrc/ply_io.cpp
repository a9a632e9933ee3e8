// ply_io.cpp -- point-cloud ingest and conformance checksums on the host.
//
// Replaces PCCPointSet3::read (reference: source/lib/PccLibCommon/source/PCCPointSet.cpp:464-757), which
// PCCGroupOfFrames::load (PccLibCommon/source/PCCGroupOfFrames.cpp:46-80) calls per frame, and PCCPointSet3::computeChecksum /
// computeMd5 / reorder (PCCPointSet.cpp:222-305), the MD5 the conformance logs and PccLibMetrics/PCCChecksum carry.
//
// The reader lands the cloud directly in the caller's buffers (int16 xyz[n][3], uint8 rgb[n][3] -- page-locked staging
// when the caller uploads next): one pass over the memory-mapped file, ASCII bodies split at line boundaries across
// threads, decimal tokens through an exact fast path (<= 15 significant digits: one correctly rounded division, what
// strtod returns) with strtod itself behind it.  Every quirk of the reference's reader that decides a value is kept:
// properties are told apart by NAME and BYTE COUNT only (a 4-byte x is read as float whatever its declared type, a 2-byte
// one as uint16), vertex properties stop counting at the first other element, short bodies leave zeros.
// What it refuses instead of misreading: big-endian bodies, unknown property types, lines beyond the reference's
// 4095-character buffer.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "internal.h"
#include "lex_order.h"

namespace tmc2 {
namespace {

// ---- MD5 (RFC 1321) ---------------------------------------------------------------------------------------------
class Md5 {
 public:
  Md5() : bytes_( 0 ), fill_( 0 ) {
    h_[0] = 0x67452301u, h_[1] = 0xefcdab89u, h_[2] = 0x98badcfeu, h_[3] = 0x10325476u;
  }
  void update( const uint8_t* p, size_t n ) {
    bytes_ += n;
    if ( fill_ ) {
      const size_t take = std::min( n, size_t( 64 ) - fill_ );
      std::memcpy( block_ + fill_, p, take );
      fill_ += take, p += take, n -= take;
      if ( fill_ < 64 ) return;
      transform( block_ );
      fill_ = 0;
    }
    for ( ; n >= 64; p += 64, n -= 64 ) transform( p );
    if ( n ) {
      std::memcpy( block_, p, n );
      fill_ = n;
    }
  }
  void finish( uint8_t out[16] ) {
    const uint64_t bits = bytes_ * 8;
    uint8_t        pad[72] = {0x80};
    const size_t   padLen = ( fill_ < 56 ? 56 : 120 ) - fill_;
    uint8_t        len[8];
    for ( int i = 0; i < 8; ++i ) len[i] = uint8_t( bits >> ( 8 * i ) );
    update( pad, padLen );
    update( len, 8 );
    for ( int i = 0; i < 4; ++i )
      for ( int k = 0; k < 4; ++k ) out[4 * i + k] = uint8_t( h_[i] >> ( 8 * k ) );
  }

 private:
  static uint32_t rol( uint32_t x, int s ) { return ( x << s ) | ( x >> ( 32 - s ) ); }
  void            transform( const uint8_t* blk ) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
        0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
        0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
        0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
        0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t w[16];
    for ( int i = 0; i < 16; ++i )
      w[i] = uint32_t( blk[4 * i] ) | ( uint32_t( blk[4 * i + 1] ) << 8 ) | ( uint32_t( blk[4 * i + 2] ) << 16 ) |
             ( uint32_t( blk[4 * i + 3] ) << 24 );
    uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3];
    for ( int i = 0; i < 64; ++i ) {
      uint32_t f;
      int      g;
      if ( i < 16 ) {
        f = ( b & c ) | ( ~b & d ), g = i;
      } else if ( i < 32 ) {
        f = ( d & b ) | ( ~d & c ), g = ( 5 * i + 1 ) & 15;
      } else if ( i < 48 ) {
        f = b ^ c ^ d, g = ( 3 * i + 5 ) & 15;
      } else {
        f = c ^ ( b | ~d ), g = ( 7 * i ) & 15;
      }
      const uint32_t t = d;
      d                = c;
      c                = b;
      b                = b + rol( a + f + K[i] + w[g], S[i] );
      a                = t;
    }
    h_[0] += a, h_[1] += b, h_[2] += c, h_[3] += d;
  }
  uint32_t h_[4];
  uint64_t bytes_;
  uint8_t  block_[64];
  size_t   fill_;
};

// ---- PLY header ---------------------------------------------------------------------------------------------------
struct Property {
  std::string name;
  int         bytes;
};
struct Header {
  bool                  ascii = false;
  uint64_t              points = 0;
  std::vector<Property> props;
  size_t                body = 0;  // offset of the first body byte
  int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, inx = -1, iny = -1, inz = -1;
  bool colors = false, normals = false;
};
inline bool isSep( char c ) { return c == ' ' || c == '\t' || c == '\r'; }

// tokens of one line (no '\n' inside [p, e))
void splitLine( const char* p, const char* e, std::vector<std::pair<const char*, const char*>>& out ) {
  out.clear();
  while ( p < e ) {
    while ( p < e && isSep( *p ) ) ++p;
    if ( p >= e ) break;
    const char* s = p;
    while ( p < e && !isSep( *p ) ) ++p;
    out.emplace_back( s, p );
  }
}
int typeBytes( const std::string& t ) {
  if ( t == "double" || t == "float64" || t == "uint64" || t == "int64" ) return 8;
  if ( t == "float" || t == "float32" || t == "uint32" || t == "int32" || t == "int" ) return 4;
  if ( t == "uint16" || t == "int16" ) return 2;
  if ( t == "uchar" || t == "uint8" || t == "char" || t == "int8" ) return 1;
  return 0;
}

int parseHeader( const char* data, size_t size, bool readNormals, Header& h ) {
  std::vector<std::pair<const char*, const char*>> tok;
  size_t                                           at = 0;
  bool                                             first = true, vertexProps = true, done = false, haveFormat = false;
  while ( !done ) {
    if ( at >= size ) {
      setError( "ply: corrupted header" );
      return TMC2_E_INVALID;
    }
    const char* nl   = static_cast<const char*>( memchr( data + at, '\n', size - at ) );
    const char* end  = nl ? nl : data + size;
    if ( end - ( data + at ) > 4095 ) {
      setError( "ply: header line longer than 4095 characters" );
      return TMC2_E_UNSUPPORTED;
    }
    splitLine( data + at, end, tok );
    at = size_t( end - data ) + ( nl ? 1 : 0 );
    auto is = [&]( size_t i, const char* s ) { return tok.size() > i && std::string( tok[i].first, tok[i].second ) == s; };
    if ( first ) {
      first = false;
      if ( !is( 0, "ply" ) ) {
        setError( "ply: not a PLY file" );
        return TMC2_E_INVALID;
      }
      continue;
    }
    if ( tok.empty() || is( 0, "comment" ) ) continue;
    if ( is( 0, "format" ) ) {
      if ( tok.size() != 3 ) {
        setError( "ply: corrupted format info" );
        return TMC2_E_INVALID;
      }
      const std::string fmt( tok[1].first, tok[1].second ), ver( tok[2].first, tok[2].second );
      h.ascii = fmt == "ascii";
      if ( !h.ascii && fmt != "binary_little_endian" ) {
        setError( "ply: format %s unsupported", fmt.c_str() );
        return TMC2_E_UNSUPPORTED;
      }
      if ( atof( ver.c_str() ) != 1.0 ) {
        setError( "ply: non-supported version" );
        return TMC2_E_UNSUPPORTED;
      }
      haveFormat = true;
    } else if ( is( 0, "element" ) ) {
      if ( tok.size() != 3 ) {
        setError( "ply: corrupted element info" );
        return TMC2_E_INVALID;
      }
      if ( is( 1, "vertex" ) )
        h.points = uint64_t( std::max( 0, atoi( std::string( tok[2].first, tok[2].second ).c_str() ) ) );
      else
        vertexProps = false;
    } else if ( is( 0, "property" ) && vertexProps ) {
      if ( tok.size() != 3 ) {
        setError( "ply: corrupted property info" );
        return TMC2_E_INVALID;
      }
      Property p;
      p.name  = std::string( tok[2].first, tok[2].second );
      p.bytes = typeBytes( std::string( tok[1].first, tok[1].second ) );
      if ( p.bytes == 0 ) {
        setError( "ply: property type %s unsupported", std::string( tok[1].first, tok[1].second ).c_str() );
        return TMC2_E_UNSUPPORTED;
      }
      h.props.push_back( p );
    } else if ( is( 0, "end_header" ) ) {
      done = true;
    }
  }
  (void)haveFormat;  // (a file without a format line is read as binary by the reference, too)
  h.body = at;
  for ( size_t a = 0; a < h.props.size(); ++a ) {
    const Property& p     = h.props[a];
    const bool      coord = p.bytes == 8 || p.bytes == 4 || p.bytes == 2;
    if ( p.name == "x" && coord )
      h.ix = int( a );
    else if ( p.name == "y" && coord )
      h.iy = int( a );
    else if ( p.name == "z" && coord )
      h.iz = int( a );
    else if ( p.name == "red" && p.bytes == 1 )
      h.ir = int( a );
    else if ( p.name == "green" && p.bytes == 1 )
      h.ig = int( a );
    else if ( p.name == "blue" && p.bytes == 1 )
      h.ib = int( a );
    else if ( p.name == "nx" && p.bytes == 4 && readNormals )
      h.inx = int( a );
    else if ( p.name == "ny" && p.bytes == 4 && readNormals )
      h.iny = int( a );
    else if ( p.name == "nz" && p.bytes == 4 && readNormals )
      h.inz = int( a );
    else if ( ( p.name == "reflectance" || p.name == "refc" ) && p.bytes <= 2 ) {
      setError( "ply: reflectance attributes are not part of this path" );
      return TMC2_E_UNSUPPORTED;
    }
  }
  if ( h.ix < 0 || h.iy < 0 || h.iz < 0 ) {
    setError( "ply: missing coordinates" );
    return TMC2_E_INVALID;
  }
  h.colors  = h.ir >= 0 && h.ig >= 0 && h.ib >= 0;
  h.normals = h.inx >= 0 && h.iny >= 0 && h.inz >= 0;
  return TMC2_OK;
}

// ---- numbers ------------------------------------------------------------------------------------------------------
// atof( token ): plain decimals with at most 15 significant digits are mantissa / 10^k with both exact in double -- one
// correctly rounded division, which is the value strtod returns; everything else goes to strtod
double tokenToDouble( const char* s, const char* e ) {
  static const double pow10[] = {1e0, 1e1, 1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* p   = s;
  bool        neg = false;
  if ( p < e && ( *p == '-' || *p == '+' ) ) neg = *p++ == '-';
  uint64_t m = 0;
  int      digits = 0, frac = 0;
  bool     any = false, simple = true;
  while ( p < e && *p >= '0' && *p <= '9' ) {
    if ( m || *p != '0' ) ++digits;
    m   = m * 10 + uint64_t( *p++ - '0' );
    any = true;
    if ( digits > 15 ) {
      simple = false;
      break;
    }
  }
  if ( simple && p < e && *p == '.' ) {
    ++p;
    while ( p < e && *p >= '0' && *p <= '9' ) {
      if ( m || *p != '0' ) ++digits;
      m = m * 10 + uint64_t( *p++ - '0' );
      ++frac;
      any = true;
      if ( digits > 15 || frac > 22 ) {
        simple = false;
        break;
      }
    }
  }
  // an exponent, "inf", "nan", hex floats, or no digits at all: not the simple form
  if ( simple && any && ( p == e || !( *p == 'e' || *p == 'E' || *p == 'x' || *p == 'X' || *p == 'p' || *p == 'P' ) ) ) {
    const double v = double( m ) / pow10[frac];
    return neg ? -v : v;
  }
  char buf[64];
  if ( size_t( e - s ) < sizeof( buf ) ) {
    std::memcpy( buf, s, size_t( e - s ) );
    buf[e - s] = 0;
    return atof( buf );
  }
  return atof( std::string( s, e ).c_str() );
}
int tokenToInt( const char* s, const char* e ) {  // atoi
  const char* p   = s;
  bool        neg = false;
  if ( p < e && ( *p == '-' || *p == '+' ) ) neg = *p++ == '-';
  long long v = 0;
  while ( p < e && *p >= '0' && *p <= '9' ) v = v * 10 + ( *p++ - '0' );
  return int( neg ? -v : v );
}
inline int16_t toCoordinate( double v ) { return int16_t( int32_t( v ) ); }  // (the reference assigns a double to an int16_t)

struct Out {
  int16_t* xyz;
  uint8_t* rgb;
  double*  normals;
};

// ASCII body: [p, e) holds whole lines; `skipLines` non-empty lines belong to earlier chunks
struct AsciiChunk {
  const char *begin, *end;
  uint64_t    lines = 0;  // non-empty lines
  uint64_t    first = 0;  // point index of the chunk's first non-empty line
  int         status = TMC2_OK;
};
inline bool emptyLine( const char* p, const char* e ) {
  for ( ; p < e; ++p )
    if ( !isSep( *p ) ) return false;
  return true;
}
void countChunk( AsciiChunk& c ) {
  const char* p = c.begin;
  while ( p < c.end ) {
    const char* nl = static_cast<const char*>( memchr( p, '\n', size_t( c.end - p ) ) );
    const char* e  = nl ? nl : c.end;
    if ( !emptyLine( p, e ) ) ++c.lines;
    p = nl ? nl + 1 : c.end;
  }
}
void parseChunk( AsciiChunk& c, const Header& h, const Out& out ) {
  const size_t                                     need = h.props.size();
  std::vector<std::pair<const char*, const char*>> tok;
  tok.reserve( 16 );
  const char* p     = c.begin;
  uint64_t    point = c.first;
  while ( p < c.end && point < h.points ) {
    const char* nl = static_cast<const char*>( memchr( p, '\n', size_t( c.end - p ) ) );
    const char* e  = nl ? nl : c.end;
    if ( e - p > 4095 ) {
      c.status = TMC2_E_UNSUPPORTED;  // the reference reads lines through a 4096-byte buffer
      return;
    }
    splitLine( p, e, tok );
    p = nl ? nl + 1 : c.end;
    if ( tok.empty() ) continue;
    if ( tok.size() < need ) {
      c.status = TMC2_E_INVALID;  // the reference gives up here
      return;
    }
    out.xyz[3 * point]     = toCoordinate( tokenToDouble( tok[size_t( h.ix )].first, tok[size_t( h.ix )].second ) );
    out.xyz[3 * point + 1] = toCoordinate( tokenToDouble( tok[size_t( h.iy )].first, tok[size_t( h.iy )].second ) );
    out.xyz[3 * point + 2] = toCoordinate( tokenToDouble( tok[size_t( h.iz )].first, tok[size_t( h.iz )].second ) );
    if ( h.colors && out.rgb ) {
      out.rgb[3 * point]     = uint8_t( tokenToInt( tok[size_t( h.ir )].first, tok[size_t( h.ir )].second ) );
      out.rgb[3 * point + 1] = uint8_t( tokenToInt( tok[size_t( h.ig )].first, tok[size_t( h.ig )].second ) );
      out.rgb[3 * point + 2] = uint8_t( tokenToInt( tok[size_t( h.ib )].first, tok[size_t( h.ib )].second ) );
    }
    // (the reference never fills normals from an ASCII body: they stay zero)
    ++point;
  }
}

void parseBinary( const char* body, size_t bytes, const Header& h, uint64_t from, uint64_t to, size_t stride, const Out& out ) {
  std::vector<size_t> offset( h.props.size() );
  size_t              o = 0;
  for ( size_t a = 0; a < h.props.size(); ++a ) {
    offset[a] = o;
    o += size_t( h.props[a].bytes );
  }
  auto coord = [&]( const char* rec, int a ) -> int16_t {
    const char* p = rec + offset[size_t( a )];
    switch ( h.props[size_t( a )].bytes ) {
      case 2: {
        uint16_t v;
        std::memcpy( &v, p, 2 );
        return int16_t( v );
      }
      case 4: {
        float v;
        std::memcpy( &v, p, 4 );
        return int16_t( int32_t( v ) );
      }
      default: {
        double v;
        std::memcpy( &v, p, 8 );
        return toCoordinate( v );
      }
    }
  };
  for ( uint64_t i = from; i < to; ++i ) {
    if ( i * stride >= bytes ) break;  // short file: the rest stays zero
    const char* rec = body + i * stride;
    // a record cut off by the end of the file: the reference still stores the properties it could read completely (in
    // file order, up to the first one that is cut)
    size_t whole = h.props.size();
    if ( ( i + 1 ) * stride > bytes ) {
      const size_t have = bytes - i * stride;
      whole             = 0;
      while ( whole < h.props.size() && offset[whole] + size_t( h.props[whole].bytes ) <= have ) ++whole;
    }
    auto got = [&]( int a ) { return size_t( a ) < whole; };
    if ( got( h.ix ) ) out.xyz[3 * i] = coord( rec, h.ix );
    if ( got( h.iy ) ) out.xyz[3 * i + 1] = coord( rec, h.iy );
    if ( got( h.iz ) ) out.xyz[3 * i + 2] = coord( rec, h.iz );
    if ( h.colors && out.rgb ) {
      if ( got( h.ir ) ) out.rgb[3 * i] = uint8_t( rec[offset[size_t( h.ir )]] );
      if ( got( h.ig ) ) out.rgb[3 * i + 1] = uint8_t( rec[offset[size_t( h.ig )]] );
      if ( got( h.ib ) ) out.rgb[3 * i + 2] = uint8_t( rec[offset[size_t( h.ib )]] );
    }
    if ( h.normals && out.normals ) {
      const int idx[3] = {h.inx, h.iny, h.inz};
      for ( int k = 0; k < 3; ++k ) {
        if ( !got( idx[k] ) ) continue;
        float v;
        std::memcpy( &v, rec + offset[size_t( idx[k] )], 4 );
        out.normals[3 * i + k] = double( v );
      }
    }
  }
}

struct Mapping {
  const char* data = nullptr;
  size_t      size = 0;
  int         fd   = -1;
  ~Mapping() {
    if ( data && size ) munmap( const_cast<char*>( data ), size );
    if ( fd >= 0 ) close( fd );
  }
  int open( const char* path ) {
    fd = ::open( path, O_RDONLY );
    if ( fd < 0 ) {
      setError( "ply: cannot open %s", path );
      return TMC2_E_INVALID;
    }
    struct stat st;
    if ( fstat( fd, &st ) != 0 || st.st_size <= 0 ) {
      setError( "ply: cannot stat %s (or empty file)", path );
      return TMC2_E_INVALID;
    }
    size       = size_t( st.st_size );
    void* addr = mmap( nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0 );
    if ( addr == MAP_FAILED ) {
      data = nullptr;
      setError( "ply: mmap of %s failed", path );
      return TMC2_E_INVALID;
    }
    data = static_cast<const char*>( addr );
    return TMC2_OK;
  }
};
}  // namespace
}  // namespace tmc2

extern "C" {

int tmc2_ply_info( const char* path, int readNormals, uint64_t* pointCount, int* hasColors, int* hasNormals ) {
  if ( !path || !pointCount ) return TMC2_E_INVALID;
  tmc2::Mapping m;
  TMC2_TRY( m.open( path ) );
  tmc2::Header h;
  TMC2_TRY( tmc2::parseHeader( m.data, m.size, readNormals != 0, h ) );
  *pointCount = h.points;
  if ( hasColors ) *hasColors = h.colors ? 1 : 0;
  if ( hasNormals ) *hasNormals = h.normals ? 1 : 0;
  return TMC2_OK;
}

int tmc2_ply_read( const char* path, int16_t* xyz, uint8_t* rgb, double* normals, uint64_t capacity, int threads,
                   uint64_t* pointCount ) {
  if ( !path || !xyz || !pointCount ) return TMC2_E_INVALID;
  const bool timing = getenv( "TMC2_PLY_TIMING" ) != nullptr;
  auto       now    = [] { return std::chrono::steady_clock::now(); };
  auto       ms     = []( std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b ) {
    return std::chrono::duration<double, std::milli>( b - a ).count();
  };
  const auto t0 = now();
  tmc2::Mapping m;
  TMC2_TRY( m.open( path ) );
  tmc2::Header h;
  TMC2_TRY( tmc2::parseHeader( m.data, m.size, normals != nullptr, h ) );
  *pointCount = h.points;
  if ( h.points > capacity ) {
    tmc2::setError( "ply: %llu points do not fit the buffers (%llu)", (unsigned long long)h.points, (unsigned long long)capacity );
    return TMC2_E_INVALID;
  }
  const uint64_t n = h.points;
  std::memset( xyz, 0, size_t( n ) * 6 );
  if ( rgb ) std::memset( rgb, 0, size_t( n ) * 3 );
  if ( normals ) std::memset( normals, 0, size_t( n ) * 24 );
  const tmc2::Out out{xyz, rgb, normals};
  const int       T    = std::max( 1, std::min( threads > 0 ? threads : 8, 64 ) );
  const char*     body = m.data + std::min( h.body, m.size );
  const size_t    left = m.size - std::min( h.body, m.size );
  if ( !h.ascii ) {
    size_t stride = 0;
    for ( auto& p : h.props ) stride += size_t( p.bytes );
    std::vector<std::thread> pool;
    for ( int t = 0; t < T; ++t ) {
      const uint64_t from = n * uint64_t( t ) / uint64_t( T ), to = n * uint64_t( t + 1 ) / uint64_t( T );
      pool.emplace_back( [=, &h] { tmc2::parseBinary( body, left, h, from, to, stride, out ); } );
    }
    for ( auto& th : pool ) th.join();
    return TMC2_OK;
  }
  // ASCII: chunks that end on line boundaries; count the non-empty lines of each, then parse them in parallel
  std::vector<tmc2::AsciiChunk> chunks;
  chunks.resize( size_t( T ) );
  {
    const char* at = body;
    for ( int t = 0; t < T; ++t ) {
      const char* target = t + 1 == T ? body + left : body + left * size_t( t + 1 ) / size_t( T );
      if ( target < at ) target = at;
      const char* nl = target >= body + left ? nullptr : static_cast<const char*>( memchr( target, '\n', size_t( body + left - target ) ) );
      const char* e  = ( t + 1 == T || !nl ) ? body + left : nl + 1;
      chunks[size_t( t )].begin = at;
      chunks[size_t( t )].end   = e;
      at                        = e;
    }
  }
  {
    std::vector<std::thread> pool;
    for ( auto& c : chunks ) pool.emplace_back( [&c] { tmc2::countChunk( c ); } );
    for ( auto& th : pool ) th.join();
  }
  const auto t1 = now();
  uint64_t first = 0;
  for ( auto& c : chunks ) {
    c.first = first;
    first += c.lines;
  }
  {
    std::vector<std::thread> pool;
    for ( auto& c : chunks ) pool.emplace_back( [&c, &h, &out] { tmc2::parseChunk( c, h, out ); } );
    for ( auto& th : pool ) th.join();
  }
  if ( timing ) fprintf( stderr, "ply_read: open + count %.2f ms, parse %.2f ms (%d threads)\n", ms( t0, t1 ), ms( t1, now() ), T );
  // the reference stops at the first bad line: only an error BEFORE the last needed point counts, in file order
  for ( auto& c : chunks ) {
    if ( c.first >= n ) break;
    if ( c.status == TMC2_E_INVALID ) {
      tmc2::setError( "ply: a body line of %s has fewer values than the header declares properties", path );
      return TMC2_E_INVALID;
    }
    if ( c.status != TMC2_OK ) {
      tmc2::setError( "ply: a body line of %s is longer than 4095 characters", path );
      return c.status;
    }
  }
  return TMC2_OK;
}

// PCCPointSet3::write (PCCPointSet.cpp:359-462): the file the reference writes for a reconstructed / decoded frame, byte for
// byte -- header wording included (float coordinates even in ASCII files, the empty face element)
int tmc2_ply_write( const char* path, const int16_t* xyz, const uint8_t* rgb, const double* normals, uint64_t n, int asAscii ) {
  if ( !path || ( n && !xyz ) ) return TMC2_E_INVALID;
  FILE* fp = fopen( path, "wb" );
  if ( !fp ) {
    tmc2::setError( "ply: cannot create %s", path );
    return TMC2_E_INVALID;
  }
  std::string head = "ply\n";
  head += asAscii ? "format ascii 1.0\n" : "format binary_little_endian 1.0\n";
  head += "element vertex " + std::to_string( (unsigned long long)n ) + "\n";
  head += "property float x\nproperty float y\nproperty float z\n";
  if ( normals ) head += "property float nx\nproperty float ny\nproperty float nz\n";
  if ( rgb ) head += "property uchar red\nproperty uchar green\nproperty uchar blue\n";
  head += "element face 0\nproperty list uint8 int32 vertex_index\nend_header\n";
  std::vector<char> out;
  out.reserve( head.size() + size_t( n ) * ( asAscii ? 48 : 27 ) );
  out.insert( out.end(), head.begin(), head.end() );
  char buf[128];
  for ( uint64_t i = 0; i < n; ++i ) {
    if ( asAscii ) {
      int len = snprintf( buf, sizeof( buf ), "%d %d %d", int( xyz[3 * i] ), int( xyz[3 * i + 1] ), int( xyz[3 * i + 2] ) );
      out.insert( out.end(), buf, buf + len );
      if ( normals ) {  // operator<<( float ) at precision max_digits10 of double: "%.17g" of the float's value
        len = snprintf( buf, sizeof( buf ), " %.17g %.17g %.17g", double( float( normals[3 * i] ) ), double( float( normals[3 * i + 1] ) ),
                        double( float( normals[3 * i + 2] ) ) );
        out.insert( out.end(), buf, buf + len );
      }
      if ( rgb ) {
        len = snprintf( buf, sizeof( buf ), " %d %d %d", int( rgb[3 * i] ), int( rgb[3 * i + 1] ), int( rgb[3 * i + 2] ) );
        out.insert( out.end(), buf, buf + len );
      }
      out.push_back( '\n' );
    } else {
      float v[3] = {float( xyz[3 * i] ), float( xyz[3 * i + 1] ), float( xyz[3 * i + 2] )};
      out.insert( out.end(), reinterpret_cast<char*>( v ), reinterpret_cast<char*>( v ) + 12 );
      if ( normals ) {
        float w[3] = {float( normals[3 * i] ), float( normals[3 * i + 1] ), float( normals[3 * i + 2] )};
        out.insert( out.end(), reinterpret_cast<char*>( w ), reinterpret_cast<char*>( w ) + 12 );
      }
      if ( rgb ) out.insert( out.end(), rgb + 3 * i, rgb + 3 * i + 3 );
    }
  }
  const bool ok = fwrite( out.data(), 1, out.size(), fp ) == out.size();
  if ( fclose( fp ) != 0 || !ok ) {
    tmc2::setError( "ply: writing %s failed", path );
    return TMC2_E_INVALID;
  }
  return TMC2_OK;
}

// PCCChecksum::write / read (PccLibMetrics/source/PCCChecksum.cpp:112-139): the .checksum file next to the bitstream: the
// number of frames, the checksum size (16), then 32 hex digits per frame.
int tmc2_checksum_file_write( const char* path, const uint8_t* digests, uint64_t frames ) {
  if ( !path || ( frames && !digests ) ) return TMC2_E_INVALID;
  FILE* fp = fopen( path, "wb" );
  if ( !fp ) {
    tmc2::setError( "checksum file: cannot create %s", path );
    return TMC2_E_INVALID;
  }
  fprintf( fp, "%llu\n%d\n", (unsigned long long)frames, frames ? 16 : 0 );
  for ( uint64_t f = 0; f < frames; ++f ) {
    for ( int k = 0; k < 16; ++k ) fprintf( fp, "%02x", digests[16 * f + uint64_t( k )] );
    fputc( '\n', fp );
  }
  if ( fclose( fp ) != 0 ) return TMC2_E_INVALID;
  return TMC2_OK;
}
int tmc2_checksum_file_read( const char* path, uint8_t* digests, uint64_t capacity, uint64_t* frames ) {
  if ( !path || !frames ) return TMC2_E_INVALID;
  FILE* fp = fopen( path, "rb" );
  if ( !fp ) {
    tmc2::setError( "checksum file: cannot open %s", path );
    return TMC2_E_INVALID;
  }
  unsigned long long n = 0, size = 0;
  if ( fscanf( fp, "%llu %llu", &n, &size ) != 2 || ( n && size < 16 ) ) {
    fclose( fp );
    tmc2::setError( "checksum file: corrupted header in %s", path );
    return TMC2_E_INVALID;
  }
  *frames = n;
  if ( n > capacity || ( n && !digests ) ) {
    fclose( fp );
    return digests ? TMC2_E_INVALID : TMC2_OK;
  }
  for ( unsigned long long f = 0; f < n; ++f )
    for ( unsigned long long k = 0; k < size; ++k ) {
      char     c[2];
      unsigned v = 0;
      if ( fscanf( fp, " %c%c", &c[0], &c[1] ) != 2 ) {
        fclose( fp );
        tmc2::setError( "checksum file: %s ends early", path );
        return TMC2_E_INVALID;
      }
      for ( int d = 0; d < 2; ++d ) v = v * 16 + unsigned( ( c[d] + ( c[d] > '9' ? 9 : 0 ) ) & 0x0F );
      if ( k < 16 ) digests[16 * f + k] = uint8_t( v );
    }
  fclose( fp );
  return TMC2_OK;
}

int tmc2_point_set_checksum( const int16_t* xyz, const uint8_t* rgb, uint64_t n, int reorderPoints, uint8_t digest[16] ) {
  if ( ( n && !xyz ) || !digest ) return TMC2_E_INVALID;
  tmc2::Md5 md5;
  if ( !reorderPoints ) {
    md5.update( reinterpret_cast<const uint8_t*>( xyz ), size_t( n ) * 6 );
    if ( rgb ) md5.update( rgb, size_t( n ) * 3 );
    md5.finish( digest );
    return TMC2_OK;
  }
  // PCCPointSet3::reorder( dropDuplicates ): positions in (x, y, z) order; with colours one point per position, its colour
  // the integer mean of the colours that share it -- without colours the reference keeps the duplicates
  std::vector<uint32_t> order;
  tmc2::lexOrderStable( xyz, size_t( n ), order );
  auto key = [&]( uint32_t i ) {
    return ( uint64_t( uint16_t( xyz[3 * size_t( i )] ) ) << 32 ) | ( uint64_t( uint16_t( xyz[3 * size_t( i ) + 1] ) ) << 16 ) |
           uint64_t( uint16_t( xyz[3 * size_t( i ) + 2] ) );
  };
  std::vector<int16_t> pos;
  std::vector<uint8_t> col;
  pos.reserve( size_t( n ) * 3 );
  if ( rgb ) col.reserve( size_t( n ) * 3 );
  for ( uint64_t i = 0; i < n; ) {
    uint64_t j = i;
    uint64_t s[3] = {0, 0, 0};
    while ( j < n && key( order[j] ) == key( order[i] ) ) {
      if ( rgb )
        for ( int k = 0; k < 3; ++k ) s[k] += rgb[3 * size_t( order[j] ) + k];
      ++j;
    }
    for ( uint64_t r = 0; r < ( rgb ? 1 : j - i ); ++r )
      for ( int k = 0; k < 3; ++k ) pos.push_back( xyz[3 * size_t( order[i] ) + k] );
    if ( rgb )
      for ( int k = 0; k < 3; ++k ) col.push_back( uint8_t( s[k] / ( j - i ) ) );
    i = j;
  }
  md5.update( reinterpret_cast<const uint8_t*>( pos.data() ), pos.size() * 2 );
  if ( rgb ) md5.update( col.data(), col.size() );
  md5.finish( digest );
  return TMC2_OK;
}
}
