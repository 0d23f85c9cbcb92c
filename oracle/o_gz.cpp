// ORACLE (test infrastructure): compressString of the reference — ref utils/utils.h:35-91.
// The reference pipes the bytes through boost::iostreams::gzip_compressor with default gzip_params and prints two lower-case hex digits per
// byte (std::hex, setw(2), setfill('0')).  Boost is absent from this image, so the member layout is restated from Boost.Iostreams' published
// gzip filter (boost/iostreams/filter/gzip.hpp, any 1.7x: basic_gzip_compressor's constructor and prepare_footer):
//   header  1f 8b | CM 08 | FLG 00 (no name / comment) | MTIME 00 00 00 00 (gzip_params::mtime = 0) | XFL 00 (level neither best_compression
//           nor best_speed) | OS ff (gzip::os_unknown)
//   body    raw deflate stream of zlib at its default level (zlib_params: level -1 -> 6, method deflated, window_bits 15 negated by
//           noheader = true, mem_level 8, strategy default) — zlib itself is what boost links, and it is in this image
//   footer  CRC-32 of the input, then its length mod 2^32, both little-endian
// PARITY: pinned to tests/golden/compress_string.json (made with Python's zlib by tests/golden/gen_obs_golden.py, same explicit layout) and by
// the round trip through Python's gzip module; the header bytes above are NOT checked against a real boost build (none available here).
#include "oracle.h"
#include <zlib.h>
#include <stdexcept>

namespace mzo {

std::string compressString(const std::string& s)
{
    if (s.empty()) { return s; } // utils.h:37
    std::string bin("\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff", 10);
    z_stream zs{};
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { throw std::runtime_error("deflateInit2"); }
    std::string body(deflateBound(&zs, s.size()) + 16, '\0');
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(s.data()));
    zs.avail_in = static_cast<uInt>(s.size());
    zs.next_out = reinterpret_cast<Bytef*>(&body[0]);
    zs.avail_out = static_cast<uInt>(body.size());
    const int rc = deflate(&zs, Z_FINISH);
    body.resize(zs.total_out);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) { throw std::runtime_error("deflate"); }
    bin += body;
    const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(s.data()), static_cast<uInt>(s.size())));
    const uint32_t len = static_cast<uint32_t>(s.size());
    for (int k = 0; k < 4; ++k) { bin += static_cast<char>((crc >> (8 * k)) & 0xFF); }
    for (int k = 0; k < 4; ++k) { bin += static_cast<char>((len >> (8 * k)) & 0xFF); }
    static const char* digits = "0123456789abcdef";
    std::string hex;
    hex.reserve(bin.size() * 2);
    for (unsigned char c : bin) { hex += digits[c >> 4]; hex += digits[c & 15]; }
    return hex;
}

} // namespace mzo
