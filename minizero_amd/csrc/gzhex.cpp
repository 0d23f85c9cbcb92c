// compressString of the reference's records (ref utils/utils.h:35-91): the Atari `OBS[...]` tag = gzip member of all observation bytes, as
// lower-case hex.  Host-side record formatting, once per finished sequence: deflate comes from the system zlib (the library the reference's
// boost::iostreams::gzip_compressor wraps), the gzip framing is written here the way boost's filter with default gzip_params writes it:
// 10-byte header without name / comment, MTIME 0, XFL 0, OS 255 (unknown); raw deflate at zlib's default level, 32 K window, mem level 8;
// CRC-32 and input length, little-endian.
#include "common.h"
#include <zlib.h>
#include <cstring>

namespace mz {

bool compressToHex(const uint8_t* data, size_t n, std::string* hex)
{
    hex->clear();
    if (n == 0) { return true; } // utils.h:37: the empty string stays empty
    if (n > 0xFFFFFFF0ull) { setError("compressToHex: %zu bytes are more than one gzip member takes here", n); return false; }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) { setError("deflateInit2 failed"); return false; }
    static const char kDigits[] = "0123456789abcdef";
    auto put = [&](const uint8_t* p, size_t k) {
        for (size_t i = 0; i < k; ++i) { hex->push_back(kDigits[p[i] >> 4]); hex->push_back(kDigits[p[i] & 15]); }
    };
    const uint8_t header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    hex->reserve(2 * (n / 8 + 64));
    put(header, sizeof(header));
    std::vector<uint8_t> chunk(1 << 16);
    zs.next_in = const_cast<Bytef*>(data);
    zs.avail_in = static_cast<uInt>(n);
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        zs.next_out = chunk.data();
        zs.avail_out = static_cast<uInt>(chunk.size());
        rc = deflate(&zs, Z_FINISH);
        if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { deflateEnd(&zs); setError("deflate failed (%d)", rc); return false; }
        put(chunk.data(), chunk.size() - zs.avail_out);
    }
    deflateEnd(&zs);
    const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), data, static_cast<uInt>(n))), len = static_cast<uint32_t>(n);
    uint8_t footer[8];
    for (int k = 0; k < 4; ++k) { footer[k] = static_cast<uint8_t>(crc >> (8 * k)); footer[4 + k] = static_cast<uint8_t>(len >> (8 * k)); }
    put(footer, sizeof(footer));
    return true;
}

} // namespace mz

extern "C" long mz_compress_string(const void* data, size_t n, char* out, size_t capacity)
{
    if (!data && n) { mz::setError("mz_compress_string: NULL data"); return MZ_ERR_ARG; }
    std::string hex;
    if (!mz::compressToHex(static_cast<const uint8_t*>(data), n, &hex)) { return MZ_ERR_ARG; }
    if (out) {
        if (capacity <= hex.size()) { mz::setError("mz_compress_string: buffer of %zu bytes, %zu needed", capacity, hex.size() + 1); return MZ_ERR_ARG; }
        memcpy(out, hex.data(), hex.size());
        out[hex.size()] = 0;
    }
    return static_cast<long>(hex.size());
}
