// tw_inflate.h — the host layer's DEFLATE (RFC 1951) / zlib (RFC 1950) decoder and PNG unfilter, written for
// throughput: image decode is what bounds the service once the flow runs on the GPU (the reference
// decodes inside OpticalFlow::calculate, /root/reference/src/opticalflow.cpp:37-48, through cv::imread -> libpng ->
// zlib).  Byte-identical to zlib / libpng on valid input; rejects what zlib rejects.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace twhost {

// Raw DEFLATE stream -> dst (at most cap bytes).  *out_len = bytes produced, *in_used = bytes of src consumed (up to the
// byte that holds the last bit of the final block).  false: malformed or truncated stream, or dst too small.
bool tw_inflate_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len, size_t* in_used);
// zlib-wrapped stream (2-byte header, DEFLATE, Adler-32): what zlib's uncompress() accepts (bytes after the stream
// are ignored, as there).
bool tw_inflate_zlib(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len);
uint32_t tw_adler32(uint32_t adler, const uint8_t* p, size_t n);

// PNG scanline filters in place.  raw = rows x (1 filter-type byte + rowbytes); fbpp = bytes per complete pixel
// (at least 1).  false: a filter type above 4.
bool tw_png_unfilter(uint8_t* raw, size_t rowbytes, size_t rows, size_t fbpp);
// one row; prev = the unfiltered previous row or nullptr for the first row of an image / interlace pass
bool tw_png_unfilter_row(int filter_type, uint8_t* cur, const uint8_t* prev, size_t rowbytes, size_t fbpp);

}  // namespace twhost
