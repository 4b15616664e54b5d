// inflate_test_api.cpp — C entry points over tw_inflate.{h,cpp} and load_gray for tests/test_png_fast.py (ctypes):
// built as build/libinflate_test.so (`make inflate_test`, plain) and build/libinflate_asan.so (`make inflate_asan`,
// AddressSanitizer + UBSan, loaded into a python that preloads libasan).  Test plumbing only.
#include <string.h>

#include <string>
#include <vector>

#include "tw_inflate.h"
#include "twhost.h"

extern "C" {
int twt_inflate_zlib(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* outlen)
{
    return twhost::tw_inflate_zlib(s, n, d, cap, outlen) ? 1 : 0;
}
int twt_inflate_raw(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* outlen, size_t* used)
{
    return twhost::tw_inflate_raw(s, n, d, cap, outlen, used) ? 1 : 0;
}
unsigned twt_adler32(unsigned a, const uint8_t* p, size_t n) { return twhost::tw_adler32(a, p, n); }
int twt_unfilter(uint8_t* raw, size_t rowbytes, size_t rows, size_t fbpp)
{
    return twhost::tw_png_unfilter(raw, rowbytes, rows, fbpp) ? 1 : 0;
}
// load_gray(path): 1 and w, h, the first min(cap, w*h) bytes on success; 0 if the file cannot be decoded
int twt_load_gray(const char* path, uint8_t* out, size_t cap, int* w, int* h)
{
    std::vector<uint8_t> img;
    int ww = 0, hh = 0;
    const bool ok = twhost::load_gray(path, img, ww, hh);
    if (!ok) {  // what a failed decode reports as its size (must be 0 x 0: it used to size the page-locked arena)
        *w = ww;
        *h = hh;
        return 0;
    }
    if (img.size() != (size_t)ww * hh) return 0;
    *w = ww;
    *h = hh;
    memcpy(out, img.data(), img.size() < cap ? img.size() : cap);
    return 1;
}
// load_gray_or_png_rows(path, want_rows = true): 1 + w, h, ch and the first min(cap, size) bytes (ch = 0: gray pixels;
// ch 1-4: filtered PNG rows); *size = the full byte count
int twt_load_rows(const char* path, uint8_t* out, size_t cap, int* w, int* h, int* ch, size_t* size)
{
    std::vector<uint8_t> img;
    if (!twhost::load_gray_or_png_rows(path, true, img, *w, *h, *ch)) return 0;
    *size = img.size();
    memcpy(out, img.data(), img.size() < cap ? img.size() : cap);
    return 1;
}
// finish_png_rows_on_host: rows -> gray (w * h bytes into out)
int twt_finish_rows(const uint8_t* rows, size_t n, int w, int h, int ch, uint8_t* out)
{
    std::vector<uint8_t> r(rows, rows + n), g;
    if (!twhost::finish_png_rows_on_host(r, w, h, ch, g) || g.size() != (size_t)w * h) return 0;
    memcpy(out, g.data(), g.size());
    return 1;
}
// resize_u8_linear (the <= 5 px size reconcile of prepare(), src/opticalflow.cpp:64-68): dw * dh bytes into out
void twt_resize_u8(const uint8_t* src, int sw, int sh, uint8_t* out, int dw, int dh)
{
    std::vector<uint8_t> s(src, src + (size_t)sw * sh), d;
    twhost::resize_u8_linear(s, sw, sh, d, dw, dh);
    memcpy(out, d.data(), (size_t)dw * dh);
}
// the same inflate-heavy loop a mutation fuzzer wants, inside the sanitised library: `iters` mutations of one zlib
// stream, each decoded into a buffer of exactly `cap` bytes; returns how many were accepted
long twt_fuzz_stream(const uint8_t* s, size_t n, size_t cap, int iters, unsigned seed)
{
    std::vector<uint8_t> b(n), out(cap);
    long ok = 0;
    for (int it = 0; it < iters; it++) {
        memcpy(b.data(), s, n);
        const int m = 1 + (int)(seed % 4);
        for (int k = 0; k < m; k++) {
            seed = seed * 1664525u + 1013904223u;
            b[(seed >> 8) % n] ^= (uint8_t)(1u << (seed >> 29));
        }
        seed = seed * 1664525u + 1013904223u;
        const size_t len = (it % 7 == 0) ? 2 + (seed >> 8) % (n - 2) : n;
        size_t produced = 0;
        if (twhost::tw_inflate_zlib(b.data(), len, out.data(), cap, &produced)) ok++;
    }
    return ok;
}
}
